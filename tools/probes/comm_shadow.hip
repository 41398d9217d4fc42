// Measurement probe, NOT part of libvipant_hip.so (it lives in tools/probes/libvipant_probes.so, built by
// vipant_amd.build.build_probes()): a stand-in for the RCCL all-reduce kernel of one gradient bucket, for measuring on ONE GPU what
// the overlapped reduction costs the step.  `nwg` workgroups of 256 threads (an RCCL channel is one workgroup) copy `bytes` from src
// to dst, each holds its CU -- 16 KiB of LDS, like a channel's staging area, so that a kernel which needs a CU's whole LDS cannot
// share it -- and none leaves before `min_us` have passed since it started (the time a collective spends waiting for its peers).
// Users: tools/comm_shadow.py (profiles/*_comm_shadow*.md) and the ticket-walk tests (CUs held on a second stream).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ __launch_bounds__(256) void comm_shadow_kernel(const uint4* src, uint4* dst, int64_t n16, int64_t hold_ticks) {
    __shared__ uint4 hold[1024];                              // 16 KiB
    const uint64_t t0 = wall_clock64();                       // 100 MHz constant clock
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint4 last = uint4{0u, 0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        last = src[i];
        dst[i] = last;
    }
    hold[threadIdx.x] = last;
    __syncthreads();
    if (hold[(threadIdx.x + 1) & 255].x == 0xFFFFFFFFu && dst != nullptr && n16 < 0) dst[0] = hold[0];     // keeps the array
    while ((int64_t)(wall_clock64() - t0) < hold_ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

// 0 on success, -1 bad arguments, -2 launch failure
extern "C" int32_t probe_comm_shadow(const void* src, void* dst, size_t bytes, int32_t nwg, float min_us, void* stream) {
    if (src == nullptr || dst == nullptr || bytes % 16 != 0 || (uintptr_t)src % 16 != 0 || (uintptr_t)dst % 16 != 0) return -1;
    if (nwg <= 0 || nwg > 256 || !(min_us >= 0.f && min_us <= 1e5f)) return -1;
    hipLaunchKernelGGL(comm_shadow_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       (int64_t)(bytes / 16), (int64_t)(min_us * 100.0f));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
