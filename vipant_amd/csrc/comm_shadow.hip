// A stand-in for the RCCL all-reduce kernel of one gradient bucket (vipant_amd/parallel.py: GradSync.reduce_async), for measuring on
// ONE GPU what the overlapped reduction costs the step: `nwg` workgroups of 256 threads (an RCCL channel is one workgroup) copy
// `bytes` from src to dst, each holds its CU -- 16 KiB of LDS, like a channel's staging area, so that a kernel which needs a CU's
// whole LDS cannot share it -- and none leaves before `min_us` have passed since it started (the time a collective spends waiting
// for its peers).  Measurement utility: nothing on the training path calls it unless VIPANT_COMM_SHADOW is set
// (profiles/r5_comm_shadow.md, tools/comm_shadow.py).
#include "common.h"

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

extern "C" int32_t vipant_comm_shadow(const void* src, void* dst, size_t bytes, int32_t nwg, float min_us, void* stream) {
    VIPANT_REQUIRE(src != nullptr && dst != nullptr && bytes % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                   VIPANT_EALIGN, "comm_shadow: src / dst / bytes must be 16-byte aligned");
    VIPANT_REQUIRE(nwg > 0 && nwg <= 256 && min_us >= 0.f && min_us <= 1e5f, VIPANT_EBADSHAPE,
                   "comm_shadow: 1..256 workgroups, 0..100 ms (nwg=%d min_us=%g)", nwg, (double)min_us);
    hipLaunchKernelGGL(comm_shadow_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       (int64_t)(bytes / 16), (int64_t)(min_us * 100.0f));
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}
