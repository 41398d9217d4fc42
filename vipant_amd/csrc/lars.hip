// LARS step over a list of tensors in two launches (cvap/module/lars.py:43-72): the reference runs ~10 tiny
// torch kernels per tensor (153 tensors -> launch-bound); here every (tensor, chunk) pair is one workgroup.
//   pass 1  partial sums of |p|^2 and |g + wd p|^2 per chunk (deterministic, no atomics)
//   pass 2  q = eta |p| / |dp| (1 if either norm is 0; adapt only), mu = momentum mu + q dp, p -= lr mu
#include "common.h"

namespace {

constexpr int CHUNKS = 32;

struct LarsArgs {
    float* const* p; const float* const* g; float* const* mu;
    const int64_t* n; const int32_t* adapt; const float* lr;
    float* partial;   // [ntensors][CHUNKS][2]
    float wd, momentum, eta;
};

__device__ __forceinline__ void chunk_range(int64_t n, int c, int64_t* b, int64_t* e) {
    const int64_t per = ((n + CHUNKS - 1) / CHUNKS + 3) & ~(int64_t)3;
    *b = per * c;
    *e = *b + per < n ? *b + per : n;
}

__global__ __launch_bounds__(256) void lars_norm_kernel(LarsArgs a) {
    __shared__ float red[2][256];
    const int t = blockIdx.y, c = blockIdx.x;
    float sp = 0.f, sd = 0.f;
    if (a.adapt[t]) {
        int64_t b, e;
        chunk_range(a.n[t], c, &b, &e);
        const float* p = a.p[t];
        const float* g = a.g[t];
        int64_t i0 = b;
        if ((((uintptr_t)p | (uintptr_t)g) & 15) == 0) {      // chunk starts are multiples of 4 elements: 16-byte loads
            const int64_t e4 = b + ((e - b) & ~(int64_t)3);
            for (int64_t i = b + 4 * (int64_t)threadIdx.x; i < e4; i += 1024) {
                const f32x4 pv = *(const f32x4*)(p + i);
                const f32x4 dv = *(const f32x4*)(g + i) + a.wd * pv;
                sp += (pv[0] * pv[0] + pv[1] * pv[1]) + (pv[2] * pv[2] + pv[3] * pv[3]);
                sd += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
            }
            i0 = e4;
        }
        for (int64_t i = i0 + threadIdx.x; i < e; i += 256) {
            const float pv = p[i], dv = g[i] + a.wd * pv;
            sp += pv * pv;
            sd += dv * dv;
        }
    }
    red[0][threadIdx.x] = sp; red[1][threadIdx.x] = sd;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.partial[((int64_t)t * CHUNKS + c) * 2 + 0] = red[0][0];
        a.partial[((int64_t)t * CHUNKS + c) * 2 + 1] = red[1][0];
    }
}

__global__ __launch_bounds__(256) void lars_update_kernel(LarsArgs a) {
    const int t = blockIdx.y, c = blockIdx.x;
    int64_t b, e;
    chunk_range(a.n[t], c, &b, &e);
    if (b >= e) return;
    const bool adapt = a.adapt[t] != 0;
    float q = 1.f;
    if (adapt) {
        float sp = 0.f, sd = 0.f;
        for (int k = 0; k < CHUNKS; ++k) {
            sp += a.partial[((int64_t)t * CHUNKS + k) * 2 + 0];
            sd += a.partial[((int64_t)t * CHUNKS + k) * 2 + 1];
        }
        const float pn = sqrtf(sp), un = sqrtf(sd);
        q = pn > 0.f ? (un > 0.f ? a.eta * pn / un : 1.f) : 1.f;
    }
    float* p = a.p[t];
    const float* g = a.g[t];
    float* mu = a.mu[t];
    const float lr = a.lr[t], wd = adapt ? a.wd : 0.f;
    // (16-byte accesses, as in the norm pass, make THIS pass slower: 305 -> 342 us over the 85 M parameters; four-byte ones stay)
    const int64_t i0 = b;
    for (int64_t i = i0 + threadIdx.x; i < e; i += 256) {
        const float pv = p[i];
        const float dp = (g[i] + wd * pv) * q;
        const float m = mu[i] * a.momentum + dp;
        mu[i] = m;
        p[i] = pv - lr * m;
    }
}

}  // namespace

extern "C" size_t vipant_lars_workspace_bytes(int64_t ntensors) { return (size_t)ntensors * CHUNKS * 2 * sizeof(float); }

extern "C" int32_t vipant_lars_step(float* const* p, const float* const* g, float* const* mu, const int64_t* n,
                                    const int32_t* adapt, const float* lr, int64_t ntensors, float weight_decay,
                                    float momentum, float eta, void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(ntensors > 0 && ntensors < 65536, VIPANT_EBADSHAPE, "lars_step: bad tensor count %ld", (long)ntensors);
    VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= vipant_lars_workspace_bytes(ntensors), VIPANT_ENOWORKSPACE,
                   "lars_step: workspace too small");
    LarsArgs a{p, g, mu, n, adapt, lr, (float*)workspace, weight_decay, momentum, eta};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lars_norm_kernel, dim3(CHUNKS, (unsigned)ntensors), dim3(256), 0, s, a);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL(lars_update_kernel, dim3(CHUNKS, (unsigned)ntensors), dim3(256), 0, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}
