// Log-mel front-end on the GPU: the Kaldi-style filter bank the reference computes per clip on CPU dataloader workers
// (cvap/data/audio/transform.py:12-35 -> torchaudio.compliance.kaldi.fbank with the parameters of
// cvap/data/image_audio.py:119-126), followed by the dataset's zero padding, (x - mean) / std and SpecAugment
// masks (image_audio.py:183-207).  Output is the [b, 1, T, F] fp32 spectrogram batch the audio tower consumes, so the
// waveform never leaves the device once it is there.
//
// One wave per frame PAIR (two real frames packed into one complex transform), four waves per workgroup, each walking
// four pairs (the workgroup's tables serve 32 frames).  A frame is framed (snip_edges), DC-removed, pre-emphasised
// (replicated first sample), windowed and zero-padded to NFFT in LDS; a radix-2 Stockham FFT (log2 NFFT passes
// ping-ponging two LDS buffers, twiddles from sincospi -- exact at the power-of-two fractions) gives the spectrum;
// every lane then owns mel bins lane, lane + 64, ... and sums power x weight over that filter's (contiguous)
// support; log with the epsilon floor, normalisation and masks are applied on the way out (coalesced row stores).
// HBM traffic = waveform read (overlapping frames hit L2) + b*T*F*4 B written: the kernel is launch/HBM bound, the
// FFT is ~25 kFLOP per frame.
#include "common.h"

namespace {

struct FbankArgs {
    const float* wave; int64_t wave_stride; const int64_t* nsamples; float* out;
    const float* window; const float* banks; const int32_t* bank_start; const int32_t* bank_len; const int32_t* masks;
    const float* clip_mean;
    int T, F, size, shift;
    float preemph, norm_mean, norm_std;
};

__global__ __launch_bounds__(256) void clip_mean_kernel(const float* __restrict__ wave, int64_t stride,
                                                        const int64_t* __restrict__ nsamples, float* __restrict__ mean) {
    __shared__ float red[256];
    const int clip = blockIdx.x;
    const int64_t n = nsamples[clip];
    const float* w = wave + (int64_t)clip * stride;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += w[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[clip] = n > 0 ? red[0] / (float)n : 0.f;
}

// LDS traffic of one wave is processed in program order, and a frame is private to its wave: between FFT passes it is
// enough to stop the compiler from moving LDS accesses across the pass boundary.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

constexpr int FPW = 4;            // frame PAIRS per wave: the workgroup's tables (twiddles, window, mel weights) serve 32 frames
constexpr int MAXF = 256;         // mel bins

template <int NFFT>
__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_off[MAXF + 1], s_start[MAXF], s_len[MAXF];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int clip = blockIdx.y;
    float2* tw = (float2*)smem;                                   // exp(-2 pi i k / NFFT), k < NFFT / 2
    float* win = (float*)(tw + NFFT / 2);                         // window, NFFT floats reserved
    float* wgt = win + NFFT;                                      // packed non-zero mel weights, <= NFFT floats
    float2* buf0 = (float2*)(wgt + NFFT) + (size_t)wave * 2 * NFFT;
    float2* buf1 = buf0 + NFFT;
    float* raw = (float*)buf1;                                    // the frame's samples before windowing

    // ---- workgroup tables
    for (int i = threadIdx.x; i < NFFT / 2; i += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)i / (float)NFFT, &sn, &cs);      // exact at the power-of-two fractions
        tw[i] = make_float2(cs, sn);
    }
    for (int i = threadIdx.x; i < a.size; i += 256) win[i] = a.window[i];
    if ((int)threadIdx.x < a.F) { s_start[threadIdx.x] = a.bank_start[threadIdx.x]; s_len[threadIdx.x] = a.bank_len[threadIdx.x]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int o = 0;
        for (int m = 0; m < a.F; ++m) { s_off[m] = o; o += s_len[m]; }
        s_off[a.F] = o;
    }
    __syncthreads();
    const int nnz = s_off[a.F] <= NFFT ? s_off[a.F] : 0;          // more than two filters per bin: read the weights from global
    if (nnz > 0) {
        for (int m = wave; m < a.F; m += 4) {
            const float* row = a.banks + (int64_t)m * (NFFT / 2 + 1) + s_start[m];
            for (int k = lane; k < s_len[m]; k += 64) wgt[s_off[m] + k] = row[k];
        }
    }
    __syncthreads();

    const int64_t n = a.nsamples[clip];
    const int nframes = n < a.size ? 0 : (int)(1 + (n - a.size) / a.shift);
    const float cmean = a.clip_mean != nullptr ? a.clip_mean[clip] : 0.f;
    const int32_t* mk = a.masks != nullptr ? a.masks + (int64_t)clip * 4 : nullptr;
    int f0 = 0, f1 = 0, t0 = 0, t1 = 0;
    if (mk != nullptr) { f0 = mk[0]; f1 = mk[1]; t0 = mk[2]; t1 = mk[3]; }

    // Two real frames share one complex transform: z = y_a + i y_b, and with Z = FFT(z)
    //   X_a[k] = (Z[k] + conj Z[N-k]) / 2,   X_b[k] = (Z[k] - conj Z[N-k]) / (2i)
    // so |X_a|^2 = ((Zr + Zr')^2 + (Zi - Zi')^2) / 4 and |X_b|^2 = ((Zi + Zi')^2 + (Zr' - Zr)^2) / 4 with Z' = Z[N-k].
    // A wave walks FPW frame pairs (2t, 2t + 1); the FFT's LDS traffic per frame halves.
#pragma unroll 1
    for (int it = 0; it < FPW; ++it) {
        const int ta = ((blockIdx.x * FPW + it) * 4 + wave) * 2, tb = ta + 1;
        if (ta >= a.T) break;
        const bool live_a = ta < nframes && !(ta >= t0 && ta < t1);
        const bool live_b = tb < a.T && tb < nframes && !(tb >= t0 && tb < t1);
        const float padv = a.norm_std != 0.f ? (0.f - a.norm_mean) / a.norm_std : 0.f;
        float* dst_a = a.out + ((int64_t)clip * a.T + ta) * a.F;
        float* dst_b = dst_a + a.F;
        if (!live_a) {                                            // padding row or time-masked row: constant
            const bool tm = ta >= t0 && ta < t1;
            for (int m = lane; m < a.F; m += 64) dst_a[m] = (tm || (m >= f0 && m < f1)) ? 0.f : padv;
        }
        if (!live_b && tb < a.T) {
            const bool tm = tb >= t0 && tb < t1;
            for (int m = lane; m < a.F; m += 64) dst_b[m] = (tm || (m >= f0 && m < f1)) ? 0.f : padv;
        }
        if (!live_a && !live_b) continue;
        const float* src = a.wave + (int64_t)clip * a.wave_stride + (int64_t)ta * a.shift;
        float* raw_b = raw + NFFT;
        float sa = 0.f, sb = 0.f;
        for (int j = lane; j < a.size; j += 64) {
            const float xa = live_a ? src[j] - cmean : 0.f;
            const float xb = live_b ? src[j + a.shift] - cmean : 0.f;
            raw[j] = xa; raw_b[j] = xb;
            sa += xa; sb += xb;
        }
        sa = wave_sum(sa); sb = wave_sum(sb);
        const float ma = sa / (float)a.size, mb = sb / (float)a.size;
        wave_lds_sync();
        for (int j = lane; j < NFFT; j += 64) {
            float ya = 0.f, yb = 0.f;
            if (j < a.size) {
                const int jp = j > 0 ? j - 1 : 0;
                const float w = win[j];
                ya = ((raw[j] - ma) - a.preemph * (raw[jp] - ma)) * w;
                yb = ((raw_b[j] - mb) - a.preemph * (raw_b[jp] - mb)) * w;
            }
            buf0[j] = make_float2(ya, yb);
        }
        wave_lds_sync();

        // Stockham autosort FFT: one radix-2 pass if log2(NFFT) is odd, then radix-4 passes (sub-transform length ns -> 4 ns)
        float2* in = buf0;
        float2* out = buf1;
        int ns = 1;
        if ((31 - __builtin_clz(NFFT)) & 1) {
            for (int j = lane; j < NFFT / 2; j += 64) {
                const float2 u = in[j], v = in[j + NFFT / 2];
                out[2 * j] = make_float2(u.x + v.x, u.y + v.y);
                out[2 * j + 1] = make_float2(u.x - v.x, u.y - v.y);
            }
            wave_lds_sync();
            float2* tmp = in; in = out; out = tmp;
            ns = 2;
        }
#pragma unroll 1
        for (; ns < NFFT; ns <<= 2) {
            const int tstep = NFFT / (4 * ns);
            for (int j = lane; j < NFFT / 4; j += 64) {
                const int k = j & (ns - 1);
                const int i1 = k * tstep;
                const float2 w1 = tw[i1], w2 = tw[2 * i1];
                float2 w3 = tw[(3 * i1) & (NFFT / 2 - 1)];
                if (3 * i1 >= NFFT / 2) w3 = make_float2(-w3.x, -w3.y);
                const float2 v0 = in[j], v1 = cmul(in[j + NFFT / 4], w1), v2 = cmul(in[j + NFFT / 2], w2),
                             v3 = cmul(in[j + 3 * NFFT / 4], w3);
                const float2 p = make_float2(v0.x + v2.x, v0.y + v2.y), q = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 r = make_float2(v1.x + v3.x, v1.y + v3.y);
                const float2 d = make_float2(v1.y - v3.y, -(v1.x - v3.x));           // (v1 - v3) * (-i)
                const int o = ((j - k) << 2) + k;
                out[o] = make_float2(p.x + r.x, p.y + r.y);
                out[o + ns] = make_float2(q.x + d.x, q.y + d.y);
                out[o + 2 * ns] = make_float2(p.x - r.x, p.y - r.y);
                out[o + 3 * ns] = make_float2(q.x - d.x, q.y - d.y);
            }
            wave_lds_sync();
            float2* tmp = in; in = out; out = tmp;
        }
        // `in` holds Z; both frames' one-sided power spectra come from Z[k] and Z[N - k]
        for (int m = lane; m < a.F; m += 64) {
            const int k0 = s_start[m], len = s_len[m];
            const float* w = nnz > 0 ? wgt + s_off[m] : a.banks + (int64_t)m * (NFFT / 2 + 1) + k0;
            float ea = 0.f, eb = 0.f;
            for (int k = 0; k < len; ++k) {
                const float2 z = in[k0 + k], zc = in[(NFFT - (k0 + k)) & (NFFT - 1)];
                const float ar = z.x + zc.x, ai = z.y - zc.y, br = z.y + zc.y, bi = zc.x - z.x;
                const float wk = 0.25f * w[k];
                ea += (ar * ar + ai * ai) * wk;
                eb += (br * br + bi * bi) * wk;
            }
            const bool fm = m >= f0 && m < f1;
            if (live_a) {
                float v = logf(fmaxf(ea, 1.1920928955078125e-07f));
                if (a.norm_std != 0.f) v = (v - a.norm_mean) / a.norm_std;
                dst_a[m] = fm ? 0.f : v;
            }
            if (live_b) {
                float v = logf(fmaxf(eb, 1.1920928955078125e-07f));
                if (a.norm_std != 0.f) v = (v - a.norm_mean) / a.norm_std;
                dst_b[m] = fm ? 0.f : v;
            }
        }
        wave_lds_sync();                                          // the next pair overwrites both buffers
    }
}

template <int NFFT>
int32_t launch_fbank(const FbankArgs& a, int64_t b, hipStream_t s) {
    static DeviceOnce once;
    const size_t lds = ((size_t)4 * 2 * NFFT + NFFT / 2) * sizeof(float2) + (size_t)2 * NFFT * sizeof(float);
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)fbank_kernel<NFFT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done_on_device(once);
    }
    hipLaunchKernelGGL(fbank_kernel<NFFT>, dim3((unsigned)((a.T + 8 * FPW - 1) / (8 * FPW)), (unsigned)b), dim3(256), lds, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

extern "C" size_t vipant_fbank_workspace_bytes(int64_t b) { return (size_t)((b * 4 + 255) / 256 * 256); }

extern "C" int32_t vipant_fbank(const float* wave, int64_t wave_stride, const int64_t* nsamples, float* out, const float* window,
                                const float* banks, const int32_t* bank_start, const int32_t* bank_len, const int32_t* masks,
                                int64_t b, int64_t T, int64_t F, int32_t window_size, int32_t window_shift, int32_t nfft,
                                float preemphasis, int32_t zero_mean, float norm_mean, float norm_std, void* workspace,
                                size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(b > 0 && b < 65536 && T > 0 && F > 0 && F <= 256, VIPANT_EBADSHAPE, "fbank: bad batch b=%ld T=%ld F=%ld", (long)b, (long)T, (long)F);
    VIPANT_REQUIRE(nfft == 512 || nfft == 1024 || nfft == 2048, VIPANT_EBADSHAPE, "fbank: padded window %d not in {512, 1024, 2048}", nfft);
    VIPANT_REQUIRE(window_size > 1 && window_size <= nfft && window_size > nfft / 2 && window_shift > 0, VIPANT_EBADSHAPE,
                   "fbank: bad window %d / shift %d for padded size %d", window_size, window_shift, nfft);
    VIPANT_REQUIRE(wave != nullptr && nsamples != nullptr && out != nullptr && window != nullptr && banks != nullptr &&
                       bank_start != nullptr && bank_len != nullptr, VIPANT_EBADSHAPE, "fbank: null argument");
    VIPANT_REQUIRE(!zero_mean || (workspace != nullptr && workspace_bytes >= vipant_fbank_workspace_bytes(b)), VIPANT_ENOWORKSPACE,
                   "fbank: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* cmean = nullptr;
    if (zero_mean) {
        cmean = (float*)workspace;
        hipLaunchKernelGGL(clip_mean_kernel, dim3((unsigned)b), dim3(256), 0, s, wave, wave_stride, nsamples, cmean);
        VIPANT_LAUNCH_CHECK();
    }
    FbankArgs a{wave, wave_stride, nsamples, out, window, banks, bank_start, bank_len, masks, cmean,
                (int)T, (int)F, window_size, window_shift, preemphasis, norm_mean, norm_std};
    switch (nfft) {
        case 512: return launch_fbank<512>(a, b, s);
        case 1024: return launch_fbank<1024>(a, b, s);
        default: return launch_fbank<2048>(a, b, s);
    }
}
