// Multi-head attention on v_mfma_f32_32x32x16_bf16 (round 4): softmax(q k^T / 8) v for the audio tower's shape, 288 < S <= 320,
// head dim 64, no mask (nn.MultiheadAttention inside ResidualAttentionBlock, cvap/module/val.py:511-517).
//
// Why the wide shape.  The 16x16x32 kernels of attention.hip are bound by instruction issue: head dim 64 makes the exp / scale /
// pack arithmetic of a score as expensive as the matrix work beside it, and beside a 16x16x32 MFMA (16 cycles) a SIMD issues ONE
// VALU instruction for free, beside a 32x32x16 MFMA (32 cycles, the same FLOP per cycle) four (profiles/r3_attention_experiments.md
// section 1).  Per FLOP the wide shape has 2.5x the VALU room and half the LDS read instructions.
//
// Forward.  One workgroup of four waves per (batch, head), two workgroups per CU, K and V resident in LDS (2 x 40 KiB at S <= 320) as
// before.  The work unit is (32 queries) x (160 keys = half of the keys): S^T = K Q^T with the QUERY on the MFMA column, so that a
// lane holds 16 of the 32 keys of a tile for one query (row max / sum: registers + one lane ^ 32 exchange) and the exponentiated
// accumulator is, packed to bf16, already the B operand of O^T = V^T P^T (registers 8s .. 8s+7 = k-step s; the k order this implies
// -- element j of lane half h = key 16s + 8(j >> 2) + 4h + (j & 3) -- is what the transposed reads of V fetch).  Ten query blocks do
// not split over four waves; twenty half-units do: a wave takes two query blocks whole (both key halves in turn, the second half
// joining the first by one online-softmax rescale of the 32 output registers) and one half of a shared block, whose two partial
// results (max, sum, O^T) are merged through LDS once the images are dead -- 2.5 blocks per wave, no idle wave.
// LDS images: 128-byte rows, the 16-byte chunk index XOR-ed with rot3((row >> 1) & 7): conflict-free both for the 32-row
// ds_read_b128 fragments of K (the 16-lane service groups of a b128 read see eight rows of each parity, which must land in eight
// different chunks) and for the transposed reads of V (a 32-lane half reads 4 rows x 64 B: rows r and r + 2 must take different
// 64-byte halves of their row).
#include "attn_common.h"

using namespace vipant_attn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int img32_swz(int r) {
    const int x = (r >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}

// 8-row blocks blk0 .. blk0 + nblk - 1 of an image, block i by wave i & 3 (nblk = 20 at S <= 320: exactly five pieces per wave)
__device__ __forceinline__ void dma_rows32(char* lds, __amdgpu_buffer_rsrc_t rs, uint32_t ld_bytes, int blk0, int nblk, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int blk = blk0 + wave + 4 * i;
        const int r = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ img32_swz(r);
        if (wave + 4 * i < nblk) lds_dma16(rs, lds + blk * 1024, (uint32_t)r * ld_bytes + (uint32_t)c * 16, 0);
    }
}

__device__ __forceinline__ bf16x8 pack8f(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    bf16x2 p0, p1, p2, p3;
    p0[0] = (bf16_t)a0; p0[1] = (bf16_t)a1; p1[0] = (bf16_t)a2; p1[1] = (bf16_t)a3;
    p2[0] = (bf16_t)a4; p2[1] = (bf16_t)a5; p3[0] = (bf16_t)a6; p3[1] = (bf16_t)a7;
    u32x4 r;
    r[0] = __builtin_bit_cast(uint32_t, p0); r[1] = __builtin_bit_cast(uint32_t, p1);
    r[2] = __builtin_bit_cast(uint32_t, p2); r[3] = __builtin_bit_cast(uint32_t, p3);
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ void settle(bf16x8& f) { lds_raw_use(f); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

#ifdef VIPANT_ATTN_STAMPS
__device__ unsigned long long g_attnw_stamps[64];
#define STAMP(i) do { if (blockIdx.x == 3000 && lane == 0 && wave == 1) g_attnw_stamps[i] = __builtin_readcyclecounter(); } while (0)
// per-workgroup trace: {hw id | xcc id << 32, realtime at start, at "second key half landed", at end} (100 MHz ticks)
__device__ unsigned long long g_attnw_trace[8192 * 4];
#define TRACE(k) do { if (lane == 0 && wave == 0 && blockIdx.x < 8192) g_attnw_trace[blockIdx.x * 4 + (k)] = \
    (k) == 0 ? ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32)) \
             : __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRACE(k) do {} while (0)
#define STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------- forward
template <int NT>                                   // 32-key tiles (even): two halves of NT / 2
__global__ __launch_bounds__(256, 2) void mha_fwd_wide_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NT * 32, HT = NT / 2, NQ = NT;          // NQ query blocks of 32 (S > (NT - 1) * 32)
    static_assert(NQ == 10, "the unit schedule below is written for ten query blocks over four waves");
    char* kimg = smem;
    char* vimg = smem + SP * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;

    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    // Phase stagger (first generation only): the two workgroups a CU holds start in the same microsecond, load their images together
    // (sharing the CU's load path) and then compute together (sharing its SIMDs) -- and since they also finish together, so does every
    // later pair.  Holding back the second workgroup of each CU once puts one workgroup's load phase under the other's arithmetic
    // for the rest of the launch.
    if (p.stagger > 0 && p.stagger < 90 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const int r = lane & 31, hh = lane >> 5;
    // The query fragments of the wave's first two blocks go out FIRST: a CU's vector-memory queue is in order and shared by its two
    // workgroups, so whatever is requested behind an image burst (80 KiB, ~7 k cycles at the CU's ~11 B/clk) waits for all of it
    // (inline asm: hipcc does not count LDS-DMA pieces in its vmcnt bookkeeping, so a compiler-visible load issued in front of
    // the pieces is awaited with vmcnt(0) -- all pieces; the waits for these fragments are the explicit ones below)
    auto load_q = [&](int qb, bf16x8 (&f)[4]) {
        const int qq = qb * 32 + r;
        const bf16_t* qp = base + (int64_t)(qq < p.S ? qq : p.S - 1) * ld + 8 * hh;
        v4i32_t t0, t1, t2, t3;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(qp) : "memory");
        f[0] = __builtin_bit_cast(bf16x8, t0); f[1] = __builtin_bit_cast(bf16x8, t1);
        f[2] = __builtin_bit_cast(bf16x8, t2); f[3] = __builtin_bit_cast(bf16x8, t3);
    };
    bf16x8 qf[4], qn1[4], qn2[4];
    load_q(wave, qf);
    load_q(wave + 4, qn1);
    STAMP(0);
    TRACE(0); TRACE(1);
    // the images in key halves -- K rows 0..159, V rows 0..159, then the second halves -- five pieces per wave each: the first
    // half-unit starts when the first two groups have landed, the rest flies under it
    {
        const __amdgpu_buffer_rsrc_t rk = uniform_rsrc(base + D, lim > (uint32_t)(D * 2) ? lim - D * 2 : 0);
        const __amdgpu_buffer_rsrc_t rv = uniform_rsrc(base + 2 * D, lim > (uint32_t)(D * 4) ? lim - D * 4 : 0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            dma_rows32(kimg, rk, ld * 2, half * (SP / 16), SP / 16, wave, lane);
            dma_rows32(vimg, rv, ld * 2, half * (SP / 16), SP / 16, wave, lane);
        }
    }

    // K row fragments (A operand: key tile row r, d = 16 s + 8 hh ..): one address per k-step, tiles by immediate offsets
    uint32_t ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = (uint32_t)(r * 128 + (((2 * s + hh) ^ img32_swz(r)) << 4));
    // V^T fragments (A operand: d = 32 dt + r, k-slot j of lane half hh = key 16 s' + 8 (j >> 2) + 4 hh + (j & 3)): lane 4 q4 + pp of a
    // 16-lane group supplies row q4, columns 4 pp .. of the group's 4-row x 16-column block
    const int q4 = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
    uint32_t va[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int rl = 8 * jj + 4 * hh + q4;
            va[dt][jj] = (uint32_t)(rl * 128 + (((dt * 4 + 2 * gsel + (pp >> 1)) ^ img32_swz(rl)) << 4) + (pp & 1) * 8);
        }

    STAMP(1);
    // queue of this wave: 8 query loads, 10 pieces of the first key half, 10 of the second
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // K, V rows 0 .. 159 have landed
    // (the compiler's own wait bookkeeping must not tie the query fragments to the pieces still in flight)
#pragma unroll
    for (int s = 0; s < 4; ++s) { settle(qf[s]); settle(qn1[s]); }
    STAMP(2);

    // O^T (two d tiles) and the row sums: fp32 adds of the unrounded exponentials, four independent chains pinned inside their tile
    // (left to itself hipcc sinks all 80 adds of a half-unit behind the last MFMA as ONE dependent chain).  [Measured and not kept:
    // the sums as a third MFMA product with an all-ones A operand -- no VALU adds, but it sums the bf16-ROUNDED P, and the
    // log-sum-exp the backward recomputes P from was then off by up to 3e-3.]
    f32x16 o0 = zero16(), o1 = zero16();
    float m_run = -INFINITY, l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;

    // bf16 pairs of a d tile, 16-byte stores: lane half 0 holds d = 8 g + (0..3), half 1 d = 8 g + 4 + (0..3) (g = register group);
    // one v_permlane32_swap per dword gives half 0 the eight d of an even group pair's first group, half 1 those of the second
    auto store_tile = [&](const f32x16& o, float inv, bf16_t* op) {
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            uint32_t w[2][2];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int i0 = 4 * (2 * G + gg);
                const bf16x4 v = f32x4_to_bf16x4(f32x4{o[i0] * inv, o[i0 + 1] * inv, o[i0 + 2] * inv, o[i0 + 3] * inv});
                const u32x2 t = __builtin_bit_cast(u32x2, v);
                w[gg][0] = t[0]; w[gg][1] = t[1];
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
            *(u32x4*)(op + 16 * G + 8 * hh) = u32x4{(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
        }
    };
    auto finalize_store = [&](int qb, float m, float l) {
        l += __shfl_xor(l, 32, 64);                                                 // the two lane halves hold different keys
        const int q = qb * 32 + r;
        const float inv = __builtin_amdgcn_rcpf(l);
        bf16_t* op = p.out + (row_base + (q < p.S ? q : p.S - 1)) * D + h * 64;    // rows >= S: a harmless duplicate of row S - 1's lanes is
        if (q < p.S) {                                                              // never stored (the branch is on the store only)
            store_tile(o0, inv, op);
            store_tile(o1, inv, op + 32);
            if (hh == 0) p.lse[((int64_t)b * p.H + h) * p.S + q] = m * SCALE + __logf(l);
        }
    };

    // five half-units per wave: (qb = wave, half 0), (wave, 1), (wave + 4, 0), (wave + 4, 1), (8 + (wave >> 1), half = wave & 1)
    const int nit = p.stagger == 99 ? 0 : (p.stagger == 98 ? 1 : 5);      // timing probes (results wrong): loads only / one half-unit
    if (nit < 5) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    for (int it = 0; it < nit; ++it) {
        STAMP(3 + 3 * it);
        const int hf = it < 4 ? (it & 1) : (wave & 1);
        const bool first = (it & 1) == 0;
        if (it == 1) {                                // the second key half (and the last block's queries behind it in the queue)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            TRACE(2);
            settle(qn2[0]); settle(qn2[1]); settle(qn2[2]); settle(qn2[3]);
        }
        if (first) {
            if (it == 0) load_q(8 + (wave >> 1), qn2);        // needed four half-units from now; awaited with the second key half
            if (it == 2) {
#pragma unroll
                for (int s = 0; s < 4; ++s) qf[s] = qn1[s];
            }
            if (it == 4) {
#pragma unroll
                for (int s = 0; s < 4; ++s) qf[s] = qn2[s];
            }
            o0 = zero16(); o1 = zero16();
            l0 = l1 = l2 = l3 = 0.f;
            m_run = -INFINITY;
        }
        const char* kb = kimg + hf * (HT * 4096);
        const char* vb = vimg + hf * (HT * 4096);

        // ---- S^T tiles of this half: five tiles x four k-steps; K fragments two tiles ahead through a register ring; the running
        // maximum of tile t - 1 is taken beside the MFMAs of tile t
        f32x16 sc[HT];
        bf16x8 kr[3][4];
        auto k_tile = [&](int t, bf16x8 (&f)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) f[s] = *(const bf16x8*)(kb + ka[s] + t * 4096);
        };
        k_tile(0, kr[0]);
        k_tile(1, kr[1]);
        __builtin_amdgcn_sched_barrier(0);
        float m = -INFINITY;
        auto max8 = [&](const f32x16& a, int i0) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i += 2) m = fmaxf(fmaxf(m, a[i]), a[i + 1]);
            asm volatile("" : "+v"(m));
        };
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            if (t + 2 < HT) k_tile(t + 2, kr[(t + 2) % 3]);
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][0], qf[0], zero16(), 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) max8(sc[t - 1], 0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][1], qf[1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) max8(sc[t - 1], 8);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][2], qf[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][3], qf[3], acc, 0, 0, 0);
            if (t == HT - 1) {
                if (hf) {                             // keys >= S live in the last tile of the second half only
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = (NT - 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (key >= p.S) acc[i] = -INFINITY;
                    }
                }
            }
            sc[t] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
        max8(sc[HT - 1], 0);
        max8(sc[HT - 1], 8);
        STAMP(4 + 3 * it);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (first) {
            m_run = m;
        } else if (__any((m - m_run) * C2 > 64.f)) {
            // second half of a whole block.  Its exponentials are taken against the FIRST half's maximum (exact all the same: bf16
            // and fp32 keep their relative precision at any magnitude); only when that would let them grow past 2^64 are the
            // first half's sums brought to the new maximum instead
            const float m_new = fmaxf(m_run, m);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C2);
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            l0 *= alpha; l1 *= alpha; l2 *= alpha; l3 *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * C2;

        // ---- exponentials + O^T += V^T P^T (+ the row sums), tile by tile; V^T fragments one tile ahead
        bf16x8 vr[2][2][2];                           // [ring][k-step s'][d tile]
        // (transposed reads through inline asm: the builtin makes hipcc wait for every LDS-DMA in flight first -- the second key half
        // at it = 0; the caller owns the lgkmcnt wait: eight reads per tile, the tile in use is the older eight)
        const uint32_t vbo = lds_offset(vb);
        auto v_tile = [&](int t, bf16x8 (&f)[2][2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    f[s][dt] = lds_read_tr16_pair_raw(vbo + va[dt][0] + t * 4096 + s * 2048, vbo + va[dt][1] + t * 4096 + s * 2048);
        };
        auto v_use = [&](bf16x8 (&f)[2][2]) {
            lds_raw_use(f[0][0]); lds_raw_use(f[0][1]); lds_raw_use(f[1][0]); lds_raw_use(f[1][1]);
        };
        v_tile(0, vr[0]);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            if (t + 1 < HT) { v_tile(t + 1, vr[(t + 1) & 1]); lds_raw_wait<8>(); } else { lds_raw_wait<0>(); }
            v_use(vr[t & 1]);
            f32x16 e = sc[t];
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(e[i] * C2 - mc);
            const bf16x8 pf0 = pack8f(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            const bf16x8 pf1 = pack8f(e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15]);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][0][0], pf0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][0][1], pf0, o1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; i += 4) { l0 += e[i]; l1 += e[i + 1]; l2 += e[i + 2]; l3 += e[i + 3]; }
            asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][1][0], pf1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][1][1], pf1, o1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(5 + 3 * it);
        if (!first && it < 4) {
            // the last block's query fragments were requested two half-units ago: wait for them HERE, in front of the first stores
            // (vmcnt counts loads and stores in one in-order queue; a wait behind the stores would also wait for their acknowledgements)
            finalize_store(it == 1 ? wave : wave + 4, m_run, (l0 + l1) + (l2 + l3));
        }
    }

    STAMP(18);
    // ---- the shared block: the odd wave of a pair hands its partial (max, sum, O^T) to the even one through the dead K image
    __syncthreads();                                  // every wave is done with both images
    float* xch = (float*)(kimg + (wave >> 1) * (34 * 256));
    if (wave & 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { xch[i * 64 + lane] = o0[i]; xch[(16 + i) * 64 + lane] = o1[i]; }
        xch[32 * 64 + lane] = m_run;
        xch[33 * 64 + lane] = (l0 + l1) + (l2 + l3);
    }
    __syncthreads();
    if (!(wave & 1)) {
        const float m_b = xch[32 * 64 + lane], l_b = xch[33 * 64 + lane];
        const float m_new = fmaxf(m_run, m_b);
        const float aa = __builtin_amdgcn_exp2f((m_run - m_new) * C2), ab = __builtin_amdgcn_exp2f((m_b - m_new) * C2);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] = o0[i] * aa + xch[i * 64 + lane] * ab;
            o1[i] = o1[i] * aa + xch[(16 + i) * 64 + lane] * ab;
        }
        finalize_store(8 + (wave >> 1), m_new, ((l0 + l1) + (l2 + l3)) * aa + l_b * ab);
    }
    STAMP(19);
    TRACE(3);
}

template <int NT>
int32_t launch_fwd_wide_nt(const MhaArgs& a, hipStream_t s) {
    constexpr int lds = NT * 32 * 128 * 2;
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_fwd_wide_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
        if (getenv("VIPANT_ATTN_DEBUG")) {
            int nb = -1;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)mha_fwd_wide_kernel<NT>, 256, lds);
            fprintf(stderr, "[vipant] mha_fwd_wide_kernel<%d>: %d B of LDS per workgroup, %d workgroups per CU by the occupancy query\n", NT, lds, nb);
        }
    }
    hipLaunchKernelGGL((mha_fwd_wide_kernel<NT>), dim3(a.batch * a.H), dim3(256), lds, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

namespace vipant_attn {

int32_t launch_fwd_wide(const MhaArgs& a, hipStream_t s) {
    VIPANT_REQUIRE(a.S > 288 && a.S <= 320, VIPANT_EBADSHAPE, "mha (wide forward): 288 < S <= 320 expected, got %d", a.S);
    return launch_fwd_wide_nt<10>(a, s);
}

}  // namespace vipant_attn

#ifdef VIPANT_ATTN_STAMPS
extern "C" int32_t vipant_debug_attnw_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnw_trace), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int32_t vipant_debug_attnw_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnw_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
