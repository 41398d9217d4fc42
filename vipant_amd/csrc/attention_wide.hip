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

// the same for a fragment that only MFMAs read: keep it in the accumulator half of the register file (one wave per SIMD: 256 + 256
// registers; what the VALU touches must be an architectural VGPR, what only the matrix pipe reads need not be)
__device__ __forceinline__ void settle_a(bf16x8& f) {
    v4i32_t t = __builtin_bit_cast(v4i32_t, f);
    asm volatile("" : "+a"(t));
    f = __builtin_bit_cast(bf16x8, t);
}

template <int V> struct Int2 { static constexpr int value = V; };

// LDS fragment reads straight into the accumulator half of the register file, through inline asm with immediate offsets: the caller
// owns the lgkmcnt wait (lds_raw_wait<N>) and then passes every fragment through settle_a before its first reader.  (A compiler-
// visible load followed by an asm pin is awaited on the spot -- lgkmcnt(0) behind every read; and the transposed-read builtin
// makes hipcc wait for every LDS-DMA in flight.)
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128_a(uint32_t addr) {
    v4i32_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(r) : "v"(addr), "n"(OFF));
    return __builtin_bit_cast(bf16x8, r);
}
template <int OFF0, int OFF1>
__device__ __forceinline__ bf16x8 lds_tr_pair_v(uint32_t a0, uint32_t a1) {
    // (VGPR destinations: two 64-bit asm results coalesce into one 128-bit operand there; in the accumulator file hipcc copies them
    // through VGPRs right behind the read -- before the data has arrived)
    v2i32_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFF1));
    v4i32_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

#ifdef VIPANT_ATTN_STAMPS
__device__ unsigned long long g_attnw_stamps[64];
#define STAMP(i) do { if (prob == 3000 && lane == 0 && wave == 1) g_attnw_stamps[i] = __builtin_readcyclecounter(); } while (0)
// per-workgroup trace: {hw id | xcc id << 32, realtime at start, at "second key half landed", at end} (100 MHz ticks)
__device__ unsigned long long g_attnw_trace[8192 * 4];
#define TRACE(k) do { if (lane == 0 && wave == 0 && prob < 8192) g_attnw_trace[prob * 4 + (k)] = \
    (k) == 0 ? ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32)) \
             : __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRACE(k) do {} while (0)
#define STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------- forward
// Persistent, one workgroup of four waves per CU, one wave per SIMD with the whole register file, the 160 KiB of LDS as TWO image
// sets: while a problem is computed from one set the next problem's K / V images (80 KiB) and query fragments stream into the other,
// so the CU's load path never idles and no arithmetic ever waits for a load (measured on the two-workgroups-per-CU build before:
// the workgroups took turns loading and computing, 1-3 us of start-up per relaunched workgroup on top: tools/attnw_trace.py).
// Per wave and problem the five half-units run as a software pipeline,
//     QK(0) | PV(0) + QK(1) | PV(1) + QK(2) | PV(2) + QK(3) | PV(3) + QK(4) | PV(4),
// because on its own a QK phase is all MFMA and a PV phase all VALU (exp, scale, pack, row sums: ~4.5 issue slots per score), and with
// one wave per SIMD only program order overlaps the two pipes.  S^T tile t of the next half-unit lands in the registers tile t of
// the current one leaves.
template <int NT>                                   // 32-key tiles (even): two halves of NT / 2
__global__ __launch_bounds__(256, 1) void mha_fwd_wide_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NT * 32, HT = NT / 2, NQ = NT;          // NQ query blocks of 32 (S > (NT - 1) * 32)
    constexpr int SET = 2 * SP * 128;                           // bytes of one image set: K then V
    static_assert(NQ == 10, "the unit schedule below is written for ten query blocks over four waves");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = p.H * 64, ld = 3 * D;
    const int r = lane & 31, hh = lane >> 5;

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
            va[dt][jj] = lds_offset(smem) + (uint32_t)(SP * 128 + rl * 128 + (((dt * 4 + 2 * gsel + (pp >> 1)) ^ img32_swz(rl)) << 4) + (pp & 1) * 8);
        }

    // Everything a problem needs from HBM, requested a whole problem ahead: the query fragments of the wave's three blocks (inline asm:
    // hipcc does not count LDS-DMA pieces in its vmcnt bookkeeping, a compiler-visible load in front of them would be awaited with
    // vmcnt(0); the waits are the explicit ones below) and the two images, 20 pieces per wave.
    bf16x8 qnext[3][4];
    __amdgpu_buffer_rsrc_t rs_next;                   // K rows of the requested problem (V rows: + 2 D bytes through the scalar offset)
    auto request_q = [&](int prob) {
        const bool any = prob < p.batch * p.H;
        const int pr = any ? prob : 0;
        const int b = pr / p.H, h = pr % p.H;
        const bf16_t* base = p.qkv + (int64_t)b * p.S * ld + h * 64;
        const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64 - D) * 2;
        const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int qq = (u == 0 ? wave : u == 1 ? wave + 4 : 8 + (wave >> 1)) * 32 + r;
            const bf16_t* qp = base + (int64_t)(qq < p.S ? qq : p.S - 1) * ld + 8 * hh;
            v4i32_t t0, t1, t2, t3;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                         "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                         : "=&a"(t0), "=&a"(t1), "=&a"(t2), "=&a"(t3) : "v"(qp) : "memory");
            qnext[u][0] = __builtin_bit_cast(bf16x8, t0); qnext[u][1] = __builtin_bit_cast(bf16x8, t1);
            qnext[u][2] = __builtin_bit_cast(bf16x8, t2); qnext[u][3] = __builtin_bit_cast(bf16x8, t3);
        }
        rs_next = uniform_rsrc(base + D, any ? lim : 0u);       // no next problem: a zero-length descriptor, the pieces read zeros
    };
    // Image piece `id` (0 .. 19) of this wave: groups of five -- K rows 0..159, K rows 160..319, V rows 0..159, V rows 160..319 --
    // 8-row block (id % 5) * 4 + wave of the group.  The swizzle of a row depends on (row >> 1) & 7, i.e. on the block's parity:
    // two per-lane source offsets, everything else rides the scalar offset.  One piece per tile step of the arithmetic: issued
    // into an (almost) empty queue a piece costs ~100 cycles; issued as a burst the 20 of them block the wave for ~7 k cycles
    // while the CU moves all 80 KiB.
    uint32_t dvo[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int rr = par * 8 + (lane >> 3);
        dvo[par] = (uint32_t)((lane >> 3) * (ld * 2) + (((lane & 7) ^ img32_swz(rr)) << 4));
    }
    auto piece = [&](int id, char* set) {
        const int g = id / 5, blk = (g & 1) * (SP / 16) + (id % 5) * 4 + wave;
        lds_dma16(rs_next, set + (g >> 1) * (SP * 128) + blk * 1024, dvo[blk & 1], (uint32_t)(blk * 8 * (ld * 2) + (g >> 1) * (D * 2)));
    };

    const int nprob = p.batch * p.H;
    request_q(blockIdx.x);
    for (int id = 0; id < 20; ++id) piece(id, smem);
    int cur = 0;
    for (int prob = blockIdx.x; prob < nprob; prob += gridDim.x, cur ^= 1) {
        const int b = prob / p.H, h = prob % p.H;
        const int64_t row_base = (int64_t)b * p.S;
        char* kimg = smem + cur * SET;
        STAMP(0);
        TRACE(0); TRACE(1);
        // This problem's fragments and images were requested during the previous problem, the last piece in its fourth phase: behind
        // it in the queue are only the stores of that problem's second block and (even waves) of the shared one, five store
        // instructions each (the debug builds' stamps are stores too: they wait for everything)
#ifdef VIPANT_ATTN_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (prob == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wave & 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();                 // the images have landed; the other set (and its exchange region) is free
        bf16x8 qf[4], qn1[4], qn2[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            settle_a(qnext[0][s]); settle_a(qnext[1][s]); settle_a(qnext[2][s]);
            qf[s] = qnext[0][s]; qn1[s] = qnext[1][s]; qn2[s] = qnext[2][s];
            settle_a(qf[s]); settle_a(qn1[s]); settle_a(qn2[s]);
        }
        request_q(prob + gridDim.x);
        char* nset = smem + (cur ^ 1) * SET;
        STAMP(2);

        f32x16 o0 = zero16(), o1 = zero16();
        float m_run = -INFINITY, l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
        f32x16 sc[HT];
        float m_acc = -INFINITY;                      // running maximum of the half-unit whose QK phase is in progress

        // bf16 pairs of a d tile, 16-byte stores: lane half 0 holds d = 8 g + (0..3), half 1 d = 8 g + 4 + (0..3) (g = register
        // group); one v_permlane32_swap per dword gives half 0 the eight d of a group pair's first group, half 1 those of the second
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
        // exactly five store instructions per call whatever the data (the counted wait at the top of the loop relies on it): rows
        // >= S of the last block write a duplicate of row S - 1's address range?  no -- they are redirected to their own row S - 1
        // copy only in the ADDRESS; the value stored there is that lane's own, so the redirect must not happen: such lanes are
        // masked by the bounds of a buffer store instead
        auto finalize_store = [&](int qb, float m, float l) {
            l += __shfl_xor(l, 32, 64);               // the two lane halves hold different keys
            const int q = qb * 32 + r;
            const float inv = __builtin_amdgcn_rcpf(l);
            bf16_t* op = p.out + (row_base + (q < p.S ? q : p.S - 1)) * D + h * 64;
            if (q < p.S && p.stagger != 97) {        // (97: timing probe without the output stores)
                store_tile(o0, inv, op);
                store_tile(o1, inv, op + 32);
                if (hh == 0) p.lse[((int64_t)b * p.H + h) * p.S + q] = m * SCALE + __logf(l);
            }
        };

        // One phase = five tile steps.  QK: S^T tile t of half-unit `hu + 1` (keys of half hfq) = K_tile Q^T, four chained MFMAs, its
        // running maximum taken one step later; PV: exponentials of tile t of half-unit `hu` against mc, O^T += V^T P^T (half hfv).
        auto phase = [&](auto ph_c, auto do_qk_c, auto do_pv_c, int hfq, int hfv, float mc) {
            constexpr int PH = decltype(ph_c)::value;
            constexpr bool DO_QK = decltype(do_qk_c)::value, DO_PV = decltype(do_pv_c)::value;
            // per-phase base addresses; tiles and k-steps are immediate offsets from here on
            uint32_t kad[4], vad[2][2];
            const uint32_t kbo = lds_offset(smem) + (uint32_t)(cur * SET + hfq * (HT * 4096));
            const uint32_t vbo = (uint32_t)(cur * SET + hfv * (HT * 4096));
#pragma unroll
            for (int s = 0; s < 4; ++s) kad[s] = kbo + ka[s];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) vad[dt][jj] = vbo + va[dt][jj];
            bf16x8 kr[3][4];
            bf16x8 vr[2][2][2];                       // [ring][k-step s'][d tile]
            auto k_tile = [&](auto tc, bf16x8 (&f)[4]) {
                constexpr int t = decltype(tc)::value;
                f[0] = lds_b128_a<t * 4096>(kad[0]); f[1] = lds_b128_a<t * 4096>(kad[1]);
                f[2] = lds_b128_a<t * 4096>(kad[2]); f[3] = lds_b128_a<t * 4096>(kad[3]);
            };
            auto v_tile = [&](auto tc, bf16x8 (&f)[2][2]) {
                constexpr int t = decltype(tc)::value;
                f[0][0] = lds_tr_pair_v<t * 4096, t * 4096>(vad[0][0], vad[0][1]);
                f[0][1] = lds_tr_pair_v<t * 4096, t * 4096>(vad[1][0], vad[1][1]);
                f[1][0] = lds_tr_pair_v<t * 4096 + 2048, t * 4096 + 2048>(vad[0][0], vad[0][1]);
                f[1][1] = lds_tr_pair_v<t * 4096 + 2048, t * 4096 + 2048>(vad[1][0], vad[1][1]);
            };
            auto max8 = [&](const f32x16& a, int i0) {
#pragma unroll
                for (int i = i0; i < i0 + 8; i += 2) m_acc = fmaxf(fmaxf(m_acc, a[i]), a[i + 1]);
                asm volatile("" : "+v"(m_acc));
            };
            // LDS queue discipline: every step issues K(t + 2) [4 reads] then V(t + 1) [8 reads] and then waits until only those are
            // outstanding -- K(t), K(t + 1), V(t) are older and therefore complete (LDS returns in order)
            if (DO_QK) { m_acc = -INFINITY; k_tile(Int2<0>{}, kr[0]); k_tile(Int2<1>{}, kr[1]); }
            if (DO_PV) v_tile(Int2<0>{}, vr[0]);
            __builtin_amdgcn_sched_barrier(0);
            auto step = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if (PH * HT + t < 20 && p.stagger != 96) piece(PH * HT + t, nset);      // (96: timing probe without the image stream)
                if (DO_QK && t + 2 < HT) k_tile(Int2<(t + 2 < HT ? t + 2 : 0)>{}, kr[(t + 2) % 3]);
                if (DO_PV && t + 1 < HT) v_tile(Int2<(t + 1 < HT ? t + 1 : 0)>{}, vr[(t + 1) & 1]);
                lds_raw_wait<(DO_QK && t + 2 < HT ? 4 : 0) + (DO_PV && t + 1 < HT ? 8 : 0)>();
                f32x16 acc = zero16(), e = zero16();
                if (DO_QK) { settle_a(kr[t % 3][0]); settle_a(kr[t % 3][1]); settle_a(kr[t % 3][2]); settle_a(kr[t % 3][3]); }
                if (DO_PV) {
                    settle(vr[t & 1][0][0]); settle(vr[t & 1][0][1]); settle(vr[t & 1][1][0]); settle(vr[t & 1][1][1]);
                    e = sc[t];
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- group A: QK 1, 2 | exponentials of k-step 0 | running maximum of the previous step's tile
                if (DO_QK) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][0], qf[0], acc, 0, 0, 0);
                if (DO_PV) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i] * C2 - mc);
                }
                if (DO_QK) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][1], qf[1], acc, 0, 0, 0);
                bf16x8 pf0, pf1;
                if (DO_PV) pf0 = pack8f(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
                if (DO_QK && t > 0) max8(sc[t > 0 ? t - 1 : 0], 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- group B: PV of k-step 0 | exponentials of k-step 1 | QK 3
                if (DO_PV) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][0][0], pf0, o0, 0, 0, 0);
                if (DO_PV) {
#pragma unroll
                    for (int i = 8; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(e[i] * C2 - mc);
                }
                if (DO_PV) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][0][1], pf0, o1, 0, 0, 0);
                if (DO_PV) pf1 = pack8f(e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15]);
                if (DO_QK) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][2], qf[2], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- group C: QK 4 | row sums | PV of k-step 1 | the other half of the running maximum
                if (DO_QK) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t % 3][3], qf[3], acc, 0, 0, 0);
                if (DO_PV) {
#pragma unroll
                    for (int i = 0; i < 16; i += 4) { l0 += e[i]; l1 += e[i + 1]; l2 += e[i + 2]; l3 += e[i + 3]; }
                    asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
                }
                if (DO_PV) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][1][0], pf1, o0, 0, 0, 0);
                if (DO_QK && t > 0) max8(sc[t > 0 ? t - 1 : 0], 8);
                if (DO_PV) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t & 1][1][1], pf1, o1, 0, 0, 0);
                if (DO_QK) {
                    if (t == HT - 1) {
                        if (hfq) {                    // keys >= S live in the last tile of the second half only
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const int key = (NT - 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                                if (key >= p.S) acc[i] = -INFINITY;
                            }
                        }
                    }
                    sc[t] = acc;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            static_assert(HT == 5, "five tile steps per phase");
            step(Int2<0>{}); step(Int2<1>{}); step(Int2<2>{}); step(Int2<3>{}); step(Int2<4>{});
            if (DO_QK) { max8(sc[HT - 1], 0); max8(sc[HT - 1], 8); }
        };
        // what follows a QK phase: the half-unit's maximum, and for the second half of a whole block the choice of the reference
        auto after_qk = [&](bool first) {
            float m = fmaxf(m_acc, __shfl_xor(m_acc, 32, 64));
            if (first) {
                m_run = m;
            } else if (__any((m - m_run) * C2 > 64.f)) {
                // Second half of a whole block: its exponentials are taken against the FIRST half's maximum (exact all the same: bf16
                // and fp32 keep their relative precision at any magnitude); only when that would let them grow past 2^64 are the first
                // half's sums brought to the new maximum instead
                const float m_new = fmaxf(m_run, m);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C2);
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
                l0 *= alpha; l1 *= alpha; l2 *= alpha; l3 *= alpha;
                m_run = m_new;
            }
        };
        auto new_block = [&]() {
            o0 = zero16(); o1 = zero16();
            l0 = l1 = l2 = l3 = 0.f;
        };

        // half-units: 0 (qb = wave, half 0), 1 (wave, 1), 2 (wave + 4, 0), 3 (wave + 4, 1), 4 (8 + (wave >> 1), half = wave & 1)
        STAMP(3);
        phase(Int2<0>{}, Int2<1>{}, Int2<0>{}, 0, 0, 0.f);                         // QK(0)
        after_qk(true);
        STAMP(4);
        phase(Int2<1>{}, Int2<1>{}, Int2<1>{}, 1, 0, m_run * C2);       // PV(0) + QK(1)
        after_qk(false);
        STAMP(5);
#pragma unroll
        for (int s = 0; s < 4; ++s) { qf[s] = qn1[s]; settle_a(qf[s]); }
        phase(Int2<2>{}, Int2<1>{}, Int2<1>{}, 0, 1, m_run * C2);       // PV(1) + QK(2)
        STAMP(6);
        finalize_store(wave, m_run, (l0 + l1) + (l2 + l3));
        new_block();
        after_qk(true);
        STAMP(7);
        phase(Int2<3>{}, Int2<1>{}, Int2<1>{}, 1, 0, m_run * C2);       // PV(2) + QK(3)
        after_qk(false);
        STAMP(8);
#pragma unroll
        for (int s = 0; s < 4; ++s) { qf[s] = qn2[s]; settle_a(qf[s]); }
        phase(Int2<4>{}, Int2<1>{}, Int2<1>{}, wave & 1, 1, m_run * C2);  // PV(3) + QK(4)
        STAMP(9);
        finalize_store(wave + 4, m_run, (l0 + l1) + (l2 + l3));
        new_block();
        after_qk(true);
        STAMP(10);
        phase(Int2<5>{}, Int2<0>{}, Int2<1>{}, 0, wave & 1, m_run * C2);  // PV(4)
        STAMP(11);

        // ---- the shared block: the odd wave of a pair hands its partial (max, sum, O^T) to the even one through the dead K image
        __builtin_amdgcn_s_barrier();                 // every wave is done with this image set (LDS reads are complete: their
        float* xch = (float*)(kimg + (wave >> 1) * (34 * 256));      // results have been consumed)
        if (wave & 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { xch[i * 64 + lane] = o0[i]; xch[(16 + i) * 64 + lane] = o1[i]; }
            xch[32 * 64 + lane] = m_run;
            xch[33 * 64 + lane] = (l0 + l1) + (l2 + l3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
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
        STAMP(12);
        TRACE(3);
    }
}

template <int NT>
int32_t launch_fwd_wide_nt(const MhaArgs& a, hipStream_t s) {
    constexpr int lds = 2 * NT * 32 * 128 * 2;        // two image sets
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_fwd_wide_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
        if (getenv("VIPANT_ATTN_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)mha_fwd_wide_kernel<NT>, 256, lds);
            fprintf(stderr, "[vipant] mha_fwd_wide_kernel<%d>: %d B of LDS per workgroup, %d workgroups per CU by the occupancy query\n", NT, lds, nb);
        }
    }
    static int slots = 0;                           // one persistent workgroup per CU (all 160 KiB of LDS)
    if (!slots) {
        int dev = 0, cus = 0;
        VIPANT_HIP_TRY(hipGetDevice(&dev));
        VIPANT_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        slots = cus;
    }
    const int nprob = a.batch * a.H;
    hipLaunchKernelGGL((mha_fwd_wide_kernel<NT>), dim3(nprob < slots ? nprob : slots), dim3(256), lds, s, a);
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
