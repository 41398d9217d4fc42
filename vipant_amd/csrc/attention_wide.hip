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
__device__ unsigned long long g_attnw_stamps[128];
#define STAMP(i) do { if (prob == 3000 && lane == 0 && wave == 1) g_attnw_stamps[i] = __builtin_readcyclecounter(); } while (0)
// per-workgroup trace: {hw id | xcc id << 32, realtime at start, at "second key half landed", at end} (100 MHz ticks)
__device__ unsigned long long g_attnw_trace[8192 * 4];
#define TRACE(k) do { if (lane == 0 && wave == 0 && prob < 8192) g_attnw_trace[prob * 4 + (k)] = \
    (k) == 0 ? ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32)) \
             : __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRACE(k) do {} while (0)
#endif
#ifdef VIPANT_ATTN_STAMPS
// backward: waves 1 (three key blocks) and 3 (two key blocks + dQ) of problem 3000, 32 stamps each
#define BSTAMP(i) do { if (prob == 3000 && lane == 0 && (wave & 1)) g_attnw_stamps[64 + 32 * (wave >> 1) + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BSTAMP(i) do {} while (0)
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
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_fwd_wide_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
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


// An MFMA whose accumulator lives in the ACCUMULATOR half of the register file, through inline asm.  The file is compiled with
// -amdgpu-mfma-vgpr-form (the scores must land where the VALU reads them); with that flag hipcc also gives the long-lived dK / dV
// accumulators VGPR-form MFMAs and, since 192 of them do not fit beside the working set, copies each one out of and back into the
// accumulator file around every update (400 v_accvgpr moves per step).  A and B operands: VGPRs.
__device__ __forceinline__ void mfma_acc_a(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// ------------------------------------------------------------------------------------------- backward
// Single pass, streamed operands (the scaffolding of mha_bwd1s_kernel in attention.hip: persistent workgroup, Q / dO rows through a
// four-stage ring, two K images, per-step delta, everything requested as LDS-DMA pieces three steps ahead), with the five
// contractions on v_mfma_f32_32x32x16_bf16.  Key on the MFMA column, 32-key blocks:
//     S = Q K^T, dP = dO V^T - delta      (A = the step's Q / dO rows, B = the block's K / V rows; rows = queries in registers)
//     dV^T += dO^T P, dK^T += Q^T dS      (the packed accumulators ARE the B operands; A = transposed reads of the step's rows)
//     dQ^T += K^T dS^T                    (sums over keys = lanes: dS crosses LDS once, [key][32 queries] bf16)
// Ten key blocks do not split over four waves: waves 0, 1 take three (192 accumulator registers), waves 2, 3 take two and the dQ^T
// product of the previous step, one 32-row d tile each (20 MFMAs against 16 per key block; their K^T fragments stay in registers).
// Per score the VALU work is what it was (scale, exp, multiply, two packs); beside a 32x32x16 MFMA four to five of those
// instructions issue for free instead of one.
struct BwdProb { const bf16_t* q; const bf16_t* dO; const bf16_t* o; const float* lse; uint32_t limq, limdo; uint32_t any; };

template <int NB, bool HAS_DQ>
__device__ __forceinline__ void bwd_wide_body(const MhaArgs& p, char* smem, const int wave, const int lane0, const int kb0) {
    constexpr int SP = 320, NU = 10, KIMG = SP * 128, XB = SP * 64, STG = 8192;
    char* const xbuf = smem + 2 * KIMG;
    char* const ring = xbuf + 2 * XB;
    float* const sdel = (float*)(ring + 4 * STG);      // [2][32]
    char* const obuf = ring + 4 * STG + 256;           // [4 waves][8 rows x 128 B]
    char* const lbuf = obuf + 4096;                    // [4 waves][64 floats]
    const int D = p.H * 64, ld = 3 * D;
    const int nprob = p.batch * p.H;
    const int dtl = wave & 1;                          // the d tile of a light wave's dQ^T product

    auto make_prob = [&](int pr) {
        BwdProb t;
        t.any = pr < nprob ? 1u : 0u;
        const int pq = t.any ? pr : 0;
        const int b = pq / p.H, h = pq % p.H;
        t.q = p.qkv + (int64_t)b * p.S * ld + h * 64;
        t.dO = p.dout + (int64_t)b * p.S * D + h * 64;
        t.o = p.out + (int64_t)b * p.S * D + h * 64;
        t.lse = p.lse + (int64_t)pq * p.S;
        const int64_t rq = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2, rd = ((int64_t)(p.batch - b) * p.S * D - h * 64) * 2;
        t.limq = t.any ? (uint32_t)(rq > 0xFFFFFFFFll ? 0xFFFFFFFFll : rq) : 0u;
        t.limdo = t.any ? (uint32_t)(rd > 0xFFFFFFFFll ? 0xFFFFFFFFll : rd) : 0u;
        return t;
    };
    // per-lane constants of the requests (the swizzle of an 8-row block depends on the block's parity; the blocks this wave brings
    // -- 4 i + wave of an image, block wave of a stage -- all have parity wave & 1)
    struct LaneK { uint32_t vq, vdo, vo, vl; };
    auto make_lanek = [&](int lane) {
        LaneK k;
        const int r8 = lane >> 3, c = (lane & 7) ^ img32_swz((wave & 1) * 8 + r8);
        k.vq = (uint32_t)(r8 * (ld * 2) + c * 16);
        k.vdo = (uint32_t)(r8 * (D * 2) + c * 16);
        k.vo = (uint32_t)(r8 * (D * 2) + (lane & 7) * 16);
        k.vl = (uint32_t)(lane * 4);
        return k;
    };
    auto stage_pieces = [&](const BwdProb& t, int v, int slot, const LaneK& k) {
        char* st = ring + slot * STG;
        lds_dma16_asm(uniform_rsrc(t.q, t.limq), st + wave * 1024, k.vq, (uint32_t)((32 * v + 8 * wave) * (ld * 2)));
        lds_dma16_asm(uniform_rsrc(t.dO, t.limdo), st + 4096 + wave * 1024, k.vdo, (uint32_t)((32 * v + 8 * wave) * (D * 2)));
    };
    auto k_piece = [&](const BwdProb& t, int i, char* img, const LaneK& k) {
        const int blk = 4 * i + wave;
        lds_dma16_asm(uniform_rsrc(t.q, t.limq), img + blk * 1024, k.vq, (uint32_t)(blk * 8 * (ld * 2) + D * 2));
    };
    auto o_piece = [&](const BwdProb& t, int v, const LaneK& k) {
        lds_dma16_asm(uniform_rsrc(t.o, t.any ? (uint32_t)(((int64_t)(p.S - 1) * D + 64) * 2) : 0u), obuf + wave * 1024, k.vo,
                      (uint32_t)((32 * v + 8 * wave) * (D * 2)));
    };
    auto lse_piece = [&](const BwdProb& t, int v, const LaneK& k) {
        lds_dma4_asm(uniform_rsrc(t.lse, t.any ? (uint32_t)p.S * 4u : 0u), lbuf + wave * 256, k.vl, (uint32_t)(v * 128));
    };
    auto delta_step = [&](int slot, int dbuf, int lane) {
        const int row = wave * 8 + (lane >> 3), c0 = lane & 7;
        const v4i32_t dw = *(const v4i32_t*)(ring + slot * STG + 4096 + row * 128 + ((c0 ^ img32_swz(row)) << 4));
        const v4i32_t ow = *(const v4i32_t*)(obuf + wave * 1024 + lane * 16);
        const bf16x8 d0 = __builtin_bit_cast(bf16x8, dw), o0 = __builtin_bit_cast(bf16x8, ow);
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sacc += (float)d0[e] * (float)o0[e];
        sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0xB1, 0xF, 0xF, true));
        sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x4E, 0xF, 0xF, true));
        sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x141, 0xF, 0xF, true));
        if (c0 == 0) sdel[dbuf * 32 + row] = -sacc;     // the consumer wants -delta (initial accumulator of dP)
    };
    // The V rows travel like the K rows: as an image, one piece per step, into the K image this problem no longer reads (its K / K^T
    // fragments are taken into registers at the problem switch); the next problem takes its V fragments from there.  (Loading them
    // straight into registers with inline asm -- the wait comes a problem later -- is not safe: with every register in use the
    // allocator moves a "loaded" fragment before its data has arrived.)
    auto v_piece = [&](const BwdProb& t, int i, char* img, const LaneK& k) {
        const int blk = 4 * i + wave;
        lds_dma16_asm(uniform_rsrc(t.q, t.limq), img + blk * 1024, k.vq, (uint32_t)(blk * 8 * (ld * 2) + 2 * D * 2));
    };

    // ---- preamble: the first problem's K image, its first three stages, its V fragments, lse of step 0, O of steps 0 and 1
    int prob = blockIdx.x;
    int gs = 0, cur = 0;
    BwdProb pc = make_prob(prob);
    {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const LaneK lk = make_lanek(lane);
        for (int i = 0; i < 10; ++i) k_piece(pc, i, smem, lk);
        for (int i = 0; i < 10; ++i) v_piece(pc, i, smem + KIMG, lk);
        stage_pieces(pc, 0, 0, lk);
        stage_pieces(pc, 1, 1, lk);
        stage_pieces(pc, 2, 2, lk);
        lse_piece(pc, 0, lk);
        o_piece(pc, 0, lk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        delta_step(0, 0, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        o_piece(pc, 1, lk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (; prob < nprob; prob += gridDim.x, cur ^= 1) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int tid = wave * 64 + lane;
        const int r = lane & 31, hh = lane >> 5;
        const LaneK lk = make_lanek(lane);
        const BwdProb pn = make_prob(prob + gridDim.x);
        char* const kimg = smem + cur * KIMG;
        char* const knext = smem + (cur ^ 1) * KIMG;

        BSTAMP(0);
        // ---- problem switch: the K image landed during the previous problem
        for (int i = p.S * 8 + tid; i < SP * 8; i += 256) *(u32x4*)(kimg + i * 16) = u32x4{0u, 0u, 0u, 0u};      // keys >= S
        __syncthreads();                               // ... and delta of step 0 is visible
        // per-lane LDS offsets: row fragment (row r, d = 16 s + 8 hh ..) and transposed fragment (see the forward)
        uint32_t ka[4], va[2][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) ka[s] = (uint32_t)(r * 128 + (((2 * s + hh) ^ img32_swz(r)) << 4));
        const int q4 = (lane >> 2) & 3, pp = lane & 3, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int rl = 8 * jj + 4 * hh + q4;
                va[dt][jj] = (uint32_t)(rl * 128 + (((dt * 4 + 2 * gsel + (pp >> 1)) ^ img32_swz(rl)) << 4) + (pp & 1) * 8);
            }
        // K and V rows of this wave's key blocks as B-operand fragments, resident; keys >= S: zero K rows, finite V rows
        bf16x8 kf[NB][4], vf[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kf[j][s] = *(const bf16x8*)(kimg + (kb0 + j) * 4096 + ka[s]);
                vf[j][s] = *(const bf16x8*)(knext + (kb0 + j) * 4096 + ka[s]);
            }
        // light waves: K^T of their d tile for all 20 key steps (A operand of the dQ^T product: row d = 32 dtl + r, k-slot j of lane
        // half hh = key 16 s + 8 hh + j), resident
        constexpr int NKT = HAS_DQ ? 20 : 1;
        bf16x8 kT[NKT];
        if (HAS_DQ) {
#pragma unroll
            for (int s = 0; s < NKT; ++s) {
                bf16x4 part[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int rl = 8 * hh + 4 * jj + q4;                      // key row inside the 16-key step (the swizzle sees only this)
                    part[jj] = lds_read_tr16(kimg + s * 2048 + rl * 128 + (((dtl * 4 + 2 * gsel + (pp >> 1)) ^ img32_swz(rl)) << 4) + (pp & 1) * 8);
                }
                bf16x8 v;
                v[0] = part[0][0]; v[1] = part[0][1]; v[2] = part[0][2]; v[3] = part[0][3];
                v[4] = part[1][0]; v[5] = part[1][1]; v[6] = part[1][2]; v[7] = part[1][3];
                kT[s] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) { settle_a(kf[j][s]); settle_a(vf[j][s]); }
        if (HAS_DQ) {
#pragma unroll
            for (int s = 0; s < NKT; ++s) settle_a(kT[s]);
        }
        __builtin_amdgcn_s_barrier();                  // both images are in registers everywhere: the next problem's pieces may land

        bf16_t* const dq_base = p.dqkv + (pc.q - p.qkv);
        const __amdgpu_buffer_rsrc_t rs_dq = uniform_rsrc(dq_base, (uint32_t)(((int64_t)(p.S - 1) * ld + 64) * 2));
        // exchange buffer X[key][32 queries] bf16, 64-byte rows, the 8-byte slot index XOR-ed with (key >> 2) & 7.  Write: this
        // lane's key row, slot of query group g' (queries 8 g' + 4 hh ..) = 2 g' + hh.
        const uint32_t fxw = (uint32_t)((r >> 2) & 7);
        uint32_t xwo[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) xwo[gq] = (uint32_t)((kb0 * 32 + r) * 64) + ((((uint32_t)(2 * gq + hh)) ^ fxw) << 3);
        // Read (light waves; B operand: k-slot j of lane half hh = key 16 s + 8 hh + j, column = query r): transposed reads of rows
        // 16 s + 8 hh + 4 jj + q4, slot 4 gsel + pp; the row's XOR term is (4 (s & 1) + 2 hh + jj)
        uint32_t xro[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
                xro[jj][sp] = (uint32_t)((8 * hh + 4 * jj + q4) * 64) + ((((uint32_t)(4 * gsel + pp)) ^ (uint32_t)(4 * sp + 2 * hh + jj)) << 3);

        f32x16 dkT[NB][2], dvT[NB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { dkT[j][dt] = zero16(); dvT[j][dt] = zero16(); }

        // row constants of a step, per accumulator register i <-> query (i & 3) + 8 (i >> 2) + 4 hh: -lse * log2e (queries >= S:
        // -inf) and -delta
        f32x16 nl, ndl;
        auto make_stats = [&](int u, int dbuf) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 lv = *(const f32x4*)(lbuf + wave * 256 + (8 * gq + 4 * hh) * 4);
                const f32x4 dv4 = *(const f32x4*)(sdel + dbuf * 32 + 8 * gq + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    nl[4 * gq + e] = -lv[e] * LOG2E;
                    ndl[4 * gq + e] = dv4[e];
                }
            }
            if (u == NU - 1) {                         // queries >= S exist in the last step only (S > 288)
                const int qlim = p.S - u * 32 - 4 * hh;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (8 * (i >> 2) + (i & 3) >= qlim) nl[i] = -INFINITY;
            }
        };
        make_stats(0, gs & 1);

        f32x16 dqa = zero16();
        BSTAMP(1);
        for (int u = 0; u < NU; ++u, ++gs) {
            const char* st = ring + (gs & 3) * STG;
            char* const xw = xbuf + (u & 1) * XB;
            const char* const xr = xbuf + ((u & 1) ^ 1) * XB;         // X of step u - 1
            const BwdProb& pr1 = u + 1 < NU ? pc : pn; const int v1 = u + 1 < NU ? u + 1 : u + 1 - NU;
            const BwdProb& pr2 = u + 2 < NU ? pc : pn; const int v2 = u + 2 < NU ? u + 2 : u + 2 - NU;
            const BwdProb& pr3 = u + 3 < NU ? pc : pn; const int v3 = u + 3 < NU ? u + 3 : u + 3 - NU;
            // delta of the NEXT step; then this step's requests, oldest first: lse of the next step, O of the step after next,
            // [three image pieces], [the dQ stores]
            delta_step((gs + 1) & 3, (gs + 1) & 1, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lse_piece(pr1, v1, lk);
            o_piece(pr2, v2, lk);
            // this step's operand fragments
            bf16x8 qrow[4], drow[4], qT[2][2], dT[2][2];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qrow[s] = *(const bf16x8*)(st + ka[s]);
                drow[s] = *(const bf16x8*)(st + 4096 + ka[s]);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x4 qlo = lds_read_tr16(st + va[dt][0] + s2 * 2048), qhi = lds_read_tr16(st + va[dt][1] + s2 * 2048);
                    const bf16x4 dlo = lds_read_tr16(st + 4096 + va[dt][0] + s2 * 2048), dhi = lds_read_tr16(st + 4096 + va[dt][1] + s2 * 2048);
                    bf16x8 a, b;
                    a[0] = qlo[0]; a[1] = qlo[1]; a[2] = qlo[2]; a[3] = qlo[3]; a[4] = qhi[0]; a[5] = qhi[1]; a[6] = qhi[2]; a[7] = qhi[3];
                    b[0] = dlo[0]; b[1] = dlo[1]; b[2] = dlo[2]; b[3] = dlo[3]; b[4] = dhi[0]; b[5] = dhi[1]; b[6] = dhi[2]; b[7] = dhi[3];
                    qT[dt][s2] = a; dT[dt][s2] = b;
                }
            stage_pieces(pr3, v3, (gs + 3) & 3, lk);
            k_piece(pn, u, knext, lk);
            v_piece(pn, u, kimg, lk);
            if (u == 5) BSTAMP(20);

            // ---- the key blocks, software-pipelined: region j = { S / dP of block j + 1, dV / dK of block j - 1 } beside the
            // arithmetic of block j
            f32x16 sa[2], dp[2];
            bf16x8 pf[2][2], dsf[2][2];               // [parity of the block][k-step]
            auto qk = [&](auto jc) {
                constexpr int j = decltype(jc)::value, par = j & 1;
                sa[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qrow[0], kf[j][0], zero16(), 0, 0, 0);
                dp[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(drow[0], vf[j][0], ndl, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    sa[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qrow[s], kf[j][s], sa[par], 0, 0, 0);
                    dp[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(drow[s], vf[j][s], dp[par], 0, 0, 0);
                }
            };
            auto arith = [&](auto jc) {
                constexpr int j = decltype(jc)::value, par = j & 1;
                f32x16 pe, de;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    pe[i] = __builtin_amdgcn_exp2f(sa[par][i] * C2 + nl[i]);
                    de[i] = pe[i] * dp[par][i];        // dp already holds dP - delta; the 1/sqrt(d) factor goes to dq / dk
                }
                pf[par][0] = pack8f(pe[0], pe[1], pe[2], pe[3], pe[4], pe[5], pe[6], pe[7]);
                pf[par][1] = pack8f(pe[8], pe[9], pe[10], pe[11], pe[12], pe[13], pe[14], pe[15]);
                dsf[par][0] = pack8f(de[0], de[1], de[2], de[3], de[4], de[5], de[6], de[7]);
                dsf[par][1] = pack8f(de[8], de[9], de[10], de[11], de[12], de[13], de[14], de[15]);
                // dS to the exchange buffer: group g' = 2 s' + jj is dwords 2 jj, 2 jj + 1 of k-step s'
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const u32x4 dw = __builtin_bit_cast(u32x4, dsf[par][s2]);
                    *(u32x2*)(xw + j * 2048 + xwo[2 * s2]) = u32x2{dw[0], dw[1]};
                    *(u32x2*)(xw + j * 2048 + xwo[2 * s2 + 1]) = u32x2{dw[2], dw[3]};
                }
            };
            auto dvdk = [&](auto jc) {
                constexpr int j = decltype(jc)::value, par = j & 1;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        mfma_acc_a(dvT[j][dt], dT[dt][s2], pf[par][s2]);
                        mfma_acc_a(dkT[j][dt], qT[dt][s2], dsf[par][s2]);
                    }
            };
            // light waves: dQ^T of the previous step, 20 key steps in NB + 1 slices
            auto dq_slice = [&](auto jc) {
                if (HAS_DQ) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int s0 = j * 20 / (NB + 1), s1 = (j + 1) * 20 / (NB + 1);
                    // all of the slice's fragments are requested before the first MFMA (a read per MFMA, awaited on the spot, exposes the
                    // LDS latency twenty times per step)
                    bf16x8 xf[s1 - s0];
#pragma unroll
                    for (int s = s0; s < s1; ++s) {
                        const bf16x4 lo = lds_read_tr16(xr + s * 1024 + xro[0][s & 1]), hi = lds_read_tr16(xr + s * 1024 + xro[1][s & 1]);
                        bf16x8 t;
                        t[0] = lo[0]; t[1] = lo[1]; t[2] = lo[2]; t[3] = lo[3]; t[4] = hi[0]; t[5] = hi[1]; t[6] = hi[2]; t[7] = hi[3];
                        xf[s - s0] = t;
                    }
#pragma unroll
                    for (int s = s0; s < s1; ++s) dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[s], xf[s - s0], dqa, 0, 0, 0);
                }
            };
            if (HAS_DQ) dqa = zero16();
            qk(Int2<0>{});
            __builtin_amdgcn_sched_barrier(0);
            // region 0
            if (NB > 1) qk(Int2<1>{});
            arith(Int2<0>{});
            dq_slice(Int2<0>{});
            __builtin_amdgcn_sched_barrier(0);
            // region 1
            if (NB > 2) qk(Int2<(NB > 2 ? 2 : 0)>{});
            dvdk(Int2<0>{});
            arith(Int2<1>{});
            dq_slice(Int2<1>{});
            __builtin_amdgcn_sched_barrier(0);
            // region 2
            dvdk(Int2<1>{});
            if (NB > 2) arith(Int2<(NB > 2 ? 2 : 0)>{});
            dq_slice(Int2<2>{});
            __builtin_amdgcn_sched_barrier(0);
            if (NB > 2) {
                dvdk(Int2<(NB > 2 ? 2 : 0)>{});
                dq_slice(Int2<3>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u == 5) BSTAMP(21);
            if (HAS_DQ) {
                // dQ rows of step u - 1 (rows >= S and the step before the first: out of the descriptor's range, dropped)
                const int q = (u - 1) * 32 + r;
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    uint32_t w[2][2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int i0 = 4 * (2 * G + gg);
                        const bf16x4 v = f32x4_to_bf16x4(f32x4{dqa[i0] * SCALE, dqa[i0 + 1] * SCALE, dqa[i0 + 2] * SCALE, dqa[i0 + 3] * SCALE});
                        const u32x2 t = __builtin_bit_cast(u32x2, v);
                        w[gg][0] = t[0]; w[gg][1] = t[1];
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                    const uint32_t off = (u > 0 && q < p.S) ? (uint32_t)(((int64_t)q * ld + 32 * dtl + 16 * G + 8 * hh) * 2) : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]}, rs_dq, off, 0, 0);
                }
            }
            if (u == 5) BSTAMP(22);
            // everything but this step's four pieces and (light waves) two stores: the lse words, the O rows, every older piece
            if (HAS_DQ) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            make_stats(v1, (gs + 1) & 1);
            BSTAMP(2 + u);
        }
        int nst = 0;
        if (HAS_DQ) {
            // dQ^T of the last step
            const char* const xr = xbuf + ((NU - 1) & 1) * XB;
            dqa = zero16();
#pragma unroll
            for (int s = 0; s < 20; ++s) {
                const bf16x4 lo = lds_read_tr16(xr + s * 1024 + xro[0][s & 1]), hi = lds_read_tr16(xr + s * 1024 + xro[1][s & 1]);
                bf16x8 xf;
                xf[0] = lo[0]; xf[1] = lo[1]; xf[2] = lo[2]; xf[3] = lo[3]; xf[4] = hi[0]; xf[5] = hi[1]; xf[6] = hi[2]; xf[7] = hi[3];
                dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[s], xf, dqa, 0, 0, 0);
            }
            const int q = (NU - 1) * 32 + r;
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                uint32_t w[2][2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int i0 = 4 * (2 * G + gg);
                    const bf16x4 v = f32x4_to_bf16x4(f32x4{dqa[i0] * SCALE, dqa[i0 + 1] * SCALE, dqa[i0 + 2] * SCALE, dqa[i0 + 3] * SCALE});
                    const u32x2 t = __builtin_bit_cast(u32x2, v);
                    w[gg][0] = t[0]; w[gg][1] = t[1];
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                const uint32_t off = q < p.S ? (uint32_t)(((int64_t)q * ld + 32 * dtl + 16 * G + 8 * hh) * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]}, rs_dq, off, 0, 0);
            }
            nst += 2;
        }
        BSTAMP(12);
        // dK / dV: lane = key, registers = d; the same pairing of the two lane halves into 16-byte pieces (rows >= S dropped)
        {
            const __amdgpu_buffer_rsrc_t rs_dkv = uniform_rsrc(dq_base, (uint32_t)(((int64_t)(p.S - 1) * ld + 2 * D + 64) * 2));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int key = (kb0 + j) * 32 + r;
#pragma unroll
                for (int which = 0; which < 2; ++which)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const f32x16& a = which ? dvT[j][dt] : dkT[j][dt];
                        const float sc = which ? 1.0f : SCALE;
#pragma unroll
                        for (int G = 0; G < 2; ++G) {
                            uint32_t w[2][2];
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg) {
                                const int i0 = 4 * (2 * G + gg);
                                const bf16x4 v = f32x4_to_bf16x4(f32x4{a[i0] * sc, a[i0 + 1] * sc, a[i0 + 2] * sc, a[i0 + 3] * sc});
                                const u32x2 t = __builtin_bit_cast(u32x2, v);
                                w[gg][0] = t[0]; w[gg][1] = t[1];
                            }
                            const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                            const uint32_t off = key < p.S ? (uint32_t)(((int64_t)key * ld + (1 + which) * D + 32 * dt + 16 * G + 8 * hh) * 2) : 0xFFFFFFF0u;
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]}, rs_dkv, off, 0, 0);
                        }
                    }
            }
        }
        BSTAMP(13);
        // every piece went out before this problem's last stores: 8 NB (+ 2 on the light waves)
        if (HAS_DQ) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * NB + 2) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * NB) : "memory");
        (void)nst;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave is done with this K image and with X
        BSTAMP(14);
        pc = pn;
    }
}

__global__ __launch_bounds__(256, 1) void mha_bwd_wide_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 2) bwd_wide_body<3, false>(p, smem, wave, lane0, 3 * wave);
    else bwd_wide_body<2, true>(p, smem, wave, lane0, 6 + 2 * (wave - 2));
}

int32_t launch_bwd_wide_impl(const MhaArgs& a, hipStream_t s) {
    constexpr int lds = 2 * 320 * 128 + 2 * 320 * 64 + 4 * 8192 + 256 + 4096 + 1024;
    static DeviceOnce once;
    static int cus = 0;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int dev = 0;
        VIPANT_HIP_TRY(hipGetDevice(&dev));
        VIPANT_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int nprob = a.batch * a.H;
    hipLaunchKernelGGL(mha_bwd_wide_kernel, dim3(nprob < cus ? nprob : cus), dim3(256), lds, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

namespace vipant_attn {

int32_t launch_fwd_wide(const MhaArgs& a, hipStream_t s) {
    VIPANT_REQUIRE(a.S > 288 && a.S <= 320, VIPANT_EBADSHAPE, "mha (wide forward): 288 < S <= 320 expected, got %d", a.S);
    return launch_fwd_wide_nt<10>(a, s);
}

int32_t launch_bwd_wide(const MhaArgs& a, hipStream_t s) {
    VIPANT_REQUIRE(a.S > 288 && a.S <= 320, VIPANT_EBADSHAPE, "mha (wide backward): 288 < S <= 320 expected, got %d", a.S);
    return launch_bwd_wide_impl(a, s);
}

}  // namespace vipant_attn

#ifdef VIPANT_ATTN_STAMPS
extern "C" int32_t vipant_debug_attnw_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnw_trace), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int32_t vipant_debug_attnw_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnw_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif
