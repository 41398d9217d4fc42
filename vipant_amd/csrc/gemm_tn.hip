// C[P,Q] (+)= A[M,P]^T . B[M,Q] -- the weight-gradient contraction (autograd of nn.Linear / conv1 weights,
// cvap/module/val.py:500-506, 245-247), reduction over the token dimension M.
//
// gfx950 design: both operands are M-major (the reduction index is the slow one), so tiles are staged
// [64 m][256 cols] by LDS-DMA exactly as they lie in HBM (full 512-B row segments, coalesced) and the MFMA
// fragments -- which need 8 consecutive reduction elements per lane -- are formed by ds_read_b64_tr_b16
// (hardware transpose read), two per 16x16x32 operand.  512-B rows are XOR-swizzled at 16-B granularity
// (on the DMA source address and on the read address) so the 32 lanes of a half-wave hit all 64 banks.
// 256x256 output tile per 512-thread workgroup; the M range is split over workgroups to fill 256 CUs and
// the fp32 partial tiles are summed by a second deterministic pass (no float atomics).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TP = 256, TQ = 256, BK = 64;
constexpr int OP_BYTES = BK * TP * 2;       // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * OP_BYTES;   // 64 KiB

struct GemmTN {
    const bf16_t* A; const bf16_t* B; float* out;  // out: slab base (splits > 1) or C
    int64_t lda, ldb, ldo;
    int M, P, Q;
    int splits, kt_per_split;
    int direct;   // 1: write straight into C (single split, no accumulate)
    float* colsum;   // optional [splits * ntq][ntp*256]: partial column sums of A (bias gradient of the same Linear)
};

__device__ __forceinline__ int tn_swz(int m) { return ((m & 3) | (((m >> 3) & 1) << 2)) << 1; }

__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wq = wave & 3;

    const int ntq = (p.Q + TQ - 1) / TQ, ntp = (p.P + TP - 1) / TP;
    const int ntiles = ntp * ntq;
    const int bid = xcd_remap(blockIdx.x, ntiles * p.splits);
    const int split = bid / ntiles, tile = bid % ntiles;
    const int tp = tile / ntq, tq = tile % ntq;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int nk_total = (p.M + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    int nk = nk_total - kt0;
    if (nk > p.kt_per_split) nk = p.kt_per_split;
    const int mbeg = kt0 * BK;

    // Descriptors based at (first row of this split, first column of the tile).  Rows >= M are past the end
    // of the range and read as zero, which is what the reduction needs for the M tail.
    const bf16_t* Ab = p.A + (int64_t)mbeg * p.lda + p0;
    const bf16_t* Bb = p.B + (int64_t)mbeg * p.ldb + q0;
    int64_t a_bytes = ((int64_t)(p.M - mbeg) * p.lda - p0) * 2;
    int64_t b_bytes = ((int64_t)(p.M - mbeg) * p.ldb - q0) * 2;
    if (a_bytes < 0) a_bytes = 0;
    if (b_bytes < 0) b_bytes = 0;
    const auto rsA = make_rsrc(Ab, (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes));
    const auto rsB = make_rsrc(Bb, (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes));

    // DMA: one wave instruction = 2 tile rows (m) x 512 B.  Wave w fills rows w*8 .. w*8+7 of A and of B.
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = (wave * 4 + i) * 2 + (lane >> 5);
        const int c = (lane & 31) ^ tn_swz(m);
        voffA[i] = (uint32_t)(m * p.lda * 2 + c * 16);
        voffB[i] = (uint32_t)(m * p.ldb * 2 + c * 16);
    }
    const uint32_t kstepA = (uint32_t)(BK * p.lda * 2), kstepB = (uint32_t)(BK * p.ldb * 2);
    auto stage_load = [&](int stage, int kt) {
        char* sA = smem + stage * STAGE_BYTES + wave * 4096;
        char* sB = sA + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, sA + i * 1024, voffA[i], (uint32_t)kt * kstepA);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsB, sB + i * 1024, voffB[i], (uint32_t)kt * kstepB);
    };

    // Transposed fragment reads.  Lane l: g = l>>4 owns k-slots 8g..8g+7 (tile rows ks*32 + 8g + 0..7),
    // within the group lane 4*qq+pp supplies row qq (first read) / 4+qq (second), columns 4pp..4pp+3 of the
    // 16-column block; it receives column (l&15), rows 0..3 / 4..7.
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    uint32_t rdoff[2][2];   // [ks][half] byte offset of this lane's address for column block 0 of the tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 8 * g + 4 * h + qq;
            rdoff[ks][h] = (uint32_t)(m * 512 + (((pp >> 1) ^ tn_swz(m)) << 4) + (pp & 1) * 8);
        }
    // Column block cb (16 columns = 2 chunks) adds (2*cb) to the logical chunk; since tn_swz only touches
    // bits 1..3 and 2*cb has bit 0 clear, (c0 + 2cb) ^ s == (c0 ^ s) ^ (2cb)  ->  XOR the byte offset with cb<<5.
    const uint32_t lds0 = lds_offset(smem);
    auto frag = [&](uint32_t tile, int ks, int cb) -> bf16x8 {      // raw (asm) reads: see lds_read_tr16_pair_raw in common.h
        return lds_read_tr16_pair_raw(tile + (rdoff[ks][0] ^ (uint32_t)(cb << 5)), tile + (rdoff[ks][1] ^ (uint32_t)(cb << 5)));
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Column sums of A over the token dimension ride along: the A tile is in LDS anyway.  Thread t owns column
    // (t & 255) and the rows of half (t >> 8) of every K-tile; only the workgroups of the first Q tile do it.
    // Column sums of A ride along.  The ntq workgroups that share an A tile take turns (K-tile kt belongs to workgroup
    // kt % ntq), and a thread sums 8 columns (one 16-B chunk) of 4 rows per K-tile with ds_read_b128 -- the first version
    // (the tq == 0 workgroups alone, 32 two-byte reads per thread and K-tile) made those workgroups, and with them the
    // whole launch, 15-25 % slower.
    const bool do_colsum = p.colsum != nullptr;
    const int cs_chunk = tid & 31, cs_rg = tid >> 5;
    float cs_acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs_acc[e] = 0.f;
    if (nk > 0) {
        stage_load(0, 0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage_load(cur ^ 1, kt + 1);
            const char* sA = smem + cur * STAGE_BYTES;
            const char* sB = sA + OP_BYTES;
            if (do_colsum && (kt % ntq) == tq) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int m = cs_rg + 16 * k4;
                    const bf16x8 v = *(const bf16x8*)(sA + m * 512 + ((cs_chunk ^ tn_swz(m)) << 4));
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs_acc[e] += (float)v[e];
                }
            }
            // Explicit fragment pipeline (same as the NT kernel): the B fragments of the whole K-tile first, the A
            // fragments in a 3-deep register ring two MFMA groups ahead of their use.  The transposed reads go through
            // inline asm (the builtin would make hipcc drain the LDS-DMA of the NEXT stage, issued just above, before
            // touching this one), so the lgkmcnt waits are explicit: LDS results return in order, a fragment = 2 reads.
            const uint32_t tA = lds0 + (uint32_t)(cur * STAGE_BYTES), tB = tA + OP_BYTES;
            bf16x8 fq[2][4], fp[3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) fq[ks][j] = frag(tB, ks, wq * 4 + j);
            fp[0] = frag(tA, 0, wp * 8 + 0);
            fp[1] = frag(tA, 0, wp * 8 + 1);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ks = t >> 3, i = t & 7;
                if (t + 2 < 16) fp[(t + 2) % 3] = frag(tA, (t + 2) >> 3, wp * 8 + ((t + 2) & 7));
                // outstanding behind fragment t: the fragments of groups t+1 and t+2 (2 reads each)
                if (t + 2 < 16) lds_raw_wait<4>(); else if (t + 1 < 16) lds_raw_wait<2>(); else lds_raw_wait<0>();
                lds_raw_use(fp[t % 3]);
                if (t == 0) {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) lds_raw_use(fq[k2][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[ks][j], fp[t % 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    if (do_colsum) {      // all LDS reads of the loop are behind its last barrier: reuse the front of the buffer
        float* red = (float*)smem;                     // [16 row groups][256 columns]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[cs_rg * 256 + cs_chunk * 8 + e] = cs_acc[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += red[r * 256 + tid];
            p.colsum[((int64_t)split * ntq + tq) * (ntp * TP) + p0 + tid] = sum;
        }
    }
    // lane holds C[p = p0 + wp*128 + i*16 + (lane&15)][q = q0 + wq*64 + j*16 + (lane>>4)*4 + 0..3]
    const int frow = lane & 15;
    if (p.direct) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = p0 + wp * 128 + i * 16 + frow;
            if (pr >= p.P) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qc = q0 + wq * 64 + j * 16 + g * 4;
                if (qc < p.Q) *(f32x4*)(p.out + (int64_t)pr * p.ldo + qc) = acc[i][j];
            }
        }
    } else {
        float* slab = p.out + ((int64_t)split * ntiles + tile) * (TP * TQ);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(slab + (wp * 128 + i * 16 + frow) * TQ + wq * 64 + j * 16 + g * 4) = acc[i][j];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong variant (same idea as gemm_nt_pp_kernel, see gemm_nt.hip): the two wave groups (wp = 0: output rows p0..p0+127,
// wp = 1: p0+128..p0+255) run the same stream half a K-tile apart, so that one group's LDS-DMA issue burst overlaps the
// other group's MFMAs on every SIMD.  A columns are private to a group (two [64 m][128 p] stages of 16 KiB each, filled by the
// group one K-tile ahead); B is read by both groups and lives in a ring of three [64 m][256 q] slots (group 0 fills tile rows
// 0-31 of slot k+1, group 1 rows 32-63 of slot k+2, each at the start of ITS K-tile k).  Waits before the barrier that ends
// an interval -- group 0: none | vmcnt(0);  group 1: vmcnt(8) | vmcnt(4).  Every wave issues 8 DMA pieces per K-tile
// unconditionally (past the split's range they read rows the loop never consumes, past the matrix the descriptor returns
// zeros), so the counts are exact.
constexpr int PP_A_STAGE = BK * 128 * 2;                 // 16 KiB
constexpr int PP_B_BASE = 4 * PP_A_STAGE;                // [group][stage]
constexpr int PP_B_SLOT = BK * TQ * 2;                   // 32 KiB
constexpr int PP_LDS_BYTES = PP_B_BASE + 3 * PP_B_SLOT;  // 160 KiB

__device__ __forceinline__ void tn_pp_body(const GemmTN& p, int block, char* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wl = wave & 3;            // grp = wp, wl = wq

    const int ntq = (p.Q + TQ - 1) / TQ, ntp = (p.P + TP - 1) / TP;
    const int ntiles = ntp * ntq;
    const int bid = xcd_remap(block, ntiles * p.splits);
    const int split = bid / ntiles, tile = bid % ntiles;
    const int tp = tile / ntq, tq = tile % ntq;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int nk_total = (p.M + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    int nk = nk_total - kt0;
    if (nk > p.kt_per_split) nk = p.kt_per_split;
    const int mbeg = kt0 * BK;

    // descriptors: A based at this group's first column, B at the tile's first column; rows >= M read as zero
    const bf16_t* Ab = p.A + (int64_t)mbeg * p.lda + p0 + grp * 128;
    const bf16_t* Bb = p.B + (int64_t)mbeg * p.ldb + q0;
    int64_t a_bytes = ((int64_t)(p.M - mbeg) * p.lda - p0 - grp * 128) * 2;
    int64_t b_bytes = ((int64_t)(p.M - mbeg) * p.ldb - q0) * 2;
    if (a_bytes < 0) a_bytes = 0;
    if (b_bytes < 0) b_bytes = 0;
    const auto rsA = make_rsrc(Ab, (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes));
    const auto rsB = make_rsrc(Bb, (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes));

    // DMA.  A: one instruction = 4 tile rows x 256 B; wave wl of a group fills rows wl*16 .. wl*16+15 of the group's stage.
    //       B: one instruction = 2 tile rows x 512 B; wave w fills rows w*8 .. w*8+7 of the slot (group 0: 0-31, group 1: 32-63).
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ma = wl * 16 + i * 4 + (lane >> 4);
        voffA[i] = (uint32_t)(ma * p.lda * 2 + (((lane & 15) ^ tn_swz(ma)) << 4));
        const int mb = (wave * 4 + i) * 2 + (lane >> 5);
        voffB[i] = (uint32_t)(mb * p.ldb * 2 + (((lane & 31) ^ tn_swz(mb)) << 4));
    }
    const uint32_t kstepA = (uint32_t)(BK * p.lda * 2), kstepB = (uint32_t)(BK * p.ldb * 2);
    char* const ldsA = smem + grp * (2 * PP_A_STAGE) + wl * 4096;
    char* const ldsB = smem + PP_B_BASE + wave * 4096;
    auto fill_a = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, ldsA + stage * PP_A_STAGE + i * 1024, voffA[i], (uint32_t)kt * kstepA);
    };
    auto fill_b = [&](int slot, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsB, ldsB + slot * PP_B_SLOT + i * 1024, voffB[i], (uint32_t)kt * kstepB);
    };

    // transposed fragment reads (see gemm_tn_kernel): A rows are 256 B here, B rows 512 B
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    uint32_t rdA[2][2], rdB[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 8 * g + 4 * h + qq;
            const uint32_t sw = (uint32_t)((((pp >> 1) ^ tn_swz(m)) << 4) + (pp & 1) * 8);
            rdA[ks][h] = (uint32_t)(m * 256) + sw;
            rdB[ks][h] = (uint32_t)(m * 512) + sw;
        }
    const uint32_t lds0 = lds_offset(smem);
    const uint32_t baseA = lds0 + (uint32_t)(grp * (2 * PP_A_STAGE)), baseB = lds0 + (uint32_t)PP_B_BASE;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // column sums of A (bias gradient): each group sums its own 128 columns; the ntq workgroups sharing an A tile take turns
    const bool do_colsum = p.colsum != nullptr;
    const int tl = tid & 255;
    const int cs_chunk = tl & 15, cs_rg = tl >> 4;
    float cs_acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs_acc[e] = 0.f;

    // 32 MFMAs of one k-step
    auto half_ktile = [&](int ks, uint32_t tA, uint32_t tB) {
        bf16x8 fq[4], fp[3];
        auto fragA = [&](int cb) { return lds_read_tr16_pair_raw(tA + (rdA[ks][0] ^ (uint32_t)(cb << 5)), tA + (rdA[ks][1] ^ (uint32_t)(cb << 5))); };
        auto fragB = [&](int cb) { return lds_read_tr16_pair_raw(tB + (rdB[ks][0] ^ (uint32_t)(cb << 5)), tB + (rdB[ks][1] ^ (uint32_t)(cb << 5))); };
#pragma unroll
        for (int j = 0; j < 4; ++j) fq[j] = fragB(wl * 4 + j);
        fp[0] = fragA(0);
        fp[1] = fragA(1);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t + 2 < 8) fp[(t + 2) % 3] = fragA(t + 2);
            if (t + 2 < 8) lds_raw_wait<4>(); else if (t + 1 < 8) lds_raw_wait<2>(); else lds_raw_wait<0>();
            lds_raw_use(fp[t % 3]);
            if (t == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) lds_raw_use(fq[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[j], fp[t % 3], acc[t][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: K-tile 0 (and, for group 1, its B rows of K-tile 1)
    fill_a(0, 0);
    fill_b(0, 0);
    if (grp == 1) {
        fill_b(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the lag

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const uint32_t tA = baseA + (uint32_t)(stage * PP_A_STAGE), tB = baseB + (uint32_t)(slot * PP_B_SLOT);
        fill_a(stage ^ 1, kt + 1);
        if (grp == 0) fill_b(slot1, kt + 1); else fill_b(slot2, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (do_colsum && (kt % ntq) == tq) {
            const char* sA = smem + grp * (2 * PP_A_STAGE) + stage * PP_A_STAGE;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int m = cs_rg + 16 * k4;
                const bf16x8 v = *(const bf16x8*)(sA + m * 256 + ((cs_chunk ^ tn_swz(m)) << 4));
#pragma unroll
                for (int e = 0; e < 8; ++e) cs_acc[e] += (float)v[e];
            }
        }
        half_ktile(0, tA, tB);
        if (grp == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        half_ktile(1, tA, tB);
        if (grp == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot1;
    }
    // the (unused) fills issued by the last K-tiles must have landed in this group's A stages before they are reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // column-sum partials: the group's own (now idle) A stages serve as the reduction buffer
    float* red = (float*)(smem + grp * (2 * PP_A_STAGE));        // [16 row groups][128 columns] fp32 = 8 KiB
    if (do_colsum) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[cs_rg * 128 + cs_chunk * 8 + e] = cs_acc[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 0) __builtin_amdgcn_s_barrier();          // pairs with group 1's last barrier
    if (do_colsum && tl < 128) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += red[r * 128 + tl];
        p.colsum[((int64_t)split * ntq + tq) * (ntp * TP) + p0 + grp * 128 + tl] = sum;
    }
    // lane holds C[p = p0 + grp*128 + i*16 + (lane&15)][q = q0 + wl*64 + j*16 + (lane>>4)*4 + 0..3]
    const int frow = lane & 15;
    if (p.direct) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = p0 + grp * 128 + i * 16 + frow;
            if (pr >= p.P) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qc = q0 + wl * 64 + j * 16 + g * 4;
                if (qc < p.Q) *(f32x4*)(p.out + (int64_t)pr * p.ldo + qc) = acc[i][j];
            }
        }
    } else {
        float* slab = p.out + ((int64_t)split * ntiles + tile) * (TP * TQ);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(slab + (grp * 128 + i * 16 + frow) * TQ + wl * 64 + j * 16 + g * 4) = acc[i][j];
    }
}

__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn_pp_body(p, (int)blockIdx.x, smem);
}

// two contractions of one shape in one launch (the two feature gradients of the InfoNCE loss: 2 x 32 tiles fill the chip with
// half the splits, i.e. twice the K-tiles per workgroup and half the partial tiles to reduce)
struct GemmTNPair { GemmTN a, b; int blocks_a; };
__global__ __launch_bounds__(512, 2) void gemm_tn_pp_pair_kernel(GemmTNPair pr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool first = (int)blockIdx.x < pr.blocks_a;
    tn_pp_body(first ? pr.a : pr.b, first ? (int)blockIdx.x : (int)blockIdx.x - pr.blocks_a, smem);
}

// ---------------------------------------------------------------------------------------------------------
// e4m3 operands (BASELINE.json configs[4]): the same ping-pong stream on v_mfma_scale_f32_16x16x128_f8f6f4 with the TOKEN axis as K.
// Both operands are token-major e4m3 matrices exactly as the NT contractions read them ([M, P] and [M, Q], one byte per element)
// whose MX scales are UNIFORM over aligned blocks of 32 tokens x 32 columns (vipant_quant_e4m3_mx32 / vipant_mx_uniform32, or a
// producer that emits them that way): one scale per (32 tokens, 32 columns) is then also one scale per 32 k of a column, which is
// what the instruction's block-scale operand needs when k runs along the tokens -- no transposed copy, no second quantisation.
//   * K-tile = 128 tokens = ONE MFMA k-step.  LDS images as in the bf16 kernel, byte for byte: A stages [128 m][128 p] = 16 KiB per
//     group and stage, B slots [128 m][256 q] = 32 KiB, filled by LDS-DMA as the rows lie in HBM (16-byte chunks XOR-swizzled by
//     row: f_A(m) = m[2:1] | m[4] << 2 on 128-byte rows, f_B(m) = m[2:0] | m[4] << 3 on 256-byte rows).
//   * fragments by ds_read_b64_tr_b8 (tools/probes/tr8_probe.hip: per 16 lanes an 8-row x 16-column byte block, lane 2 q + h gives
//     the address of row q, columns 8 h .. 8 h + 7, lane i receives column i, rows 0-7 in bytes 0-7).  The instruction wants, in
//     lane (r, qd), k = 16 qd .. + 15 in dwords 0-3 and k = 64 + 16 qd .. in dwords 4-7 of column r: four transposed reads per
//     16-column fragment; with the swizzles above the 16 row segments of a 32-lane half fall on 16 distinct 16-byte bank groups.
//   * scales: lane (r, qd) supplies the scale of (its column's 32-column block, token block 4 kt + qd).  In the MX layout (common.h)
//     the bytes of one 128-token group and one 128-column group are 32 contiguous bytes [kb & 3][row tile 0-7]; rows 32 qd of the
//     group are row tile 2 qd: one 32-byte (A) / 16-byte (B) load per wave and K-tile, a K-tile ahead, in front of that K-tile's DMA
//     pieces (older than all of them: the counted waits cover it).  Token blocks at or beyond M carry 2^0 (their bytes read as zero).
// The two barrier intervals of a K-tile take the wave's row tiles 0-3 and 4-7 (16 MFMAs of 32 cycles each); the B fragments are
// read in the first interval and kept.  DMA piece counts, lag and waits are the bf16 kernel's.
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int V> struct Int { static constexpr int value = V; };

struct GemmTN8 {
    const uint8_t* A; const uint8_t* B; const uint8_t* sa; const uint8_t* sb; float* out;
    int64_t lda, ldb, ldo;       // lda / ldb: bytes per token row (= columns of the quantised matrices: their scale layouts' row length)
    int M, P, Q;
    int splits, kt_per_split;    // in K-tiles of 128 tokens
    int direct;
    float* colsum;               // optional [splits * ntq][ntp * 256]: partial column sums of dequant(A) (the bias gradient of the same Linear)
};

constexpr int BK8 = 128;

// the four transposed reads of one fragment: one address register, the other three rows 8 / 64 / 72 tile rows further as instruction
// offsets (STEP = 8 rows in bytes: 1024 on the 128-byte rows of A, 2048 on the 256-byte rows of B)
template <int STEP>
__device__ __forceinline__ i32x8 lds_read_tr8_quad_raw(uint32_t a) {
    v2i32_t r0, r1, r2, r3;
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(r0) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r1) : "v"(a), "n"(STEP));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r2) : "v"(a), "n"(8 * STEP));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r3) : "v"(a), "n"(9 * STEP));
    return i32x8{r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
}
__device__ __forceinline__ void lds_raw_use8(i32x8& f) { asm volatile("" : "+v"(f)); }

__device__ __forceinline__ int tn8_swz_a(int m) { return ((m >> 1) & 3) | (((m >> 4) & 1) << 2); }
__device__ __forceinline__ int tn8_swz_b(int m) { return (m & 7) | (((m >> 4) & 1) << 3); }

__global__ __launch_bounds__(512, 2) void gemm_tn8_pp_kernel(GemmTN8 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wl = wave & 3;

    const int ntq = (p.Q + TQ - 1) / TQ, ntp = (p.P + TP - 1) / TP;
    const int ntiles = ntp * ntq;
    const int bid = xcd_remap((int)blockIdx.x, ntiles * p.splits);
    const int split = bid / ntiles, tile = bid % ntiles;
    const int tp = tile / ntq, tq = tile % ntq;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int nk_total = (p.M + BK8 - 1) / BK8;
    const int kt0 = split * p.kt_per_split;
    int nk = nk_total - kt0;
    if (nk > p.kt_per_split) nk = p.kt_per_split;
    const int mbeg = kt0 * BK8;

    const uint8_t* Ab = p.A + (int64_t)mbeg * p.lda + p0 + grp * 128;
    const uint8_t* Bb = p.B + (int64_t)mbeg * p.ldb + q0;
    int64_t a_bytes = (int64_t)(p.M - mbeg) * p.lda - p0 - grp * 128;
    int64_t b_bytes = (int64_t)(p.M - mbeg) * p.ldb - q0;
    if (a_bytes < 0) a_bytes = 0;
    if (b_bytes < 0) b_bytes = 0;
    const auto rsA = make_rsrc(Ab, (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes));
    const auto rsB = make_rsrc(Bb, (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes));

    // DMA.  A: one instruction = 8 tile rows x 128 B; wave wl of a group fills rows wl*32 .. wl*32+31 of the group's stage.
    //       B: one instruction = 4 tile rows x 256 B; wave w fills rows w*16 .. w*16+15 of the slot (group 0: 0-63, group 1: 64-127).
    // (piece i of a wave lies i * 8 rows (A) / i * 4 rows (B) below piece 0 -- a scalar offset -- and its swizzle differs from piece
    // 0's in chunk bit 2 alone: pieces 2, 3 of A and pieces 1, 3 of B; two address registers per operand instead of four)
    const int ma0 = wl * 32 + (lane >> 3), mb0 = wave * 16 + (lane >> 4);
    const uint32_t voffA0 = (uint32_t)(ma0 * p.lda + (((lane & 7) ^ tn8_swz_a(ma0)) << 4)), voffA2 = voffA0 ^ 64u;
    const uint32_t voffB0 = (uint32_t)(mb0 * p.ldb + (((lane & 15) ^ tn8_swz_b(mb0)) << 4)), voffB1 = voffB0 ^ 64u;
    const uint32_t kstepA = (uint32_t)(BK8 * p.lda), kstepB = (uint32_t)(BK8 * p.ldb);
    const uint32_t pieceA = (uint32_t)(8 * p.lda), pieceB = (uint32_t)(4 * p.ldb);
    char* const ldsA = smem + grp * (2 * PP_A_STAGE) + wl * 4096;
    char* const ldsB = smem + PP_B_BASE + wave * 4096;
    auto fill_a = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            lds_dma16(rsA, ldsA + stage * PP_A_STAGE + i * 1024, i < 2 ? voffA0 : voffA2, (uint32_t)kt * kstepA + (uint32_t)i * pieceA);
    };
    auto fill_b = [&](int slot, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            lds_dma16(rsB, ldsB + slot * PP_B_SLOT + i * 1024, (i & 1) ? voffB1 : voffB0, (uint32_t)kt * kstepB + (uint32_t)i * pieceB);
    };

    // transposed fragment reads: read n of a fragment takes tile rows kb_n + 0..7, kb_n = 64 (n >> 1) + 16 qd + 8 (n & 1); the
    // swizzle term is the same for the four (it depends on m[2:1] / m[2:0] and m[4] = qd & 1 only)
    const int qd = lane >> 4, rq = (lane & 15) >> 1, rh = lane & 1;
    const int m_rd = 16 * qd + rq;
    const uint32_t rdA = (uint32_t)(m_rd * 128 + (tn8_swz_a(m_rd) << 4) + rh * 8);
    const uint32_t rdB = (uint32_t)(m_rd * 256 + (tn8_swz_b(m_rd) << 4) + rh * 8);
    const uint32_t lds0 = lds_offset(smem);
    const uint32_t baseA = lds0 + (uint32_t)(grp * (2 * PP_A_STAGE)), baseB = lds0 + (uint32_t)PP_B_BASE;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // block scales of one K-tile: A -> byte (i >> 1) of a dword, B -> byte (j >> 1)
    const int ktpr_a = (int)(p.lda >> 7), ktpr_b = (int)(p.ldb >> 7);
    const int ca = (p0 + grp * 128) >> 7;                    // the group's 128-column group of A
    const int kb_b = (q0 + wl * 64) >> 5;                    // the wave's first 32-column block of B (even)
    const bool a_cols_ok = p0 + grp * 128 < p.P, b_cols_ok = q0 + wl * 64 < p.Q;
    // (the six dwords a lane needs -- byte 2 qd of the 8-byte words of its four / two column blocks -- are requested as six dword loads
    // and kept RAW until the K-tile's last barrier: `scales_ready` is the first use the compiler sees, so its vmcnt wait lands behind the
    // counted waits that have retired the loads anyway.  Used where they are requested, the loads' full latency -- they are the only
    // HBM round trip of the loop that is not an LDS-DMA -- stood in front of every K-tile's first MFMA: 1755 against 1196 us per launch
    // at [323584, 1024] x [323584, 4096].)
    // Buffer loads (descriptor + 32-bit lane offset + scalar offset): with plain pointers the lane-dependent 64-bit addresses were two
    // more register pairs across the K-loop, and once the column sums were added they were spilled -- reloaded at the top of every
    // K-tile behind a vmcnt(0).  Out-of-range K-tiles read as byte 0; token blocks at or beyond M get 2^0 by a select (their bytes in
    // the array were never written).
    const int sh = 16 * (qd & 1), hi = qd >> 1;
    struct RawScales { uint32_t a[4], b[2]; };
    const int64_t tk_groups = ((int64_t)p.M + 127) >> 7;
    const auto rsSA = make_rsrc(p.sa, (uint32_t)(tk_groups * ktpr_a * 512));
    const auto rsSB = make_rsrc(p.sb, (uint32_t)(tk_groups * ktpr_b * 512));
    const uint32_t sv_a = (uint32_t)(hi * 4), sv_b = (uint32_t)((kb_b & 3) * 8 + hi * 4);
    auto request_scales = [&](int kt_global, RawScales& r) {
        const uint32_t so_a = (uint32_t)((kt_global * ktpr_a + ca) * 512), so_b = (uint32_t)((kt_global * ktpr_b + (kb_b >> 2)) * 512);
#pragma unroll
        for (int k = 0; k < 4; ++k) r.a[k] = __builtin_amdgcn_raw_buffer_load_b32(rsSA, sv_a + 8 * k, so_a, 0);
        r.b[0] = __builtin_amdgcn_raw_buffer_load_b32(rsSB, sv_b, so_b, 0);
        r.b[1] = __builtin_amdgcn_raw_buffer_load_b32(rsSB, sv_b + 8, so_b, 0);
    };
    auto scales_ready = [&](RawScales& r, uint32_t& sa_out, uint32_t& sb_out) {
        asm volatile("" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]), "+v"(r.b[1]));
        sa_out = ((r.a[0] >> sh) & 255u) | (((r.a[1] >> sh) & 255u) << 8) | (((r.a[2] >> sh) & 255u) << 16) | (((r.a[3] >> sh) & 255u) << 24);
        sb_out = ((r.b[0] >> sh) & 255u) | (((r.b[1] >> sh) & 255u) << 8);
    };
    auto scales_valid = [&](int kt_global, uint32_t& sa_io, uint32_t& sb_io) {      // token block in range, columns of the tile in range
        const bool tok = kt_global * BK8 + 32 * qd < p.M;
        sa_io = (tok && a_cols_ok) ? sa_io : 0x7F7F7F7Fu;
        sb_io = (tok && b_cols_ok) ? sb_io : 0x7F7F7F7Fu;
    };

    // Column sums of A ride along, as in the bf16 kernel: the ntq workgroups that share an A tile take turns (K-tile kt belongs to
    // workgroup kt % ntq).  Wave wl of a group takes token block wl of the K-tile (rows 32 wl .. + 31): a lane sums 4 columns (one
    // dword: cs_c = lane & 31) of every second row and applies the block's scale, which lane 16 wl of the wave holds (a readlane).
    // Four accumulators per lane: eight or sixteen made the K-loop spill.
    const bool do_colsum = p.colsum != nullptr;
    const int cs_c = lane & 31, cs_row0 = 32 * wl + (lane >> 5);
    float cs_acc[4] = {0.f, 0.f, 0.f, 0.f};
    int cs_turn = tq;                    // K-tiles until this workgroup's next turn (scalar)

    i32x8 fq[4];
    uint32_t sav = 0x7F7F7F7Fu, sbv = 0x7F7F7F7Fu, sav_n = 0x7F7F7F7Fu, sbv_n = 0x7F7F7F7Fu;
    RawScales raw;
#define VIPANT_TN8_MX(ACC, BF, AF, OB, OA) \
    ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(BF, AF, ACC, 0, 0, OB, (int)sbv, OA, (int)sav)
    // row tiles 4 h .. 4 h + 3 of the wave (h = 0: the B fragments are read first and kept)
    auto half_ktile = [&](auto hc, uint32_t tA, uint32_t tB) {
        constexpr int h = decltype(hc)::value;
        i32x8 fp[2];
        // (the twelve fragment addresses are one XOR and one add away from rdA / rdB; laundered per interval so that the compiler forms
        // them here instead of keeping twelve loop-invariant address registers alive across the K-loop)
        uint32_t ra = rdA, rb = rdB;
        asm volatile("" : "+v"(ra), "+v"(rb));
        auto fragA = [&](int cb) { return lds_read_tr8_quad_raw<1024>(tA + (ra ^ (uint32_t)(cb << 4))); };
        auto fragB = [&](int cb) { return lds_read_tr8_quad_raw<2048>(tB + (rb ^ (uint32_t)(cb << 4))); };
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fq[j] = fragB(wl * 4 + j);
        }
        fp[0] = fragA(h * 4 + 0);
        // the A fragments run one row tile (four MFMAs, 128 cycles) ahead
#define VIPANT_TN8_ROW(T)                                                                           \
        if (T + 1 < 4) fp[(T + 1) & 1] = fragA(h * 4 + T + 1);                                      \
        if (T + 1 < 4) lds_raw_wait<4>(); else lds_raw_wait<0>();                                   \
        lds_raw_use8(fp[T & 1]);                                                                    \
        if (T == 0 && h == 0) { lds_raw_use8(fq[0]); lds_raw_use8(fq[1]); lds_raw_use8(fq[2]); lds_raw_use8(fq[3]); } \
        VIPANT_TN8_MX(acc[h * 4 + T][0], fq[0], fp[T & 1], 0, (h * 4 + T) >> 1);                    \
        VIPANT_TN8_MX(acc[h * 4 + T][1], fq[1], fp[T & 1], 0, (h * 4 + T) >> 1);                    \
        VIPANT_TN8_MX(acc[h * 4 + T][2], fq[2], fp[T & 1], 1, (h * 4 + T) >> 1);                    \
        VIPANT_TN8_MX(acc[h * 4 + T][3], fq[3], fp[T & 1], 1, (h * 4 + T) >> 1);                    \
        __builtin_amdgcn_sched_barrier(0);
        VIPANT_TN8_ROW(0) VIPANT_TN8_ROW(1) VIPANT_TN8_ROW(2) VIPANT_TN8_ROW(3)
#undef VIPANT_TN8_ROW
    };

    // prologue: scales and K-tile 0 (and, for group 1, its B rows of K-tile 1)
    request_scales(kt0, raw);
    scales_ready(raw, sav, sbv);                             // awaited before any DMA is in flight
    scales_valid(kt0, sav, sbv);
    fill_a(0, 0);
    fill_b(0, 0);
    if (grp == 1) {
        fill_b(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the lag

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const uint32_t tA = baseA + (uint32_t)(stage * PP_A_STAGE), tB = baseB + (uint32_t)(slot * PP_B_SLOT);
        request_scales(kt0 + kt + 1, raw);               // older than this K-tile's DMA pieces
        fill_a(stage ^ 1, kt + 1);
        if (grp == 0) fill_b(slot1, kt + 1); else fill_b(slot2, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (do_colsum && cs_turn == 0) {
            const char* sA = smem + grp * (2 * PP_A_STAGE) + stage * PP_A_STAGE;
            const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)sav, wl * 16);         // token block wl: the four column blocks' bytes
            const float sc = __uint_as_float(((s4 >> (8 * (cs_c >> 3))) & 255u) << 23);        // 2^(byte - 127); byte 0 (an all-zero block): 0
            float part[4] = {0.f, 0.f, 0.f, 0.f};
            // (row m = cs_row0 + 2 i has swizzle (i & 3) | (i >> 3) << 2 -- a constant of the unrolled loop -- so the sixteen addresses
            // are one lane base XOR a constant, plus a row offset; the base is laundered here so that the compiler forms them inside this
            // branch instead of keeping sixteen loop-invariant address registers alive across the K-loop, which made it spill)
            int cs_base = cs_row0 * 128 + ((cs_c >> 2) << 4) + (cs_c & 3) * 4;
            asm volatile("" : "+v"(cs_base));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int v = *(const int*)(sA + ((cs_base ^ (((i & 3) | ((i >> 3) << 2)) << 4)) + 2 * i * 128));
                const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(v, false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8(v, true);
                part[0] += lo[0]; part[1] += lo[1]; part[2] += hi2[0]; part[3] += hi2[1];
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);          // four loads in flight at a time: registers, not latency, are scarce here
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) cs_acc[e] += part[e] * sc;
        }
        half_ktile(Int<0>{}, tA, tB);
        if (grp == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the next K-tile's scales: requested in front of this K-tile's eight DMA pieces, first used here -- the wait the compiler
        // puts in front of this is vmcnt(8), which leaves those pieces in flight (at the K-tile's end it was vmcnt(0))
        scales_ready(raw, sav_n, sbv_n);
        scales_valid(kt0 + kt + 1, sav_n, sbv_n);
        half_ktile(Int<1>{}, tA, tB);
        if (grp == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot1;
        sav = sav_n; sbv = sbv_n;
        cs_turn = cs_turn == 0 ? ntq - 1 : cs_turn - 1;
    }
#undef VIPANT_TN8_MX
    // the (unused) fills issued by the last K-tiles must have landed in this group's A stages before they are reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // column-sum partials: the group's own (now idle) A stages serve as the reduction buffer, [8 row groups][128 columns] fp32 = 4 KiB
    float* red = (float*)(smem + grp * (2 * PP_A_STAGE));
    const int tl = tid & 255;
    if (do_colsum) *(f32x4*)(red + (wl * 2 + (lane >> 5)) * 128 + cs_c * 4) = f32x4{cs_acc[0], cs_acc[1], cs_acc[2], cs_acc[3]};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 0) __builtin_amdgcn_s_barrier();          // pairs with group 1's last barrier
    if (do_colsum && tl < 128) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += red[r * 128 + tl];
        p.colsum[((int64_t)split * ntq + tq) * (ntp * TP) + p0 + grp * 128 + tl] = sum;
    }
    // lane holds C[p = p0 + grp*128 + i*16 + (lane&15)][q = q0 + wl*64 + j*16 + (lane>>4)*4 + 0..3]
    const int frow = lane & 15, g = lane >> 4;
    if (p.direct) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = p0 + grp * 128 + i * 16 + frow;
            if (pr >= p.P) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qc = q0 + wl * 64 + j * 16 + g * 4;
                if (qc < p.Q) *(f32x4*)(p.out + (int64_t)pr * p.ldo + qc) = acc[i][j];
            }
        }
    } else {
        float* slab = p.out + ((int64_t)split * ntiles + tile) * (TP * TQ);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(slab + (grp * 128 + i * 16 + frow) * TQ + wl * 64 + j * 16 + g * 4) = acc[i][j];
    }
}

__device__ __forceinline__ void tn_reduce_blocks(const float* slab, float* C, int64_t ldc, int P, int Q, int splits, int accumulate,
                                                 int block, int nblocks) {
    const int ntq = (Q + TQ - 1) / TQ, ntp = (P + TP - 1) / TP;
    const int ntiles = ntp * ntq;
    const int64_t total4 = (int64_t)ntiles * TP * TQ / 4;
    for (int64_t idx = (int64_t)block * blockDim.x + threadIdx.x; idx < total4; idx += (int64_t)nblocks * blockDim.x) {
        const int64_t e = idx * 4;
        const int tile = (int)(e / (TP * TQ));
        const int r = (int)(e % (TP * TQ)) / TQ, c = (int)(e % TQ);
        const int pr = (tile / ntq) * TP + r, qc = (tile % ntq) * TQ + c;
        if (pr >= P || qc >= Q) continue;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) s += *(const f32x4*)(slab + ((int64_t)k * ntiles + tile) * (TP * TQ) + r * TQ + c);
        float* dst = C + (int64_t)pr * ldc + qc;
        if (accumulate) s += *(const f32x4*)dst;
        *(f32x4*)dst = s;
    }
}

// blockIdx.y selects the problem of a pair (slab2 / C2: NULL for a single problem)
__global__ void gemm_tn_reduce_kernel(const float* slab, float* C, int64_t ldc, int P, int Q, int splits,
                                      int accumulate, const float* slab2 = nullptr, float* C2 = nullptr) {
    if (blockIdx.y == 1) { slab = slab2; C = C2; }
    tn_reduce_blocks(slab, C, ldc, P, Q, splits, accumulate, (int)blockIdx.x, (int)gridDim.x);
}

// out[i] (+)= sum over the `parts` partial vectors.  64 columns per 256-thread block, 4 threads per column each taking every
// fourth partial with independent loads in flight (a single thread walking 20-80 partials one after the other made this tiny
// kernel 20-90 us long).
__device__ __forceinline__ void colsum_reduce_block(const float* part, float* out, int P, int stride, int parts, int accumulate,
                                                    int block) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int i = block * 64 + c;
    float s0 = 0.f, s1 = 0.f;
    if (i < P) {
        int k = sub;
        for (; k + 4 < parts; k += 8) { s0 += part[(int64_t)k * stride + i]; s1 += part[(int64_t)(k + 4) * stride + i]; }
        if (k < parts) s0 += part[(int64_t)k * stride + i];
    }
    red[sub][c] = s0 + s1;
    __syncthreads();
    if (sub == 0 && i < P) {
        const float s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        out[i] = accumulate ? out[i] + s : s;
    }
}

__global__ __launch_bounds__(256) void gemm_tn_colsum_reduce_kernel(const float* part, float* out, int P, int stride, int parts,
                                                                    int accumulate) {
    colsum_reduce_block(part, out, P, stride, parts, accumulate, (int)blockIdx.x);
}

// the split reduction of C and the column-sum reduction of the same launch as ONE launch: blocks [0, nred) reduce C, the rest
// take 64 columns of the column sums each
__global__ __launch_bounds__(256) void gemm_tn_reduce_both_kernel(const float* slab, float* C, int64_t ldc, int P, int Q, int splits,
                                                                  int accumulate, int nred, const float* cs_part, float* cs_out,
                                                                  int cs_stride, int cs_parts) {
    if ((int)blockIdx.x < nred) tn_reduce_blocks(slab, C, ldc, P, Q, splits, accumulate, (int)blockIdx.x, nred);
    else colsum_reduce_block(cs_part, cs_out, P, cs_stride, cs_parts, accumulate, (int)blockIdx.x - nred);
}

void plan_tiles(int64_t M, int64_t tiles, int* splits, int* kt_per_split) {
    const int64_t nk = ceil_div(M, BK);
    int64_t s = 256 / tiles;
    if (s < 1) s = 1;
    if (s > nk) s = nk;
    const int64_t per = ceil_div(nk, s);
    *splits = (int)ceil_div(nk, per);
    *kt_per_split = (int)per;
}

// VIPANT_TN_SPLIT=k (experiment, default 1): k times as many, k times shorter workgroups per weight-gradient launch -- several per CU
// instead of one long one, so that CUs another stream's kernel holds (the replica group's all-reduce) cost a launch 1/k of a workgroup's
// time at its tail instead of a whole one; the price is k times the partial tiles to write and reduce (profiles/r6_comm_shadow_cfg5.md)
int64_t split_factor() {
    const char* e = getenv("VIPANT_TN_SPLIT");
    const int k = e ? atoi(e) : 1;
    return k < 1 ? 1 : (k > 8 ? 8 : k);
}

void plan(int64_t M, int64_t P, int64_t Q, int* splits, int* kt_per_split) {
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    const int64_t nk = ceil_div(M, BK);
    int64_t s = 256 / tiles * (tiles <= 256 ? split_factor() : 1);
    if (s < 1) s = 1;
    if (s > nk) s = nk;
    const int64_t per = ceil_div(nk, s);
    *splits = (int)ceil_div(nk, per);
    *kt_per_split = (int)per;
}

// the e4m3 kernel: K-tiles of 128 tokens
void plan8(int64_t M, int64_t P, int64_t Q, int* splits, int* kt_per_split) {
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    const int64_t nk = ceil_div(M, BK8);
    int64_t s = 256 / tiles * (tiles <= 256 ? split_factor() : 1);
    if (s < 1) s = 1;
    if (s > nk) s = nk;
    const int64_t per = ceil_div(nk, s);
    *splits = (int)ceil_div(nk, per);
    *kt_per_split = (int)per;
}

}  // namespace

extern "C" size_t vipant_gemm_tn_workspace_bytes(int64_t M, int64_t P, int64_t Q) {
    int splits, per;
    plan(M, P, Q, &splits, &per);
    return (size_t)splits * (size_t)(ceil_div(P, TP) * ceil_div(Q, TQ)) * TP * TQ * sizeof(float) +
           (size_t)splits * (size_t)ceil_div(Q, TQ) * (size_t)ceil_div(P, TP) * TP * sizeof(float);
}

extern "C" int32_t vipant_gemm_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C,
                                  int64_t ldc, int64_t M, int64_t P, int64_t Q, int32_t accumulate, float* a_colsum,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(M > 0 && P > 0 && Q > 0, VIPANT_EBADSHAPE, "gemm_tn: empty problem");
    VIPANT_REQUIRE(Q % 4 == 0, VIPANT_EBADSHAPE, "gemm_tn: need Q%%4==0 (P=%ld Q=%ld)",
                   (long)P, (long)Q);
    VIPANT_REQUIRE(lda >= P && ldb >= Q && ldc >= Q && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, VIPANT_EALIGN,
                   "gemm_tn: bad leading dims");
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), VIPANT_EALIGN,
                   "gemm_tn: operands must be 16-byte aligned");
    int splits, per;
    plan(M, P, Q, &splits, &per);
    VIPANT_REQUIRE((int64_t)(per + 3) * BK * (lda > ldb ? lda : ldb) * 2 < (1ll << 32), VIPANT_EBADSHAPE,
                   "gemm_tn: per-split byte range exceeds 4 GiB");
    const size_t need = vipant_gemm_tn_workspace_bytes(M, P, Q);
    const int direct = (splits == 1 && !accumulate) ? 1 : 0;
    if (!direct || a_colsum != nullptr)
        VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= need, VIPANT_ENOWORKSPACE,
                       "gemm_tn: workspace too small (%zu < %zu)", workspace_bytes, need);
    static DeviceOnce once;
    static int variant = 0;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * STAGE_BYTES));
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           PP_LDS_BYTES));
        variant = getenv("VIPANT_GEMM_VARIANT") ? atoi(getenv("VIPANT_GEMM_VARIANT")) : 0;     // bit 16: the two-stage kernel
        done_on_device(once);
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    const size_t slab_bytes = (size_t)splits * (size_t)tiles * TP * TQ * sizeof(float);
    float* cs_part = a_colsum != nullptr ? (float*)((char*)workspace + slab_bytes) : nullptr;
    GemmTN p{(const bf16_t*)A, (const bf16_t*)B, direct ? C : (float*)workspace, lda, ldb, direct ? ldc : TQ,
             (int)M, (int)P, (int)Q, splits, per, direct, cs_part};
    if (variant & 16) hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(tiles * splits)), dim3(512), 2 * STAGE_BYTES, s, p);
    else hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3((unsigned)(tiles * splits)), dim3(512), PP_LDS_BYTES, s, p);
    VIPANT_LAUNCH_CHECK();
    const int cs_parts = (int)(splits * ceil_div(Q, TQ)), cs_stride = (int)(ceil_div(P, TP) * TP);
    if (!direct) {
        const int64_t total4 = tiles * TP * TQ / 4;
        int blocks = (int)ceil_div(total4, 256);
        if (blocks > 2048) blocks = 2048;
        if (a_colsum != nullptr) {       // one launch for both reductions
            hipLaunchKernelGGL(gemm_tn_reduce_both_kernel, dim3((unsigned)(blocks + ceil_div(P, 64))), dim3(256), 0, s,
                               (const float*)workspace, C, ldc, (int)P, (int)Q, splits, accumulate, blocks, (const float*)cs_part, a_colsum,
                               cs_stride, cs_parts);
            VIPANT_LAUNCH_CHECK();
            return VIPANT_OK;
        }
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, C, ldc,
                           (int)P, (int)Q, splits, accumulate);
        VIPANT_LAUNCH_CHECK();
    }
    if (a_colsum != nullptr) {
        hipLaunchKernelGGL(gemm_tn_colsum_reduce_kernel, dim3((unsigned)ceil_div(P, 64)), dim3(256), 0, s,
                           (const float*)cs_part, a_colsum, (int)P, cs_stride, cs_parts, accumulate);
        VIPANT_LAUNCH_CHECK();
    }
    return VIPANT_OK;
}

// C[P, Q] (+)= dequant(A, sa)^T dequant(B, sb) on e4m3 operands whose MX scales are uniform over 32-token x 32-column blocks.
extern "C" size_t vipant_gemm_tn_e4m3_workspace_bytes(int64_t M, int64_t P, int64_t Q) {
    int splits, per;
    plan8(M, P, Q, &splits, &per);
    return (size_t)splits * (size_t)(ceil_div(P, TP) * ceil_div(Q, TQ)) * TP * TQ * sizeof(float) +
           (size_t)splits * (size_t)ceil_div(Q, TQ) * (size_t)ceil_div(P, TP) * TP * sizeof(float);
}

extern "C" int32_t vipant_gemm_tn_e4m3(const uint8_t* A, int64_t lda, const uint8_t* sa, const uint8_t* B, int64_t ldb, const uint8_t* sb,
                                       float* C, int64_t ldc, int64_t M, int64_t P, int64_t Q, int32_t accumulate, float* a_colsum,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(M > 0 && P > 0 && Q > 0, VIPANT_EBADSHAPE, "gemm_tn_e4m3: empty problem");
    VIPANT_REQUIRE(P % 128 == 0 && Q % 128 == 0, VIPANT_EBADSHAPE, "gemm_tn_e4m3: need P %% 128 == 0 and Q %% 128 == 0 (P=%ld Q=%ld)",
                   (long)P, (long)Q);
    VIPANT_REQUIRE(lda >= P && ldb >= Q && ldc >= Q && lda % 128 == 0 && ldb % 128 == 0 && ldc % 4 == 0, VIPANT_EALIGN,
                   "gemm_tn_e4m3: bad leading dims (lda, ldb: the full row length of the quantised matrices, a multiple of 128)");
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && ((uintptr_t)sa % 16 == 0) &&
                       ((uintptr_t)sb % 16 == 0) && sa != nullptr && sb != nullptr,
                   VIPANT_EALIGN, "gemm_tn_e4m3: operands and their block scales must be 16-byte aligned");
    int splits, per;
    plan8(M, P, Q, &splits, &per);
    VIPANT_REQUIRE((int64_t)(per + 3) * BK8 * (lda > ldb ? lda : ldb) < (1ll << 32), VIPANT_EBADSHAPE,
                   "gemm_tn_e4m3: per-split byte range exceeds 4 GiB");
    const size_t need = vipant_gemm_tn_e4m3_workspace_bytes(M, P, Q);
    const int direct = (splits == 1 && !accumulate) ? 1 : 0;
    if (!direct || a_colsum != nullptr)
        VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= need, VIPANT_ENOWORKSPACE,
                       "gemm_tn_e4m3: workspace too small (%zu < %zu)", workspace_bytes, need);
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn8_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
        done_on_device(once);
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    const size_t slab_bytes = (size_t)splits * (size_t)tiles * TP * TQ * sizeof(float);
    float* cs_part = a_colsum != nullptr ? (float*)((char*)workspace + slab_bytes) : nullptr;
    GemmTN8 p{A, B, sa, sb, direct ? C : (float*)workspace, lda, ldb, direct ? ldc : TQ, (int)M, (int)P, (int)Q, splits, per, direct, cs_part};
    hipLaunchKernelGGL(gemm_tn8_pp_kernel, dim3((unsigned)(tiles * splits)), dim3(512), PP_LDS_BYTES, s, p);
    VIPANT_LAUNCH_CHECK();
    const int cs_parts = (int)(splits * ceil_div(Q, TQ)), cs_stride = (int)(ceil_div(P, TP) * TP);
    if (!direct) {
        const int64_t total4 = tiles * TP * TQ / 4;
        int blocks = (int)ceil_div(total4, 256);
        if (blocks > 2048) blocks = 2048;
        if (a_colsum != nullptr) {       // one launch for both reductions
            hipLaunchKernelGGL(gemm_tn_reduce_both_kernel, dim3((unsigned)(blocks + ceil_div(P, 64))), dim3(256), 0, s,
                               (const float*)workspace, C, ldc, (int)P, (int)Q, splits, accumulate, blocks, (const float*)cs_part, a_colsum,
                               cs_stride, cs_parts);
            VIPANT_LAUNCH_CHECK();
            return VIPANT_OK;
        }
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, C, ldc, (int)P, (int)Q, splits,
                           accumulate);
        VIPANT_LAUNCH_CHECK();
    }
    if (a_colsum != nullptr) {
        hipLaunchKernelGGL(gemm_tn_colsum_reduce_kernel, dim3((unsigned)ceil_div(P, 64)), dim3(256), 0, s, (const float*)cs_part, a_colsum,
                           (int)P, cs_stride, cs_parts, accumulate);
        VIPANT_LAUNCH_CHECK();
    }
    return VIPANT_OK;
}

// Two contractions of the same shape in one launch: C0 = A0^T . B0, C1 = A1^T . B1 (no accumulate, no column sums).
extern "C" size_t vipant_gemm_tn_pair_workspace_bytes(int64_t M, int64_t P, int64_t Q) {
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    int splits, per;
    plan_tiles(M, 2 * tiles, &splits, &per);
    return 2 * (size_t)splits * (size_t)tiles * TP * TQ * sizeof(float);
}

extern "C" int32_t vipant_gemm_tn_pair(const uint16_t* A0, const uint16_t* B0, float* C0, const uint16_t* A1, const uint16_t* B1,
                                       float* C1, int64_t lda, int64_t ldb, int64_t ldc, int64_t M, int64_t P, int64_t Q,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(M > 0 && P > 0 && Q > 0 && Q % 4 == 0, VIPANT_EBADSHAPE, "gemm_tn_pair: bad shape");
    VIPANT_REQUIRE(lda >= P && ldb >= Q && ldc >= Q && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, VIPANT_EALIGN,
                   "gemm_tn_pair: bad leading dims");
    VIPANT_REQUIRE(((uintptr_t)A0 % 16 == 0) && ((uintptr_t)B0 % 16 == 0) && ((uintptr_t)C0 % 16 == 0) && ((uintptr_t)A1 % 16 == 0) &&
                       ((uintptr_t)B1 % 16 == 0) && ((uintptr_t)C1 % 16 == 0), VIPANT_EALIGN, "gemm_tn_pair: operands must be 16-byte aligned");
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    int splits, per;
    plan_tiles(M, 2 * tiles, &splits, &per);
    VIPANT_REQUIRE((int64_t)(per + 3) * BK * (lda > ldb ? lda : ldb) * 2 < (1ll << 32), VIPANT_EBADSHAPE,
                   "gemm_tn_pair: per-split byte range exceeds 4 GiB");
    VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= vipant_gemm_tn_pair_workspace_bytes(M, P, Q), VIPANT_ENOWORKSPACE,
                   "gemm_tn_pair: workspace too small");
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_pp_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
        done_on_device(once);
    }
    hipStream_t s = (hipStream_t)stream;
    const int direct = splits == 1 ? 1 : 0;
    float* slab0 = (float*)workspace;
    float* slab1 = slab0 + (size_t)splits * (size_t)tiles * TP * TQ;
    GemmTNPair pr;
    pr.a = GemmTN{(const bf16_t*)A0, (const bf16_t*)B0, direct ? C0 : slab0, lda, ldb, direct ? ldc : TQ, (int)M, (int)P, (int)Q, splits, per,
                  direct, nullptr};
    pr.b = GemmTN{(const bf16_t*)A1, (const bf16_t*)B1, direct ? C1 : slab1, lda, ldb, direct ? ldc : TQ, (int)M, (int)P, (int)Q, splits, per,
                  direct, nullptr};
    pr.blocks_a = (int)(tiles * splits);
    hipLaunchKernelGGL(gemm_tn_pp_pair_kernel, dim3((unsigned)(2 * tiles * splits)), dim3(512), PP_LDS_BYTES, s, pr);
    VIPANT_LAUNCH_CHECK();
    if (!direct) {
        const int64_t total4 = tiles * TP * TQ / 4;
        int blocks = (int)ceil_div(total4, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks, 2), dim3(256), 0, s, (const float*)slab0, C0, ldc, (int)P, (int)Q, splits, 0,
                           (const float*)slab1, C1);
        VIPANT_LAUNCH_CHECK();
    }
    return VIPANT_OK;
}
