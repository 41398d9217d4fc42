// Shared device/host helpers for libvipant_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vipant_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LDS_AS __attribute__((address_space(3)))
#define WAVE 64

// ---- host side error plumbing -------------------------------------------------------------------
void vipant_set_error(const char* fmt, ...);

#define VIPANT_HIP_TRY(expr)                                                                    \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            vipant_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                             __LINE__);                                                         \
            return VIPANT_EHIP;                                                                 \
        }                                                                                       \
    } while (0)

#define VIPANT_REQUIRE(cond, code, ...)   \
    do {                                  \
        if (!(cond)) {                    \
            vipant_set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

#define VIPANT_LAUNCH_CHECK()                                                               \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) {                                                             \
            vipant_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),     \
                             __FILE__, __LINE__);                                           \
            return VIPANT_EHIP;                                                             \
        }                                                                                   \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// One-time set-up of a launcher (the dynamic-LDS opt-in of its kernels, hipFuncSetAttribute) is a property of the DEVICE, not of the
// process: bit d of the launcher's mask = done on device d (one process per GPU is how the package runs, but nothing here may
// break the day one process drives two).  Races between host threads are benign: the set-up is idempotent.
struct DeviceOnce { uint64_t mask[2] = {0, 0}; };
// true until done_on_device() has been called for the current device: the caller runs its set-up and marks it done only when every
// step succeeded (a failed hipFuncSetAttribute returns through VIPANT_HIP_TRY before the mark, so the next call tries again instead
// of launching a kernel whose LDS opt-in never happened)
static inline bool first_on_device(const DeviceOnce& o) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    return !(o.mask[(dev >> 6) & 1] & (1ull << (dev & 63)));
}
static inline void done_on_device(DeviceOnce& o) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) o.mask[(dev >> 6) & 1] |= 1ull << (dev & 63);
}
// the same for a launcher whose LDS request grows with the problem: true when `bytes` exceeds what this device was set up for;
// raised_on_device() records the new value after the set-up succeeded
struct DeviceMax { int v[128] = {}; };
static inline bool raise_on_device(const DeviceMax& o, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    return bytes > o.v[dev & 127];
}
static inline void raised_on_device(DeviceMax& o, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && bytes > o.v[dev & 127]) o.v[dev & 127] = bytes;
}
// CUs of the current device (one persistent workgroup per CU), queried once per device
static inline int device_cus() {
    static int cus[128] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 127];
    if (c <= 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 0;
    return c > 0 ? c : 256;
}

// ---- MX block scales of an e4m3 activation operand (round 5) --------------------------------------
// The A operand [M, K] of vipant_gemm_nt_e4m3 carries one power-of-two scale (E8M0 byte, value 2^(byte - 127)) per 32 consecutive k of a
// row -- the block format v_mfma_scale_f32_16x16x128_f8f6f4 takes -- so that a PRODUCER can quantise what it has in hand (a tile
// epilogue's 32 columns, a LayerNorm lane group) without knowing the rest of the row.  The bytes are stored in the order the
// contraction's waves consume them: for the 128-row group G = m / 128 and the K-tile kt = k / 128, lane (r = m % 16, q = (k % 128) / 32)
// of a wave finds the bytes of its eight 16-row tiles i = (m % 128) / 16 as ONE 8-byte word -- 512 contiguous bytes per wave and K-tile.
// Size: ceil(M / 128) * (K / 128) * 512 bytes (K % 128 == 0); bytes of rows >= M are never used for a stored result.
__host__ __device__ static inline int64_t mx_scale_offset(int64_t m, int kb /* = k / 32 */, int kt_per_row /* = K / 128 */) {
    return (((m >> 7) * kt_per_row + (kb >> 2)) * 16 + (m & 15)) * 32 + (kb & 3) * 8 + ((m >> 4) & 7);
}
static inline size_t mx_scale_bytes(int64_t M, int64_t K) { return (size_t)ceil_div(M, 128) * (size_t)(K / 128) * 512; }

// ---- ticket walk of the persistent kernels ------------------------------------------------------
// A persistent grid that walks its tiles with a static stride assumes that all of its workgroups start together: a workgroup that
// finds its CU held by another stream's kernel (the RCCL all-reduce of a gradient bucket, vipant_amd/parallel.py) starts when
// that kernel ends and then still owns 1/256 of the launch.  With tickets a late workgroup takes what is left.  The block is
// 4 KiB of device memory per (device, stream): words [0, 8) and [8, 16) = two sets of eight queue heads (one queue per XCD:
// workgroup b draws from queue b & 7, whose positions map to the tiles the static walk gave that XCD, so the L2 locality of the
// walk is kept, and only workgroups of one XCD ever touch a counter), words [16, 16 + 2 * 256) = two mailbox words per workgroup
// through which its first wave hands a ticket to the other seven (the kernels' LDS is full).  The ticket launches of a stream
// alternate between the two sets and each zeroes the set it does NOT use: launches of a stream are ordered, so that set's last
// user is complete and its next user has not started -- no "last workgroup" has to be found, nothing is done between launches.
// Concurrent streams have their own blocks.
constexpr int VIPANT_TICKET_MBOX = 16, VIPANT_TICKET_WORDS = 1024;
uint32_t* vipant_ticket_block(hipStream_t stream, uint32_t** other);        // nullptr on failure (vipant_last_error is set)

// ---- device helpers -----------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// the shared exponent of a block: the smallest e with amax / 2^e <= 448 (the largest e4m3 value); `inv` = 2^-e
__device__ __forceinline__ int mx_exponent(float amax, float* inv) {
    int e = 0;
    if (amax > 0.f) {
        e = (int)((__float_as_uint(amax) >> 23) & 255u) - 127 - 8;            // floor(log2(amax)) - 8: amax / 2^e in [256, 512)
        if (amax * __uint_as_float((uint32_t)(127 - e) << 23) > 448.f) e += 1;
        e = e < -127 ? -127 : (e > 127 ? 127 : e);
    }
    *inv = __uint_as_float((uint32_t)(127 - e) << 23);
    return e;
}
// maximum over the 2 / 4 / 8 consecutive lanes that share a block (DPP row operations: no LDS traffic)
template <int NL>
__device__ __forceinline__ float mx_lane_max(float v) {
    static_assert(NL == 2 || NL == 4 || NL == 8, "lanes per block");
#define VIPANT_MAXD(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false)))
    VIPANT_MAXD(0xB1);                          // quad_perm [1, 0, 3, 2]: xor 1
    if (NL >= 4) VIPANT_MAXD(0x4E);             // quad_perm [2, 3, 0, 1]: xor 2
#undef VIPANT_MAXD
    if (NL == 8) v = fmaxf(v, __shfl_xor(v, 4, 64));
    return v;
}
__device__ __forceinline__ int mx_pack4(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
// The same quantiser on values that ARE bf16 (every producer rounds to bf16 first), in integer / packed arithmetic -- the two-output
// epilogues are bound by their own VALU work: |x| as a 15-bit integer orders like the value, so the block maximum is an AND and a
// packed 16-bit max per two elements; the exponent comes from its bit fields (amax / 2^e in [256, 512): above 448 exactly when the
// 7-bit mantissa exceeds 0x60); v_cvt_scalef32_pk_fp8_bf16 divides two bf16 by the scale and rounds to e4m3 in one instruction.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mx_absmax2(uint32_t acc, uint32_t two_bf16) {       // packed max of |.| over two more elements
    const uint32_t a = two_bf16 & 0x7FFF7FFFu;
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, acc), __builtin_bit_cast(u16x2, a)));
}
template <int NL>
__device__ __forceinline__ uint32_t mx_lane_max_u(uint32_t v) {
#define VIPANT_MAXU(ctrl) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xF, 0xF, false))
    VIPANT_MAXU(0xB1);
    if (NL >= 4) VIPANT_MAXU(0x4E);
#undef VIPANT_MAXU
    if (NL == 8) v = max(v, (uint32_t)__shfl_xor((int)v, 4, 64));
    return v;
}
// maximum over the whole wave of a non-negative value (or of |float| bits, which order like the value): four DPP row operations and
// four readlanes -- no LDS traffic, and the result is wave-uniform (a scalar for what follows)
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
#define VIPANT_MAXU(ctrl) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xF, 0xF, false))
    VIPANT_MAXU(0xB1);                          // quad_perm [1, 0, 3, 2]
    VIPANT_MAXU(0x4E);                          // quad_perm [2, 3, 0, 1]
    VIPANT_MAXU(0x141);                         // row_half_mirror: the other quad of the 8
    VIPANT_MAXU(0x140);                         // row_mirror: the other 8 of the 16
#undef VIPANT_MAXU
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
// the block's maximum |.| as bf16 bits -> its scale byte (E8M0) and the float 2^e the conversion divides by
// (round 6: an ALL-ZERO block carries the smallest scale there is, byte 0 = 2^-127, and converts with 2^0 -- not byte 127 as a block
// whose largest element is ~1 does: vipant_mx_uniform32 takes the maximum of 32 rows' scales, and a row of zeros -- 960 of the 992 rows
// of the top block's stream gradient in a step whose read-out takes one row per item -- must not set it)
__device__ __forceinline__ uint32_t mx_scale_of_max(uint32_t m, float* scale) {
    if (m == 0u) {
        *scale = 1.0f;
        return 0u;
    }
    int e = (int)(m >> 7) - 127 - 8 + ((m & 0x7Fu) > 0x60u ? 1 : 0);
    e = e < -127 ? -127 : e;                        // (a bf16 below 2^127 never needs e > 119)
    const uint32_t byte = (uint32_t)(e + 127);
    *scale = __uint_as_float(byte ? byte << 23 : 0x00400000u);          // 2^e (2^-127 is a subnormal)
    return byte;
}
// packed |.| maxima of a lane -> the block's scale byte over NL adjacent lanes
template <int NL>
__device__ __forceinline__ uint32_t mx_scale_byte(uint32_t packed_max, float* scale) {
    uint32_t m = max(packed_max & 0xFFFFu, packed_max >> 16);
    m = mx_lane_max_u<NL>(m);
    return mx_scale_of_max(m, scale);
}
__device__ __forceinline__ int mx_pack4_bf16(uint32_t lo2, uint32_t hi2, float scale) {      // four bf16 (two packed words) -> four e4m3 bytes
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(bf16x2, lo2), scale, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(bf16x2, hi2), scale, true);
    return __builtin_bit_cast(int, w);
}

namespace tickets {
__device__ __forceinline__ void put(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The draw and the mailbox store on a GLOBAL address carried as an integer in vector registers (a pointer laundered through
// `asm("" : "+v")` loses its address space: hipcc then emits flat accesses, which wait on vmcnt AND lgkmcnt and make it put vmcnt(0) in
// front of LDS reads).  The mailbox is written and read by waves of ONE workgroup (one CU, one vector L1): plain accesses --
// wavefront-scope relaxed atomics are `global_store` / `global_load` without cache-policy bits (`volatile` would make them
// system-scope flat accesses; an agent-scope read, which bypasses the L1, cost +45 us on a 870 us launch) -- ordered by their
// distance: the store goes out at the start of an epilogue, the word is read behind the NEXT epilogue's stores (csrc/gemm_nt.hip).
typedef __attribute__((address_space(1))) uint32_t gu32_t;
__device__ __forceinline__ uint32_t take_g(uint64_t a, uint32_t n) {     // n consecutive positions per draw
    return __hip_atomic_fetch_add((gu32_t*)a, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void post_g(uint64_t a, uint32_t v) { __hip_atomic_store((gu32_t*)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
// position s of queue q -> the tile the static walk of a 256-workgroup grid gave workgroup q + 8 (s & 31) in its round s >> 5
__device__ __forceinline__ int tile_of(int q, int s) { return (s >> 5) * 256 + q * 32 + (s & 31); }
}  // namespace tickets

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }

__device__ __forceinline__ bf16x4 f32x4_to_bf16x4(f32x4 v) {
    bf16x4 r;
    r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    return r;
}

// sigmoid(1.702 u), the gate of QuickGELU (clip/model.py:163-165), as v_exp_f32 + v_rcp_f32 (about 1 ulp each; the outputs are
// rounded to bf16 anyway).  `__frcp_rn(x)` / `1.0f / x` expand to the IEEE division sequence -- 11 instructions per element,
// which made the two-output epilogues VALU-bound (c_fc: 1167 -> 850 us with the arithmetic removed, round-2 probe).
__device__ __forceinline__ float quickgelu_gate(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * u));      // 1.702 * log2(e)
}

// 128-bit buffer resource over [base, base+bytes): out-of-range lanes of a buffer load return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// One LDS-DMA instruction: each lane copies 16 B from rsrc[voff + soff] to lds_base + lane*16.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, uint32_t voff,
                                          uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_base, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ bf16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)p);
}

// The same transposed read issued through inline asm.  hipcc treats the builtin above as "may alias any LDS-DMA in flight"
// and puts `s_waitcnt vmcnt(0)` in front of it, which makes a kernel that has just issued the NEXT stage's LDS-DMA wait for
// it before reading the CURRENT stage (plain ds_read_b128 loads do not have this problem).  The asm form is invisible to that
// analysis, so the caller owns both waits: the LDS-DMA that filled the bytes must be complete (vmcnt + barrier), and the result
// must be awaited with lds_raw_wait<N>() + lds_raw_use() before it is consumed.
typedef int v2i32_t __attribute__((ext_vector_type(2)));
typedef int v4i32_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_offset(const void* p) { return (uint32_t)(uintptr_t)(LDS_AS const char*)p; }
__device__ __forceinline__ bf16x8 lds_read_tr16_pair_raw(uint32_t addr_lo, uint32_t addr_hi) {
    v2i32_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr_lo));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(addr_hi));
    v4i32_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return __builtin_bit_cast(bf16x8, r);
}
// wait until at most N LDS operations of this wave are outstanding (they complete in order)
template <int N>
__device__ __forceinline__ void lds_raw_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// order a fragment's consumers behind the preceding lds_raw_wait (asm volatile statements keep their order; this one
// "produces" the value the MFMA reads)
__device__ __forceinline__ void lds_raw_use(bf16x8& f) {
    v4i32_t t = __builtin_bit_cast(v4i32_t, f);
    asm volatile("" : "+v"(t));
    f = __builtin_bit_cast(bf16x8, t);
}

// XCD-aware bijective remap of a linear block id: blocks id, id+8, ... share an XCD, give each XCD a
// contiguous run of tiles so neighbouring tiles (shared operand panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int xcd = id & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}
#endif
