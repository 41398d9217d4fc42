// Helpers of the attention kernels (attention.hip): the argument block, the softmax constants, the wave-uniform buffer descriptor
// and the inline-asm LDS-DMA pieces.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace vipant_attn {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCALE = 0.125f;              // 1/sqrt(64)
constexpr float C2 = SCALE * LOG2E;

struct MhaArgs {
    const bf16_t* qkv; bf16_t* out; float* lse;
    const bf16_t* dout; float* delta; bf16_t* dqkv;
    int batch, S, H;
    int stagger;
    uint32_t* tk = nullptr;            // ticket counters of the stream (common.h) for the persistent backward; NULL = static walk
    uint32_t* tk_other = nullptr;      // the stream's other counter set, zeroed by this launch
    // e4m3 + MX block scales of the result beside its bf16 form (BASELINE configs[4]: the next contraction's operand, common.h):
    uint8_t* gq = nullptr; uint8_t* gq_scale = nullptr;        // backward: `dqkv` [M, 3 D] (the dK | dV columns; dQ by a pass of its own)
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, uint32_t bytes) {
    // descriptor inputs made provably wave-uniform, otherwise hipcc wraps every buffer op in a waterfall loop
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return make_rsrc((const void*)(((uint64_t)hi << 32) | lo), (uint32_t)__builtin_amdgcn_readfirstlane(bytes));
}
// LDS-DMA through inline asm: invisible to hipcc's wait insertion.  (With a builtin piece in flight every transposed-read builtin --
// no memory operand: "may alias" -- gets an s_waitcnt vmcnt(0) in front; inside a loop that streams its operands that is an HBM
// round trip per region.)  The caller owns the vmcnt wait and the barrier that publish the bytes.  M0 carries the LDS base.
__device__ __forceinline__ void lds_dma16_asm(__amdgpu_buffer_rsrc_t rs, const void* lds_base, uint32_t voff, uint32_t soff) {
    const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_offset(lds_base));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(la), "v"(voff), "s"(rs), "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)soff)) : "memory", "m0");
}

__device__ __forceinline__ void lds_dma4_asm(__amdgpu_buffer_rsrc_t rs, const void* lds_base, uint32_t voff, uint32_t soff) {
    const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_offset(lds_base));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(la), "v"(voff), "s"(rs), "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)soff)) : "memory", "m0");
}

#endif

}  // namespace vipant_attn
