// 16-bit MFMA helpers shared by the fused GP evaluation kernels (gp_eval_bf16.hip, gp_eval_compat_mfma.hip): fragment
// types, fp16 packing, and the inline-asm LDS-DMA whose completion the kernels count by hand.
#pragma once
#include "gp_common.hpp"

namespace scasml {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float v, uint32_t &h, uint32_t &m, uint32_t &l) {
    h = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2) & 0xFFFF0000u;
}

union Frag {
    s16x8 v;
    h16x8 h;
    uint32_t u[4];
    float4 f;
};

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {   // two fp16 (RNE) in one dword, a low
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    return (uint32_t)__builtin_bit_cast(unsigned short, ha) | ((uint32_t)__builtin_bit_cast(unsigned short, hb) << 16);
}

// One 16-byte-per-lane LDS-DMA (global_load_lds_dwordx4) issued from inline asm.  hipcc models the builtin
// form as an LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read, i.e. it drains the DMA the
// moment it was issued; an asm statement is opaque to that pass, so completion is counted by hand (counted
// vmcnt + raw s_barrier in `rendezvous`).  M0 carries the wave-uniform LDS byte address and is saved/restored
// inside the same statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16_asm(const float *gsrc_uniform, uint32_t lane_byte_offset, uint32_t lds_byte_addr_uniform) {
    // SADDR form: wave-uniform 64-bit base in an SGPR pair + a 32-bit per-lane byte offset, so the per-tile address
    // arithmetic is scalar (a 64-bit VGPR pointer per chunk costs ~7 VALU instructions in an issue-bound kernel)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_offset), "s"(gsrc_uniform), "s"(lds_byte_addr_uniform)
                 : "memory");
}

}  // namespace scasml
