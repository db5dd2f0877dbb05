// Device helpers shared by the two LDS-DMA pipelined fused separable-conv kernels (sep_pipe.hip: two barriers per 32-channel chunk,
// the phases in lockstep; sep_pipe2.hip: the depthwise stage of the NEXT half chunk issued between the MFMAs of the current one).
#pragma once

#include "sep_params.hpp"

namespace emd {
namespace sp {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Epilogue accesses: scalar base (a pixel of the tile row, uniform) + 32-bit lane offset, 16 bytes per lane; the stores non-temporal
// (the outputs are not re-read by this launch: L2 is kept for the patch halos).  Inline asm: the address form costs one VGPR per lane
// instead of a 64-bit pointer per access, and the loads are waited for by hand (wait_vm) -- the compiler does not see them.
// 16 bytes per lane, not 4: tools/dmabench.hip measures dword stores of this shape at 2.9 TB/s against 6.0 for dwordx4.
// The s_nop covers the store-data hazard the compiler cannot see inside the asm: a store of more than 64 bits reads its data late, and a
// VALU write to those registers must stay >= 2 wait states behind it on this ISA (LLVM's hazard recognizer: 2 for gfx940+).  Rounds 1-3
// had "s_nop 0" (ONE wait state), which held while the compiler happened to put other work first; round 4's epilogue with plain residual
// loads got a v_lshl_add_u64 (the next load's address, allocated INTO the dead store-data registers) two instructions behind the
// store, and the last four lanes of every 16 stored the address words instead of two channels -- intermittently, 64 values per tile
// (tools/sep2_debug.py).  s_nop 3 = four wait states, in every copy of this helper (conv3_pipe.hip, deconv_pipe.hip, mfma_common.hpp).
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 3" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 3" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ f32x4 load_s(const void* sbase, unsigned voff) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
__device__ __forceinline__ float load_s1(const void* sbase, unsigned voff) {
    float v;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
// One dword per lane (EPI = 1: a lane = one channel, the layout the MFMA leaves): no data hazard to cover, no transpose before it.
__device__ __forceinline__ void store_nt_d(const void* sbase, unsigned voff, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ float dpp_f(float v, int ctrl_is_xor2) {
    return ctrl_is_xor2 ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true))
                        : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
// 4 x 4 transpose inside a lane quad (two DPP exchange rounds): in, lane i holds column i of the block (r[k] = a[k][i]); out, row i
// (r[k] = a[i][k]).  It turns the MFMA C/D layout (a lane = one channel, four pixels) into one pixel's four consecutive channels.
__device__ __forceinline__ void quad_transpose(float (&r)[4], int li) {
    const bool b0 = li & 1, b1 = li & 2;
    float s0 = b0 ? r[0] : r[1], s1 = b0 ? r[2] : r[3];
    s0 = dpp_f(s0, 0);
    s1 = dpp_f(s1, 0);
    r[0] = b0 ? s0 : r[0]; r[1] = b0 ? r[1] : s0;
    r[2] = b0 ? s1 : r[2]; r[3] = b0 ? r[3] : s1;
    float t0 = b1 ? r[0] : r[2], t1 = b1 ? r[1] : r[3];
    t0 = dpp_f(t0, 1);
    t1 = dpp_f(t1, 1);
    r[0] = b1 ? t0 : r[0]; r[2] = b1 ? r[2] : t0;
    r[1] = b1 ? t1 : r[1]; r[3] = b1 ? r[3] : t1;
}
// the value of the lane 4 further on (lanes of an even channel quad) / 4 back (odd quad): the split32 pair exchange
__device__ __forceinline__ unsigned xchg4(unsigned v, bool oddq) {
    const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true);   // row_shl:4
    const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    return oddq ? dn : up;
}

}  // namespace sp
}  // namespace emd
