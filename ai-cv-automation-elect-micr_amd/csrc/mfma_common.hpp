// Shared device helpers of the matrix-core kernels (gemm_conv.hip, sep_fused.hip).
#pragma once

#include "emd_common.hpp"

namespace emd {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;      // native vectors for the staging registers:
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // HIP's float4/uint4 structs end up in scratch

// 16-byte global store with the non-temporal hint.  Inline asm on purpose: behind a run-time flag, "if (nt) __builtin_nontemporal_store
// else plain store" is merged into ONE plain store by the optimizer (the merged store keeps only the metadata both sides share).
// The s_nop covers the store-data hazard the compiler cannot see inside the asm.
__device__ __forceinline__ void store_nt16(void* dst, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 3" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store_nt16(void* dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 3" ::"v"(dst), "v"(v) : "memory");
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int kBK = 64;       // channel padding unit of the packed weights (and the GEMM's K step)
constexpr int kNPadTo = 128;  // packed weights are padded to a multiple of the widest BN

// a = hi + lo (+ O(2^-17)): two packed bf16 words for two floats
// q = v / d, rem = v % d for the index decompositions of the element-wise kernels.  A 64-bit integer division is a ~100
// instruction emulation on this ISA and those kernels do three of them per 16-byte output (the bilinear resize of graph G
// ran at 1.4 TB/s because of it); an index below 2^32 -- every shape of the reference's graphs -- takes the 32-bit form.
__device__ __forceinline__ long divmod(long v, int d, int& rem) {
    if (((unsigned long)v >> 32) == 0) {
        const unsigned q = (unsigned)v / (unsigned)d;
        rem = (int)((unsigned)v - q * (unsigned)d);
        return (long)q;
    }
    const long q = v / d;
    rem = (int)(v - q * d);
    return q;
}

__device__ __forceinline__ void split2(float a0, float a1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a0, a1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    const bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// The value of the other lane of an (even, odd) lane pair: a DPP quad permute [1,0,3,2] -- what __shfl_xor(v, 1) returns, without
// the trip through the LDS crossbar (ds_bpermute) that the HIP shuffle compiles to.
__device__ __forceinline__ unsigned swap_pair(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// Output stage of the depthwise kernels.  SPLIT = false: 4 fp32 channels at y + pix*ldy + 4*c4.  SPLIT = true: the
// split32 layout consumed by emd_conv1x1_split32_f32 (gemm_split.hip): the value is split into bf16 hi + lo here, once,
// instead of in every N-tile of the GEMM; pixel pitch ldy 4-byte units, channel group g = c/32 at byte 128 g:
// 32 x hi | 32 x lo.  Threads with c4 >= C4 (the padding up to a multiple of 32 channels) store zeros.
template <bool SPLIT>
__device__ __forceinline__ void dw_store(float* __restrict__ y, long pix, int ldy, int c4, float4 v) {
    if (!SPLIT) {
        *reinterpret_cast<float4*>(y + pix * ldy + c4 * 4) = v;
    } else {
        // 16-byte stores: the two lanes of an (even, odd) pair of channel quads swap halves -- the even lane stores both
        // quads' hi words, the odd lane both quads' lo words (8-byte stores cost 1606 vs 1365 us on the 256^2 x 384 maps).
        // Callers keep both lanes of a pair active together (the quad count per pixel is even, and so is every early exit).
        unsigned h0, l0, h1, l1;
        split2(v.x, v.y, h0, l0);
        split2(v.z, v.w, h1, l1);
        const bool odd = c4 & 1;
        const unsigned r0 = emd::swap_pair(odd ? h0 : l0), r1 = emd::swap_pair(odd ? h1 : l1);
        unsigned char* g = reinterpret_cast<unsigned char*>(y) + pix * (long)ldy * 4 + (c4 >> 3) * 128;
        if (!odd) *reinterpret_cast<u32x4*>(g + (c4 & 7) * 8) = u32x4{h0, h1, r0, r1};
        else *reinterpret_cast<u32x4*>(g + 64 + ((c4 - 1) & 7) * 8) = u32x4{r0, r1, l0, l1};
    }
}
}  // namespace emd
