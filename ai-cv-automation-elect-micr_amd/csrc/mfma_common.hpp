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
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int kBK = 64;       // channel padding unit of the packed weights (and the GEMM's K step)
constexpr int kNPadTo = 128;  // packed weights are padded to a multiple of the widest BN

// a = hi + lo (+ O(2^-17)): two packed bf16 words for two floats
__device__ __forceinline__ void split2(float a0, float a1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a0, a1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    const bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

}  // namespace emd
