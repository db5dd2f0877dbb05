// dev_bench.hip -- on-box peak micro-benchmarks ("peaks = vendor nominal AND measured on the box (stream-copy
// kernel; MFMA peak micro-benchmark), both stated").  Development entry points (include/emdenoise_dev.h): bench.py times them
// with HIP events and prints the figures next to the nominal 8 TB/s / 2.5 PFLOP/s; no product path calls them.
#include "mfma_common.hpp"

using namespace emd;

namespace {

// Plain stream copy, ONE float4 per lane and no loop: the workgroups in flight cover one contiguous window that moves through the
// buffer, which is what the DRAM pages like -- the practical HBM roof of a read + write kernel (6.1-6.2 TB/s at 1 GiB each way).
// A grid-stride loop over the same bytes (every lane touching addresses a whole grid apart, the form this benchmark had first)
// reaches 3.5-5.0 TB/s: tools/membench.hip, tools/membench2.hip.
__global__ void __launch_bounds__(256) stream_copy_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ b, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) b[i] = a[i];
}

// Back-to-back v_mfma_f32_32x32x16_bf16 on register operands, 4 independent accumulators per wave, one wave per SIMD when
// launched with 256 threads x (number of CUs) workgroups: the matrix pipe's issue rate at the clock the chip holds under this
// load.  Operands come from memory (random data: an all-zero operand lets the chip hold a higher clock than real work does).
__global__ void __launch_bounds__(256) mfma_peak_kernel(const bf16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a0 = ops[lane], a1 = ops[64 + lane], b0 = ops[128 + lane], b1 = ops[192 + lane];
    f32x16 c0, c1, c2, c3;
#pragma unroll
    for (int e = 0; e < 16; ++e) c0[e] = c1[e] = c2[e] = c3[e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;   // keeps the chain alive; practically never taken
}

}  // namespace

extern "C" int emd_debug_stream_copy_f32(const float* src, float* dst, long n, emd_stream_t stream) {
    EMD_REQUIRE(src && dst && n > 0 && n % 4 == 0, EMD_E_INVALID, "emd_debug_stream_copy_f32: n must be a positive multiple of 4");
    EMD_REQUIRE(emd::aligned16(src) && emd::aligned16(dst), EMD_E_ALIGN, "emd_debug_stream_copy_f32: 16-byte alignment");
    const long n4 = n / 4;
    const long blocks = (n4 + 255) / 256;
    EMD_REQUIRE(blocks <= 0x7fffffffL, EMD_E_UNSUPPORTED, "emd_debug_stream_copy_f32: more than 2^31 workgroups");
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), n4);
    return emd::check_launch("stream_copy_kernel");
}

// ops: 256 x 16 bytes of bf16 operands (device); out: >= workgroups * 256 floats (never written in practice).
// Issues workgroups * 4 waves * iters * 32 MFMAs of 32*32*16*2 flop each.
extern "C" int emd_debug_mfma_peak_bf16(const void* ops, float* out, int workgroups, int iters, emd_stream_t stream) {
    EMD_REQUIRE(ops && out && workgroups > 0 && iters > 0, EMD_E_INVALID, "emd_debug_mfma_peak_bf16: bad arguments");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3((unsigned)workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16x8*>(ops), out, iters);
    return emd::check_launch("mfma_peak_kernel");
}
