// Backward ("training") kernels of graph D' (misc_py/denoiser-multi-gpu.py:752-782 tf.gradients of the tower
// loss; :1011-1077 the Nesterov train op).  Data gradients of the convolutions reuse the forward implicit
// GEMM (gemm_conv.hip) with transposed packed weights; this file holds what has no forward twin:
//   emd_conv_wgrad_f32        dW[t][k][n] = sum_m A[src_t(m)][k] * dY[m][n]   (any 1x1 / 3x3 / transposed conv)
//   emd_pack_weights_dev      fp32 weights on the device -> bf16 hi/lo planes, either orientation, any tap subset
// (bn_train.hip: training-mode batch norm forward fold / backward; bwd_misc.hip: depthwise, resize, pooling,
// 1-channel conv backward, the loss and the optimizer step.)
// Numerics: fp32 VALU, fp32 accumulation per block, float atomics across blocks.
#include "mfma_common.hpp"

using namespace emd;

namespace {

// ------------------------------------------------------------------------------------------------
// Weight gradient.  Block = 256 threads, output tile 64 (k) x 64 (n), thread = 4x4 micro-tile; the block walks
// its slice of M in chunks of 32 positions staged in LDS (A chunk [32][64], dY chunk [32][64]); per position a
// thread does 2 ds_read_b128 and 16 FMAs.  Slices of M are combined with float atomics into dW (zeroed by the
// caller).  Row map as in gemm_conv.hip: m -> (b,i,j); A is read at (i*sa+dy_t, j*sa+dx_t), dY at m.
struct WgradParams {
    const float* A;   // [.., lda]  activations (or, for the transposed conv, the upstream gradient image)
    const float* dY;  // [M, ldd]   gradient w.r.t. the conv output (or, for the transposed conv, its input x)
    float* dW;        // [taps][K][N]
    long M;
    int K, N, lda, ldd, ntaps, msplit;
    int flat, Hg, Wg, Ha, Wa, sa;
    unsigned long long dyp, dxp;
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int TK = 64, TN = 64, MC = 32;
    __shared__ __attribute__((aligned(16))) float As[MC][TK + 4];
    __shared__ __attribute__((aligned(16))) float Ds[MC][TN + 4];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * TK, n0 = blockIdx.y * TN;
    const int tap = blockIdx.z / p.msplit, split = blockIdx.z % p.msplit;
    const long mper = ((p.M + p.msplit - 1) / p.msplit + MC - 1) / MC * MC;
    const long mbeg = (long)split * mper, mend = mbeg + mper < p.M ? mbeg + mper : p.M;
    const int dyo = (int)((p.dyp >> (7 * tap)) & 127) - 64, dxo = (int)((p.dxp >> (7 * tap)) & 127) - 64;
    const int tk = (tid >> 4) * 4, tn = (tid & 15) * 4;  // micro-tile origin
    const int lr = tid >> 4, lc = (tid & 15) * 4;        // loader: rows lr, lr+16; 4 consecutive channels
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (long mc = mbeg; mc < mend; mc += MC) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = lr + h * 16;
            const long m = mc + r;
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f), dv = av;
            if (m < mend) {
                long src = m;
                if (!p.flat) {
                    const int j = (int)(m % p.Wg);
                    const long t = m / p.Wg;
                    const int iy = (int)(t % p.Hg) * p.sa + dyo, ix = j * p.sa + dxo;
                    src = (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) ? ((t / p.Hg) * p.Ha + iy) * (long)p.Wa + ix : -1;
                }
                if (src >= 0 && k0 + lc < p.K) av = *reinterpret_cast<const float4*>(p.A + src * p.lda + k0 + lc);
                if (n0 + lc < p.N) dv = *reinterpret_cast<const float4*>(p.dY + m * p.ldd + n0 + lc);
            }
            *reinterpret_cast<float4*>(&As[r][lc]) = av;
            *reinterpret_cast<float4*>(&Ds[r][lc]) = dv;
        }
        __syncthreads();
#pragma unroll 8
        for (int mm = 0; mm < MC; ++mm) {
            const float4 a = *reinterpret_cast<const float4*>(&As[mm][tk]);
            const float4 d = *reinterpret_cast<const float4*>(&Ds[mm][tn]);
            const float av[4] = {a.x, a.y, a.z, a.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], dv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* out = p.dW + (long)tap * p.K * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + tk + i;
        if (k >= p.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn + j;
            if (n < p.N) atomicAdd(out + (long)k * p.N + n, acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 weights on the DEVICE, [src_taps][Cin][Cout] (cout_major = 0) or [src_taps][Cout][Cin] (cout_major = 1)
// -> packed bf16 hi/lo planes [Npad][ntaps][Cpad] (the layout of emd_pack_weights_bf16).  Packed tap t comes from
// source tap (sel >> 4t) & 15, so one kernel serves the forward pack, the flipped pack of the data gradient
// and the four tap subsets of the transposed conv.  One thread per packed element (padding written as zero).
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int ntaps, unsigned long long sel,
                                                           int Cin, int Cout, int cout_major, int cpad, long total,
                                                           uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cpad);
    const long r = idx / cpad;
    const int t = (int)(r % ntaps);
    const int n = (int)(r / ntaps);
    float v = 0.f;
    if (c < Cin && n < Cout) {
        const long st = (long)((sel >> (4 * t)) & 15);
        v = cout_major ? w[(st * Cout + n) * Cin + c] : w[(st * Cin + c) * Cout + n];
    }
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    hi[idx] = __builtin_bit_cast(uint16_t, h);
    lo[idx] = __builtin_bit_cast(uint16_t, l);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int emd_conv_wgrad_f32(const float* a, int lda, const float* dy, int ldd, float* dw, int B, int Hg, int Wg,
                                  int Ha, int Wa, int K, int N, int ntaps, const int* tap_dy, const int* tap_dx, int sa,
                                  emd_stream_t stream) {
    EMD_REQUIRE(a && dy && dw, EMD_E_INVALID, "emd_conv_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && Hg >= 1 && Wg >= 1 && Ha >= 1 && Wa >= 1 && K >= 4 && N >= 4, EMD_E_INVALID, "emd_conv_wgrad_f32: bad shape");
    EMD_REQUIRE(ntaps >= 1 && ntaps <= 9 && (ntaps == 1 || (tap_dy && tap_dx)) && sa >= 1, EMD_E_INVALID, "emd_conv_wgrad_f32: bad taps");
    EMD_REQUIRE(K % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldd % 4 == 0 && lda >= K && ldd >= N && emd::aligned16(a) &&
                    emd::aligned16(dy), EMD_E_ALIGN, "emd_conv_wgrad_f32: K, N, lda, ldd multiples of 4; 16-byte aligned pointers");
    if (B == 0) return EMD_OK;
    WgradParams p{};
    p.A = a; p.dY = dy; p.dW = dw;
    p.M = (long)B * Hg * Wg; p.K = K; p.N = N; p.lda = lda; p.ldd = ldd; p.ntaps = ntaps;
    p.Hg = Hg; p.Wg = Wg; p.Ha = Ha; p.Wa = Wa; p.sa = sa;
    p.flat = ntaps == 1 && sa == 1 && Ha == Hg && Wa == Wg && (!tap_dy || (tap_dy[0] == 0 && tap_dx[0] == 0));
    for (int t = 0; t < ntaps; ++t) {
        const int dyv = tap_dy ? tap_dy[t] : 0, dxv = tap_dx ? tap_dx[t] : 0;
        EMD_REQUIRE(dyv >= -64 && dyv < 64 && dxv >= -64 && dxv < 64, EMD_E_UNSUPPORTED, "emd_conv_wgrad_f32: tap offset out of range");
        p.dyp |= (unsigned long long)(dyv + 64) << (7 * t);
        p.dxp |= (unsigned long long)(dxv + 64) << (7 * t);
    }
    const int kt = (K + 63) / 64, nt = (N + 63) / 64;
    long want = 2048 / ((long)kt * nt * ntaps);  // enough workgroups to fill the chip a few times
    if (want < 1) want = 1;
    const long maxsplit = (p.M + 2047) / 2048;
    p.msplit = (int)(want < maxsplit ? want : maxsplit);
    if (p.msplit < 1) p.msplit = 1;
    EMD_REQUIRE((long)ntaps * p.msplit <= 65535, EMD_E_UNSUPPORTED, "emd_conv_wgrad_f32: grid too large");
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(kt, nt, ntaps * p.msplit), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return emd::check_launch("conv_wgrad_kernel");
}

extern "C" int emd_pack_weights_dev(const float* w, int src_taps, int ntaps, const int* tap_sel, int Cin, int Cout,
                                    int cout_major, uint16_t* hi, uint16_t* lo, emd_stream_t stream) {
    EMD_REQUIRE(w && hi && lo, EMD_E_INVALID, "emd_pack_weights_dev: null pointer");
    EMD_REQUIRE(src_taps >= 1 && src_taps <= 9 && ntaps >= 1 && ntaps <= 9 && Cin >= 1 && Cout >= 1, EMD_E_INVALID,
                "emd_pack_weights_dev: bad shape");
    EMD_REQUIRE(tap_sel || ntaps == src_taps, EMD_E_INVALID, "emd_pack_weights_dev: a tap subset needs tap_sel");
    unsigned long long sel = 0;
    for (int t = 0; t < ntaps; ++t) {
        const int s = tap_sel ? tap_sel[t] : t;
        EMD_REQUIRE(s >= 0 && s < src_taps, EMD_E_INVALID, "emd_pack_weights_dev: tap_sel out of range");
        sel |= (unsigned long long)s << (4 * t);
    }
    const int cpad = (Cin + kBK - 1) / kBK * kBK;
    const long npad = (Cout + kNPadTo - 1) / kNPadTo * kNPadTo;
    const long total = npad * ntaps * cpad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, ntaps, sel, Cin, Cout, cout_major, cpad, total, hi, lo);
    return emd::check_launch("pack_weights_kernel");
}
