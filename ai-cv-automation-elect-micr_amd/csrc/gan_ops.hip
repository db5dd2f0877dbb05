// Kernels specific to graph G, the in-filling GAN's generator (misc_py/gan-infilling-100.py:133-374).  Its
// separable convs are tf.pad(REFLECT) + VALID (:209-216), its first layer is a 7x7 separable conv on the 1-channel
// image (:343-347) and its output is tanh(instance_norm(3x3 conv)) (:362-372); activations are leaky_relu(0.2).
// The pointwise halves, the SAME-padded separable convs of deconv_block and the resizes are the graph-D kernels.
// All HBM-bound, fp32.
#include "mfma_common.hpp"

namespace {

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
// tf.pad(mode="REFLECT"): index -1 -> 1, n -> n-2 (the border sample is not repeated)
__device__ __forceinline__ int reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.2f * v; }

// Depthwise 3x3 over the reflect-padded (1 px) input, VALID, stride 1 or 2: output (oy,ox) reads rows
// oy*s-1 .. oy*s+1 (reflected).  One output pixel x 4 channels per thread.
template <bool SPLIT = false>
__global__ __launch_bounds__(256) void dw3x3_reflect_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            float* __restrict__ y, int ldy, int H, int W, int C4, int Ho,
                                                            int Wo, int stride, long nthreads, int C4t) {
    // C4t = channel quads per pixel that have a thread: C4, or ceil32(C)/4 when the split32 padding is written too.
    // A workgroup = 4 x 4 output pixels x 16 channel quads: overlapping windows are served by the workgroup's L1.
    (void)nthreads;
    const int ncb = (C4t + 15) >> 4, npx = (Wo + 3) >> 2, npy = (Ho + 3) >> 2;
    int bidx = blockIdx.x;
    const int cblk = bidx % ncb;
    bidx /= ncb;
    const int bx = bidx % npx;
    bidx /= npx;
    const int by = bidx % npy;
    const long b = bidx / npy;
    const int c4o = cblk * 16 + (threadIdx.x & 15);
    const int ox = bx * 4 + ((threadIdx.x >> 4) & 3), oy = by * 4 + (threadIdx.x >> 6);
    if (c4o >= C4t || ox >= Wo || oy >= Ho) return;
    const bool padq = SPLIT && c4o >= C4;
    const int c4 = padq ? C4 - 1 : c4o;
    const int C = C4 * 4;
    const float* xb = x + (b * H) * (long)W * ldx + c4 * 4;
    float4 acc = f4zero();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int iy = reflect(oy * stride - 1 + i, H);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ix = reflect(ox * stride - 1 + j, W);
            acc = fma4(*reinterpret_cast<const float4*>(w + (i * 3 + j) * C + c4 * 4),
                       *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx), acc);
        }
    }
    emd::dw_store<SPLIT>(y, (b * Ho + oy) * (long)Wo + ox, ldy, c4o, padq ? f4zero() : acc);
}

// dw3x3_reflect_kernel on a GENERATED input: the C-channel tensor it reads is act(d[pixel] * a[c] + t[c]) with d one value per
// pixel (pitch ldd) -- the first layer's output (cin1_k7_reflect_kernel) rebuilt in registers for the layer that follows it
// (enc0 -> enc1, misc_py/gan-infilling-100.py:343-349), so that [B,H,W,C] tensor is neither written nor read.  Same
// arithmetic in the same order as the two kernels it replaces.
__global__ __launch_bounds__(256) void dw3x3_reflect_gen_kernel(const float* __restrict__ d, int ldd, const float* __restrict__ a,
                                                                const float* __restrict__ tsh, int gen_act,
                                                                const float* __restrict__ w, float* __restrict__ y, int ldy,
                                                                int H, int W, int C4, int Ho, int Wo, int stride, int qs) {
    // a workgroup = 2^qs channel quads x (4 rows x 2^(6-qs) columns) of output pixels; qs = 3 for the 32-channel first layer,
    // so that no lane idles on a channel quad that does not exist
    const int QB = 1 << qs, tw = 64 >> qs;
    const int ncb = (C4 + QB - 1) >> qs, npx = (Wo + tw - 1) / tw, npy = (Ho + 3) >> 2;
    int bidx = blockIdx.x;
    const int cblk = bidx % ncb;
    bidx /= ncb;
    const int bx = bidx % npx;
    bidx /= npx;
    const int by = bidx % npy;
    const long b = bidx / npy;
    const int c4 = cblk * QB + (threadIdx.x & (QB - 1));
    const int pp = threadIdx.x >> qs;
    const int ox = bx * tw + (pp & (tw - 1)), oy = by * 4 + pp / tw;
    if (c4 >= C4 || ox >= Wo || oy >= Ho) return;
    const int C = C4 * 4;
    const float4 av = *reinterpret_cast<const float4*>(a + c4 * 4);
    const float4 tv = *reinterpret_cast<const float4*>(tsh + c4 * 4);
    const float* db = d + (b * H) * (long)W * ldd;
    float4 acc = f4zero();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int iy = reflect(oy * stride - 1 + i, H);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ix = reflect(ox * stride - 1 + j, W);
            const float dv = db[((long)iy * W + ix) * ldd];
            float4 v = make_float4(fmaf(dv, av.x, tv.x), fmaf(dv, av.y, tv.y), fmaf(dv, av.z, tv.z), fmaf(dv, av.w, tv.w));
            if (gen_act) v = make_float4(leaky(v.x), leaky(v.y), leaky(v.z), leaky(v.w));
            acc = fma4(*reinterpret_cast<const float4*>(w + (i * 3 + j) * C + c4 * 4), v, acc);
        }
    }
    *reinterpret_cast<float4*>(y + ((b * Ho + oy) * (long)Wo + ox) * ldy + c4 * 4) = acc;
}

// First layer: d = (7x7 depthwise of the reflect-padded 1-channel image), y[pix][n] = leaky(d*a[n] + shift[n]).
// One pixel per lane for the stencil, then N4 lanes share a pixel for the 16-byte stores (as cin1_kernel).
__global__ __launch_bounds__(256) void cin1_k7_reflect_kernel(const float* __restrict__ x, const float* __restrict__ w49,
                                                              const float* __restrict__ a, const float* __restrict__ tsh,
                                                              float* __restrict__ y, int ldy, int H, int W, int N4,
                                                              long npix, int act) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long p0 = wave * 64;
    if (p0 >= npix) return;
    const long pix = p0 + lane;
    float d = 0.f;
    if (pix < npix) {
        int ox, oy;
        const long t = emd::divmod(pix, W, ox);
        const float* img = x + emd::divmod(t, H, oy) * (long)H * W;
        for (int i = 0; i < 7; ++i) {
            const float* row = img + (long)reflect(oy - 3 + i, H) * W;
#pragma unroll
            for (int j = 0; j < 7; ++j) d = fmaf(w49[i * 7 + j], row[reflect(ox - 3 + j, W)], d);
        }
    }
    const int ppi = 64 / N4;
    const int sub = lane / N4, n4 = lane % N4;
    const float4 av = *reinterpret_cast<const float4*>(a + n4 * 4);
    const float4 tv = *reinterpret_cast<const float4*>(tsh + n4 * 4);
    for (int q = 0; q < 64; q += ppi) {
        const float dv = __shfl(d, q + sub);
        const long op = p0 + q + sub;
        if (op < npix) {
            float4 o = make_float4(fmaf(dv, av.x, tv.x), fmaf(dv, av.y, tv.y), fmaf(dv, av.z, tv.z), fmaf(dv, av.w, tv.w));
            if (act) o = make_float4(leaky(o.x), leaky(o.y), leaky(o.z), leaky(o.w));
            *reinterpret_cast<float4*>(y + op * ldy + n4 * 4) = o;
        }
    }
}

// 3x3 conv to ONE channel over the reflect-padded input + bias: LP = Cin/4 lanes share a pixel and reduce with
// wave shuffles (the reflect twin of conv3x3_cout1_kernel).
__global__ __launch_bounds__(256) void conv3x3_cout1_reflect_kernel(const float* __restrict__ x, int ldx,
                                                                    const float* __restrict__ w, float bias,
                                                                    float* __restrict__ y, int H, int W, int LP, long npix) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int ppw = 64 / LP;
    const int c4 = lane % LP, sub = lane / LP;
    const int C = LP * 4;
    const long pix = wave * ppw + sub;
    float4 acc = f4zero();
    if (pix < npix) {
        int ox, oy;
        const long t = emd::divmod(pix, W, ox);
        const float* xb = x + (emd::divmod(t, H, oy) * H) * (long)W * ldx + c4 * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = reflect(oy - 1 + i, H);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ix = reflect(ox - 1 + j, W);
                acc = fma4(*reinterpret_cast<const float4*>(w + (i * 3 + j) * C + c4 * 4),
                           *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx), acc);
            }
        }
    }
    float s = (acc.x + acc.y) + (acc.z + acc.w);
    for (int m = 1; m < LP; m <<= 1) s += __shfl_xor(s, m);
    if (c4 == 0 && pix < npix) y[pix] = s + bias;
}

// y = tanh( (x - mean[b]) * rsqrt(var[b] + eps) ) for a 1-channel image batch [B, npix_per_image]
// (_instance_norm with its fixed scale 1 / shift 0, :140-148, then tf.tanh :372)
__global__ __launch_bounds__(256) void instnorm_tanh_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ var, float* __restrict__ y,
                                                            long npix_img, long total, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / npix_img;
    y[i] = tanhf((x[i] - mean[b]) * rsqrtf(var[b] + eps));
}

// Discriminator head (:560-567, :708): y[b] = sum_k x[b][k]*w[k] + bias -- one wave per row, shuffle reduction.
__global__ __launch_bounds__(64) void fc_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     float bias, float* __restrict__ y, int K) {
    const float* row = x + (long)blockIdx.x * ldx;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) s = fmaf(row[k], w[k], s);
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if (threadIdx.x == 0) y[blockIdx.x] = s + bias;
}

// output = sigmoid(max(small, medium, large))   (:708)
__global__ void max3_sigmoid_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                    float* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = fmaxf(a[i], fmaxf(b[i], c[i]));
    y[i] = 1.f / (1.f + __expf(-m));
}

int blocks_for(long nthreads, unsigned* nb) {
    const long b = (nthreads + 255) / 256;
    if (b <= 0 || b > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "grid too large");
    *nb = (unsigned)b;
    return EMD_OK;
}

}  // namespace

template <bool SPLIT>
static int dw3x3_reflect_launch(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C,
                                int stride, emd_stream_t stream) {
    EMD_REQUIRE(x && w && y, EMD_E_INVALID, "emd_dw3x3_reflect_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2 && C >= 4 && (stride == 1 || stride == 2), EMD_E_INVALID,
                "emd_dw3x3_reflect_f32: bad shape (reflect padding needs H, W >= 2)");
    const int Cp = (C + 31) / 32 * 32;
    if (SPLIT)
        EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldx >= C && ldy % 32 == 0 && ldy >= Cp && emd::aligned16(x) &&
                        (reinterpret_cast<uintptr_t>(y) & 127u) == 0 && emd::aligned16(w),
                    EMD_E_ALIGN, "emd_dw3x3_reflect_split32_f32: C, ldx multiples of 4; ldy a multiple of 32, >= ceil32(C); y 128-byte aligned");
    else
        EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && emd::aligned16(x) &&
                        emd::aligned16(y) && emd::aligned16(w), EMD_E_ALIGN, "emd_dw3x3_reflect_f32: alignment");
    if (B == 0) return EMD_OK;
    if (stride == 1)   // the rolling kernel (each input row read once per strip): 70 -> 40 us on the 32 x 32 x 768 maps
        return emd::launch_dw3x3_reflect_roll(x, ldx, w, y, ldy, B, H, W, C, SPLIT, static_cast<hipStream_t>(stream));
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;  // VALID on the (H+2) x (W+2) padded input
    const int C4t = SPLIT ? Cp / 4 : C / 4;
    const long nthreads = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4) * ((C4t + 15) / 16) * 256;
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL((dw3x3_reflect_kernel<SPLIT>), dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, w, y, ldy,
                       H, W, C / 4, Ho, Wo, stride, nthreads, C4t);
    return emd::check_launch("dw3x3_reflect_kernel");
}

extern "C" int emd_dw3x3_reflect_gen_f32(const float* d, int ldd, const float* gen_a, const float* gen_t, int leaky_act,
                                         const float* w, float* y, int ldy, int B, int H, int W, int C, int stride,
                                         emd_stream_t stream) {
    EMD_REQUIRE(d && gen_a && gen_t && w && y, EMD_E_INVALID, "emd_dw3x3_reflect_gen_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2 && C >= 4 && ldd >= 1 && (stride == 1 || stride == 2), EMD_E_INVALID,
                "emd_dw3x3_reflect_gen_f32: bad shape (reflect padding needs H, W >= 2)");
    EMD_REQUIRE(C % 4 == 0 && ldy % 4 == 0 && ldy >= C && emd::aligned16(y) && emd::aligned16(w) && emd::aligned16(gen_a) &&
                    emd::aligned16(gen_t), EMD_E_ALIGN, "emd_dw3x3_reflect_gen_f32: alignment");
    if (B == 0) return EMD_OK;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int C4 = C / 4, qs = C4 >= 16 ? 4 : (C4 >= 8 ? 3 : 2), QB = 1 << qs, tw = 64 >> qs;
    const long nthreads = (long)B * ((Ho + 3) / 4) * ((Wo + tw - 1) / tw) * ((C4 + QB - 1) / QB) * 256;
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(dw3x3_reflect_gen_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), d, ldd, gen_a, gen_t,
                       leaky_act ? 1 : 0, w, y, ldy, H, W, C4, Ho, Wo, stride, qs);
    return emd::check_launch("dw3x3_reflect_gen_kernel");
}

extern "C" int emd_dw3x3_reflect_f32(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C,
                                     int stride, emd_stream_t stream) {
    return dw3x3_reflect_launch<false>(x, ldx, w, y, ldy, B, H, W, C, stride, stream);
}

extern "C" int emd_dw3x3_reflect_split32_f32(const float* x, int ldx, const float* w, void* y, int ldy, int B, int H, int W,
                                             int C, int stride, emd_stream_t stream) {
    return dw3x3_reflect_launch<true>(x, ldx, w, static_cast<float*>(y), ldy, B, H, W, C, stride, stream);
}

extern "C" int emd_cin1_k7_reflect_f32(const float* x, const float* w49, const float* a, const float* shift, float* y,
                                       int ldy, int B, int H, int W, int Cout, int act, emd_stream_t stream) {
    EMD_REQUIRE(x && w49 && a && shift && y, EMD_E_INVALID, "emd_cin1_k7_reflect_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 4 && W >= 4, EMD_E_INVALID, "emd_cin1_k7_reflect_f32: reflect padding by 3 needs H, W >= 4");
    const int N4 = Cout / 4;
    EMD_REQUIRE(Cout % 4 == 0 && N4 >= 1 && N4 <= 64 && (64 % N4) == 0, EMD_E_UNSUPPORTED,
                "emd_cin1_k7_reflect_f32: Cout/4 must divide 64");
    EMD_REQUIRE(ldy % 4 == 0 && ldy >= Cout && emd::aligned16(y) && emd::aligned16(a) && emd::aligned16(shift), EMD_E_ALIGN,
                "emd_cin1_k7_reflect_f32: alignment");
    if (B == 0) return EMD_OK;
    const long npix = (long)B * H * W;
    unsigned nb;
    int rc = blocks_for((npix + 63) / 64 * 64, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(cin1_k7_reflect_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, w49, a, shift, y,
                       ldy, H, W, N4, npix, act ? 1 : 0);
    return emd::check_launch("cin1_k7_reflect_kernel");
}

extern "C" int emd_conv3x3_cout1_reflect_f32(const float* x, int ldx, const float* w, float bias, float* y, int B, int H,
                                             int W, int Cin, emd_stream_t stream) {
    EMD_REQUIRE(x && w && y, EMD_E_INVALID, "emd_conv3x3_cout1_reflect_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2, EMD_E_INVALID, "emd_conv3x3_cout1_reflect_f32: bad shape");
    const int LP = Cin / 4;
    EMD_REQUIRE(Cin % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, EMD_E_UNSUPPORTED,
                "emd_conv3x3_cout1_reflect_f32: Cin/4 must be a power of two <= 64");
    EMD_REQUIRE(ldx % 4 == 0 && ldx >= Cin && emd::aligned16(x) && emd::aligned16(w), EMD_E_ALIGN,
                "emd_conv3x3_cout1_reflect_f32: alignment");
    if (B == 0) return EMD_OK;
    const long npix = (long)B * H * W;
    const int ppw = 64 / LP;
    if (W % ppw == 0)   // a wave never straddles an image row: rolling kernel (an input row is read once per 8-row strip)
        return emd::launch_conv3x3_cout1_reflect_roll(x, ldx, w, bias, y, B, H, W, Cin, static_cast<hipStream_t>(stream));
    unsigned nb;
    int rc = blocks_for((npix + ppw - 1) / ppw * 64, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(conv3x3_cout1_reflect_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, w, bias,
                       y, H, W, LP, npix);
    return emd::check_launch("conv3x3_cout1_reflect_kernel");
}

extern "C" int emd_instnorm_tanh_f32(const float* x, const float* mean, const float* var, float* y, int B, long npix_img,
                                     float eps, emd_stream_t stream) {
    EMD_REQUIRE(x && mean && var && y, EMD_E_INVALID, "emd_instnorm_tanh_f32: null pointer");
    EMD_REQUIRE(B >= 0 && npix_img >= 1, EMD_E_INVALID, "emd_instnorm_tanh_f32: bad shape");
    if (B == 0) return EMD_OK;
    const long total = (long)B * npix_img;
    unsigned nb;
    int rc = blocks_for(total, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(instnorm_tanh_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, mean, var, y,
                       npix_img, total, eps);
    return emd::check_launch("instnorm_tanh_kernel");
}

extern "C" int emd_fc_rows_f32(const float* x, int ldx, const float* w, float bias, float* y, int B, int K,
                               emd_stream_t stream) {
    EMD_REQUIRE(x && w && y, EMD_E_INVALID, "emd_fc_rows_f32: null pointer");
    EMD_REQUIRE(B >= 0 && K >= 1 && ldx >= K, EMD_E_INVALID, "emd_fc_rows_f32: bad shape");
    if (B == 0) return EMD_OK;
    hipLaunchKernelGGL(fc_rows_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), x, ldx, w, bias, y, K);
    return emd::check_launch("fc_rows_kernel");
}

extern "C" int emd_max3_sigmoid_f32(const float* a, const float* b, const float* c, float* y, int n, emd_stream_t stream) {
    EMD_REQUIRE(a && b && c && y, EMD_E_INVALID, "emd_max3_sigmoid_f32: null pointer");
    EMD_REQUIRE(n >= 0, EMD_E_INVALID, "emd_max3_sigmoid_f32: bad size");
    if (n == 0) return EMD_OK;
    hipLaunchKernelGGL(max3_sigmoid_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, c, y, n);
    return emd::check_launch("max3_sigmoid_kernel");
}
