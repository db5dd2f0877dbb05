// Backward kernels of the HBM-bound layers of graph D' plus its loss and optimizer step
// (misc_py/denoiser-multi-gpu.py:752-782 _tower_fn, :1011-1077 _train_op).  fp32 VALU; float atomics where several
// workgroups add into one small result; double accumulation for the scalar loss.
//   emd_dw3x3_wgrad_f32 / emd_dw3x3_bwd_data_f32        depthwise 3x3 (the depthwise half of slim.separable_convolution2d)
//   emd_conv3x3_cout1_wgrad_f32 / _bwd_data_f32          the final 3x3 conv to one channel
//   emd_resize_bilinear_bwd_f32, emd_avgpool2x2_bwd_f32  decoder / image-level-branch resampling
//   emd_axpy_f32                                         y += alpha*x (gradient fan-in)
//   emd_denoise_loss_f32                                 mse, the capped loss and dLoss/dOutput
//   emd_nesterov_step_f32                                tf.train.MomentumOptimizer(use_nesterov=True)
#include "emd_common.hpp"

namespace {

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
// relu (hi = +inf) / relu6 (hi = 6) in affine_relu6_kernel's order (dw_misc.hip): the PRE forms rebuild its bits
__device__ __forceinline__ float4 clamp4(float4 a, float hi) {
    return make_float4(fminf(fmaxf(a.x, 0.f), hi), fminf(fmaxf(a.y, 0.f), hi), fminf(fmaxf(a.z, 0.f), hi), fminf(fmaxf(a.w, 0.f), hi));
}
__device__ __forceinline__ float4 fma4s(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}

inline int same_pad_before(int n, int s, int r) {  // TF SAME, k = 3
    const int o = (n + s - 1) / s;
    int total = (o - 1) * s + 2 * r + 1 - n;
    if (total < 0) total = 0;
    return total / 2;
}

// dw[t][c] += sum_{b,oy,ox} x[b, oy*s + ky*r - pt, ox*s + kx*r - pl, c] * dy[b,oy,ox,c]      (t = 3*ky + kx)
// SCALAR: dy has ONE channel (the weight gradient of the 3x3 conv to one output channel, w[t][c]).
// Block: 16 channel-quads x 16 pixel lanes; grid (ceil(C/64), slabs of output pixels).
// PRE: x is act(r * pre_s + pre_t) of the tensor given (the forward pass left the previous layer's affine + activation to its consumer's
// loads, emd_dw3x3_pre_act_f32: the same values are rebuilt here); pre_ld != 0: pre_s / pre_t are [image][pre_ld]; pre_hi 6 / inf.
template <bool SCALAR, bool PRE = false>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                       int ldd, float* __restrict__ dw, int H, int W, int Ho, int Wo, int C,
                                                       int s, int r, int pt, int pl, long npix, long pix_per_slab,
                                                       const float* __restrict__ pre_s = nullptr, const float* __restrict__ pre_t = nullptr,
                                                       long pre_ld = 0, float pre_hi = 0.f) {
    const int c = (blockIdx.x * 16 + (threadIdx.x & 15)) * 4;
    const int plane = threadIdx.x >> 4;
    const long p0 = (long)blockIdx.y * pix_per_slab;
    const long p1 = min(p0 + pix_per_slab, npix);
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f4zero();
    if (c < C) {
        for (long p = p0 + plane; p < p1; p += 16) {
            const int ox = (int)(p % Wo);
            const long q = p / Wo;
            const int oy = (int)(q % Ho);
            const long b = q / Ho;
            float4 g;
            if (SCALAR) {
                const float v = dy[p];
                g = make_float4(v, v, v, v);
            } else {
                g = *reinterpret_cast<const float4*>(dy + p * ldd + c);
            }
            const float* xb = x + b * H * (long)W * ldx + c;
            float4 ps = f4zero(), pq = f4zero();
            if (PRE) {
                ps = *reinterpret_cast<const float4*>(pre_s + b * pre_ld + c);
                pq = *reinterpret_cast<const float4*>(pre_t + b * pre_ld + c);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * s + ky * r - pt;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * s + kx * r - pl;
                    if (ix < 0 || ix >= W) continue;
                    float4 xv = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx);
                    if (PRE) xv = clamp4(fma4(xv, ps, pq), pre_hi);
                    acc[ky * 3 + kx] = fma4(xv, g, acc[ky * 3 + kx]);
                }
            }
        }
    }
    // 16 pixel lanes -> one sum per channel through LDS, then one atomic per (tap, channel) and block
    __shared__ float red[16][9][64 + 1];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int cl = (threadIdx.x & 15) * 4;
        red[plane][t][cl + 0] = acc[t].x; red[plane][t][cl + 1] = acc[t].y;
        red[plane][t][cl + 2] = acc[t].z; red[plane][t][cl + 3] = acc[t].w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        const int t = i / 64, cl = i % 64;
        const int cc = blockIdx.x * 64 + cl;
        if (cc >= C) continue;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += red[k][t][cl];
        atomicAdd(dw + (long)t * C + cc, sum);
    }
}

// Stride-1 (rate 1, SAME) form of the above for the big layers: a thread owns (column ox, 4 channels) and walks down
// a strip of TH output rows keeping the three live input rows x[.][ox-1..ox+1] in registers, so every output pixel
// costs 4 vector loads (3 of x, 1 of dy) instead of 10; neighbouring lanes share the x loads through L1.
// Block = 16 channel quads x 16 columns; grid (ceil(C/64), ceil(W/16), B * strips).
// SCALAR (round 4): dy has ONE channel -- the weight gradient of the final 3x3 conv to one output channel ran on the generic kernel, nine
// gathers of x per pixel: 204 us for a pair of 512^2 x 64 maps this form reads once.
template <int TH, bool PRE = false, bool SCALAR = false>
__global__ __launch_bounds__(256) void dw_wgrad_roll_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                            int ldd, float* __restrict__ dw, int H, int W, int C, int nstrip,
                                                            const float* __restrict__ pre_s = nullptr, const float* __restrict__ pre_t = nullptr,
                                                            long pre_ld = 0, float pre_hi = 0.f) {
    const int cl = (threadIdx.x & 15) * 4;
    const int c = blockIdx.x * 64 + cl;
    const int ox = blockIdx.y * 16 + (threadIdx.x >> 4);
    const int b = blockIdx.z / nstrip, y0 = (blockIdx.z % nstrip) * TH;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f4zero();
    if (c < C && ox < W) {
        const float* xb = x + ((long)b * H) * W * ldx + c;
        const float* db = dy + ((long)b * H) * W * ldd + (SCALAR ? 0 : c);
        const bool hl = ox > 0, hr = ox + 1 < W;
        float4 ps = f4zero(), pq = f4zero();
        if (PRE) {
            ps = *reinterpret_cast<const float4*>(pre_s + b * pre_ld + c);
            pq = *reinterpret_cast<const float4*>(pre_t + b * pre_ld + c);
        }
        auto row = [&](int iy, float4& l, float4& m, float4& r) {
            l = m = r = f4zero();
            if (iy >= 0 && iy < H) {
                const float* rp = xb + ((long)iy * W + ox) * ldx;
                m = *reinterpret_cast<const float4*>(rp);
                if (PRE) m = clamp4(fma4(m, ps, pq), pre_hi);
                if (hl) {
                    l = *reinterpret_cast<const float4*>(rp - ldx);
                    if (PRE) l = clamp4(fma4(l, ps, pq), pre_hi);
                }
                if (hr) {
                    r = *reinterpret_cast<const float4*>(rp + ldx);
                    if (PRE) r = clamp4(fma4(r, ps, pq), pre_hi);
                }
            }
        };
        float4 a0, a1, a2, b0, b1, b2, c0, c1, c2;   // rows oy-1, oy, oy+1
        row(y0 - 1, a0, a1, a2);
        row(y0, b0, b1, b2);
        const int y1 = y0 + TH < H ? y0 + TH : H;
        for (int oy = y0; oy < y1; ++oy) {
            row(oy + 1, c0, c1, c2);
            float4 g;
            if constexpr (SCALAR) {
                const float v = db[(long)oy * W + ox];
                g = make_float4(v, v, v, v);
            } else {
                g = *reinterpret_cast<const float4*>(db + ((long)oy * W + ox) * ldd);
            }
            acc[0] = fma4(a0, g, acc[0]); acc[1] = fma4(a1, g, acc[1]); acc[2] = fma4(a2, g, acc[2]);
            acc[3] = fma4(b0, g, acc[3]); acc[4] = fma4(b1, g, acc[4]); acc[5] = fma4(b2, g, acc[5]);
            acc[6] = fma4(c0, g, acc[6]); acc[7] = fma4(c1, g, acc[7]); acc[8] = fma4(c2, g, acc[8]);
            a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
        }
    }
    __shared__ float red[16][9][64 + 1];
    const int plane = threadIdx.x >> 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        red[plane][t][cl + 0] = acc[t].x; red[plane][t][cl + 1] = acc[t].y;
        red[plane][t][cl + 2] = acc[t].z; red[plane][t][cl + 3] = acc[t].w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        const int t = i / 64, l = i % 64;
        const int cc = blockIdx.x * 64 + l;
        if (cc >= C) continue;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += red[k][t][l];
        atomicAdd(dw + (long)t * C + cc, sum);
    }
}

// dx[b,iy,ix,c] = sum_{ky,kx} dy[b,oy,ox,c] * w[t][c]  over the (oy,ox) with oy*s + ky*r - pt == iy (same for x).
// SCALAR: dy has one channel (data gradient of the 3x3 conv to one output channel).
template <bool SCALAR>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ w,
                                                          float* __restrict__ dx, int ldx, int H, int W, int Ho, int Wo,
                                                          int C4, int s, int r, int pt, int pl, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    long q = tid / C4;
    const int ix = (int)(q % W);
    q /= W;
    const int iy = (int)(q % H);
    const long b = q / H;
    const int C = C4 * 4;
    float4 acc = f4zero();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ny = iy + pt - ky * r;
        if (ny < 0 || ny % s != 0 || ny / s >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int nx = ix + pl - kx * r;
            if (nx < 0 || nx % s != 0 || nx / s >= Wo) continue;
            const long op = (b * Ho + ny / s) * (long)Wo + nx / s;
            const float4 wv = *reinterpret_cast<const float4*>(w + (ky * 3 + kx) * C + c);
            if (SCALAR)
                acc = fma4s(wv, dy[op], acc);
            else
                acc = fma4(*reinterpret_cast<const float4*>(dy + op * ldd + c), wv, acc);
        }
    }
    *reinterpret_cast<float4*>(dx + ((b * H + iy) * (long)W + ix) * ldx + c) = acc;
}

// The SCALAR stride-1 case above for the large maps (the final conv's data gradient: 64 channels at 512^2): a thread = four pixels along
// W x 4 channels, the nine taps' weights and the 3 x 6 window of the 1-channel gradient image in registers -- nine weight loads per four
// pixels instead of per pixel.  Sums in dw_bwd_data_kernel<true>'s order (ky, then kx): its bits.  W % 4 == 0.
__global__ __launch_bounds__(256) void cout1_bwd_data_x4_kernel(const float* __restrict__ g1, const float* __restrict__ w, float* __restrict__ dx,
                                                                int ldx, int H, int W, int C4, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    long q = tid / C4;
    const int W4 = W >> 2;
    const int ix0 = (int)(q % W4) * 4;
    q /= W4;
    const int iy = (int)(q % H);
    const long b = q / H;
    const int C = C4 * 4;
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(w + k * C + c);
    const float* gi = g1 + b * (long)H * W;
    // window value (row iy + 1 - ky, column ix + 1 - kx) for output pixel ix = ix0 + j: columns ix0 - 1 .. ix0 + 4
    float gw[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ny = iy + 1 - ky;
        const bool rowok = ny >= 0 && ny < H;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
            const int nx = ix0 - 1 + cc;
            gw[ky][cc] = (rowok && nx >= 0 && nx < W) ? gi[(long)ny * W + nx] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 acc = f4zero();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int nx = ix0 + j + 1 - kx;       // column of the tap; a padded tap adds nothing in the generic kernel: skip it
                const int ny = iy + 1 - ky;
                if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
                acc = fma4s(wk[ky * 3 + kx], gw[ky][j + 2 - kx], acc);
            }
        *reinterpret_cast<float4*>(dx + ((b * H + iy) * (long)W + ix0 + j) * ldx + c) = acc;
    }
}

// Gradient of the legacy (align_corners=False, no half-pixel) bilinear resize: every INPUT pixel gathers from
// the output pixels whose 2x2 footprint touches it, with the forward kernel's own coordinate arithmetic.
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ dy, int ldd,
                                                                  float* __restrict__ dx, int ldx, int Hi, int Wi, int Ho,
                                                                  int Wo, int C4, float sy, float sx, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    long q = tid / C4;
    const int ix = (int)(q % Wi);
    q /= Wi;
    const int iy = (int)(q % Hi);
    const long b = q / Hi;
    // candidate output rows/cols: source coordinate o*s within (i-1, i+1), widened by one for rounding
    const int oy_lo = max(0, (int)floorf((float)(iy - 1) / sy) - 1), oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) / sy) + 1);
    const int ox_lo = max(0, (int)floorf((float)(ix - 1) / sx) - 1), ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) / sx) + 1);
    float4 acc = f4zero();
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float fy = (float)oy * sy;
        const int y0 = (int)floorf(fy), y1 = min(y0 + 1, Hi - 1);
        const float ly = fy - (float)y0;
        float wy = 0.f;
        if (y0 == iy) wy += 1.f - ly;
        if (y1 == iy) wy += ly;
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const float fx = (float)ox * sx;
            const int x0 = (int)floorf(fx), x1 = min(x0 + 1, Wi - 1);
            const float lx = fx - (float)x0;
            float wx = 0.f;
            if (x0 == ix) wx += 1.f - lx;
            if (x1 == ix) wx += lx;
            if (wx == 0.f) continue;
            acc = fma4s(*reinterpret_cast<const float4*>(dy + ((b * Ho + oy) * (long)Wo + ox) * ldd + c), wy * wx, acc);
        }
    }
    *reinterpret_cast<float4*>(dx + ((b * Hi + iy) * (long)Wi + ix) * ldx + c) = acc;
}

// Gradient of the 2x2 stride-2 SAME average pool: dx[iy,ix] = dy[iy/2, ix/2] / (window elements inside the image).
__global__ __launch_bounds__(256) void avgpool2x2_bwd_kernel(const float* __restrict__ dy, int ldd, float* __restrict__ dx,
                                                             int ldx, int H, int W, int Ho, int Wo, int C4, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    long q = tid / C4;
    const int ix = (int)(q % W);
    q /= W;
    const int iy = (int)(q % H);
    const long b = q / H;
    const int oy = iy >> 1, ox = ix >> 1;
    const int cnt = ((2 * oy + 1 < H) ? 2 : 1) * ((2 * ox + 1 < W) ? 2 : 1);
    const float inv = 1.0f / (float)cnt;
    const float4 g = *reinterpret_cast<const float4*>(dy + ((b * Ho + oy) * (long)Wo + ox) * ldd + c);
    *reinterpret_cast<float4*>(dx + ((b * H + iy) * (long)W + ix) * ldx + c) = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, int ldx, float* y, int ldy, int C4,
                                                   long nthreads, float alpha) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    const long r = tid / C4;
    const float4 a = *reinterpret_cast<const float4*>(x + r * ldx + c);
    float4 b = *reinterpret_cast<const float4*>(y + r * ldy + c);
    b = fma4s(a, alpha, b);
    *reinterpret_cast<float4*>(y + r * ldy + c) = b;
}

// ---- loss.  mse = mean((out-truth)^2);  loss = 1000*mse if mse < 1e-3 else sqrt(1000*mse)   (:768-775; the weight
// decay term is multiplied by weight_decay = 0, :117).  d(loss)/d(out) = f * (out - truth),
// f = (2/n) * (1000 | 500/sqrt(1000*mse)).
__global__ __launch_bounds__(256) void sqdiff_partial(const float* __restrict__ out, const float* __restrict__ truth, long n,
                                                      double* __restrict__ part) {
    __shared__ double sm[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = out[i] - truth[i];
        s += (double)d * (double)d;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

__global__ void loss_final(const double* __restrict__ part, int nblk, long n, float grad_scale, float* __restrict__ res) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += part[k];
    const float mse = (float)(s / (double)n);
    float loss, dl;
    if (mse < 0.001f) {
        loss = 1000.f * mse;
        dl = 1000.f;
    } else {
        loss = sqrtf(1000.f * mse);
        dl = 500.f / loss;
    }
    res[0] = mse;
    res[1] = loss;
    res[2] = grad_scale * dl * 2.0f / (float)n;
}

__global__ __launch_bounds__(256) void loss_grad(const float* __restrict__ out, const float* __restrict__ truth, long n,
                                                 const float* __restrict__ res, float* __restrict__ dout) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dout[i] = res[2] * (out[i] - truth[i]);
}

// accum = momentum*accum + g;  param -= lr*g + lr*momentum*accum;   g = grad*grad_scale
// (ApplyMomentum with use_nesterov=true, as tf.train.MomentumOptimizer runs it, :1064-1066)
__global__ __launch_bounds__(256) void nesterov_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                       float* __restrict__ accum, long n, float lr, float momentum,
                                                       float grad_scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i] * grad_scale;
    const float a = momentum * accum[i] + g;
    accum[i] = a;
    param[i] -= g * lr + a * momentum * lr;
}

int blocks_for(long nthreads, unsigned* nb) {
    const long b = (nthreads + 255) / 256;
    if (b <= 0 || b > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "grid too large");
    *nb = (unsigned)b;
    return EMD_OK;
}

template <bool SCALAR, bool PRE = false>
int launch_wgrad(const float* x, int ldx, const float* dy, int ldd, float* dw, int B, int H, int W, int C, int stride,
                 int rate, hipStream_t st, const float* pre_s = nullptr, const float* pre_t = nullptr, long pre_ld = 0, float pre_hi = 0.f) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const long npix = (long)B * Ho * Wo;
    if (stride == 1 && rate == 1 && H >= 64 && W >= 64) {  // the large maps: rolling-window form
        constexpr int TH = 32;
        const int nstrip = (H + TH - 1) / TH;
        if ((long)B * nstrip <= 65535 && (W + 15) / 16 <= 65535) {
            hipLaunchKernelGGL((dw_wgrad_roll_kernel<TH, PRE, SCALAR>), dim3((C + 63) / 64, (W + 15) / 16, B * nstrip), dim3(256), 0, st, x, ldx,
                               dy, ldd, dw, H, W, C, nstrip, pre_s, pre_t, pre_ld, pre_hi);
            return emd::check_launch("dw_wgrad_roll_kernel");
        }
    }
    long nslab = (npix + 63) / 64;  // >= 4 pixels per pixel lane; few enough slabs to keep the atomics cheap
    if (nslab > 512) nslab = 512;
    const long pps = (npix + nslab - 1) / nslab;
    hipLaunchKernelGGL((dw_wgrad_kernel<SCALAR, PRE>), dim3((C + 63) / 64, (unsigned)nslab), dim3(256), 0, st, x, ldx, dy, ldd, dw,
                       H, W, Ho, Wo, C, stride, rate, same_pad_before(H, stride, rate), same_pad_before(W, stride, rate), npix,
                       pps, pre_s, pre_t, pre_ld, pre_hi);
    return emd::check_launch("dw_wgrad_kernel");
}

template <bool SCALAR>
int launch_bwd_data(const float* dy, int ldd, const float* w, float* dx, int ldx, int B, int H, int W, int C, int stride,
                    int rate, hipStream_t st) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const long nthreads = (long)B * H * W * (C / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(dw_bwd_data_kernel<SCALAR>, dim3(nb), dim3(256), 0, st, dy, ldd, w, dx, ldx, H, W, Ho, Wo, C / 4,
                       stride, rate, same_pad_before(H, stride, rate), same_pad_before(W, stride, rate), nthreads);
    return emd::check_launch("dw_bwd_data_kernel");
}

bool dw_args_ok(const float* a, int lda, int C) { return C >= 4 && C % 4 == 0 && lda % 4 == 0 && lda >= C && emd::aligned16(a); }

}  // namespace

// The depthwise weight gradient with the layer's input given as the PRE-activation tensor r of the layer before it: x = act(r * pre_scale +
// pre_shift) is rebuilt in the loads (the forward pass did the same, emd_dw3x3_pre_act_f32, and never wrote x).  pre_images != 0:
// [B][C] scale / shift (per-image statistics); act EMD_ACT_RELU6 or EMD_ACT_RELU.  Bits of emd_affine_act[_images]_f32 + emd_dw3x3_wgrad_f32.
extern "C" int emd_dw3x3_wgrad_pre_f32(const float* r, int ldx, const float* pre_scale, const float* pre_shift, int pre_images, int act,
                                       const float* dy, int ldd, float* dw, int B, int H, int W, int C, int stride, int rate,
                                       emd_stream_t stream) {
    EMD_REQUIRE(r && dy && dw && pre_scale && pre_shift, EMD_E_INVALID, "emd_dw3x3_wgrad_pre_f32: null pointer");
    EMD_REQUIRE(emd::aligned16(pre_scale) && emd::aligned16(pre_shift), EMD_E_ALIGN, "emd_dw3x3_wgrad_pre_f32: pre_scale / pre_shift 16-byte aligned");
    EMD_REQUIRE(act == 1 || act == 2, EMD_E_INVALID, "emd_dw3x3_wgrad_pre_f32: act must be EMD_ACT_RELU6 or EMD_ACT_RELU");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (stride == 1 || stride == 2) && rate >= 1 && (rate == 1 || stride == 1),
                EMD_E_INVALID, "emd_dw3x3_wgrad_pre_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(r, ldx, C) && dw_args_ok(dy, ldd, C), EMD_E_ALIGN, "emd_dw3x3_wgrad_pre_f32: C, ldx, ldd multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    return launch_wgrad<false, true>(r, ldx, dy, ldd, dw, B, H, W, C, stride, rate, static_cast<hipStream_t>(stream), pre_scale, pre_shift,
                                     pre_images ? (long)C : 0L, act == 1 ? 6.f : __builtin_inff());
}

extern "C" int emd_dw3x3_wgrad_f32(const float* x, int ldx, const float* dy, int ldd, float* dw, int B, int H, int W, int C,
                                   int stride, int rate, emd_stream_t stream) {
    EMD_REQUIRE(x && dy && dw, EMD_E_INVALID, "emd_dw3x3_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (stride == 1 || stride == 2) && rate >= 1 && (rate == 1 || stride == 1),
                EMD_E_INVALID, "emd_dw3x3_wgrad_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(x, ldx, C) && dw_args_ok(dy, ldd, C), EMD_E_ALIGN, "emd_dw3x3_wgrad_f32: C, ldx, ldd multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    return launch_wgrad<false>(x, ldx, dy, ldd, dw, B, H, W, C, stride, rate, static_cast<hipStream_t>(stream));
}

extern "C" int emd_dw3x3_bwd_data_f32(const float* dy, int ldd, const float* w, float* dx, int ldx, int B, int H, int W,
                                      int C, int stride, int rate, emd_stream_t stream) {
    EMD_REQUIRE(dy && w && dx, EMD_E_INVALID, "emd_dw3x3_bwd_data_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (stride == 1 || stride == 2) && rate >= 1 && (rate == 1 || stride == 1),
                EMD_E_INVALID, "emd_dw3x3_bwd_data_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(dx, ldx, C) && dw_args_ok(dy, ldd, C) && emd::aligned16(w), EMD_E_ALIGN,
                "emd_dw3x3_bwd_data_f32: C, ldx, ldd multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    return launch_bwd_data<false>(dy, ldd, w, dx, ldx, B, H, W, C, stride, rate, static_cast<hipStream_t>(stream));
}

extern "C" int emd_conv3x3_cout1_wgrad_f32(const float* x, int ldx, const float* dy, float* dw, int B, int H, int W, int Cin,
                                           emd_stream_t stream) {
    EMD_REQUIRE(x && dy && dw, EMD_E_INVALID, "emd_conv3x3_cout1_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_cout1_wgrad_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(x, ldx, Cin), EMD_E_ALIGN, "emd_conv3x3_cout1_wgrad_f32: Cin, ldx multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    return launch_wgrad<true>(x, ldx, dy, 1, dw, B, H, W, Cin, 1, 1, static_cast<hipStream_t>(stream));
}

extern "C" int emd_conv3x3_cout1_bwd_data_f32(const float* dy, const float* w, float* dx, int ldx, int B, int H, int W,
                                              int Cin, emd_stream_t stream) {
    EMD_REQUIRE(dy && w && dx, EMD_E_INVALID, "emd_conv3x3_cout1_bwd_data_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_cout1_bwd_data_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(dx, ldx, Cin) && emd::aligned16(w), EMD_E_ALIGN, "emd_conv3x3_cout1_bwd_data_f32: Cin, ldx multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    if (W % 4 == 0 && (long)H * W >= 64 * 64) {   // the large maps
        const long nthreads = (long)B * H * (W / 4) * (Cin / 4);
        unsigned nb;
        int rc = blocks_for(nthreads, &nb);
        if (rc != EMD_OK) return rc;
        hipLaunchKernelGGL(cout1_bwd_data_x4_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dy, w, dx, ldx, H, W, Cin / 4, nthreads);
        return emd::check_launch("cout1_bwd_data_x4_kernel");
    }
    return launch_bwd_data<true>(dy, 1, w, dx, ldx, B, H, W, Cin, 1, 1, static_cast<hipStream_t>(stream));
}

extern "C" int emd_resize_bilinear_bwd_f32(const float* dy, int ldd, float* dx, int ldx, int B, int Hi, int Wi, int Ho,
                                           int Wo, int C, emd_stream_t stream) {
    EMD_REQUIRE(dy && dx, EMD_E_INVALID, "emd_resize_bilinear_bwd_f32: null pointer");
    EMD_REQUIRE(B >= 0 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1, EMD_E_INVALID, "emd_resize_bilinear_bwd_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(dx, ldx, C) && dw_args_ok(dy, ldd, C), EMD_E_ALIGN, "emd_resize_bilinear_bwd_f32: C, ldx, ldd multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    const long nthreads = (long)B * Hi * Wi * (C / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dy, ldd, dx, ldx,
                       Hi, Wi, Ho, Wo, C / 4, (float)Hi / (float)Ho, (float)Wi / (float)Wo, nthreads);
    return emd::check_launch("resize_bilinear_bwd_kernel");
}

extern "C" int emd_avgpool2x2_bwd_f32(const float* dy, int ldd, float* dx, int ldx, int B, int H, int W, int C,
                                      emd_stream_t stream) {
    EMD_REQUIRE(dy && dx, EMD_E_INVALID, "emd_avgpool2x2_bwd_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_avgpool2x2_bwd_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(dx, ldx, C) && dw_args_ok(dy, ldd, C), EMD_E_ALIGN, "emd_avgpool2x2_bwd_f32: C, ldx, ldd multiples of 4, 16-byte aligned");
    if (B == 0) return EMD_OK;
    const long nthreads = (long)B * H * W * (C / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(avgpool2x2_bwd_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dy, ldd, dx, ldx, H, W,
                       (H + 1) / 2, (W + 1) / 2, C / 4, nthreads);
    return emd::check_launch("avgpool2x2_bwd_kernel");
}

extern "C" int emd_axpy_f32(const float* x, int ldx, float* y, int ldy, long npix, int C, float alpha, emd_stream_t stream) {
    EMD_REQUIRE(x && y, EMD_E_INVALID, "emd_axpy_f32: null pointer");
    EMD_REQUIRE(npix >= 0, EMD_E_INVALID, "emd_axpy_f32: bad shape");
    EMD_REQUIRE(dw_args_ok(x, ldx, C) && dw_args_ok(y, ldy, C), EMD_E_ALIGN, "emd_axpy_f32: C, ldx, ldy multiples of 4, 16-byte aligned");
    if (npix == 0) return EMD_OK;
    const long nthreads = npix * (C / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(axpy_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, y, ldy, C / 4, nthreads, alpha);
    return emd::check_launch("axpy_kernel");
}

extern "C" size_t emd_denoise_loss_workspace_bytes(void) { return 1024 * sizeof(double); }

extern "C" int emd_denoise_loss_f32(const float* out, const float* truth, long n, float grad_scale, float* result3,
                                    float* dout, void* workspace, emd_stream_t stream) {
    EMD_REQUIRE(out && truth && result3 && workspace, EMD_E_INVALID, "emd_denoise_loss_f32: null pointer");
    EMD_REQUIRE(n >= 1, EMD_E_INVALID, "emd_denoise_loss_f32: empty input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    long nblk = (n + 256 * 16 - 1) / (256 * 16);
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(sqdiff_partial, dim3((unsigned)nblk), dim3(256), 0, st, out, truth, n, static_cast<double*>(workspace));
    hipLaunchKernelGGL(loss_final, dim3(1), dim3(64), 0, st, static_cast<const double*>(workspace), (int)nblk, n, grad_scale, result3);
    if (dout) {
        unsigned nb;
        int rc = blocks_for(n, &nb);
        if (rc != EMD_OK) return rc;
        hipLaunchKernelGGL(loss_grad, dim3(nb), dim3(256), 0, st, out, truth, n, result3, dout);
    }
    return emd::check_launch("denoise_loss");
}

extern "C" int emd_nesterov_step_f32(float* param, const float* grad, float* accum, long n, float lr, float momentum,
                                     float grad_scale, emd_stream_t stream) {
    EMD_REQUIRE(param && grad && accum, EMD_E_INVALID, "emd_nesterov_step_f32: null pointer");
    EMD_REQUIRE(n >= 0, EMD_E_INVALID, "emd_nesterov_step_f32: bad size");
    if (n == 0) return EMD_OK;
    unsigned nb;
    int rc = blocks_for(n, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(nesterov_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, accum, n, lr,
                       momentum, grad_scale);
    return emd::check_launch("nesterov_kernel");
}
