// Training-side kernels of graph G, the in-filling GAN (misc_py/gan-infilling-100.py:982-1088 tower functions,
// :1378-1379 / :1429-1431 Adam wrapped in clip_gradients_by_norm).  Everything here is tiny or HBM-bound; the
// convolution gradients are the graph-D' kernels (wgrad.hip, bn_train.hip, bwd_misc.hip) with a leaky-relu mask.
#include "emd_common.hpp"

namespace {

// Head of one tower (batch_size = 1, :74): out = sigmoid(max(logit[0..2])).
//   mode 0 (discriminator, :1080): loss = -log(clip(1 - |label - out|, 1e-8, 1 - 1e-8))
//   mode 1 (generator, :1037):     loss = -log(clip(out, 1e-8, 1))
// result = {out, loss}; dlogit[k] = grad_scale * dloss/dlogit[k] (non-zero only for the arg-max branch; the first
// maximum wins ties, as tf.reduce_max's gradient splits them -- ties do not occur with real-valued logits).
__global__ void gan_head_kernel(const float* __restrict__ logit, float label, int mode, float grad_scale,
                                float* __restrict__ result, float* __restrict__ dlogit) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int k = 0;
    if (logit[1] > logit[k]) k = 1;
    if (logit[2] > logit[k]) k = 2;
    const float o = 1.f / (1.f + __expf(-logit[k]));
    float loss, dldo;
    if (mode == 0) {
        const float d = label - o, t = 1.f - fabsf(d);
        const float tc = fminf(fmaxf(t, 1e-8f), 1.f - 1e-8f);
        loss = -logf(tc);
        dldo = (t >= 1e-8f && t <= 1.f - 1e-8f) ? -(d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / t : 0.f;
    } else {
        const float oc = fminf(fmaxf(o, 1e-8f), 1.f);
        loss = -logf(oc);
        dldo = (o >= 1e-8f && o <= 1.f) ? -1.f / o : 0.f;
    }
    result[0] = o;
    result[1] = loss;
    dlogit[0] = dlogit[1] = dlogit[2] = 0.f;
    dlogit[k] = grad_scale * dldo * o * (1.f - o);
}

// Fully connected layer to one output, backward for one row: dw[k] += x[k]*g, db += g, dx[k] = w[k]*g, g = *dlogit
__global__ __launch_bounds__(256) void fc_row_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dlogit, float* __restrict__ dw,
                                                         float* __restrict__ db, float* __restrict__ dx, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const float g = *dlogit;
    if (k == 0) atomicAdd(db, g);
    if (k >= K) return;
    atomicAdd(dw + k, x[k] * g);
    dx[k] = w[k] * g;
}

// Gradient of the global mean over H, W (tf.reduce_mean(x, [1,2]), :578): y[p][c] = v[c] * alpha for every pixel
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ v, float* __restrict__ y, int ldy, int C4,
                                                         long nthreads, float alpha) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    const float4 a = *reinterpret_cast<const float4*>(v + c);
    *reinterpret_cast<float4*>(y + (tid / C4) * ldy + c) = make_float4(a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha);
}

// sum of squares of a flat vector (the global gradient norm of clip_gradients_by_norm), double accumulation
__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ x, long n, double* __restrict__ part) {
    __shared__ double sm[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double v = x[i];
        s += v * v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

__global__ void sumsq_final(const double* __restrict__ part, int nblk, float scale, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += part[k];
    out[0] = (float)(s * (double)scale * (double)scale);   // |scale * x|^2
}

// tf.train.AdamOptimizer(lr, beta1) on a flat vector, gradient g = grad * grad_scale * clip_norm / max(|g|, clip_norm)
// with |g|^2 read from the device (no host round trip):  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g^2;
// param -= lr_t * m / (sqrt(v) + eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr_t,
                                                   float beta1, float beta2, float eps, float grad_scale,
                                                   const float* __restrict__ gnorm_sq, float clip_norm,
                                                   const float* __restrict__ lr_t_dev) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (lr_t_dev) lr_t = lr_t_dev[0];
    float s = grad_scale;
    if (gnorm_sq) {
        const float gn = sqrtf(gnorm_sq[0]);
        s *= clip_norm / fmaxf(gn, clip_norm);
    }
    const float g = grad[i] * s;
    const float mi = beta1 * m[i] + (1.f - beta1) * g;
    const float vi = beta2 * v[i] + (1.f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    param[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

__device__ __forceinline__ int reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 fma4s(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}

// ---- reflect-padded depthwise 3x3 (tf.pad REFLECT 1 + VALID, stride 1 or 2), backward.
// Weight gradient: dw[t][c] += sum x[reflect(oy*s-1+ky), reflect(ox*s-1+kx), c] * dy[oy,ox,c].
// SCALAR: dy has one channel (the final 3x3 conv to one output channel).
template <bool SCALAR>
__global__ __launch_bounds__(256) void dw_reflect_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                               int ldd, float* __restrict__ dw, int H, int W, int Ho, int Wo,
                                                               int C, int s, long npix, long pix_per_slab) {
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
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = reflect(oy * s - 1 + ky, H);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = reflect(ox * s - 1 + kx, W);
                    acc[ky * 3 + kx] = fma4(*reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx), g, acc[ky * 3 + kx]);
                }
            }
        }
    }
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

// Data gradient: the gradient w.r.t. the PADDED input at padded position q is gp(q) = sum_k dy[(q-k)/s] * w[k] (where
// divisible and in range); the padding mirrors row -1 onto row 1 and row H onto row H-2, so
// dx[i] = gp(i+1) + [i == 1] gp(0) + [i == H-2] gp(H+1), the same along x.
template <bool SCALAR>
__global__ __launch_bounds__(256) void dw_reflect_bwd_data_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int ldx, int H, int W, int Ho, int Wo,
                                                                  int C4, int s, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    const int c = (int)(tid % C4) * 4;
    long q = tid / C4;
    const int ix = (int)(q % W);
    q /= W;
    const int iy = (int)(q % H);
    const long b = q / H;
    const int C = C4 * 4;
    int qy[3], qx[3], ny = 0, nx = 0;
    qy[ny++] = iy + 1;
    if (iy == 1) qy[ny++] = 0;
    if (iy == H - 2) qy[ny++] = H + 1;
    qx[nx++] = ix + 1;
    if (ix == 1) qx[nx++] = 0;
    if (ix == W - 2) qx[nx++] = W + 1;
    float4 acc = f4zero();
    for (int a = 0; a < ny; ++a)
        for (int e = 0; e < nx; ++e) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int vy = qy[a] - ky;
                if (vy < 0 || vy % s != 0 || vy / s >= Ho) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int vx = qx[e] - kx;
                    if (vx < 0 || vx % s != 0 || vx / s >= Wo) continue;
                    const long op = (b * Ho + vy / s) * (long)Wo + vx / s;
                    const float4 wv = *reinterpret_cast<const float4*>(w + (ky * 3 + kx) * C + c);
                    if (SCALAR)
                        acc = fma4s(wv, dy[op], acc);
                    else
                        acc = fma4(*reinterpret_cast<const float4*>(dy + op * ldd + c), wv, acc);
                }
            }
        }
    *reinterpret_cast<float4*>(dx + ((b * H + iy) * (long)W + ix) * ldx + c) = acc;
}

// ---- the generator's first layer (7x7 depthwise on the 1-channel image, reflect pad 3): forward into channel 0 of a
// 4-channel tensor (the pointwise half then runs as a K = 4 GEMM), and the gradient of its 49 weights.
__global__ __launch_bounds__(256) void dw7_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w49,
                                                         float* __restrict__ d4, int H, int W, long npix) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const int ox = (int)(pix % W);
    const long t = pix / W;
    const int oy = (int)(t % H);
    const float* img = x + (t / H) * (long)H * W;
    float d = 0.f;
    for (int i = 0; i < 7; ++i) {
        const float* row = img + (long)reflect(oy - 3 + i, H) * W;
#pragma unroll
        for (int j = 0; j < 7; ++j) d = fmaf(w49[i * 7 + j], row[reflect(ox - 3 + j, W)], d);
    }
    *reinterpret_cast<float4*>(d4 + pix * 4) = make_float4(d, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void dw7_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dd4,
                                                           float* __restrict__ dw49, int H, int W, long npix) {
    float acc[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) acc[t] = 0.f;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
        const int ox = (int)(pix % W);
        const long t = pix / W;
        const int oy = (int)(t % H);
        const float* img = x + (t / H) * (long)H * W;
        const float g = dd4[pix * 4];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float* row = img + (long)reflect(oy - 3 + i, H) * W;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[i * 7 + j] = fmaf(row[reflect(ox - 3 + j, W)], g, acc[i * 7 + j]);
        }
    }
    __shared__ float red[4][49];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 49; ++t) {
        float v = acc[t];
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wv][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 49) atomicAdd(dw49 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// g = dy * (1 - y^2)   (tf.tanh, :372)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ g, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = dy[i] * (1.f - y[i] * y[i]);
}

// Feature-matching term of the generator loss (:1027-1035): weight * mean|a - b| over one feature map.
// dy[i] (+)= weight * sign(a[i] - b[i]) / n  (the gradient w.r.t. a);  *loss_acc += weight * mean|a - b|.
__global__ __launch_bounds__(256) void l1_feature_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                         float wn, float* __restrict__ dy, int accumulate,
                                                         float* __restrict__ loss_acc) {
    __shared__ float sm[256];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s += fabsf(d);
        const float g = d > 0.f ? wn : (d < 0.f ? -wn : 0.f);
        dy[i] = (accumulate ? dy[i] : 0.f) + g;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss_acc, sm[0] * wn);
}

// Gradient of get_multiscale_crops (:957-980) for one crop: channel 0 of dcrop [n,n,ldc] is added into dimg [S,S]
// at the mirrored position of padded coordinate (y0+i, x0+j), pad = 3S/4.
__global__ __launch_bounds__(256) void crop_scatter_kernel(const float* __restrict__ dcrop, int ldc, float* __restrict__ dimg,
                                                           int y0, int x0, const int* __restrict__ yx_dev, int n, int S, int pad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    if (yx_dev) {  // offsets living on the device: a captured hipGraph is replayed with new crops
        y0 = yx_dev[0];
        x0 = yx_dev[1];
    }
    const int i = idx / n, j = idx % n;
    const int iy = reflect(y0 + i - pad, S), ix = reflect(x0 + j - pad, S);
    atomicAdd(dimg + (long)iy * S + ix, dcrop[(long)idx * ldc]);
}

// Inference-mode double batch norm of a generator separable conv as ONE affine of r, plus what its backward needs:
//   z1 = a1*r + c1 (inner BN, moving statistics), z = a2*(z1 - mu2) + beta2 (outer BN)  =>  scale = a1*a2, shift.
//   mprime, rprime: (z1 - mu2)/s2 = (r - mprime)*rprime, for dgamma2 = sum g*(z1-mu2)/s2;  rstd1 = 1/s1; a2 = gamma2/s2.
__global__ __launch_bounds__(256) void bn_infer_fold2_kernel(const float* __restrict__ g1, const float* __restrict__ b1,
                                                             const float* __restrict__ m1, const float* __restrict__ v1,
                                                             const float* __restrict__ g2, const float* __restrict__ b2,
                                                             const float* __restrict__ m2, const float* __restrict__ v2,
                                                             float eps, int C, float* __restrict__ scale, float* __restrict__ shift,
                                                             float* __restrict__ mprime, float* __restrict__ rprime,
                                                             float* __restrict__ rstd1, float* __restrict__ a2o) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float r1 = rsqrtf(v1[c] + eps), r2 = rsqrtf(v2[c] + eps);
    const float a1 = g1[c] * r1, c1 = b1[c] - m1[c] * a1, a2 = g2[c] * r2;
    scale[c] = a1 * a2;
    shift[c] = (c1 - m2[c]) * a2 + b2[c];
    // (a1*r + c1 - mu2)*r2 = (r - (mu2 - c1)/a1) * (a1*r2); a1 == 0 makes z1 constant: the sum is then s1*(c1-mu2)*r2,
    // which mprime = -(c1 - mu2)*r2 / tiny reproduces only approximately -- gamma1 is never exactly 0 in practice
    const float a1s = fabsf(a1) > 1e-30f ? a1 : 1e-30f;
    mprime[c] = (m2[c] - c1) / a1s;
    rprime[c] = a1s * r2;
    rstd1[c] = r1;
    a2o[c] = a2;
}

// dbeta2 += s1; dgamma2 += t2; dbeta1 += a2*s1; dgamma1 += a2*t1    (s1 = sum g, t2 = sum g*(z1-mu2)/s2, t1 = sum g*(r-mu1)/s1)
__global__ __launch_bounds__(256) void bn_infer_grads_kernel(const float* __restrict__ s1, const float* __restrict__ t1,
                                                             const float* __restrict__ t2, const float* __restrict__ a2, int C,
                                                             float* dg1, float* db1, float* dg2, float* db2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    atomicAdd(db2 + c, s1[c]);
    atomicAdd(dg2 + c, t2[c]);
    atomicAdd(db1 + c, a2[c] * s1[c]);
    atomicAdd(dg1 + c, a2[c] * t1[c]);
}

int blocks_for(long nthreads, unsigned* nb) {
    const long b = (nthreads + 255) / 256;
    if (b <= 0 || b > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "grid too large");
    *nb = (unsigned)b;
    return EMD_OK;
}

}  // namespace

extern "C" int emd_gan_head_f32(const float* logit3, float label, int mode, float grad_scale, float* result2, float* dlogit3,
                                emd_stream_t stream) {
    EMD_REQUIRE(logit3 && result2 && dlogit3, EMD_E_INVALID, "emd_gan_head_f32: null pointer");
    EMD_REQUIRE(mode == 0 || mode == 1, EMD_E_INVALID, "emd_gan_head_f32: mode must be 0 (discriminator) or 1 (generator)");
    hipLaunchKernelGGL(gan_head_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), logit3, label, mode, grad_scale,
                       result2, dlogit3);
    return emd::check_launch("gan_head_kernel");
}

extern "C" int emd_fc_row_bwd_f32(const float* x, const float* w, const float* dlogit, float* dw, float* db, float* dx, int K,
                                  emd_stream_t stream) {
    EMD_REQUIRE(x && w && dlogit && dw && db && dx, EMD_E_INVALID, "emd_fc_row_bwd_f32: null pointer");
    EMD_REQUIRE(K >= 1, EMD_E_INVALID, "emd_fc_row_bwd_f32: bad size");
    hipLaunchKernelGGL(fc_row_bwd_kernel, dim3((K + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, dlogit, dw,
                       db, dx, K);
    return emd::check_launch("fc_row_bwd_kernel");
}

extern "C" int emd_bcast_rows_f32(const float* v, float* y, int ldy, long npix, int C, float alpha, emd_stream_t stream) {
    EMD_REQUIRE(v && y, EMD_E_INVALID, "emd_bcast_rows_f32: null pointer");
    EMD_REQUIRE(npix >= 0 && C >= 4 && C % 4 == 0 && ldy % 4 == 0 && ldy >= C && emd::aligned16(v) && emd::aligned16(y),
                EMD_E_ALIGN, "emd_bcast_rows_f32: C, ldy multiples of 4; 16-byte aligned pointers");
    if (npix == 0) return EMD_OK;
    const long nthreads = npix * (C / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), v, y, ldy, C / 4, nthreads, alpha);
    return emd::check_launch("bcast_rows_kernel");
}

extern "C" size_t emd_sumsq_workspace_bytes(void) { return 1024 * sizeof(double); }

extern "C" int emd_sumsq_f32(const float* x, long n, float scale, float* out, void* workspace, emd_stream_t stream) {
    EMD_REQUIRE(x && out && workspace, EMD_E_INVALID, "emd_sumsq_f32: null pointer");
    EMD_REQUIRE(n >= 1, EMD_E_INVALID, "emd_sumsq_f32: empty input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    long nblk = (n + 256 * 16 - 1) / (256 * 16);
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(sumsq_partial, dim3((unsigned)nblk), dim3(256), 0, st, x, n, static_cast<double*>(workspace));
    hipLaunchKernelGGL(sumsq_final, dim3(1), dim3(64), 0, st, static_cast<const double*>(workspace), (int)nblk, scale, out);
    return emd::check_launch("sumsq");
}

static int adam_entry(float* param, const float* grad, float* m, float* v, long n, float lr_t, const float* lr_t_dev,
                      float beta1, float beta2, float eps, float grad_scale, const float* gnorm_sq, float clip_norm,
                      emd_stream_t stream) {
    EMD_REQUIRE(param && grad && m && v, EMD_E_INVALID, "emd_adam_step_f32: null pointer");
    EMD_REQUIRE(n >= 0 && (!gnorm_sq || clip_norm > 0.f), EMD_E_INVALID, "emd_adam_step_f32: bad argument");
    if (n == 0) return EMD_OK;
    unsigned nb;
    int rc = blocks_for(n, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, m, v, n, lr_t, beta1,
                       beta2, eps, grad_scale, gnorm_sq, clip_norm, lr_t_dev);
    return emd::check_launch("adam_kernel");
}

extern "C" int emd_adam_step_f32(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1,
                                 float beta2, float eps, float grad_scale, const float* gnorm_sq, float clip_norm,
                                 emd_stream_t stream) {
    return adam_entry(param, grad, m, v, n, lr_t, nullptr, beta1, beta2, eps, grad_scale, gnorm_sq, clip_norm, stream);
}

extern "C" int emd_adam_step_dev_f32(float* param, const float* grad, float* m, float* v, long n, const float* lr_t_dev,
                                     float beta1, float beta2, float eps, float grad_scale, const float* gnorm_sq,
                                     float clip_norm, emd_stream_t stream) {
    EMD_REQUIRE(lr_t_dev, EMD_E_INVALID, "emd_adam_step_dev_f32: null pointer");
    return adam_entry(param, grad, m, v, n, 0.f, lr_t_dev, beta1, beta2, eps, grad_scale, gnorm_sq, clip_norm, stream);
}

static bool vec_ok(const float* a, int ld, int C) { return C >= 4 && C % 4 == 0 && ld % 4 == 0 && ld >= C && emd::aligned16(a); }

extern "C" int emd_dw3x3_reflect_wgrad_f32(const float* x, int ldx, const float* dy, int ldd, float* dw, int B, int H, int W,
                                           int C, int stride, emd_stream_t stream) {
    EMD_REQUIRE(x && dy && dw, EMD_E_INVALID, "emd_dw3x3_reflect_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2 && (stride == 1 || stride == 2), EMD_E_INVALID, "emd_dw3x3_reflect_wgrad_f32: bad shape");
    EMD_REQUIRE(vec_ok(x, ldx, C) && vec_ok(dy, ldd, C), EMD_E_ALIGN, "emd_dw3x3_reflect_wgrad_f32: alignment");
    if (B == 0) return EMD_OK;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long npix = (long)B * Ho * Wo;
    long nslab = (npix + 63) / 64;
    if (nslab > 512) nslab = 512;
    hipLaunchKernelGGL(dw_reflect_wgrad_kernel<false>, dim3((C + 63) / 64, (unsigned)nslab), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, dy, ldd, dw, H, W, Ho, Wo, C, stride, npix, (npix + nslab - 1) / nslab);
    return emd::check_launch("dw_reflect_wgrad_kernel");
}

extern "C" int emd_dw3x3_reflect_bwd_data_f32(const float* dy, int ldd, const float* w, float* dx, int ldx, int B, int H, int W,
                                              int C, int stride, emd_stream_t stream) {
    EMD_REQUIRE(dy && w && dx, EMD_E_INVALID, "emd_dw3x3_reflect_bwd_data_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2 && (stride == 1 || stride == 2), EMD_E_INVALID, "emd_dw3x3_reflect_bwd_data_f32: bad shape");
    EMD_REQUIRE(vec_ok(dx, ldx, C) && vec_ok(dy, ldd, C) && emd::aligned16(w), EMD_E_ALIGN, "emd_dw3x3_reflect_bwd_data_f32: alignment");
    if (B == 0) return EMD_OK;
    const long nthreads = (long)B * H * W * (C / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(dw_reflect_bwd_data_kernel<false>, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dy, ldd, w, dx,
                       ldx, H, W, (H - 1) / stride + 1, (W - 1) / stride + 1, C / 4, stride, nthreads);
    return emd::check_launch("dw_reflect_bwd_data_kernel");
}

extern "C" int emd_conv3x3_cout1_reflect_wgrad_f32(const float* x, int ldx, const float* dy, float* dw, int B, int H, int W,
                                                   int Cin, emd_stream_t stream) {
    EMD_REQUIRE(x && dy && dw, EMD_E_INVALID, "emd_conv3x3_cout1_reflect_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2, EMD_E_INVALID, "emd_conv3x3_cout1_reflect_wgrad_f32: bad shape");
    EMD_REQUIRE(vec_ok(x, ldx, Cin), EMD_E_ALIGN, "emd_conv3x3_cout1_reflect_wgrad_f32: alignment");
    if (B == 0) return EMD_OK;
    const long npix = (long)B * H * W;
    long nslab = (npix + 63) / 64;
    if (nslab > 512) nslab = 512;
    hipLaunchKernelGGL(dw_reflect_wgrad_kernel<true>, dim3((Cin + 63) / 64, (unsigned)nslab), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, dy, 1, dw, H, W, H, W, Cin, 1, npix, (npix + nslab - 1) / nslab);
    return emd::check_launch("dw_reflect_wgrad_kernel");
}

extern "C" int emd_conv3x3_cout1_reflect_bwd_data_f32(const float* dy, const float* w, float* dx, int ldx, int B, int H, int W,
                                                      int Cin, emd_stream_t stream) {
    EMD_REQUIRE(dy && w && dx, EMD_E_INVALID, "emd_conv3x3_cout1_reflect_bwd_data_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2, EMD_E_INVALID, "emd_conv3x3_cout1_reflect_bwd_data_f32: bad shape");
    EMD_REQUIRE(vec_ok(dx, ldx, Cin) && emd::aligned16(w), EMD_E_ALIGN, "emd_conv3x3_cout1_reflect_bwd_data_f32: alignment");
    if (B == 0) return EMD_OK;
    const long nthreads = (long)B * H * W * (Cin / 4);
    unsigned nb;
    int rc = blocks_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(dw_reflect_bwd_data_kernel<true>, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dy, 1, w, dx, ldx,
                       H, W, H, W, Cin / 4, 1, nthreads);
    return emd::check_launch("dw_reflect_bwd_data_kernel");
}

extern "C" int emd_dw7_c1_reflect_f32(const float* x, const float* w49, float* d4, int B, int H, int W, emd_stream_t stream) {
    EMD_REQUIRE(x && w49 && d4, EMD_E_INVALID, "emd_dw7_c1_reflect_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 4 && W >= 4 && emd::aligned16(d4), EMD_E_INVALID, "emd_dw7_c1_reflect_f32: bad shape");
    if (B == 0) return EMD_OK;
    const long npix = (long)B * H * W;
    unsigned nb;
    int rc = blocks_for(npix, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(dw7_c1_fwd_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, w49, d4, H, W, npix);
    return emd::check_launch("dw7_c1_fwd_kernel");
}

extern "C" int emd_dw7_c1_reflect_wgrad_f32(const float* x, const float* dd4, float* dw49, int B, int H, int W, emd_stream_t stream) {
    EMD_REQUIRE(x && dd4 && dw49, EMD_E_INVALID, "emd_dw7_c1_reflect_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 4 && W >= 4, EMD_E_INVALID, "emd_dw7_c1_reflect_wgrad_f32: bad shape");
    if (B == 0) return EMD_OK;
    const long npix = (long)B * H * W;
    long nb = (npix + 256 * 8 - 1) / (256 * 8);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(dw7_c1_wgrad_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, dd4, dw49, H, W, npix);
    return emd::check_launch("dw7_c1_wgrad_kernel");
}

extern "C" int emd_tanh_bwd_f32(const float* dy, const float* y, float* g, long n, emd_stream_t stream) {
    EMD_REQUIRE(dy && y && g && n >= 0, EMD_E_INVALID, "emd_tanh_bwd_f32: bad argument");
    if (n == 0) return EMD_OK;
    unsigned nb;
    int rc = blocks_for(n, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, g, n);
    return emd::check_launch("tanh_bwd_kernel");
}

extern "C" int emd_l1_feature_f32(const float* a, const float* b, long n, float weight, float* dy, int accumulate,
                                  float* loss_acc, emd_stream_t stream) {
    EMD_REQUIRE(a && b && dy && loss_acc && n >= 1, EMD_E_INVALID, "emd_l1_feature_f32: bad argument");
    long nb = (n + 256 * 4 - 1) / (256 * 4);
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(l1_feature_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, n,
                       weight / (float)n, dy, accumulate, loss_acc);
    return emd::check_launch("l1_feature_kernel");
}

extern "C" int emd_crop_scatter_f32(const float* dcrop, int ldc, float* dimg, int y0, int x0, int n, int S, emd_stream_t stream) {
    EMD_REQUIRE(dcrop && dimg, EMD_E_INVALID, "emd_crop_scatter_f32: null pointer");
    const int pad = (3 * S) / 4;
    EMD_REQUIRE(S >= 4 && n >= 1 && ldc >= 1 && y0 >= 0 && x0 >= 0 && y0 + n <= S + 2 * pad && x0 + n <= S + 2 * pad, EMD_E_INVALID,
                "emd_crop_scatter_f32: crop outside the padded image");
    hipLaunchKernelGGL(crop_scatter_kernel, dim3((n * n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dcrop, ldc,
                       dimg, y0, x0, nullptr, n, S, pad);
    return emd::check_launch("crop_scatter_kernel");
}

extern "C" int emd_crop_scatter_dev_f32(const float* dcrop, int ldc, float* dimg, const int* yx_dev, int n, int S,
                                        emd_stream_t stream) {
    EMD_REQUIRE(dcrop && dimg && yx_dev, EMD_E_INVALID, "emd_crop_scatter_dev_f32: null pointer");
    EMD_REQUIRE(S >= 4 && n >= 1 && ldc >= 1, EMD_E_INVALID, "emd_crop_scatter_dev_f32: bad shape");
    hipLaunchKernelGGL(crop_scatter_kernel, dim3((n * n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dcrop, ldc,
                       dimg, 0, 0, yx_dev, n, S, (3 * S) / 4);
    return emd::check_launch("crop_scatter_kernel");
}

extern "C" int emd_bn_infer_fold2_f32(const float* g1, const float* b1, const float* m1, const float* v1, const float* g2,
                                      const float* b2, const float* m2, const float* v2, float eps, int C, float* scale,
                                      float* shift, float* mprime, float* rprime, float* rstd1, float* a2, emd_stream_t stream) {
    EMD_REQUIRE(g1 && b1 && m1 && v1 && g2 && b2 && m2 && v2 && scale && shift && mprime && rprime && rstd1 && a2 && C >= 1,
                EMD_E_INVALID, "emd_bn_infer_fold2_f32: bad argument");
    hipLaunchKernelGGL(bn_infer_fold2_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), g1, b1, m1, v1,
                       g2, b2, m2, v2, eps, C, scale, shift, mprime, rprime, rstd1, a2);
    return emd::check_launch("bn_infer_fold2_kernel");
}

extern "C" int emd_bn_infer_grads_f32(const float* s1, const float* t1, const float* t2, const float* a2, int C, float* dg1,
                                      float* db1, float* dg2, float* db2, emd_stream_t stream) {
    EMD_REQUIRE(s1 && t1 && t2 && a2 && dg1 && db1 && dg2 && db2 && C >= 1, EMD_E_INVALID, "emd_bn_infer_grads_f32: bad argument");
    hipLaunchKernelGGL(bn_infer_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), s1, t1, t2, a2,
                       C, dg1, db1, dg2, db2);
    return emd::check_launch("bn_infer_grads_kernel");
}
