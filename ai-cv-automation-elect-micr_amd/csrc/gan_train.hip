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
                                                   const float* __restrict__ gnorm_sq, float clip_norm) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
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

extern "C" int emd_adam_step_f32(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1,
                                 float beta2, float eps, float grad_scale, const float* gnorm_sq, float clip_norm,
                                 emd_stream_t stream) {
    EMD_REQUIRE(param && grad && m && v, EMD_E_INVALID, "emd_adam_step_f32: null pointer");
    EMD_REQUIRE(n >= 0 && (!gnorm_sq || clip_norm > 0.f), EMD_E_INVALID, "emd_adam_step_f32: bad argument");
    if (n == 0) return EMD_OK;
    unsigned nb;
    int rc = blocks_for(n, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad, m, v, n, lr_t, beta1,
                       beta2, eps, grad_scale, gnorm_sq, clip_norm);
    return emd::check_launch("adam_kernel");
}
