// Training-mode batch normalisation of graph D' (misc_py/denoiser-multi-gpu.py:210-223: tf.contrib.layers.batch_norm
// with is_training=phase, fused=True, decay 0.999 (the contrib default), eps 1e-3) -- forward fold and backward.
//
// A layer is  r = conv(x)  ->  [BN1 inside the separable conv]  ->  BN2 (batch_then_activ)  ->  relu6 [-> clip].
// With batch statistics both norms are per-channel affines of r, and the statistics of BN2's input follow from
// those of r analytically (mean2 = beta1, var2 = gamma1^2 * q, q = var1/(var1+eps)), so one statistics pass over r
// serves the whole chain, forward and backward:
//   forward :  z = r*scale + shift,            scale = g1*g2*rstd1*rstd2 (double BN) | g*rstd1 (single BN)
//   backward:  g = dy * mask(z);  s1 = sum g;  t = sum g*rhat,  rhat = (r-mean1)*rstd1
//              dr = K * ( g - s1/N - rhat * (t/N) * c ),   K = scale,
//              c = 1 (single BN) | a^2 + eps*rstd2^2, a = g1*rstd2 (double BN)
//              dbeta2 = s1, dgamma2 = a*t, dgamma1 = g2*rstd2^3*eps*t, dbeta1 = 0      (double BN)
//              dbeta  = s1, dgamma  = t                                                 (single BN)
// (a bias added before a training-mode batch norm cancels: its gradient is zero and it only shifts the moving mean).
#include "emd_common.hpp"
#include "bn_chain_dev.hpp"

namespace {

__device__ __forceinline__ float grad_mask(float dy, float z, int mask) {
    if (mask == 1) return (z > 0.f && z < 6.f) ? dy : 0.f;   // tf.nn.relu6 (Relu6Grad: 0 < z < 6)
    if (mask == 2) return (z > 0.f && z <= 1.f) ? dy : 0.f;  // relu6 then tf.clip_by_value(., 0, 1) (passes on [0,1])
    if (mask == 3) return z > 0.f ? dy : 0.2f * dy;           // tf.nn.leaky_relu, alpha 0.2 (graph G)
    return dy;
}

// Round 4: dy given as the data gradient of a 3x3 conv to ONE output channel (the network's final conv: dy[p][c] = sum_taps g1[p + (1-ky,
// 1-kx)] * w9[ky*3+kx][c], TF SAME) -- never written; formed here from the 1-channel image g1 in dw_bwd_data_kernel<true>'s order (its bits).
struct Cout1Src {
    const float* g1;   // [images][H*W] (NULL: dy is a tensor)
    const float* w9;   // [9][C]
    int H, W;
};
__device__ __forceinline__ float4 cout1_dy(const Cout1Src& s, const float4 (&wk)[9], long r) {
    const long HW = (long)s.H * s.W;
    const long img = r / HW;
    const int rem = (int)(r - img * HW), iy = rem / s.W, ix = rem - iy * s.W;
    const float* gi = s.g1 + img * HW;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ny = iy + 1 - ky;
        if (ny < 0 || ny >= s.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int nx = ix + 1 - kx;
            if (nx < 0 || nx >= s.W) continue;
            const float g = gi[(long)ny * s.W + nx];
            const float4 w = wk[ky * 3 + kx];
            acc = make_float4(fmaf(w.x, g, acc.x), fmaf(w.y, g, acc.y), fmaf(w.z, g, acc.z), fmaf(w.w, g, acc.w));
        }
    }
    return acc;
}

// s1[c] = sum_pix g,  s2[c] = sum_pix g * (x-mean[c])*rstd[c];  g = dy * mask(x*mscale[c] + mshift[c]).
// Two passes, double accumulation (as the forward statistics in dw_misc.hip).
__global__ __launch_bounds__(256) void chan_reduce_partial(const float* __restrict__ dy, int ldd,
                                                           const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                           int mask, long npix, int C, long rows_per_slab,
                                                           double* __restrict__ part) {
    // per-image form (gridDim.z images of npix pixels each, per-image statistics vectors [B][C]): image b = blockIdx.z
    {
        const long b = blockIdx.z;
        dy += b * npix * ldd;
        if (x) { x += b * npix * ldx; mean += b * C; rstd += b * C; }
        if (mask) { mscale += b * C; mshift += b * C; }
        part += b * (long)gridDim.y * 2 * C;
    }
    __shared__ double sm[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    const long r1 = min(r0 + rows_per_slab, npix);
    double s = 0.0, q = 0.0;
    if (c < C) {
        const float mu = x ? mean[c] : 0.f, rs = x ? rstd[c] : 0.f;
        const float ms = mask ? mscale[c] : 0.f, mh = mask ? mshift[c] : 0.f;
        for (long r = r0 + rsub; r < r1; r += 4) {
            const float xv = x ? x[r * ldx + c] : 0.f;
            const float g = grad_mask(dy[r * ldd + c], fmaf(xv, ms, mh), mask);
            s += (double)g;
            q += (double)g * (double)((xv - mu) * rs);
        }
    }
    sm[0][rsub][threadIdx.x & 63] = s;
    sm[1][rsub][threadIdx.x & 63] = q;
    __syncthreads();
    if (rsub == 0 && c < C) {
        const int l = threadIdx.x;
        part[((long)blockIdx.y * 2 + 0) * C + c] = sm[0][0][l] + sm[0][1][l] + sm[0][2][l] + sm[0][3][l];
        part[((long)blockIdx.y * 2 + 1) * C + c] = sm[1][0][l] + sm[1][1][l] + sm[1][2][l] + sm[1][3][l];
    }
}

// The same for C % 4 == 0: 16 channel quads x 16 row lanes per workgroup, 16-byte loads.  C1: dy from a Cout1Src (a template parameter: a
// run-time test inside the unrolled loop kept its loads from being batched).
template <bool C1 = false>
__global__ __launch_bounds__(256) void chan_reduce_partial_v4(const float* __restrict__ dy, int ldd,
                                                              const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                              int mask, long npix, int C, long rows_per_slab,
                                                              double* __restrict__ part, Cout1Src c1 = Cout1Src{nullptr, nullptr, 0, 0}) {
    // per-image form (gridDim.z images of npix pixels each, per-image statistics vectors [B][C]): image b = blockIdx.z
    {
        const long b = blockIdx.z;
        if (C1) c1.g1 += b * npix;
        else dy += b * npix * ldd;
        if (x) { x += b * npix * ldx; mean += b * C; rstd += b * C; }
        if (mask) { mscale += b * C; mshift += b * C; }
        part += b * (long)gridDim.y * 2 * C;
    }
    __shared__ double sm[2][16][64 + 1];
    const int cl = (threadIdx.x & 15) * 4;
    const int c = blockIdx.x * 64 + cl;
    const int rl = threadIdx.x >> 4;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    const long r1 = min(r0 + rows_per_slab, npix);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, ms[4] = {0.f, 0.f, 0.f, 0.f}, mh[4] = {0.f, 0.f, 0.f, 0.f};
        if (x) {
            const float4 a = *reinterpret_cast<const float4*>(mean + c), b4 = *reinterpret_cast<const float4*>(rstd + c);
            mu[0] = a.x; mu[1] = a.y; mu[2] = a.z; mu[3] = a.w;
            rs[0] = b4.x; rs[1] = b4.y; rs[2] = b4.z; rs[3] = b4.w;
        }
        if (mask) {
            const float4 a = *reinterpret_cast<const float4*>(mscale + c), b4 = *reinterpret_cast<const float4*>(mshift + c);
            ms[0] = a.x; ms[1] = a.y; ms[2] = a.z; ms[3] = a.w;
            mh[0] = b4.x; mh[1] = b4.y; mh[2] = b4.z; mh[3] = b4.w;
        }
        float4 wk[C1 ? 9 : 1];
        if constexpr (C1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(c1.w9 + k * C + c);
        }
        auto term = [&](const float4 d, const float4 xv) {
            const float dd[4] = {d.x, d.y, d.z, d.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = grad_mask(dd[k], fmaf(xx[k], ms[k], mh[k]), mask);
                s[k] += (double)g;
                q[k] += (double)g * (double)((xx[k] - mu[k]) * rs[k]);
            }
        };
        long r = r0 + rl;
        if constexpr (!C1) {
            if (x) {   // four rows' loads in flight at a time (same rows, same order of the sums: same bits)
                for (; r + 48 < r1; r += 64) {
                    float4 d4[4], x4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        d4[u] = *reinterpret_cast<const float4*>(dy + (r + 16 * u) * ldd + c);
                        x4[u] = *reinterpret_cast<const float4*>(x + (r + 16 * u) * ldx + c);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) term(d4[u], x4[u]);
                }
            }
        }
        for (; r < r1; r += 16) {
            float4 d;
            if constexpr (C1) d = cout1_dy(c1, wk, r);
            else d = *reinterpret_cast<const float4*>(dy + r * ldd + c);
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x) xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
            term(d, xv);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sm[0][rl][cl + k] = s[k];
        sm[1][rl][cl + k] = q[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, l = threadIdx.x & 63;
        const int cc = blockIdx.x * 64 + l;
        if (cc < C) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += sm[which][k][l];
            part[((long)blockIdx.y * 2 + which) * C + cc] = t;
        }
    }
}

// do_prep: the per-channel step bn_bwd_prep_kernel would run next (K, m1, m2 and the parameter gradients) in the same launch.
// CL channels x (256 / CL) slab lanes per workgroup (16 x 16 originally; 4 / 1 channels for the long partial lists of the large maps: the
// fused depthwise-gradient reduction delivers 1 024 per 512^2 image and a 64-channel layer had four workgroups; see bn_stats_final).
inline int reduce_final_cl(int nslab) { return nslab < 128 ? 16 : (nslab < 1024 ? 4 : 1); }
template <int CL>
__global__ __launch_bounds__(256) void chan_reduce_final(const double* __restrict__ part, int nslab, int C,
                                                         float* __restrict__ s1, float* __restrict__ s2, int accumulate,
                                                         int do_prep, emd::BnPrepArgs pa, float inv_n) {
    constexpr int SL = 256 / CL;
    __shared__ double sm[2][SL][CL + 1];
    {   // per-image form: image b = blockIdx.y
        const long b = blockIdx.y;
        part += b * (long)nslab * 2 * C;
        s1 += b * C;
        if (s2) s2 += b * C;
    }
    const int l = threadIdx.x % CL, k0 = threadIdx.x / CL;
    const int c = blockIdx.x * CL + l;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int k = k0; k < nslab; k += SL) {
            s += part[((long)k * 2 + 0) * C + c];
            q += part[((long)k * 2 + 1) * C + c];
        }
    sm[0][k0][l] = s;
    sm[1][k0][l] = q;
    __syncthreads();
    if constexpr (SL > 16) {
        if (k0 < 16) {
            double s2v = 0.0, q2v = 0.0;
#pragma unroll 4
            for (int k = 0; k < SL / 16; ++k) {
                s2v += sm[0][k0 * (SL / 16) + k][l];
                q2v += sm[1][k0 * (SL / 16) + k][l];
            }
            s = s2v;
            q = q2v;
        }
        __syncthreads();
        if (k0 < 16) {
            sm[0][k0][l] = s;
            sm[1][k0][l] = q;
        }
        __syncthreads();
    }
    if (k0 != 0 || c >= C) return;
    s = q = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        s += sm[0][k][l];
        q += sm[1][k][l];
    }
    if (accumulate) atomicAdd(s1 + c, (float)s);  // bias gradients: towers on different streams add concurrently
    else s1[c] = (float)s;
    if (s2) s2[c] = (float)q;
    if (do_prep) emd::bn_bwd_prep_one(pa, (int)blockIdx.y * C + c, c, (float)s, (float)q, inv_n);
}

static void launch_final(const double* part, int nslab, int C, int B, float* s1, float* s2, int accumulate, const emd::BnPrepArgs* prep, float inv_n,
                         hipStream_t st) {
    const int cl = reduce_final_cl(nslab);
    const emd::BnPrepArgs pa = prep ? *prep : emd::BnPrepArgs{};
    const int dp = prep ? 1 : 0;
    if (cl == 16)
        hipLaunchKernelGGL(chan_reduce_final<16>, dim3((C + 15) / 16, (unsigned)B), dim3(256), 0, st, part, nslab, C, s1, s2, accumulate, dp, pa, inv_n);
    else if (cl == 4)
        hipLaunchKernelGGL(chan_reduce_final<4>, dim3((C + 3) / 4, (unsigned)B), dim3(256), 0, st, part, nslab, C, s1, s2, accumulate, dp, pa, inv_n);
    else
        hipLaunchKernelGGL(chan_reduce_final<1>, dim3(C, (unsigned)B), dim3(256), 0, st, part, nslab, C, s1, s2, accumulate, dp, pa, inv_n);
}

// dx = K * ( g - m1 - (x-mean)*m2 ),  g = dy * mask(x*mscale + mshift); dx may alias dy (elementwise).
// V = 4: a thread keeps the six per-channel vectors of its channel quad in registers and walks ROWS rows with them.
template <int V, bool C1 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, int ldd, const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ K, const float* __restrict__ m1,
                                                           const float* __restrict__ mean, const float* __restrict__ m2,
                                                           const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                           int mask, float* dx, int ldo, long npix, int CV,
                                                           Cout1Src c1 = Cout1Src{nullptr, nullptr, 0, 0}) {
    constexpr int ROWS = V == 4 ? 8 : 1;
    {   // per-image form: image b = blockIdx.y, npix pixels per image, per-image vectors [B][C]
        const long b = blockIdx.y;
        if (C1) c1.g1 += b * npix;
        else dy += b * npix * ldd;
        x += b * npix * ldx; dx += b * npix * ldo;
        K += b * CV * V; m1 += b * CV * V; mean += b * CV * V; m2 += b * CV * V;
        if (mask) { mscale += b * CV * V; mshift += b * CV * V; }
    }
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const long ngroups = (npix + ROWS - 1) / ROWS;
    if (tid >= ngroups * CV) return;
    const int c = (int)(tid % CV) * V;
    const long r0 = (tid / CV) * ROWS;
    float kk[V], mm1[V], mu[V], mm2[V], ms[V], mh[V];
    if constexpr (V == 4) {   // six 16-byte loads (the vectors are 16-byte aligned device allocations, C % 4 == 0), not twenty-four
        const float4 a = *reinterpret_cast<const float4*>(K + c), b4 = *reinterpret_cast<const float4*>(m1 + c);
        const float4 d4 = *reinterpret_cast<const float4*>(mean + c), e = *reinterpret_cast<const float4*>(m2 + c);
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f), g4 = f;
        if (mask) {
            f = *reinterpret_cast<const float4*>(mscale + c);
            g4 = *reinterpret_cast<const float4*>(mshift + c);
        }
        kk[0] = a.x; kk[1] = a.y; kk[2] = a.z; kk[3] = a.w;
        mm1[0] = b4.x; mm1[1] = b4.y; mm1[2] = b4.z; mm1[3] = b4.w;
        mu[0] = d4.x; mu[1] = d4.y; mu[2] = d4.z; mu[3] = d4.w;
        mm2[0] = e.x; mm2[1] = e.y; mm2[2] = e.z; mm2[3] = e.w;
        ms[0] = f.x; ms[1] = f.y; ms[2] = f.z; ms[3] = f.w;
        mh[0] = g4.x; mh[1] = g4.y; mh[2] = g4.z; mh[3] = g4.w;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            kk[k] = K[c + k]; mm1[k] = m1[c + k]; mu[k] = mean[c + k]; mm2[k] = m2[c + k];
            ms[k] = mask ? mscale[c + k] : 0.f; mh[k] = mask ? mshift[c + k] : 0.f;
        }
    }
    float4 wk[C1 ? 9 : 1];
    if constexpr (V == 4 && C1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(c1.w9 + k * (CV * 4) + c);
    }
    // (V == 4, dy a tensor: all sixteen loads of the thread's eight rows first -- rows past the end re-read the last one --, then the
    // arithmetic and the stores: with a load, a wait and a break per row the small maps of the 1/16-resolution flow, a handful of waves
    // per CU, paid eight dependent round trips per thread)
    float4 dpre[(V == 4 && !C1) ? ROWS : 1], xpre[(V == 4 && !C1) ? ROWS : 1];
    if constexpr (V == 4 && !C1) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const long r = r0 + i < npix ? r0 + i : npix - 1;
            dpre[i] = *reinterpret_cast<const float4*>(dy + r * ldd + c);
            xpre[i] = *reinterpret_cast<const float4*>(x + r * ldx + c);
        }
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const long r = r0 + i;
        if (r >= npix) break;
        float dd[V], xx[V], o[V];
        if constexpr (V == 4 && !C1) {
            const float4 d = dpre[i], xv = xpre[i];
            dd[0] = d.x; dd[1] = d.y; dd[2] = d.z; dd[3] = d.w;
            xx[0] = xv.x; xx[1] = xv.y; xx[2] = xv.z; xx[3] = xv.w;
        } else if constexpr (V == 4) {
            float4 d;
            if constexpr (C1) d = cout1_dy(c1, wk, r);
            else d = *reinterpret_cast<const float4*>(dy + r * ldd + c);
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
            dd[0] = d.x; dd[1] = d.y; dd[2] = d.z; dd[3] = d.w;
            xx[0] = xv.x; xx[1] = xv.y; xx[2] = xv.z; xx[3] = xv.w;
        } else {
            dd[0] = dy[r * ldd + c];
            xx[0] = x[r * ldx + c];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float g = grad_mask(dd[k], fmaf(xx[k], ms[k], mh[k]), mask);
            o[k] = kk[k] * (g - mm1[k] - (xx[k] - mu[k]) * mm2[k]);
        }
        if constexpr (V == 4)
            *reinterpret_cast<float4*>(dx + r * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
        else
            dx[r * ldo + c] = o[0];
    }
}

// Forward fold for training: batch statistics -> the affine of the whole BN chain, what backward needs, and the
// moving-average updates in assign_moving_average's own form, variable -= (variable - value) * float32(1 - decay)
// (decay 0.999; the moving variance takes the unbiased batch variance, as TF's fused batch norm reports it).
__global__ __launch_bounds__(256) void bn_train_fold_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                            const float* __restrict__ gamma1, const float* __restrict__ beta1,
                                                            const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                            const float* __restrict__ bias, float eps, float n, int C,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ rstd1, float* __restrict__ rstd2,
                                                            float* mm1, float* mv1, float* mm2, float* mv2, float omd, int period) {
    // per-image form: C = B * period entries [B][period]; the parameters repeat with `period`, the moving statistics follow
    // image 0 only (the first tower, misc_py/denoiser-multi-gpu.py:701-707)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C) return;
    const emd::BnFoldArgs a{gamma1, beta1, gamma2, beta2, bias, eps, omd, scale, shift, rstd1, rstd2, mm1, mv1, mm2, mv2};
    emd::bn_train_fold_one(a, i, i % period, i < period, mean[i], var[i], n);
}

// Per-channel step between the backward reduction and the elementwise apply; parameter gradients ACCUMULATE
// (several towers / micro-batches add into one gradient set, misc_py/denoiser-multi-gpu.py:1040).
__global__ __launch_bounds__(256) void bn_bwd_prep_kernel(const float* __restrict__ s1, const float* __restrict__ t,
                                                          const float* __restrict__ gamma1, const float* __restrict__ gamma2,
                                                          const float* __restrict__ rstd1, const float* __restrict__ rstd2,
                                                          float eps, float inv_n, int C, float* __restrict__ K,
                                                          float* __restrict__ m1, float* __restrict__ m2,
                                                          float* dgamma1, float* dgamma2, float* dbeta2, int period) {
    // per-image form: C = B * period entries; parameters and their gradients are indexed by the channel i % period
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C) return;
    const emd::BnPrepArgs a{gamma1, gamma2, rstd1, rstd2, eps, K, m1, m2, dgamma1, dgamma2, dbeta2};
    emd::bn_bwd_prep_one(a, i, i % period, s1[i], t[i], inv_n);
}

// ------------------------------------------------------------------------------------------------
// Round 4: the batch norm of a SMALL map as one launch per direction.  A tower's 1/16- and 1/8-resolution layers (32 x 32 and 64 x 64
// pixels per image, 36 + 9 of graph D's 60 norms) ran as statistics partial + final + fold + affine forward and reduction partial +
// final + prep + apply backward: eight launches of 4-20 us on 3-12 MB, none of which fills the chip.  Here one workgroup owns 64
// channels of ONE image for all of its pixels (16 channel quads x 16 row lanes, 16-byte accesses): pass 1 reduces in double, the
// per-channel step runs in the workgroup, pass 2 re-reads the slice (L2) and writes.  Same formulas, statement for statement, as
// bn_stats_final + bn_train_fold_kernel + affine_relu6_kernel and chan_reduce_* + bn_bwd_prep_kernel + bn_bwd_apply_kernel; the sums
// are cut differently (16 row lanes over the whole image instead of slabs), so results agree to double rounding, not bit for bit.
// blockIdx.y = image; every per-channel vector is [B][C]; parameters and moving statistics are indexed by the channel, the moving
// statistics follow image 0 (the first tower, misc_py/denoiser-multi-gpu.py:701-707).
// Geometry: CB = 16 channels (CQ = 4 quads) x RL = 64 row lanes per workgroup -- a first version with 64 channels x 16 row lanes had 24
// workgroups for a pair of 32 x 32 x 728 maps, each walking 64 rows twice: 50 us of latency in a chain of 5-20 us kernels, and the
// step got 5 ms SLOWER (the step is bound by the length of each stream's chain, not by the work).
constexpr int CB = 16, CQ = CB / 4, RL = 256 / CQ;
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(const float* r, int ldr, long npix, int C, const float* __restrict__ gamma1,
                                                           const float* __restrict__ beta1, const float* __restrict__ gamma2,
                                                           const float* __restrict__ beta2, const float* __restrict__ bias, float eps,
                                                           float omd, float* mm1, float* mv1, float* mm2, float* mv2,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ rstd1, float* __restrict__ rstd2,
                                                           float* __restrict__ mean_out, const float* res, int ldres, float* out,
                                                           int ldo, int act) {
    __shared__ double sm[2][RL][CB + 1];
    __shared__ float fs[2][CB];          // scale, shift of this workgroup's channels
    const long b = blockIdx.y;
    r += b * npix * ldr;
    out += b * npix * ldo;
    if (res) res += b * npix * ldres;
    const int cl = (threadIdx.x % CQ) * 4, rl = threadIdx.x / CQ;
    const int c = blockIdx.x * CB + cl;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
#pragma unroll 4
        for (long p = rl; p < npix; p += RL) {
            const float4 v = *reinterpret_cast<const float4*>(r + p * ldr + c);
            const float xx[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double d = (double)xx[k];
                s[k] += d;
                q[k] += d * d;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sm[0][rl][cl + k] = s[k];
        sm[1][rl][cl + k] = q[k];
    }
    __syncthreads();
    if (threadIdx.x < CB) {
        const int l = threadIdx.x, cc = blockIdx.x * CB + l;
        if (cc < C) {
            double ss = 0.0, qq = 0.0;
            for (int k = 0; k < RL; ++k) {
                ss += sm[0][k][l];
                qq += sm[1][k][l];
            }
            // bn_stats_final
            const double m = ss / (double)npix;
            const double vd = qq / (double)npix - m * m;
            const float mu = (float)m, v = (float)(vd > 0.0 ? vd : 0.0);
            // bn_train_fold_kernel
            const long i = b * C + cc;
            const float n = (float)npix;
            const float r1 = rsqrtf(v + eps);
            const float bessel = n > 1.f ? n / (n - 1.f) : 1.f;
            const bool mov = b == 0 && mm2 != nullptr;
            rstd1[i] = r1;
            mean_out[i] = mu;
            float sc, sh;
            if (gamma1) {
                const float g1 = gamma1[cc];
                const float var2 = g1 * g1 * v * r1 * r1;
                const float r2 = rsqrtf(var2 + eps);
                rstd2[i] = r2;
                sc = g1 * gamma2[cc] * r1 * r2;
                sh = beta2[cc] - mu * sc;
                if (mov) {
                    mm1[cc] -= (mm1[cc] - mu) * omd;
                    mv1[cc] -= (mv1[cc] - v * bessel) * omd;
                    mm2[cc] -= (mm2[cc] - beta1[cc]) * omd;
                    mv2[cc] -= (mv2[cc] - var2 * bessel) * omd;
                }
            } else {
                sc = gamma2[cc] * r1;
                sh = beta2[cc] - mu * sc;
                if (mov) {
                    mm2[cc] -= (mm2[cc] - (mu + (bias ? bias[cc] : 0.f))) * omd;
                    mv2[cc] -= (mv2[cc] - v * bessel) * omd;
                }
            }
            scale[i] = sc;
            shift[i] = sh;
            fs[0][l] = sc;
            fs[1][l] = sh;
        }
    }
    __syncthreads();
    if (c >= C) return;
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sc[k] = fs[0][cl + k]; sh[k] = fs[1][cl + k]; }
    const float hi = act == 2 ? __builtin_inff() : (act == 3 ? 1.f : 6.f);   // affine_relu6_kernel's codes (3: relu6 then clip to [0,1])
#pragma unroll 4
    for (long p = rl; p < npix; p += RL) {
        const float4 v = *reinterpret_cast<const float4*>(r + p * ldr + c);
        float o[4] = {fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]), fmaf(v.w, sc[3], sh[3])};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (act == 4) o[k] = o[k] > 0.f ? o[k] : 0.2f * o[k];
            else if (act) o[k] = fminf(fmaxf(o[k], 0.f), hi);
        }
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + p * ldres + c);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
        }
        *reinterpret_cast<float4*>(out + p * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const float* dy, int ldd, const float* x, int ldx, long npix, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd1,
                                                           const float* __restrict__ rstd2, const float* __restrict__ mscale,
                                                           const float* __restrict__ mshift, int mask,
                                                           const float* __restrict__ gamma1, const float* __restrict__ gamma2, float eps,
                                                           float* dgamma1, float* dgamma2, float* dbeta2, float* dx, int ldo) {
    __shared__ double sm[2][RL][CB + 1];
    __shared__ float cf[3][CB];          // K, m1, m2 of this workgroup's channels
    const long b = blockIdx.y;
    dy += b * npix * ldd;
    x += b * npix * ldx;
    dx += b * npix * ldo;
    const long vo = b * C;
    const int cl = (threadIdx.x % CQ) * 4, rl = threadIdx.x / CQ;
    const int c = blockIdx.x * CB + cl;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, ms[4] = {0.f, 0.f, 0.f, 0.f}, mh[4] = {0.f, 0.f, 0.f, 0.f};
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mu[k] = mean[vo + c + k]; rs[k] = rstd1[vo + c + k];
            if (mask) { ms[k] = mscale[vo + c + k]; mh[k] = mshift[vo + c + k]; }
        }
#pragma unroll 4
        for (long p = rl; p < npix; p += RL) {
            const float4 d = *reinterpret_cast<const float4*>(dy + p * ldd + c);
            const float4 xv = *reinterpret_cast<const float4*>(x + p * ldx + c);
            const float dd[4] = {d.x, d.y, d.z, d.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // chan_reduce_partial_v4
                const float g = grad_mask(dd[k], fmaf(xx[k], ms[k], mh[k]), mask);
                s[k] += (double)g;
                q[k] += (double)g * (double)((xx[k] - mu[k]) * rs[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sm[0][rl][cl + k] = s[k];
        sm[1][rl][cl + k] = q[k];
    }
    __syncthreads();
    if (threadIdx.x < CB) {
        const int l = threadIdx.x, cc = blockIdx.x * CB + l;
        if (cc < C) {
            double ss = 0.0, qq = 0.0;
            for (int k = 0; k < RL; ++k) {
                ss += sm[0][k][l];
                qq += sm[1][k][l];
            }
            // chan_reduce_final -> bn_bwd_prep_kernel
            const float sv = (float)ss, tv = (float)qq;
            const float inv_n = 1.f / (float)npix;
            const float r1 = rstd1[vo + cc];
            float Kc, m2c;
            atomicAdd(dbeta2 + cc, sv);
            if (gamma1) {
                const float g1 = gamma1[cc], g2 = gamma2[cc], r2 = rstd2[vo + cc];
                const float a = g1 * r2;
                const float e2 = eps * r2 * r2;
                Kc = g1 * g2 * r1 * r2;
                m2c = r1 * tv * inv_n * (a * a + e2);
                atomicAdd(dgamma2 + cc, a * tv);
                atomicAdd(dgamma1 + cc, g2 * r2 * e2 * tv);
            } else {
                Kc = gamma2[cc] * r1;
                m2c = r1 * tv * inv_n;
                atomicAdd(dgamma2 + cc, tv);
            }
            cf[0][l] = Kc;
            cf[1][l] = sv * inv_n;
            cf[2][l] = m2c;
        }
    }
    __syncthreads();
    if (c >= C) return;
    float kk[4], m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { kk[k] = cf[0][cl + k]; m1[k] = cf[1][cl + k]; m2[k] = cf[2][cl + k]; }
#pragma unroll 4
    for (long p = rl; p < npix; p += RL) {   // bn_bwd_apply_kernel; dx may alias dy or x (each element is read before it is written, by this thread)
        const float4 d = *reinterpret_cast<const float4*>(dy + p * ldd + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + p * ldx + c);
        const float dd[4] = {d.x, d.y, d.z, d.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = grad_mask(dd[k], fmaf(xx[k], ms[k], mh[k]), mask);
            o[k] = kk[k] * (g - m1[k] - (xx[k] - mu[k]) * m2[k]);
        }
        *reinterpret_cast<float4*>(dx + p * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

extern "C" size_t emd_chan_reduce_workspace_bytes(long npix, int C) {
    if (npix <= 0 || C <= 0) return 0;
    return (size_t)emd::reduce_slabs(npix) * 2 * C * sizeof(double);
}

// B = 1: the batch forms (npix = all pixels of the tower).  B > 1: the per-image forms -- npix pixels PER IMAGE, statistics /
// coefficient vectors [B][C]; image b is reduced exactly as it would be alone (same slabs, same order), so a batch of one-image
// towers (misc_py/denoiser-multi-gpu.py:763) can run as one batched pass with identical arithmetic per image.
static int bwd_reduce_impl(const float* dy, int ldd, const float* x, int ldx, const float* mean, const float* rstd,
                           const float* mscale, const float* mshift, int mask, int B, long npix, int C, float* s1, float* s2,
                           int accumulate_s1, void* workspace, emd_stream_t stream, const emd::BnPrepArgs* prep = nullptr,
                           Cout1Src c1 = Cout1Src{nullptr, nullptr, 0, 0}) {
    EMD_REQUIRE((dy || c1.g1) && s1 && workspace, EMD_E_INVALID, "emd_bn_bwd_reduce_f32: null pointer");
    EMD_REQUIRE(!c1.g1 || (c1.w9 && x && C % 4 == 0 && ldx % 4 == 0 && emd::aligned16(x) && emd::aligned16(c1.w9) && c1.H >= 1 && c1.W >= 1 &&
                           npix % ((long)c1.H * c1.W) == 0), EMD_E_INVALID, "emd_bn_bwd_reduce_prep_cout1_f32: bad argument");
    EMD_REQUIRE(!prep || (x && !accumulate_s1), EMD_E_INVALID, "emd_bn_bwd_reduce_prep_f32: the per-channel step needs x (both sums)");
    EMD_REQUIRE(B >= 1 && B <= 65535 && npix >= 1 && C >= 1 && mask >= 0 && mask <= 3, EMD_E_INVALID, "emd_bn_bwd_reduce_f32: bad argument");
    EMD_REQUIRE(!x || (mean && rstd && s2), EMD_E_INVALID, "emd_bn_bwd_reduce_f32: x needs mean, rstd and s2");
    EMD_REQUIRE(!mask || (x && mscale && mshift), EMD_E_INVALID, "emd_bn_bwd_reduce_f32: a mask needs x, mscale, mshift");
    EMD_REQUIRE(B == 1 || !accumulate_s1, EMD_E_INVALID, "emd_bn_bwd_reduce_images_f32: accumulate_s1 is a batch-form option");
    const long ns = emd::reduce_slabs(npix), rps = emd::reduce_rows_per_slab(npix);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (c1.g1 || (C % 4 == 0 && ldd % 4 == 0 && (!x || ldx % 4 == 0) && emd::aligned16(dy) && (!x || emd::aligned16(x))))
        if (c1.g1)
            hipLaunchKernelGGL(chan_reduce_partial_v4<true>, dim3((C + 63) / 64, (unsigned)ns, (unsigned)B), dim3(256), 0, st, dy, ldd, x, ldx, mean,
                               rstd, mscale, mshift, mask, npix, C, rps, ws, c1);
        else
            hipLaunchKernelGGL(chan_reduce_partial_v4<false>, dim3((C + 63) / 64, (unsigned)ns, (unsigned)B), dim3(256), 0, st, dy, ldd, x, ldx, mean,
                               rstd, mscale, mshift, mask, npix, C, rps, ws, c1);
    else
        hipLaunchKernelGGL(chan_reduce_partial, dim3((C + 63) / 64, (unsigned)ns, (unsigned)B), dim3(256), 0, st, dy, ldd, x, ldx, mean,
                           rstd, mscale, mshift, mask, npix, C, rps, ws);
    launch_final(static_cast<const double*>(ws), (int)ns, C, B, s1, x ? s2 : nullptr, accumulate_s1, prep, 1.0f / (float)npix, st);
    return emd::check_launch("chan_reduce");
}

extern "C" int emd_bn_bwd_reduce_f32(const float* dy, int ldd, const float* x, int ldx, const float* mean,
                                     const float* rstd, const float* mscale, const float* mshift, int mask, long npix,
                                     int C, float* s1, float* s2, int accumulate_s1, void* workspace, emd_stream_t stream) {
    return bwd_reduce_impl(dy, ldd, x, ldx, mean, rstd, mscale, mshift, mask, 1, npix, C, s1, s2, accumulate_s1, workspace, stream);
}

extern "C" int emd_bn_bwd_reduce_images_f32(const float* dy, int ldd, const float* x, int ldx, const float* mean,
                                            const float* rstd, const float* mscale, const float* mshift, int mask, int B,
                                            long npix, int C, float* s1, float* s2, void* workspace, emd_stream_t stream) {
    return bwd_reduce_impl(dy, ldd, x, ldx, mean, rstd, mscale, mshift, mask, B, npix, C, s1, s2, 0, workspace, stream);
}

// chan_reduce_final on partials another kernel produced in chan_reduce_partial_v4's layout ([image][slab][2][C] doubles): dw_bn_bwd.hip
int emd::launch_chan_reduce_final(const double* part, int nslab, int C, int B, float* s1, float* s2, hipStream_t st,
                                  const emd::BnPrepArgs* prep, long npix) {
    EMD_REQUIRE(part && s1 && nslab >= 1 && C >= 1 && B >= 1 && B <= 65535, EMD_E_INVALID, "chan_reduce_final: bad argument");
    EMD_REQUIRE(!prep || (s2 && npix >= 1), EMD_E_INVALID, "chan_reduce_final: the per-channel step needs both sums and the pixel count");
    launch_final(part, nslab, C, B, s1, s2, 0, prep, prep ? 1.0f / (float)npix : 0.f, st);
    return emd::check_launch("chan_reduce_final");
}

int emd::bn_fold_args(const emd_bn_train_fold_t* p, emd::BnFoldArgs* out) {
    EMD_REQUIRE(p && p->gamma2 && p->beta2 && p->scale && p->shift && p->rstd1, EMD_E_INVALID, "emd_bn_train_fold_t: null pointer");
    EMD_REQUIRE((p->gamma1 == nullptr) == (p->beta1 == nullptr) && (!p->gamma1 || p->rstd2), EMD_E_INVALID,
                "emd_bn_train_fold_t: the double batch norm needs gamma1, beta1 and rstd2");
    EMD_REQUIRE(!p->mm2 || p->mv2, EMD_E_INVALID, "emd_bn_train_fold_t: moving mean and variance come together");
    EMD_REQUIRE(!p->gamma1 || ((p->mm1 == nullptr) == (p->mm2 == nullptr) && (!p->mm1 || p->mv1)), EMD_E_INVALID,
                "emd_bn_train_fold_t: the double batch norm updates both sets of moving statistics or none");
    *out = emd::BnFoldArgs{p->gamma1, p->beta1, p->gamma2, p->beta2, p->bias, p->eps, (float)(1.0 - p->decay), p->scale, p->shift, p->rstd1,
                           p->rstd2, p->mm1, p->mv1, p->mm2, p->mv2};
    return EMD_OK;
}

// emd_bn_bwd_prep_t -> the device-side argument block, checked
int emd::bn_prep_args(const emd_bn_bwd_prep_t* p, emd::BnPrepArgs* out) {
    EMD_REQUIRE(p && p->gamma2 && p->rstd1 && p->K && p->m1 && p->m2 && p->dgamma2 && p->dbeta2, EMD_E_INVALID, "emd_bn_bwd_prep_t: null pointer");
    EMD_REQUIRE(!p->gamma1 || (p->rstd2 && p->dgamma1), EMD_E_INVALID, "emd_bn_bwd_prep_t: the double batch norm needs rstd2 and dgamma1");
    *out = emd::BnPrepArgs{p->gamma1, p->gamma2, p->rstd1, p->rstd2, p->eps, p->K, p->m1, p->m2, p->dgamma1, p->dgamma2, p->dbeta2};
    return EMD_OK;
}

// emd_bn_bwd_reduce[_images]_f32 + emd_bn_bwd_prep[_images]_f32 in two launches instead of three (round 4: the per-channel step runs in
// the reduction's final kernel).  images = 0: batch form (vectors [C], npix = all pixels); images = B: per-image form.
extern "C" int emd_bn_bwd_reduce_prep_f32(const float* dy, int ldd, const float* x, int ldx, const float* mean, const float* rstd,
                                          const float* mscale, const float* mshift, int mask, int images, long npix, int C, float* s1,
                                          float* s2, void* workspace, const emd_bn_bwd_prep_t* prep, emd_stream_t stream) {
    emd::BnPrepArgs pa;
    int rc = emd::bn_prep_args(prep, &pa);
    if (rc != EMD_OK) return rc;
    return bwd_reduce_impl(dy, ldd, x, ldx, mean, rstd, mscale, mshift, mask, images ? images : 1, npix, C, s1, s2, 0, workspace, stream, &pa);
}

static int bwd_apply_impl(const float* dy, int ldd, const float* x, int ldx, const float* K, const float* m1,
                          const float* mean, const float* m2, const float* mscale, const float* mshift,
                          int mask, float* dx, int ldo, int B, long npix, int C, emd_stream_t stream,
                          Cout1Src c1 = Cout1Src{nullptr, nullptr, 0, 0}) {
    EMD_REQUIRE((dy || c1.g1) && x && K && m1 && mean && m2 && dx, EMD_E_INVALID, "emd_bn_bwd_apply_f32: null pointer");
    EMD_REQUIRE(!c1.g1 || (c1.w9 && C % 4 == 0 && emd::aligned16(c1.w9) && c1.H >= 1 && c1.W >= 1 && npix % ((long)c1.H * c1.W) == 0),
                EMD_E_INVALID, "emd_bn_bwd_apply_cout1_f32: bad argument");
    EMD_REQUIRE(B >= 1 && B <= 65535 && npix >= 1 && C >= 1 && mask >= 0 && mask <= 3 && (!mask || (mscale && mshift)), EMD_E_INVALID,
                "emd_bn_bwd_apply_f32: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C % 4 == 0 && (c1.g1 || (ldd % 4 == 0 && emd::aligned16(dy))) && ldx % 4 == 0 && ldo % 4 == 0 && emd::aligned16(x) &&
        emd::aligned16(dx)) {
        const long n = ((npix + 7) / 8) * (C / 4);
        if (c1.g1)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, dy, ldd, x, ldx, K,
                               m1, mean, m2, mscale, mshift, mask, dx, ldo, npix, C / 4, c1);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<4, false>), dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, dy, ldd, x, ldx, K,
                               m1, mean, m2, mscale, mshift, mask, dx, ldo, npix, C / 4, c1);
    } else {
        EMD_REQUIRE(!c1.g1, EMD_E_ALIGN, "emd_bn_bwd_apply_cout1_f32: C, pitches multiples of 4; 16-byte aligned tensors");
        const long n = npix * C;
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, dy, ldd, x, ldx, K,
                           m1, mean, m2, mscale, mshift, mask, dx, ldo, npix, C);
    }
    return emd::check_launch("bn_bwd_apply_kernel");
}

extern "C" int emd_bn_bwd_apply_f32(const float* dy, int ldd, const float* x, int ldx, const float* K, const float* m1,
                                    const float* mean, const float* m2, const float* mscale, const float* mshift,
                                    int mask, float* dx, int ldo, long npix, int C, emd_stream_t stream) {
    return bwd_apply_impl(dy, ldd, x, ldx, K, m1, mean, m2, mscale, mshift, mask, dx, ldo, 1, npix, C, stream);
}

extern "C" int emd_bn_bwd_apply_images_f32(const float* dy, int ldd, const float* x, int ldx, const float* K, const float* m1,
                                           const float* mean, const float* m2, const float* mscale, const float* mshift,
                                           int mask, float* dx, int ldo, int B, long npix, int C, emd_stream_t stream) {
    return bwd_apply_impl(dy, ldd, x, ldx, K, m1, mean, m2, mscale, mshift, mask, dx, ldo, B, npix, C, stream);
}

// The two passes for a gradient that is the data gradient of a 3x3 conv to one output channel (the final conv): dy never exists.
// g1 [B][H][W] (images = B: per-image vectors, else one reduction over all B images), w9 [9][C] the conv's weights.
extern "C" int emd_bn_bwd_reduce_prep_cout1_f32(const float* g1, const float* w9, int B, int H, int W, const float* x, int ldx, const float* mean,
                                                const float* rstd, const float* mscale, const float* mshift, int mask, int images, int C,
                                                float* s1, float* s2, void* workspace, const emd_bn_bwd_prep_t* prep, emd_stream_t stream) {
    EMD_REQUIRE(g1 && B >= 1 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_bn_bwd_reduce_prep_cout1_f32: bad argument");
    emd::BnPrepArgs pa;
    int rc = emd::bn_prep_args(prep, &pa);
    if (rc != EMD_OK) return rc;
    const long hw = (long)H * W;
    return bwd_reduce_impl(nullptr, 0, x, ldx, mean, rstd, mscale, mshift, mask, images ? B : 1, images ? hw : hw * B, C, s1, s2, 0, workspace, stream,
                           &pa, Cout1Src{g1, w9, H, W});
}

extern "C" int emd_bn_bwd_apply_cout1_f32(const float* g1, const float* w9, int B, int H, int W, const float* x, int ldx, const float* K,
                                          const float* m1, const float* mean, const float* m2, const float* mscale, const float* mshift,
                                          int mask, int images, float* dx, int ldo, int C, emd_stream_t stream) {
    EMD_REQUIRE(g1 && B >= 1 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_bn_bwd_apply_cout1_f32: bad argument");
    const long hw = (long)H * W;
    return bwd_apply_impl(nullptr, 0, x, ldx, K, m1, mean, m2, mscale, mshift, mask, dx, ldo, images ? B : 1, images ? hw : hw * B, C, stream,
                          Cout1Src{g1, w9, H, W});
}

static int train_fold_impl(const float* mean, const float* var, const float* gamma1, const float* beta1,
                           const float* gamma2, const float* beta2, const float* bias, float eps, long npix, int B,
                           int C, float* scale, float* shift, float* rstd1, float* rstd2, float* mm1, float* mv1,
                           float* mm2, float* mv2, double decay, emd_stream_t stream) {
    EMD_REQUIRE(mean && var && gamma2 && beta2 && scale && shift && rstd1, EMD_E_INVALID, "emd_bn_train_fold_f32: null pointer");
    EMD_REQUIRE((gamma1 == nullptr) == (beta1 == nullptr) && (!gamma1 || rstd2), EMD_E_INVALID,
                "emd_bn_train_fold_f32: the double batch norm needs gamma1, beta1 and rstd2");
    EMD_REQUIRE(!mm2 || mv2, EMD_E_INVALID, "emd_bn_train_fold_f32: moving mean and variance come together");
    EMD_REQUIRE(!gamma1 || ((mm1 == nullptr) == (mm2 == nullptr) && (!mm1 || mv1)), EMD_E_INVALID,
                "emd_bn_train_fold_f32: the double batch norm updates both sets of moving statistics or none");
    EMD_REQUIRE(npix >= 1 && C >= 1 && B >= 1 && (long)B * C <= 0x7fffffffL, EMD_E_INVALID, "emd_bn_train_fold_f32: bad shape");
    const int n = B * C;
    hipLaunchKernelGGL(bn_train_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), mean,
                       var, gamma1, beta1, gamma2, beta2, bias, eps, (float)npix, n, scale, shift, rstd1, rstd2, mm1, mv1,
                       mm2, mv2, (float)(1.0 - decay), C);
    return emd::check_launch("bn_train_fold_kernel");
}

extern "C" int emd_bn_train_fold_f32(const float* mean, const float* var, const float* gamma1, const float* beta1,
                                     const float* gamma2, const float* beta2, const float* bias, float eps, long npix,
                                     int C, float* scale, float* shift, float* rstd1, float* rstd2, float* mm1, float* mv1,
                                     float* mm2, float* mv2, double decay, emd_stream_t stream) {
    return train_fold_impl(mean, var, gamma1, beta1, gamma2, beta2, bias, eps, npix, 1, C, scale, shift, rstd1, rstd2, mm1, mv1, mm2, mv2,
                           decay, stream);
}

extern "C" int emd_bn_train_fold_images_f32(const float* mean, const float* var, const float* gamma1, const float* beta1,
                                            const float* gamma2, const float* beta2, const float* bias, float eps, long npix,
                                            int B, int C, float* scale, float* shift, float* rstd1, float* rstd2, float* mm1,
                                            float* mv1, float* mm2, float* mv2, double decay, emd_stream_t stream) {
    return train_fold_impl(mean, var, gamma1, beta1, gamma2, beta2, bias, eps, npix, B, C, scale, shift, rstd1, rstd2, mm1, mv1, mm2, mv2,
                           decay, stream);
}

static int bwd_prep_impl(const float* s1, const float* t, const float* gamma1, const float* gamma2,
                         const float* rstd1, const float* rstd2, float eps, long npix, int B, int C, float* K,
                         float* m1, float* m2, float* dgamma1, float* dgamma2, float* dbeta2, emd_stream_t stream) {
    EMD_REQUIRE(s1 && t && gamma2 && rstd1 && K && m1 && m2 && dgamma2 && dbeta2, EMD_E_INVALID, "emd_bn_bwd_prep_f32: null pointer");
    EMD_REQUIRE(!gamma1 || (rstd2 && dgamma1), EMD_E_INVALID, "emd_bn_bwd_prep_f32: the double batch norm needs rstd2 and dgamma1");
    EMD_REQUIRE(npix >= 1 && C >= 1 && B >= 1 && (long)B * C <= 0x7fffffffL, EMD_E_INVALID, "emd_bn_bwd_prep_f32: bad shape");
    const int n = B * C;
    hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), s1, t,
                       gamma1, gamma2, rstd1, rstd2, eps, 1.0f / (float)npix, n, K, m1, m2, dgamma1, dgamma2, dbeta2, C);
    return emd::check_launch("bn_bwd_prep_kernel");
}

extern "C" int emd_bn_bwd_prep_f32(const float* s1, const float* t, const float* gamma1, const float* gamma2,
                                   const float* rstd1, const float* rstd2, float eps, long npix, int C, float* K,
                                   float* m1, float* m2, float* dgamma1, float* dgamma2, float* dbeta2,
                                   emd_stream_t stream) {
    return bwd_prep_impl(s1, t, gamma1, gamma2, rstd1, rstd2, eps, npix, 1, C, K, m1, m2, dgamma1, dgamma2, dbeta2, stream);
}

extern "C" int emd_bn_bwd_prep_images_f32(const float* s1, const float* t, const float* gamma1, const float* gamma2,
                                          const float* rstd1, const float* rstd2, float eps, long npix, int B, int C, float* K,
                                          float* m1, float* m2, float* dgamma1, float* dgamma2, float* dbeta2,
                                          emd_stream_t stream) {
    return bwd_prep_impl(s1, t, gamma1, gamma2, rstd1, rstd2, eps, npix, B, C, K, m1, m2, dgamma1, dgamma2, dbeta2, stream);
}


// ---- the small-map one-launch forms (round 4; kernels above).  Per-image statistics only (a tower of one image, or B such towers as
// one batched pass: B = 1 is the plain batch norm of one image).  npix <= 4096 (EMD_E_UNSUPPORTED above: one workgroup per 64 channels
// and image would leave the chip idle -- use the slab forms), C, pitches multiples of 4, 16-byte aligned tensors.
extern "C" int emd_bn_train_small_supported(long npix, int C) { return npix >= 1 && npix <= 4096 && C >= 4 && C % 4 == 0; }

extern "C" int emd_bn_train_fwd_small_f32(const float* r, int ldr, int B, long npix, int C, const float* gamma1, const float* beta1,
                                          const float* gamma2, const float* beta2, const float* bias, float eps, float* scale,
                                          float* shift, float* rstd1, float* rstd2, float* mean, float* mm1, float* mv1, float* mm2,
                                          float* mv2, double decay, const float* res, int ldres, float* out, int ldo, int act,
                                          emd_stream_t stream) {
    EMD_REQUIRE(r && gamma2 && beta2 && scale && shift && rstd1 && mean && out, EMD_E_INVALID, "emd_bn_train_fwd_small_f32: null pointer");
    EMD_REQUIRE((gamma1 == nullptr) == (beta1 == nullptr) && (!gamma1 || rstd2), EMD_E_INVALID, "emd_bn_train_fwd_small_f32: BN1 needs gamma1, beta1, rstd2");
    EMD_REQUIRE(B >= 1 && B <= 65535 && act >= 0 && act <= 4, EMD_E_INVALID, "emd_bn_train_fwd_small_f32: bad argument");
    EMD_REQUIRE(emd_bn_train_small_supported(npix, C), EMD_E_UNSUPPORTED, "emd_bn_train_fwd_small_f32: needs npix <= 4096 and C % 4 == 0");
    EMD_REQUIRE(ldr % 4 == 0 && ldo % 4 == 0 && ldr >= C && ldo >= C && emd::aligned16(r) && emd::aligned16(out) &&
                    (!res || (ldres % 4 == 0 && ldres >= C && emd::aligned16(res))), EMD_E_ALIGN, "emd_bn_train_fwd_small_f32: alignment");
    EMD_REQUIRE(!mm2 || (mv2 && (!gamma1 || (mm1 && mv1))), EMD_E_INVALID, "emd_bn_train_fwd_small_f32: moving statistics come in pairs");
    hipLaunchKernelGGL(bn_fwd_small_kernel, dim3((C + CB - 1) / CB, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), r, ldr, npix, C,
                       gamma1, beta1, gamma2, beta2, bias, eps, (float)(1.0 - decay), mm1, mv1, mm2, mv2, scale, shift, rstd1, rstd2, mean,
                       res, ldres, out, ldo, act);
    return emd::check_launch("bn_fwd_small_kernel");
}

extern "C" int emd_bn_train_bwd_small_f32(const float* dy, int ldd, const float* x, int ldx, int B, long npix, int C, const float* mean,
                                          const float* rstd1, const float* rstd2, const float* mscale, const float* mshift, int mask,
                                          const float* gamma1, const float* gamma2, float eps, float* dgamma1, float* dgamma2,
                                          float* dbeta2, float* dx, int ldo, emd_stream_t stream) {
    EMD_REQUIRE(dy && x && mean && rstd1 && gamma2 && dgamma2 && dbeta2 && dx, EMD_E_INVALID, "emd_bn_train_bwd_small_f32: null pointer");
    EMD_REQUIRE(!gamma1 || (rstd2 && dgamma1), EMD_E_INVALID, "emd_bn_train_bwd_small_f32: BN1 needs rstd2 and dgamma1");
    EMD_REQUIRE(B >= 1 && B <= 65535 && mask >= 0 && mask <= 3 && (!mask || (mscale && mshift)), EMD_E_INVALID, "emd_bn_train_bwd_small_f32: bad argument");
    EMD_REQUIRE(emd_bn_train_small_supported(npix, C), EMD_E_UNSUPPORTED, "emd_bn_train_bwd_small_f32: needs npix <= 4096 and C % 4 == 0");
    EMD_REQUIRE(ldd % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldd >= C && ldx >= C && ldo >= C && emd::aligned16(dy) && emd::aligned16(x) &&
                    emd::aligned16(dx), EMD_E_ALIGN, "emd_bn_train_bwd_small_f32: alignment");
    hipLaunchKernelGGL(bn_bwd_small_kernel, dim3((C + CB - 1) / CB, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), dy, ldd, x, ldx,
                       npix, C, mean, rstd1, rstd2, mscale, mshift, mask, gamma1, gamma2, eps, dgamma1, dgamma2, dbeta2, dx, ldo);
    return emd::check_launch("bn_bwd_small_kernel");
}
