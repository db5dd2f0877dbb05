// Training step (graph D', misc_py/denoiser-multi-gpu.py:752-782): the data gradient of a stride-1 depthwise 3x3 fused with the batch-norm
// backward of the layer BEFORE it -- round 4, for the separable convs whose output feeds exactly one separable conv (ops.PreAct): the
// gradient dy of that output exists only as the depthwise data gradient of the consumer, so it is never written:
//   pass 1 (EPI = 1)  dy = dw3x3(dd, flipped taps) on the fly;  g = dy * mask(r*ms + mh);  per-workgroup double sums of g and
//                     g * (r - mean) * rstd  -> partials in chan_reduce_partial_v4's layout (bn_train.hip), finished by chan_reduce_final
//   [bn_bwd_prep_kernel: K, m1, m2 and the parameter gradients, as for any layer]
//   pass 2 (EPI = 2)  dy again;  dr = K * (g - m1 - (r - mean) * m2)   written over r.
// Against  depthwise data gradient (read dd, write dy) + reduction (read dy, r) + apply (read dy, r, write dr): 5 passes instead of 7 over
// the tensor, two launches instead of three.  dy has the bits of emd_dw3x3_f32 (same window sums in the same order as dw3x3_s1_roll,
// dw_misc.hip); the reduction is cut into other slabs than chan_reduce_partial_v4's, so s1 / s2 agree with the unfused route to double
// rounding, not bit for bit.
// replaces: the gradient of slim.separable_convolution2d's depthwise stage w.r.t. its input followed by the gradient of
//           _batch_norm_fn(is_training) + relu6 of the previous block (machine_learning/denoiser.py:110-136 under tf.gradients).
#include "emd_common.hpp"
#include "bn_chain_dev.hpp"

namespace {

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float grad_mask(float dy, float z, int mask) {   // bn_train.hip
    if (mask == 1) return (z > 0.f && z < 6.f) ? dy : 0.f;
    if (mask == 2) return (z > 0.f && z <= 1.f) ? dy : 0.f;
    if (mask == 3) return z > 0.f ? dy : 0.2f * dy;
    return dy;
}

struct DwBnArgs {
    const float* dd;      // [B,H,W,C] gradient w.r.t. the consumer's depthwise output, pitch ldd
    const float* w;       // [9][C] the consumer's depthwise taps, FLIPPED (tap t = original tap 8 - t)
    const float* r;       // [B,H,W,C] this layer's conv output (the batch norm's input), pitch ldr
    float* dr;            // EPI 2: [B,H,W,C], pitch ldo (may be r)
    const float *mean, *rstd, *ms, *mh;   // EPI 1: statistics and the mask's affine; [C], or [B][C] when vld = C
    const float *K, *m1, *m2;             // EPI 2 (with mean, ms, mh)
    double* part;         // EPI 1: [B][nslab][2][C]
    float* dwg;           // EPI 1, WG: [9][C] += the CONSUMER's depthwise weight gradient, sum_p x[p + tap] * dd[p] with x = act(r*ms + mh)
    int ldd, ldr, ldo, H, W, C4, nstrip, mask;
    long vld;             // floats between two images' per-channel vectors (0: shared)
};

// The window arithmetic is dw3x3_s1_roll's (dw_misc.hip): a thread owns (image, column ox, channel quad) and rolls down TH output rows.
// WG (EPI 1 only): the pass streams exactly the two operands of the consumer's depthwise WEIGHT gradient -- dd with its halo, and r, from
// which x = relu6(r*ms + mh) is the mask's own argument clamped -- so that gradient is accumulated here too (the raw 3 x 3 window of dd
// around the pixel is kept beside the window sums): dW[3ky + kx] += x[q] * dd[q - (ky-1, kx-1)], 16 columns summed through LDS, one
// float atomic per (tap, channel) and workgroup as in dw_wgrad_roll_kernel (bwd_misc.hip).  emd_dw3x3_wgrad_pre_f32's launch is gone.
// EPI = 0 (always with WG): no batch norm at all -- the depthwise stage's two gradients in ONE pass for a consumer whose input x WAS written
// (a block's first separable conv): dx = dw3x3(dd, flipped taps) stored to a.dr, dW += sum x[q] * dd[q - tap] with x = a.r as it is
// (emd_dw3x3_f32 on the reversed taps + emd_dw3x3_wgrad_f32, which read dd twice).
template <int TH, int EPI, bool WG = false>
__global__ __launch_bounds__(256) void dw_bn_bwd_kernel(const DwBnArgs a) {
    static_assert(!WG || EPI <= 1, "the weight gradient rides in the reduction pass (or in the plain data-gradient pass)");
    static_assert(EPI != 0 || WG, "EPI 0 is the two depthwise gradients in one pass");
    const int ncb = (a.C4 + 15) >> 4, npb = (a.W + 15) >> 4;
    int bidx = blockIdx.x;
    const int cblk = bidx % ncb;
    bidx /= ncb;
    const int pblk = bidx % npb;
    bidx /= npb;
    const int strip = bidx % a.nstrip;
    const long b = bidx / a.nstrip;
    const int H = a.H, W = a.W, C = a.C4 * 4;
    const int c4o = cblk * 16 + (threadIdx.x & 15);
    const int ox = pblk * 16 + (threadIdx.x >> 4);
    const bool live = c4o < a.C4 && ox < W;
    if (EPI == 2 && !live) return;
    const int c4 = c4o < a.C4 ? c4o : a.C4 - 1, oxc = ox < W ? ox : W - 1;   // (EPI 1: idle threads stay for the reduction, on clamped addresses)

    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(a.w + k * C + c4 * 4);
    const long vo = b * a.vld + c4 * 4;
    float4 mu = f4zero(), ms = f4zero(), mh = f4zero();
    if (EPI != 0) mu = *reinterpret_cast<const float4*>(a.mean + vo);
    if (EPI != 0 && a.mask) {
        ms = *reinterpret_cast<const float4*>(a.ms + vo);
        mh = *reinterpret_cast<const float4*>(a.mh + vo);
    }
    float4 e0 = f4zero(), e1 = f4zero(), e2 = f4zero();   // EPI 1: rstd; EPI 2: K, m1, m2
    if (EPI == 0) {
    } else if (EPI == 1) {
        e0 = *reinterpret_cast<const float4*>(a.rstd + vo);
    } else {
        e0 = *reinterpret_cast<const float4*>(a.K + vo);
        e1 = *reinterpret_cast<const float4*>(a.m1 + vo);
        e2 = *reinterpret_cast<const float4*>(a.m2 + vo);
    }

    const float* xb = a.dd + (b * H) * (long)W * a.ldd + c4 * 4;
    const float* rb = a.r + (b * H) * (long)W * a.ldr + c4 * 4;
    const int oy0 = strip * TH;
    const bool hasl = oxc > 0, hasr = oxc + 1 < W;
    constexpr int PF = 3, NR = TH + 2;
    const long xl = hasl ? -(long)a.ldd : 0, xr = hasr ? (long)a.ldd : 0;
    auto row_ptr = [&](int tt) {
        int iy = oy0 - 1 + tt;
        iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        return xb + ((long)iy * W + oxc) * a.ldd;
    };
    float4 rc[PF], rl[PF], rr[PF];
#pragma unroll
    for (int t0 = 0; t0 < PF && t0 < NR; ++t0) {
        const float* row = row_ptr(t0);
        rc[t0] = *reinterpret_cast<const float4*>(row);
        rl[t0] = *reinterpret_cast<const float4*>(row + xl);
        rr[t0] = *reinterpret_cast<const float4*>(row + xr);
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    float4 s0 = f4zero(), s1 = f4zero();
    float4 wa[WG ? 9 : 1];                                  // the consumer's weight gradient, this thread's share
    float4 pl0 = f4zero(), pc0 = f4zero(), pr0 = f4zero();  // raw dd of window row 0 (input row tt - 2) ...
    float4 pl1 = f4zero(), pc1 = f4zero(), pr1 = f4zero();  // ... and row 1 (tt - 1)
#pragma unroll
    for (int k = 0; k < (WG ? 9 : 1); ++k) wa[k] = f4zero();
#pragma unroll
    for (int tt = 0; tt < NR; ++tt) {
        const int iy = oy0 - 1 + tt;
        const bool ok = iy >= 0 && iy < H;
        float4 c = rc[tt % PF], l = rl[tt % PF], r = rr[tt % PF];
        c = ok ? c : f4zero();
        l = ok && hasl ? l : f4zero();
        r = ok && hasr ? r : f4zero();
        if (tt + PF < NR) {
            const float* row = row_ptr(tt + PF);
            rc[tt % PF] = *reinterpret_cast<const float4*>(row);
            rl[tt % PF] = *reinterpret_cast<const float4*>(row + xl);
            rr[tt % PF] = *reinterpret_cast<const float4*>(row + xr);
        }
        const float4 h0 = fma4(wk[0], l, fma4(wk[1], c, fma4(wk[2], r, f4zero())));
        const float4 h1 = fma4(wk[3], l, fma4(wk[4], c, fma4(wk[5], r, f4zero())));
        const float4 h2 = fma4(wk[6], l, fma4(wk[7], c, fma4(wk[8], r, f4zero())));
        if (tt >= 2) {
            const int oy = oy0 + tt - 2;
            if (oy < H && live) {
                const float4 dyv = add4(s0, h2);
                const long pix = (b * H + oy) * (long)W + ox;
                const float4 rv = *reinterpret_cast<const float4*>(rb + ((long)oy * W + ox) * a.ldr);
                const float dyk[4] = {dyv.x, dyv.y, dyv.z, dyv.w}, rk[4] = {rv.x, rv.y, rv.z, rv.w};
                const float msk[4] = {ms.x, ms.y, ms.z, ms.w}, mhk[4] = {mh.x, mh.y, mh.z, mh.w}, muk[4] = {mu.x, mu.y, mu.z, mu.w};
                const float e0k[4] = {e0.x, e0.y, e0.z, e0.w}, e1k[4] = {e1.x, e1.y, e1.z, e1.w}, e2k[4] = {e2.x, e2.y, e2.z, e2.w};
                float o[4];
                if (EPI == 0) *reinterpret_cast<float4*>(a.dr + pix * a.ldo + c4 * 4) = dyv;
#pragma unroll
                for (int k = 0; k < (EPI == 0 ? 0 : 4); ++k) {
                    const float g = grad_mask(dyk[k], fmaf(rk[k], msk[k], mhk[k]), a.mask);
                    if (EPI == 1) {     // chan_reduce_partial_v4's terms
                        s[k] += (double)g;
                        q[k] += (double)g * (double)((rk[k] - muk[k]) * e0k[k]);
                    } else {            // bn_bwd_apply_kernel's statement
                        o[k] = e0k[k] * (g - e1k[k] - (rk[k] - muk[k]) * e2k[k]);
                    }
                }
                if (EPI == 2) *reinterpret_cast<float4*>(a.dr + pix * a.ldo + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
                if constexpr (WG) {   // x = relu6(r*ms + mh): affine_relu6_kernel's bits (what the forward's loads rebuilt); window rows 0, 1, 2 = tt - 2, tt - 1, tt
                    const float4 x = EPI == 0 ? rv : make_float4(fminf(fmaxf(fmaf(rk[0], msk[0], mhk[0]), 0.f), 6.f), fminf(fmaxf(fmaf(rk[1], msk[1], mhk[1]), 0.f), 6.f),
                                                 fminf(fmaxf(fmaf(rk[2], msk[2], mhk[2]), 0.f), 6.f), fminf(fmaxf(fmaf(rk[3], msk[3], mhk[3]), 0.f), 6.f));
                    wa[0] = fma4(x, r, wa[0]);   wa[1] = fma4(x, c, wa[1]);   wa[2] = fma4(x, l, wa[2]);      // ky = 0: window row 2
                    wa[3] = fma4(x, pr1, wa[3]); wa[4] = fma4(x, pc1, wa[4]); wa[5] = fma4(x, pl1, wa[5]);    // ky = 1: row 1
                    wa[6] = fma4(x, pr0, wa[6]); wa[7] = fma4(x, pc0, wa[7]); wa[8] = fma4(x, pl0, wa[8]);    // ky = 2: row 0
                }
            }
        }
        s0 = add4(s1, h1);
        s1 = h0;
        if constexpr (WG) {
            pl0 = pl1; pc0 = pc1; pr0 = pr1;
            pl1 = l; pc1 = c; pr1 = r;
        }
    }
    if constexpr (WG) {   // 16 columns -> one sum per (tap, channel) through LDS, then one atomic each (dw_wgrad_roll_kernel's tail)
        __shared__ float red[16][9][64 + 1];
        const int cl = (threadIdx.x & 15) * 4, col = threadIdx.x >> 4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            red[col][t][cl + 0] = wa[t].x; red[col][t][cl + 1] = wa[t].y;
            red[col][t][cl + 2] = wa[t].z; red[col][t][cl + 3] = wa[t].w;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * 64; i += 256) {
            const int t = i / 64, lc = i % 64;
            const int cc = cblk * 64 + lc;
            if (cc >= C) continue;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k][t][lc];
            atomicAdd(a.dwg + (long)t * C + cc, sum);
        }
    }
    if (EPI == 1) {   // 16 columns -> one sum per channel and workgroup = one slab of chan_reduce_final's input
        __shared__ double sm[2][16][64 + 1];
        const int cl = (threadIdx.x & 15) * 4, col = threadIdx.x >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sm[0][col][cl + k] = s[k];
            sm[1][col][cl + k] = q[k];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, lc = threadIdx.x & 63;
            const int cc = cblk * 64 + lc;
            if (cc < C) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) t += sm[which][k][lc];
                const long nslab = (long)a.nstrip * npb, slab = (long)strip * npb + pblk;
                a.part[((b * nslab + slab) * 2 + which) * C + cc] = t;
            }
        }
    }
}

// Any stride / dilation (the stride-2 consumers cnn*_strided; same passes, gather form): a thread walks every 16th pixel q of its slab of
// one image for one channel quad and gathers dy[q] = sum over the output pixels whose window holds q, in dw_bwd_data_kernel's order
// (bwd_misc.hip: its bits); WG adds x[q] * dd[that output pixel] into the tap's weight-gradient accumulator in the same loop.
// Grid (ceil(C / 64), slabs per image, B); a.H, a.W: the grid of r (the consumer's INPUT); dd is [B, Ho, Wo, C].
template <int EPI, bool WG>
__global__ __launch_bounds__(256) void dw_bn_bwd_gen_kernel(const DwBnArgs a, int Ho, int Wo, int st, int rate, int pt, int pl, long pps) {
    const int H = a.H, W = a.W, C = a.C4 * 4;
    const int cl = (threadIdx.x & 15) * 4, plane = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + cl;
    const bool live = c0 < C;
    const int c = live ? c0 : C - 4;
    const long b = blockIdx.z, npix = (long)H * W;
    const long p0 = (long)blockIdx.y * pps, p1 = p0 + pps < npix ? p0 + pps : npix;
    const long vo = b * a.vld + c;
    const float4 mu = *reinterpret_cast<const float4*>(a.mean + vo);
    float4 ms = f4zero(), mh = f4zero();
    if (a.mask) {
        ms = *reinterpret_cast<const float4*>(a.ms + vo);
        mh = *reinterpret_cast<const float4*>(a.mh + vo);
    }
    float4 e0 = f4zero(), e1 = f4zero(), e2 = f4zero();
    if (EPI == 1) {
        e0 = *reinterpret_cast<const float4*>(a.rstd + vo);
    } else {
        e0 = *reinterpret_cast<const float4*>(a.K + vo);
        e1 = *reinterpret_cast<const float4*>(a.m1 + vo);
        e2 = *reinterpret_cast<const float4*>(a.m2 + vo);
    }
    const float* ddb = a.dd + (b * Ho) * (long)Wo * a.ldd + c;
    const float* rb = a.r + b * npix * a.ldr + c;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    float4 wa[WG ? 9 : 1];
#pragma unroll
    for (int k = 0; k < (WG ? 9 : 1); ++k) wa[k] = f4zero();
    if (live) {
        for (long p = p0 + plane; p < p1; p += 16) {
            const int ix = (int)(p % W), iy = (int)(p / W);
            const float4 rv = *reinterpret_cast<const float4*>(rb + p * a.ldr);
            const float rk[4] = {rv.x, rv.y, rv.z, rv.w};
            const float msk[4] = {ms.x, ms.y, ms.z, ms.w}, mhk[4] = {mh.x, mh.y, mh.z, mh.w}, muk[4] = {mu.x, mu.y, mu.z, mu.w};
            float4 x = f4zero();
            if constexpr (WG)
                x = make_float4(fminf(fmaxf(fmaf(rk[0], msk[0], mhk[0]), 0.f), 6.f), fminf(fmaxf(fmaf(rk[1], msk[1], mhk[1]), 0.f), 6.f),
                                fminf(fmaxf(fmaf(rk[2], msk[2], mhk[2]), 0.f), 6.f), fminf(fmaxf(fmaf(rk[3], msk[3], mhk[3]), 0.f), 6.f));
            float4 acc = f4zero();
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ny = iy + pt - ky * rate;
                if (ny < 0 || ny % st != 0 || ny / st >= Ho) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int nx = ix + pl - kx * rate;
                    if (nx < 0 || nx % st != 0 || nx / st >= Wo) continue;
                    const float4 v = *reinterpret_cast<const float4*>(ddb + ((long)(ny / st) * Wo + nx / st) * a.ldd);
                    acc = fma4(v, *reinterpret_cast<const float4*>(a.w + (8 - (ky * 3 + kx)) * C + c), acc);   // a.w holds the taps reversed
                    if constexpr (WG) wa[ky * 3 + kx] = fma4(x, v, wa[ky * 3 + kx]);
                }
            }
            const float dyk[4] = {acc.x, acc.y, acc.z, acc.w};
            const float e0k[4] = {e0.x, e0.y, e0.z, e0.w}, e1k[4] = {e1.x, e1.y, e1.z, e1.w}, e2k[4] = {e2.x, e2.y, e2.z, e2.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = grad_mask(dyk[k], fmaf(rk[k], msk[k], mhk[k]), a.mask);
                if (EPI == 1) {
                    s[k] += (double)g;
                    q[k] += (double)g * (double)((rk[k] - muk[k]) * e0k[k]);
                } else {
                    o[k] = e0k[k] * (g - e1k[k] - (rk[k] - muk[k]) * e2k[k]);
                }
            }
            if (EPI == 2) *reinterpret_cast<float4*>(a.dr + (b * npix + p) * a.ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if constexpr (WG) {
        __shared__ float red[16][9][64 + 1];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            red[plane][t][cl + 0] = wa[t].x; red[plane][t][cl + 1] = wa[t].y;
            red[plane][t][cl + 2] = wa[t].z; red[plane][t][cl + 3] = wa[t].w;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * 64; i += 256) {
            const int t = i / 64, lc = i % 64;
            const int cc = blockIdx.x * 64 + lc;
            if (cc >= C) continue;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k][t][lc];
            atomicAdd(a.dwg + (long)t * C + cc, sum);
        }
    }
    if (EPI == 1) {
        __shared__ double sm[2][16][64 + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sm[0][plane][cl + k] = s[k];
            sm[1][plane][cl + k] = q[k];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, lc = threadIdx.x & 63;
            const int cc = blockIdx.x * 64 + lc;
            if (cc < C) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) t += sm[which][k][lc];
                a.part[((b * gridDim.y + blockIdx.y) * 2 + which) * C + cc] = t;
            }
        }
    }
}

int strip_height(int H) { return H >= 64 ? 16 : 8; }
constexpr long kGenPps = 512;    // pixels per slab of the gather form
inline int gen_slabs(long npix) { return (int)((npix + kGenPps - 1) / kGenPps); }
inline int same_pad_before(int n, int s, int r) {  // TF SAME, k = 3 (bwd_misc.hip)
    const int o = (n + s - 1) / s;
    int total = (o - 1) * s + 2 * r + 1 - n;
    if (total < 0) total = 0;
    return total / 2;
}

bool args_ok(const float* p, int ld, int C) { return p && C >= 4 && C % 4 == 0 && ld % 4 == 0 && ld >= C && emd::aligned16(p); }

int launch_plain(const DwBnArgs& a0, int B, hipStream_t st) {   // EPI 0: stride 1, rate 1
    DwBnArgs a = a0;
    const int TH = a.H >= 64 ? 16 : 8;
    a.nstrip = (a.H + TH - 1) / TH;
    const long nb = (long)B * a.nstrip * ((a.W + 15) / 16) * ((a.C4 + 15) / 16);
    EMD_REQUIRE(nb >= 1 && nb <= 0x7fffffffL, EMD_E_UNSUPPORTED, "emd_dw3x3_bwd_both_f32: grid too large");
    if (TH == 16) hipLaunchKernelGGL((dw_bn_bwd_kernel<16, 0, true>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_bn_bwd_kernel<8, 0, true>), dim3((unsigned)nb), dim3(256), 0, st, a);
    return emd::check_launch("dw_bn_bwd_kernel<both depthwise gradients>");
}

template <int EPI>
int launch(const DwBnArgs& a0, int B, hipStream_t st, int stride = 1, int rate = 1) {
    DwBnArgs a = a0;
    if (stride != 1 || rate != 1) {   // the gather form
        const int Ho = (a.H + stride - 1) / stride, Wo = (a.W + stride - 1) / stride;
        const int pt = same_pad_before(a.H, stride, rate), pl = same_pad_before(a.W, stride, rate);
        const dim3 grid((a.C4 * 4 + 63) / 64, gen_slabs((long)a.H * a.W), B);
        if constexpr (EPI == 1) {
            if (a.dwg) hipLaunchKernelGGL((dw_bn_bwd_gen_kernel<1, true>), grid, dim3(256), 0, st, a, Ho, Wo, stride, rate, pt, pl, kGenPps);
            else hipLaunchKernelGGL((dw_bn_bwd_gen_kernel<1, false>), grid, dim3(256), 0, st, a, Ho, Wo, stride, rate, pt, pl, kGenPps);
        } else {
            hipLaunchKernelGGL((dw_bn_bwd_gen_kernel<2, false>), grid, dim3(256), 0, st, a, Ho, Wo, stride, rate, pt, pl, kGenPps);
        }
        return emd::check_launch("dw_bn_bwd_gen_kernel");
    }
    const int TH = strip_height(a.H);
    a.nstrip = (a.H + TH - 1) / TH;
    const long nb = (long)B * a.nstrip * ((a.W + 15) / 16) * ((a.C4 + 15) / 16);
    EMD_REQUIRE(nb >= 1 && nb <= 0x7fffffffL, EMD_E_UNSUPPORTED, "emd_dw3x3_bn_bwd: grid too large");
    if constexpr (EPI == 1) {
        if (a.dwg) {
            if (TH == 16) hipLaunchKernelGGL((dw_bn_bwd_kernel<16, 1, true>), dim3((unsigned)nb), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((dw_bn_bwd_kernel<8, 1, true>), dim3((unsigned)nb), dim3(256), 0, st, a);
            return emd::check_launch("dw_bn_bwd_kernel<weight gradient>");
        }
    }
    if (TH == 16) hipLaunchKernelGGL((dw_bn_bwd_kernel<16, EPI>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_bn_bwd_kernel<8, EPI>), dim3((unsigned)nb), dim3(256), 0, st, a);
    return emd::check_launch("dw_bn_bwd_kernel");
}

}  // namespace

extern "C" size_t emd_dw3x3_bn_bwd_workspace_bytes(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const int TH = strip_height(H);
    const long roll = (long)((H + TH - 1) / TH) * ((W + 15) / 16), gen = gen_slabs((long)H * W);   // slabs per image, either form
    return (size_t)B * (roll > gen ? roll : gen) * 2 * C * sizeof(double);
}

// s1[c] = sum g, s2[c] = sum g * (r - mean) * rstd with g = dw3x3(dd, w_flipped) * mask(r * mscale + mshift): emd_dw3x3_f32(stride 1) followed
// by emd_bn_bwd_reduce[_images]_f32, without the tensor between them.  images != 0: statistics vectors, s1, s2 are [B][C].
// dw_consumer != NULL (needs mask = relu6): [9][C] += the consumer's depthwise weight gradient, emd_dw3x3_wgrad_pre_f32(r, mscale, mshift,
// relu6, dd) -- the pass reads exactly its operands.
extern "C" int emd_dw3x3_bn_bwd_reduce_f32(const float* dd, int ldd, const float* w_flipped, const float* r, int ldr, const float* mean,
                                           const float* rstd, const float* mscale, const float* mshift, int mask, int images, int B, int H,
                                           int W, int C, int stride, int rate, float* s1, float* s2, float* dw_consumer, void* workspace,
                                           const emd_bn_bwd_prep_t* prep, emd_stream_t stream) {
    // prep != NULL: emd_bn_bwd_prep[_images]_f32's per-channel step in the reduction's final kernel (one launch less)
    emd::BnPrepArgs pa;
    if (prep) {
        int rcp = emd::bn_prep_args(prep, &pa);
        if (rcp != EMD_OK) return rcp;
    }
    const emd::BnPrepArgs* pp = prep ? &pa : nullptr;
    EMD_REQUIRE((stride == 1 || stride == 2) && rate >= 1 && (rate == 1 || stride == 1), EMD_E_INVALID, "emd_dw3x3_bn_bwd_reduce_f32: stride 1 or 2; rate > 1 needs stride 1");
    EMD_REQUIRE(!dw_consumer || mask == 1, EMD_E_INVALID, "emd_dw3x3_bn_bwd_reduce_f32: the consumer's weight gradient needs the relu6 mask (x = relu6(r*mscale + mshift))");
    EMD_REQUIRE(w_flipped && mean && rstd && s1 && s2 && workspace, EMD_E_INVALID, "emd_dw3x3_bn_bwd_reduce_f32: null pointer");
    EMD_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && mask >= 0 && mask <= 3 && (!mask || (mscale && mshift)), EMD_E_INVALID,
                "emd_dw3x3_bn_bwd_reduce_f32: bad argument");
    EMD_REQUIRE(args_ok(dd, ldd, C) && args_ok(r, ldr, C) && emd::aligned16(w_flipped) && emd::aligned16(mean) && emd::aligned16(rstd) &&
                    (!mask || (emd::aligned16(mscale) && emd::aligned16(mshift))) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                EMD_E_ALIGN, "emd_dw3x3_bn_bwd_reduce_f32: C, pitches multiples of 4; 16-byte aligned tensors and vectors");
    DwBnArgs a{};
    a.dd = dd; a.w = w_flipped; a.r = r; a.mean = mean; a.rstd = rstd; a.ms = mscale; a.mh = mshift; a.part = static_cast<double*>(workspace);
    a.dwg = dw_consumer;
    a.ldd = ldd; a.ldr = ldr; a.H = H; a.W = W; a.C4 = C / 4; a.mask = mask; a.vld = images ? C : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = launch<1>(a, B, st, stride, rate);
    if (rc != EMD_OK) return rc;
    const int TH = strip_height(H);
    const int nslab = (stride != 1 || rate != 1) ? gen_slabs((long)H * W) : ((H + TH - 1) / TH) * ((W + 15) / 16);
    if (images) return emd::launch_chan_reduce_final(a.part, nslab, C, B, s1, s2, st, pp, (long)H * W);
    // batch statistics: the B images' slabs are one list
    return emd::launch_chan_reduce_final(a.part, nslab * B, C, 1, s1, s2, st, pp, (long)B * H * W);
}

// dr = K * (g - m1 - (r - mean) * m2), g as above: emd_dw3x3_f32(stride 1) followed by emd_bn_bwd_apply[_images]_f32; dr may be r.
extern "C" int emd_dw3x3_bn_bwd_apply_f32(const float* dd, int ldd, const float* w_flipped, const float* r, int ldr, const float* K,
                                          const float* m1, const float* mean, const float* m2, const float* mscale, const float* mshift,
                                          int mask, int images, float* dr, int ldo, int B, int H, int W, int C, int stride, int rate,
                                          emd_stream_t stream) {
    EMD_REQUIRE((stride == 1 || stride == 2) && rate >= 1 && (rate == 1 || stride == 1), EMD_E_INVALID, "emd_dw3x3_bn_bwd_apply_f32: stride 1 or 2; rate > 1 needs stride 1");
    EMD_REQUIRE(w_flipped && K && m1 && mean && m2 && dr, EMD_E_INVALID, "emd_dw3x3_bn_bwd_apply_f32: null pointer");
    EMD_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && mask >= 0 && mask <= 3 && (!mask || (mscale && mshift)), EMD_E_INVALID,
                "emd_dw3x3_bn_bwd_apply_f32: bad argument");
    EMD_REQUIRE(args_ok(dd, ldd, C) && args_ok(r, ldr, C) && args_ok(dr, ldo, C) && emd::aligned16(w_flipped) && emd::aligned16(K) &&
                    emd::aligned16(m1) && emd::aligned16(mean) && emd::aligned16(m2) && (!mask || (emd::aligned16(mscale) && emd::aligned16(mshift))),
                EMD_E_ALIGN, "emd_dw3x3_bn_bwd_apply_f32: C, pitches multiples of 4; 16-byte aligned tensors and vectors");
    DwBnArgs a{};
    a.dd = dd; a.w = w_flipped; a.r = r; a.dr = dr; a.mean = mean; a.ms = mscale; a.mh = mshift; a.K = K; a.m1 = m1; a.m2 = m2;
    a.ldd = ldd; a.ldr = ldr; a.ldo = ldo; a.H = H; a.W = W; a.C4 = C / 4; a.mask = mask; a.vld = images ? C : 0;
    return launch<2>(a, B, static_cast<hipStream_t>(stream), stride, rate);
}

// Both gradients of a stride-1 depthwise 3x3 in one pass: dx = emd_dw3x3_f32(dd, w_flipped) and dw[9][C] += emd_dw3x3_wgrad_f32(x, dd) -- dd is
// read once (with its halo) instead of twice, one launch instead of two.  dx has emd_dw3x3_f32's bits.
extern "C" int emd_dw3x3_bwd_both_f32(const float* dd, int ldd, const float* w_flipped, const float* x, int ldx, float* dx, int ldo, float* dw,
                                      int B, int H, int W, int C, emd_stream_t stream) {
    EMD_REQUIRE(w_flipped && dw, EMD_E_INVALID, "emd_dw3x3_bwd_both_f32: null pointer");
    EMD_REQUIRE(B >= 0 && B <= 65535 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_dw3x3_bwd_both_f32: bad shape");
    EMD_REQUIRE(args_ok(dd, ldd, C) && args_ok(x, ldx, C) && args_ok(dx, ldo, C) && emd::aligned16(w_flipped), EMD_E_ALIGN,
                "emd_dw3x3_bwd_both_f32: C, pitches multiples of 4; 16-byte aligned tensors");
    if (B == 0) return EMD_OK;
    DwBnArgs a{};
    a.dd = dd; a.w = w_flipped; a.r = x; a.dr = dx; a.dwg = dw;
    a.ldd = ldd; a.ldr = ldx; a.ldo = ldo; a.H = H; a.W = W; a.C4 = C / 4;
    return launch_plain(a, B, static_cast<hipStream_t>(stream));
}
