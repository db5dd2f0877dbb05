// Bandwidth-bound NHWC fp32 kernels of graph D (machine_learning/denoiser.py):
//   emd_dw3x3_f32          depthwise half of slim.separable_convolution2d (:113-131), stride 1/2, any rate
//   emd_cin1_f32           layers whose input is the 1-channel image: cnn0 (depthwise 3x3 on C=1 then
//                          1->Cout pointwise, :252) and residual0 (1x1 stride-2 conv 1->128, :263)
//   emd_conv3x3_cout1_f32  the final slim.conv2d(64->1, kernel 3) + bias + BN + relu6 (:387)
//   emd_resize_bilinear_f32  tf.image.resize_images, legacy bilinear (:199, :350)
//   emd_affine_relu6_f32   a lone batch_then_activ (:200)
// Layout rule for all of them: channels are innermost, a lane owns 4 consecutive channels (one 16-B
// access) and consecutive lanes own consecutive channel groups, then consecutive pixels along W, so
// a wavefront's access is one contiguous run of NHWC memory whenever C >= 4*64/pixels-per-wave.
#include <cstdlib>

#include "mfma_common.hpp"
#include "bn_chain_dev.hpp"

namespace {

// Workgroup ids are dealt to the 8 XCDs round-robin (id mod 8).  xcd_run gives XCD k the k-th contiguous eighth of the tile
// list instead, so the tiles that share halo rows / columns (neighbours in the list) run on ONE XCD at about the same time and the
// shared input lines are fetched into that L2 once (tools/dw_traffic.sh: fetched bytes per shape).  the dev knob dw_xcd = 0 turns it off.
__device__ __forceinline__ int xcd_run(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
// Rule: maps up to 128 x 128 (there the shared halo lines are a large part of a tile's bytes: 32 x 32 x 728 runs 40.5 -> 31.5 us); on the
// 256^2 / 512^2 maps eight separate sweeps cost DRAM locality more than the halo hits save (565 -> 581 us at 512^2 x 64, stride 2).
// Dev knob dw_xcd (emd_debug_knob): 0 = never, 2 = always.
inline int dw_xcd(int H, int W) {
    const int v = emd::g_knobs.dw_xcd;
    return v == 2 || (v == 1 && (long)H * W <= 128L * 128);
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
// relu (hi = +inf) or relu6 (hi = 6), in the order affine_relu6_kernel applies them: the PRE forms produce its bits
__device__ __forceinline__ float4 clamp4(float4 a, float hi) {
    return make_float4(fminf(fmaxf(a.x, 0.f), hi), fminf(fmaxf(a.y, 0.f), hi), fminf(fmaxf(a.z, 0.f), hi), fminf(fmaxf(a.w, 0.f), hi));
}
// act code of the Cout=1 kernels: 0 none, 1 relu6, 2 relu6 followed by tf.clip_by_value(0,1) = clamp to [0,1]
__device__ __forceinline__ float act_out(float v, int act) { return act == 0 ? v : fminf(fmaxf(v, 0.f), act == 2 ? 1.f : 6.f); }
// the Cout=1 kernels' output stage: optional "+pre_bias, relu" first (tf.layers.conv2d(activation=relu) before the
// batch norm, misc_py/modified_Xception.py:215-229), then the scalar affine and the act code above
__device__ __forceinline__ float cout1_out(float s, float pre_bias, int pre_relu, float scale, float shift, int act) {
    if (pre_relu) s = fmaxf(s + pre_bias, 0.f);
    return act_out(fmaf(s, scale, shift), act);
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3, stride 1, rate 1: each thread owns (image b, column ox, channel group c4) and rolls
// down a strip of TH output rows with the three live input rows' horizontal partial sums in registers:
// an input row is read once per strip (3 shifted 16-B loads) and turned into its contribution as the
// top / middle / bottom row of a window.  TF SAME: pad 1 on every side.
template <int TH, bool SPLIT = false, bool REFLECT = false, bool PRE = false>
__global__ __launch_bounds__(256) void dw3x3_s1_roll(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ w, float* __restrict__ y,
                                                     int ldy, int H, int W, int C4, long nthreads, int nstrip, int C4t,
                                                     const float* __restrict__ pre_s = nullptr,
                                                     const float* __restrict__ pre_t = nullptr, int xcd = 0, long pre_ld = 0,
                                                     float pre_hi = __builtin_inff()) {
    // PRE: the input is relu(x * pre_s + pre_t) per channel -- the batch-statistics norm + relu of the previous separable
    // block (misc_py/modified_Xception.py:302-323) applied on the fly instead of in a pass of its own.  pre_ld != 0: pre_s / pre_t are
    // [image][pre_ld] (per-image statistics: a batched pass of one-image towers); pre_hi = 6: relu6 (graph D', round 4)
    // C4t = channel quads per pixel that have a thread: C4, or ceil32(C)/4 when the split32 padding is written too.
    // A workgroup = 16 adjacent pixel columns x 16 channel quads (64 channels, 256 contiguous bytes per pixel): the left /
    // right neighbours of a pixel are loaded by the SAME workgroup (L1 hits).  With one thread per (pixel, quad) in
    // channel-major order a 728-channel pixel filled 0.7 of a workgroup, its neighbours sat in adjacent workgroups -- which
    // the dispatcher deals to different XCDs -- and every input line crossed the fabric three times (53 us for the
    // 32 x 32 x 728 maps; a copy of the same bytes takes 30).
    (void)nthreads;
    const int ncb = (C4t + 15) >> 4, npb = (W + 15) >> 4;
    int bidx = xcd ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
    const int cblk = bidx % ncb;
    bidx /= ncb;
    const int pblk = bidx % npb;
    bidx /= npb;
    const int strip = bidx % nstrip;
    const long b = bidx / nstrip;
    const int c4o = cblk * 16 + (threadIdx.x & 15);
    const int ox = pblk * 16 + (threadIdx.x >> 4);
    if (c4o >= C4t || ox >= W) return;
    const bool padq = SPLIT && c4o >= C4;
    const int c4 = padq ? C4 - 1 : c4o;
    const int C = C4 * 4;

    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(w + k * C + c4 * 4);
    float4 ps4 = f4zero(), pt4 = f4zero();
    if (PRE) {
        ps4 = *reinterpret_cast<const float4*>(pre_s + b * pre_ld + c4 * 4);
        pt4 = *reinterpret_cast<const float4*>(pre_t + b * pre_ld + c4 * 4);
    }

    const float* xb = x + (b * H) * (long)W * ldx + c4 * 4;
    const int oy0 = strip * TH;
    const bool hasl = ox > 0, hasr = ox + 1 < W;

    // Every load is unconditional (row and column indices clamped into the image, the value replaced by zero afterwards
    // where TF SAME pads) and the loads of input row t+3 are issued before row t is used: the first version branched
    // around each of a row's three loads and waited for them at the end of every row -- ten dependent memory round
    // trips per thread, 53 us for the 32 x 32 x 728 maps that a plain copy moves in 30.
    constexpr int PF = 3, NR = TH + 2;
    // REFLECT (graph G: tf.pad(REFLECT, 1) + VALID): index -1 -> 1, H -> H-2, in both directions; nothing is zeroed
    const long xl = hasl ? -(long)ldx : (REFLECT ? (long)ldx : 0), xr = hasr ? (long)ldx : (REFLECT ? -(long)ldx : 0);
    auto row_ptr = [&](int tt) {
        int iy = oy0 - 1 + tt;
        if (REFLECT) iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        return xb + ((long)iy * W + ox) * ldx;
    };
    float4 rc[PF], rl[PF], rr[PF];
#pragma unroll
    for (int t0 = 0; t0 < PF && t0 < NR; ++t0) {
        const float* row = row_ptr(t0);
        rc[t0] = *reinterpret_cast<const float4*>(row);
        rl[t0] = *reinterpret_cast<const float4*>(row + xl);
        rr[t0] = *reinterpret_cast<const float4*>(row + xr);
    }
    float4 s0 = f4zero(), s1 = f4zero();  // s0: top+mid of output (t-2); s1: top of output (t-1)
#pragma unroll
    for (int tt = 0; tt < NR; ++tt) {
        const int iy = oy0 - 1 + tt;
        const bool ok = REFLECT || (iy >= 0 && iy < H);
        float4 c = rc[tt % PF], l = rl[tt % PF], r = rr[tt % PF];
        if (PRE) {
            c = clamp4(fma4(c, ps4, pt4), pre_hi);
            l = clamp4(fma4(l, ps4, pt4), pre_hi);
            r = clamp4(fma4(r, ps4, pt4), pre_hi);
        }
        c = ok ? c : f4zero();
        l = ok && (REFLECT || hasl) ? l : f4zero();
        r = ok && (REFLECT || hasr) ? r : f4zero();
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
            if (oy < H) emd::dw_store<SPLIT>(y, (b * H + oy) * (long)W + ox, ldy, c4o, padq ? f4zero() : add4(s0, h2));
        }
        s0 = add4(s1, h1);
        s1 = h0;
    }
}

// Depthwise 3x3, any stride / rate: one output pixel x 4 channels per thread (9 loads).
template <bool SPLIT = false, bool PRE = false>
__global__ __launch_bounds__(256) void dw3x3_generic(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ w, float* __restrict__ y,
                                                     int ldy, int H, int W, int C4, int Ho, int Wo,
                                                     int stride, int rate, int pt, int pl, long nthreads, int C4t,
                                                     const float* __restrict__ pre_s = nullptr,
                                                     const float* __restrict__ pre_t = nullptr, int xcd = 0, long pre_ld = 0,
                                                     float pre_hi = __builtin_inff()) {
    // a workgroup = 4 x 4 output pixels x 16 channel quads: the overlapping windows of neighbouring outputs are served by
    // the workgroup's L1 instead of by neighbouring workgroups on other XCDs (see dw3x3_s1_roll)
    (void)nthreads;
    const int ncb = (C4t + 15) >> 4, npx = (Wo + 3) >> 2, npy = (Ho + 3) >> 2;
    int bidx = xcd ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
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
        const int iy = oy * stride - pt + i * rate;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ix = ox * stride - pl + j * rate;
            if (ix < 0 || ix >= W) continue;
            float4 v = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx);
            if (PRE) v = clamp4(fma4(v, *reinterpret_cast<const float4*>(pre_s + b * pre_ld + c4 * 4), *reinterpret_cast<const float4*>(pre_t + b * pre_ld + c4 * 4)), pre_hi);
            const float4 wk = *reinterpret_cast<const float4*>(w + (i * 3 + j) * C + c4 * 4);
            acc = fma4(wk, v, acc);
        }
    }
    emd::dw_store<SPLIT>(y, (b * Ho + oy) * (long)Wo + ox, ldy, c4o, padq ? f4zero() : acc);
}

// ------------------------------------------------------------------------------------------------
// Cin == 1 layers.  out[pix][n] = relu6( d[pix] * a[n] + t[n] ),  d = (3x3 depthwise of the image) or the
// (strided) sample itself.  A wave first evaluates d for 64 consecutive output pixels (lane = pixel),
// then writes them out N4 = Cout/4 lanes per pixel so that each store instruction covers one contiguous
// 1-KiB run of the NHWC output; the pixel's d travels to its writer lanes by wave shuffle.
__global__ __launch_bounds__(256) void cin1_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                   const float* __restrict__ a, const float* __restrict__ tsh,
                                                   float* __restrict__ y, int ldy, int H, int W, int Ho, int Wo,
                                                   int stride, int N4, int use_dw, long npix, int act) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long p0 = wave * 64;
    if (p0 >= npix) return;
    const long pix = p0 + lane;
    float d = 0.f;
    if (pix < npix) {
        int ox, oy;
        const long t = emd::divmod(pix, Wo, ox);
        const long b = emd::divmod(t, Ho, oy);
        const float* img = x + b * (long)H * W;
        if (use_dw) {  // 3x3, stride 1, SAME (pad 1)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int iy = oy - 1 + i;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ix = ox - 1 + j;
                    if (ix < 0 || ix >= W) continue;
                    d = fmaf(w9[i * 3 + j], img[(long)iy * W + ix], d);
                }
            }
        } else {
            d = img[(long)(oy * stride) * W + ox * stride];
        }
    }
    const int ppi = 64 / N4;  // pixels written per store instruction
    const int sub = lane / N4, n4 = lane % N4;
    const float4 av = *reinterpret_cast<const float4*>(a + n4 * 4);
    const float4 tv = *reinterpret_cast<const float4*>(tsh + n4 * 4);
    for (int q = 0; q < 64; q += ppi) {
        const float dv = __shfl(d, q + sub);
        const long op = p0 + q + sub;
        if (op < npix) {
            float4 o = make_float4(fmaf(dv, av.x, tv.x), fmaf(dv, av.y, tv.y), fmaf(dv, av.z, tv.z), fmaf(dv, av.w, tv.w));
            if (act) o = make_float4(relu6f(o.x), relu6f(o.y), relu6f(o.z), relu6f(o.w));
            *reinterpret_cast<float4*>(y + op * ldy + n4 * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 conv of a ONE-channel image (graph X's entry conv, misc_py/modified_Xception.py:356-364: tf.layers.conv2d(1 -> 32, k 3,
// stride 2) + bias -> BN -> relu): 9 inputs per output pixel -- nothing for the matrix cores (the 9-tap GEMM ran it with the channel
// padded to 64 per tap: 0.57 ms for 0.4 GB of traffic).  A thread = one output pixel x 8 output channels (fp32 FMAs); Cout / 8 lanes
// share a pixel, so a wave's store instruction covers whole pixels' channel runs.  TF SAME.  SPLIT: the output as a split32 tensor
// (a thread's 8 channels = 16 bytes of hi words + 16 bytes of lo words of the pixel's line).
template <bool SPLIT>
__global__ __launch_bounds__(256) void conv3x3_cin1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, int ldy, int H, int W, int Ho, int Wo, int stride,
                                                           int pt, int pl, int N8, int Cout, long npix, int act) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(idx % N8);
    const long pix = idx / N8;
    if (pix >= npix) return;
    int ox, oy;
    const long t = emd::divmod(pix, Wo, ox);
    const long b = emd::divmod(t, Ho, oy);
    const float* img = x + b * (long)H * W;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    const bool live = c8 * 8 < Cout;       // (SPLIT: the lanes of the padding channels up to a multiple of 32 store zeros)
    if (live) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = oy * stride - pt + i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ix = ox * stride - pl + j;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[(long)iy * W + ix] : 0.f;
                const float4 w0 = *reinterpret_cast<const float4*>(w + (i * 3 + j) * Cout + c8 * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(w + (i * 3 + j) * Cout + c8 * 8 + 4);
                acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]); acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]); acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
            }
        }
        const float lo = (act == 1 || act == 2) ? 0.f : -__builtin_inff(), hi = act == 1 ? 6.f : __builtin_inff();
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fminf(fmaxf(fmaf(acc[k], scale[c8 * 8 + k], shift[c8 * 8 + k]), lo), hi);
    }
    if (!SPLIT) {
        if (!live) return;
        *reinterpret_cast<float4*>(y + pix * ldy + c8 * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(y + pix * ldy + c8 * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) emd::split2(acc[2 * k], acc[2 * k + 1], h[k], l[k]);
        unsigned char* g = reinterpret_cast<unsigned char*>(y) + pix * (long)ldy * 4 + (c8 >> 2) * 128 + (c8 & 3) * 16;
        *reinterpret_cast<emd::u32x4*>(g) = emd::u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<emd::u32x4*>(g + 64) = emd::u32x4{l[0], l[1], l[2], l[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 conv to ONE output channel: LP = Cin/4 lanes share a pixel (each owns 4 input channels of
// all 9 taps) and reduce their partial dot products with wave shuffles; 64/LP pixels per wave.
__global__ __launch_bounds__(256) void conv3x3_cout1_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, float scale,
                                                            float shift, float* __restrict__ y, int H, int W,
                                                            int LP, long npix, int act, float pre_bias, int pre_relu) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int ppw = 64 / LP;
    const int c4 = lane % LP, sub = lane / LP;
    const int C = LP * 4;
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(w + k * C + c4 * 4);
    const long pix = wave * ppw + sub;
    float4 acc = f4zero();
    if (pix < npix) {
        int ox, oy;
        const long t = emd::divmod(pix, W, ox);
        const long b = emd::divmod(t, H, oy);
        const float* xb = x + (b * H) * (long)W * ldx + c4 * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = oy - 1 + i;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ix = ox - 1 + j;
                if (ix < 0 || ix >= W) continue;
                acc = fma4(wk[i * 3 + j], *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx), acc);
            }
        }
    }
    float s = (acc.x + acc.y) + (acc.z + acc.w);
    for (int m = 1; m < LP; m <<= 1) s += __shfl_xor(s, m);
    if (c4 == 0 && pix < npix) {
        y[pix] = cout1_out(s, pre_bias, pre_relu, scale, shift, act);
    }
}

// Rolling form of the above (W % (64/LP) == 0): a lane owns (pixel column, 4 channels) and walks down TH
// output rows keeping the vector partial sums of the three live input rows, so an input row is read once
// per strip (3 shifted 16-B loads per lane) instead of 9 loads per output; the cross-lane reduction over the
// LP channel groups happens once per output pixel.
template <int TH, bool REFLECT>
__global__ __launch_bounds__(256) void conv3x3_cout1_roll(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ w, float scale, float shift,
                                                          float* __restrict__ y, int H, int W, int LP, int nstrip,
                                                          long nthreads, int act, float pre_bias, int pre_relu) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;  // nthreads is a multiple of 64: whole waves leave together
    int c4, ox, strip;
    long t = emd::divmod(tid, LP, c4);
    t = emd::divmod(t, W, ox);
    const long b = emd::divmod(t, nstrip, strip);
    const int C = LP * 4;
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const float4*>(w + k * C + c4 * 4);
    const float* xb = x + (b * H) * (long)W * ldx + c4 * 4;
    float* yb = y + (b * H) * (long)W;
    const int oy0 = strip * TH;
    const bool hasl = ox > 0, hasr = ox + 1 < W;
    // As in dw3x3_s1_roll: every load unconditional (indices clamped or reflected into the image, a padding value replaced by
    // zero where it is USED) and issued PF rows ahead; the first version branched around each load and waited row by row
    // (629 us for the 64-channel 512^2 x 32 maps of graph D, 2.1 GB).  REFLECT: tf.pad(REFLECT, 1) + VALID (graph G).
    constexpr int PF = 3, NR = TH + 2;
    const long xl = hasl ? -(long)ldx : (REFLECT ? (long)ldx : 0), xr = hasr ? (long)ldx : (REFLECT ? -(long)ldx : 0);
    auto row_ptr = [&](int tt) {
        int iy = oy0 - 1 + tt;
        if (REFLECT) iy = iy < 0 ? -iy : (iy >= H ? 2 * H - 2 - iy : iy);
        iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        return xb + ((long)iy * W + ox) * ldx;
    };
    float4 rc[PF], rl[PF], rr[PF];
#pragma unroll
    for (int t0 = 0; t0 < PF && t0 < NR; ++t0) {
        const float* row = row_ptr(t0);
        rc[t0] = *reinterpret_cast<const float4*>(row);
        rl[t0] = *reinterpret_cast<const float4*>(row + xl);
        rr[t0] = *reinterpret_cast<const float4*>(row + xr);
    }
    float4 s0 = f4zero(), s1 = f4zero();
#pragma unroll
    for (int tt = 0; tt < NR; ++tt) {
        const int iy = oy0 - 1 + tt;
        const bool ok = REFLECT || (iy >= 0 && iy < H);
        const float4 c = ok ? rc[tt % PF] : f4zero();
        const float4 l = ok && (REFLECT || hasl) ? rl[tt % PF] : f4zero();
        const float4 r = ok && (REFLECT || hasr) ? rr[tt % PF] : f4zero();
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
            const float4 a = add4(s0, h2);
            float s = (a.x + a.y) + (a.z + a.w);
            if (LP == 16) {
                // 16 lanes = one DPP row: quad butterflies, then the neighbouring quad and the other half by row rotation (data
                // parallel primitives, no LDS crossbar).  Lane c4 == 0 -- the one that stores -- sums in the butterfly's order:
                // ((q0 + q1) + (q2 + q3)).
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x124, 0xF, 0xF, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xF, 0xF, true));
            } else {
                for (int m = 1; m < LP; m <<= 1) s += __shfl_xor(s, m);
            }
            if (c4 == 0 && oy < H) yb[(long)oy * W + ox] = cout1_out(s, pre_bias, pre_relu, scale, shift, act);
        }
        s0 = add4(s1, h1);
        s1 = h0;
    }
}

// ------------------------------------------------------------------------------------------------
// tf.image.resize_images: bilinear, align_corners=False, legacy sampling src = dst * (in/out).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int ldx,
                                                              float* __restrict__ y, int ldy, int Hi, int Wi,
                                                              int Ho, int Wo, int C4, float sy, float sx,
                                                              long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    int c4, ox, oy;
    long t = emd::divmod(tid, C4, c4);
    t = emd::divmod(t, Wo, ox);
    const long b = emd::divmod(t, Ho, oy);
    const float fy = (float)oy * sy, fx = (float)ox * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* xb = x + (b * Hi) * (long)Wi * ldx + c4 * 4;
    const float4 tl = *reinterpret_cast<const float4*>(xb + ((long)y0 * Wi + x0) * ldx);
    const float4 tr = *reinterpret_cast<const float4*>(xb + ((long)y0 * Wi + x1) * ldx);
    const float4 bl = *reinterpret_cast<const float4*>(xb + ((long)y1 * Wi + x0) * ldx);
    const float4 br = *reinterpret_cast<const float4*>(xb + ((long)y1 * Wi + x1) * ldx);
    auto lerp = [](float a, float b2, float l) { return a + (b2 - a) * l; };
    float4 o;
    o.x = lerp(lerp(tl.x, tr.x, lx), lerp(bl.x, br.x, lx), ly);
    o.y = lerp(lerp(tl.y, tr.y, lx), lerp(bl.y, br.y, lx), ly);
    o.z = lerp(lerp(tl.z, tr.z, lx), lerp(bl.z, br.z, lx), ly);
    o.w = lerp(lerp(tl.w, tr.w, lx), lerp(bl.w, br.w, lx), ly);
    *reinterpret_cast<float4*>(y + ((b * Ho + oy) * (long)Wo + ox) * ldy + c4 * 4) = o;
}

// The same for an exact 2x upsampling (Ho = 2 Hi, Wo = 2 Wi: every resize of graph G's generator): src = dst / 2, so the
// weights are 0 or 1/2 and one SOURCE pixel feeds a 2 x 2 block of outputs.  A thread owns (source pixel, channel quad):
// four loads and one index decomposition for four 16-byte stores (the kernel above: four loads and a decomposition per
// store).  Same expression per output, so the same bits.
__global__ __launch_bounds__(256) void resize_up2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                         int Hi, int Wi, int C4, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    int c4, sx, sy;
    long t = emd::divmod(tid, C4, c4);
    t = emd::divmod(t, Wi, sx);
    const long b = emd::divmod(t, Hi, sy);
    const int y1 = min(sy + 1, Hi - 1), x1 = min(sx + 1, Wi - 1);
    const float* xb = x + (b * Hi) * (long)Wi * ldx + c4 * 4;
    const float4 tl = *reinterpret_cast<const float4*>(xb + ((long)sy * Wi + sx) * ldx);
    const float4 tr = *reinterpret_cast<const float4*>(xb + ((long)sy * Wi + x1) * ldx);
    const float4 bl = *reinterpret_cast<const float4*>(xb + ((long)y1 * Wi + sx) * ldx);
    const float4 br = *reinterpret_cast<const float4*>(xb + ((long)y1 * Wi + x1) * ldx);
    auto lerp = [](float a, float b2, float l) { return a + (b2 - a) * l; };
    const int Wo = 2 * Wi;
    float* yb = y + ((b * 2 * Hi + 2 * sy) * (long)Wo + 2 * sx) * ldy + c4 * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float ly = 0.5f * (float)i, lx = 0.5f * (float)j;
            float4 o;
            o.x = lerp(lerp(tl.x, tr.x, lx), lerp(bl.x, br.x, lx), ly);
            o.y = lerp(lerp(tl.y, tr.y, lx), lerp(bl.y, br.y, lx), ly);
            o.z = lerp(lerp(tl.z, tr.z, lx), lerp(bl.z, br.z, lx), ly);
            o.w = lerp(lerp(tl.w, tr.w, lx), lerp(bl.w, br.w, lx), ly);
            *reinterpret_cast<float4*>(yb + ((long)i * Wo + j) * ldy) = o;
        }
}

// y = act(x*scale + shift) [+ res]; x and y may be the same buffer (elementwise, same index).
__global__ __launch_bounds__(256) void affine_relu6_kernel(const float* x, int ldx, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, const float* res, int ldres,
                                                           float* y, int ldy, int C4, long nthreads, int act,
                                                           long npix_img, const float* __restrict__ rsc = nullptr,
                                                           const float* __restrict__ rsh = nullptr, float rhi = 0.f) {
    // rsc != NULL: the residual operand is itself given before ITS affine + activation, res = min(max(res*rsc + rsh, 0), rhi) (round 4: the
    // 1x1 residual projection's norm + relu6 applied here instead of in a pass of its own; vectors indexed like sc / sh)
    // npix_img != 0: scale / shift are [image][C] (per-image statistics: instance norms, graph S), image = pix / npix_img
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    int c4;
    const long pix = emd::divmod(tid, C4, c4);
    const long so = npix_img ? (((unsigned long)pix >> 32) == 0 && npix_img <= 0x7fffffffL ? (long)((unsigned)pix / (unsigned)npix_img) : pix / npix_img) * C4 * 4 : 0;
    const float4 v = *reinterpret_cast<const float4*>(x + pix * ldx + c4 * 4);
    const float4 s = *reinterpret_cast<const float4*>(sc + so + c4 * 4);
    const float4 t = *reinterpret_cast<const float4*>(sh + so + c4 * 4);
    float4 o = make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
    if (act == 4) {  // tf.nn.leaky_relu, alpha 0.2
        o = make_float4(o.x > 0.f ? o.x : 0.2f * o.x, o.y > 0.f ? o.y : 0.2f * o.y, o.z > 0.f ? o.z : 0.2f * o.z,
                        o.w > 0.f ? o.w : 0.2f * o.w);
    } else if (act) {
        const float hi = act == 2 ? __builtin_inff() : (act == 3 ? 1.f : 6.f);  // 3: relu6 then clip to [0,1]
        o = make_float4(fminf(fmaxf(o.x, 0.f), hi), fminf(fmaxf(o.y, 0.f), hi), fminf(fmaxf(o.z, 0.f), hi),
                        fminf(fmaxf(o.w, 0.f), hi));
    }
    if (res) {
        float4 rv = *reinterpret_cast<const float4*>(res + pix * ldres + c4 * 4);
        if (rsc) rv = clamp4(fma4(rv, *reinterpret_cast<const float4*>(rsc + so + c4 * 4), *reinterpret_cast<const float4*>(rsh + so + c4 * 4)), rhi);
        o = add4(o, rv);
    }
    *reinterpret_cast<float4*>(y + pix * ldy + c4 * 4) = o;
}

// ------------------------------------------------------------------------------------------------
// Batch statistics of a [npix, C] tensor (tf.contrib.layers.batch_norm with is_training=True, as the
// separable convs of misc_py/modified_Xception.py:302-323 call it even at inference): per-channel mean and
// BIASED variance over N,H,W.  Pass 1: each block sums a slab of rows for 64 channels in double; pass 2
// combines the slabs.  Double keeps E[x^2]-E[x]^2 safe from cancellation.
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, int ldx, long npix, int C,
                                                        long rows_per_slab, double* __restrict__ part) {
    __shared__ double sm[2][4][64];
    x += (long)blockIdx.z * npix * ldx;                    // blockIdx.z = image (per-image statistics); 0 for batch statistics
    part += (long)blockIdx.z * gridDim.y * 2 * C;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;  // 4 row phases
    const long r0 = (long)blockIdx.y * rows_per_slab;
    const long r1 = min(r0 + rows_per_slab, npix);
    double s = 0.0, q = 0.0;
    if (c < C)
        for (long r = r0 + rsub; r < r1; r += 4) {
            const double v = (double)x[r * ldx + c];
            s += v;
            q += v * v;
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

// The same for ONE dense channel (C == 1, ldx == 1: the instance norm of the generator's output image, misc_py/gan-infilling-100.py:367):
// all 256 threads of the workgroup share the slab (the kernel above would leave 252 of them idle); fixed shuffle tree, so
// the sum is reproducible run to run.
__global__ __launch_bounds__(256) void bn_stats_partial_c1(const float* __restrict__ x, long npix, long rows_per_slab,
                                                           double* __restrict__ part) {
    __shared__ double sm[2][4];
    x += (long)blockIdx.z * npix;
    part += (long)blockIdx.z * gridDim.y * 2;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    const long r1 = min(r0 + rows_per_slab, npix);
    double s = 0.0, q = 0.0;
    for (long r = r0 + threadIdx.x; r < r1; r += 256) {
        const double v = (double)x[r];
        s += v;
        q += v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_down(s, d);
        q += __shfl_down(q, d);
    }
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = s;
        sm[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x < 2) part[(long)blockIdx.y * 2 + threadIdx.x] = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// The same for C % 4 == 0: 16 channel quads x 16 row lanes per workgroup, 16-byte loads, 4 rows in flight per thread.
__global__ __launch_bounds__(256) void bn_stats_partial_v4(const float* __restrict__ x, int ldx, long npix, int C,
                                                           long rows_per_slab, double* __restrict__ part) {
    __shared__ double sm[2][16][64 + 1];
    x += (long)blockIdx.z * npix * ldx;                    // blockIdx.z = image (per-image statistics); 0 for batch statistics
    part += (long)blockIdx.z * gridDim.y * 2 * C;
    const int cl = (threadIdx.x & 15) * 4;
    const int c = blockIdx.x * 64 + cl;
    const int rl = threadIdx.x >> 4;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    const long r1 = min(r0 + rows_per_slab, npix);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        const float* xp = x + c;
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(xp + r * ldx);
            const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
            s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
            q[0] += d0 * d0; q[1] += d1 * d1; q[2] += d2 * d2; q[3] += d3 * d3;
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

// FOLD: the batch norm that uses these statistics is folded right here as well (scale = gamma / sqrt(var + eps), shift = beta -
// mean * scale, from the float mean / var exactly as bn_fold_kernel computes them): one launch instead of two between a GEMM
// and the kernel that applies the norm (graph X has 63 such pairs on its critical path).
// TRAIN (round 4): the training-mode fold of the whole BN chain (bn_train_fold_kernel's step, bn_chain_dev.hpp: scale, shift, rstd1,
// rstd2 and the moving-average updates from image 0) in the same launch.
inline int final_cl(int nslab) { return nslab < 128 ? 16 : (nslab < 1024 ? 4 : 1); }   // channels per workgroup of bn_stats_final
// CL channels x (256 / CL) slab lanes per workgroup.  CL = 16 is the original geometry (<= 512 slabs of a channel summed by 16 lanes); the
// statistics epilogues of the GEMMs deliver one partial per 128-row tile -- 2 048 per 512^2 image -- and with 16 channels per workgroup a
// 64-channel layer had FOUR workgroups walking 128-256 dependent rounds: 87 us behind a 53 us GEMM (tools/small_gemm_bench.py, round 4).
// CL = 4 / 1 (launch rule: final_cl) give those layers 16 / 64 workgroups of 64 / 256 lanes.  The order of the sum depends on CL, which
// depends on nslab only: an image alone and the same image in a batch still get the same bits.
template <bool FOLD, bool TRAIN = false, int CL = 16>
__global__ __launch_bounds__(256) void bn_stats_final(const double* __restrict__ part, int nslab, int C, long npix,
                                                      float* __restrict__ mean, float* __restrict__ var,
                                                      const float* __restrict__ gamma = nullptr,
                                                      const float* __restrict__ beta = nullptr, float eps = 0.f,
                                                      float* __restrict__ scale = nullptr, float* __restrict__ shift = nullptr,
                                                      emd::BnFoldArgs fa = emd::BnFoldArgs{}) {
    constexpr int SL = 256 / CL;
    __shared__ double sm[2][SL][CL + 1];
    part += (long)blockIdx.y * nslab * 2 * C;              // blockIdx.y = image
    mean += (long)blockIdx.y * C;
    var += (long)blockIdx.y * C;
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
    if constexpr (SL > 16) {   // SL lanes -> 16 (lane j takes the SL / 16 consecutive entries from j * SL / 16), then as before
        if (k0 < 16) {
            double s2 = 0.0, q2 = 0.0;
#pragma unroll 4
            for (int k = 0; k < SL / 16; ++k) {
                s2 += sm[0][k0 * (SL / 16) + k][l];
                q2 += sm[1][k0 * (SL / 16) + k][l];
            }
            s = s2;
            q = q2;
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
    const double m = s / (double)npix;
    double v = q / (double)npix - m * m;
    const float mf = (float)m, vf = (float)(v > 0.0 ? v : 0.0);
    mean[c] = mf;
    var[c] = vf;
    if (FOLD) {
        const float g = (gamma ? gamma[c] : 1.0f) / sqrtf(vf + eps);
        scale[c] = g;
        shift[c] = (beta ? beta[c] : 0.0f) - mf * g;
    }
    if (TRAIN) emd::bn_train_fold_one(fa, (int)blockIdx.y * C + c, c, blockIdx.y == 0, mf, vf, (float)npix);
}

// scale = gamma / sqrt(var + eps) (gamma may be NULL = 1), shift = beta - mean*scale: a batch norm as one affine
__global__ void bn_fold_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                               float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float g = (gamma ? gamma[c] : 1.0f) / sqrtf(var[c] + eps);
    scale[c] = g;
    shift[c] = (beta ? beta[c] : 0.0f) - mean[c] * g;
}

// tf.nn.pool(window (2,2), "AVG", "SAME", strides (2,2)): mean over the window's in-image samples.
__global__ __launch_bounds__(256) void avgpool2x2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                         int ldy, int H, int W, int Ho, int Wo, int C4, long nthreads) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nthreads) return;
    int c4, ox, oy;
    long t = emd::divmod(tid, C4, c4);
    t = emd::divmod(t, Wo, ox);
    const long b = emd::divmod(t, Ho, oy);
    const float* xb = x + (b * H) * (long)W * ldx + c4 * 4;
    float4 acc = f4zero();
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = 2 * oy + i, ix = 2 * ox + j;  // SAME with k=2,s=2: pad 0 before (only "after" on odd sizes)
            if (iy < H && ix < W) {
                acc = add4(acc, *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx));
                ++cnt;
            }
        }
    const float inv = 1.0f / (float)cnt;
    *reinterpret_cast<float4*>(y + ((b * Ho + oy) * (long)Wo + ox) * ldy + c4 * 4) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

inline int same_pad_before(int n, int k, int s, int r, int* out) {
    const int o = (n + s - 1) / s;
    const int eff = (k - 1) * r + 1;
    int total = (o - 1) * s + eff - n;
    if (total < 0) total = 0;
    *out = o;
    return total / 2;
}

inline int grid_for(long nthreads, unsigned* blocks) {
    const long nb = (nthreads + 255) / 256;
    if (nb <= 0 || nb > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "grid too large");
    *blocks = (unsigned)nb;
    return EMD_OK;
}

template <bool SPLIT, bool PRE = false>
int dw3x3_launch(const char* who, const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C,
                 int stride, int rate, emd_stream_t stream, const float* pre_s = nullptr, const float* pre_t = nullptr, long pre_ld = 0,
                 float pre_hi = __builtin_inff()) {
    (void)who;
    if (PRE) EMD_REQUIRE(pre_s && pre_t && emd::aligned16(pre_s) && emd::aligned16(pre_t), EMD_E_INVALID,
                         "emd_dw3x3_pre: pre_scale / pre_shift must be non-null and 16-byte aligned");
    EMD_REQUIRE(x && w && y, EMD_E_INVALID, "emd_dw3x3: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= 4, EMD_E_INVALID, "emd_dw3x3: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_dw3x3: stride must be 1 or 2");
    EMD_REQUIRE(rate >= 1 && (rate == 1 || stride == 1), EMD_E_UNSUPPORTED, "emd_dw3x3: rate > 1 needs stride 1");
    const int Cp = (C + 31) / 32 * 32;
    if (SPLIT) {
        EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldx >= C && ldy % 32 == 0 && ldy >= Cp, EMD_E_ALIGN,
                    "emd_dw3x3_split32_f32: C, ldx multiples of 4; ldy a multiple of 32, >= ceil32(C)");
        EMD_REQUIRE(emd::aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 127u) == 0 && emd::aligned16(w), EMD_E_ALIGN,
                    "emd_dw3x3_split32_f32: x, w 16-byte and y 128-byte aligned");
    } else {
        EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, EMD_E_ALIGN,
                    "emd_dw3x3_f32: C, ldx, ldy must be multiples of 4 and ld >= C");
        EMD_REQUIRE(emd::aligned16(x) && emd::aligned16(y) && emd::aligned16(w), EMD_E_ALIGN,
                    "emd_dw3x3_f32: pointers must be 16-byte aligned");
    }
    if (B == 0) return EMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int Ho, Wo;
    const int pt = same_pad_before(H, 3, stride, rate, &Ho);
    const int pl = same_pad_before(W, 3, stride, rate, &Wo);
    const int C4 = C / 4, C4t = SPLIT ? Cp / 4 : C4;
    unsigned nb;
    if (stride == 1 && rate == 1) {
        // strip height: 16 rows (input re-read factor 18/16) measured 1-4 % faster than 8 on the 256^2/512^2 layers;
        // short images keep 8 so that small maps still spread over the chip
        const int th_force = emd::g_knobs.dw_th;
        // (round 3: 16 rows also on the short maps while that still leaves >= 1024 workgroups -- 32 x 32 x 728 at a batch of 32: 1536
        // workgroups, 2.38 -> 2.28 ms over graph D's 43 standalone launches)
        const long wg16 = (long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((C4t + 15) / 16);
        int TH = th_force == 4 || th_force == 8 || th_force == 16 || th_force == 32 ? th_force : ((H >= 64 || (H >= 16 && wg16 >= 1024)) ? 16 : 8);
        // small batches of small maps (4 images of 32 x 32 x 728: 384 workgroups of 8 rows): a thread's 10 dependent row loads are the
        // kernel's whole life and most CUs hold one workgroup -- 4-row strips double the workgroups and halve the chain (same bits:
        // the three row contributions of an output are added in the same order at every strip height)
        if (!th_force && TH == 8 && (long)B * ((H + 7) / 8) * ((W + 15) / 16) * ((C4t + 15) / 16) < 1024) TH = 4;
        const int nstrip = (H + TH - 1) / TH;
        const long nblocks = (long)B * nstrip * ((W + 15) / 16) * ((C4t + 15) / 16);
        const long nthreads = nblocks * 256;
        int rc = grid_for(nthreads, &nb);
        if (rc != EMD_OK) return rc;
        if (TH == 32)
            hipLaunchKernelGGL((dw3x3_s1_roll<32, SPLIT, false, PRE>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t, pre_s, pre_t, dw_xcd(H, W), pre_ld, pre_hi);
        else if (TH == 16)
            hipLaunchKernelGGL((dw3x3_s1_roll<16, SPLIT, false, PRE>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t, pre_s, pre_t, dw_xcd(H, W), pre_ld, pre_hi);
        else if (TH == 4)
            hipLaunchKernelGGL((dw3x3_s1_roll<4, SPLIT, false, PRE>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t, pre_s, pre_t, dw_xcd(H, W), pre_ld, pre_hi);
        else
            hipLaunchKernelGGL((dw3x3_s1_roll<8, SPLIT, false, PRE>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t, pre_s, pre_t, dw_xcd(H, W), pre_ld, pre_hi);
        return emd::check_launch("dw3x3_s1_roll");
    }
    const long nthreads = (long)B * ((Ho + 3) / 4) * ((Wo + 3) / 4) * ((C4t + 15) / 16) * 256;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL((dw3x3_generic<SPLIT, PRE>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, Ho, Wo, stride, rate, pt,
                       pl, nthreads, C4t, pre_s, pre_t, dw_xcd(H, W), pre_ld, pre_hi);
    return emd::check_launch("dw3x3_generic");
}

}  // namespace

// second stage of the batch statistics (sum over `nslab` partials per channel, fixed order), for producers of partials
// outside this file (the statistics epilogue of gemm_split.hip)
int emd::launch_bn_stats_final(const double* part, int nslab, int C, long npix, float* mean, float* var, hipStream_t st,
                               const float* gamma, const float* beta, float eps, float* scale, float* shift, int images,
                               const emd::BnFoldArgs* train_fold) {
    // images > 1: per-image statistics -- `nslab` partials and `npix` pixels PER IMAGE, part [image][nslab][2][C], mean / var [image][C]
    const int cl = final_cl(nslab);
    const unsigned ni = images > 1 ? images : 1;
    const emd::BnFoldArgs fa = train_fold ? *train_fold : emd::BnFoldArgs{};
#define EMD_FINAL(FOLD, TRAIN, CLV, GY) \
    hipLaunchKernelGGL((bn_stats_final<FOLD, TRAIN, CLV>), dim3((C + CLV - 1) / CLV, GY), dim3(256), 0, st, part, nslab, C, npix, mean, var, gamma, \
                       beta, eps, scale, shift, fa)
    if (train_fold) {
        if (cl == 16) EMD_FINAL(false, true, 16, ni); else if (cl == 4) EMD_FINAL(false, true, 4, ni); else EMD_FINAL(false, true, 1, ni);
        return emd::check_launch("bn_stats_final<training fold>");
    }
    if (scale) {
        if (cl == 16) EMD_FINAL(true, false, 16, 1); else if (cl == 4) EMD_FINAL(true, false, 4, 1); else EMD_FINAL(true, false, 1, 1);
    } else {
        if (cl == 16) EMD_FINAL(false, false, 16, ni); else if (cl == 4) EMD_FINAL(false, false, 4, ni); else EMD_FINAL(false, false, 1, ni);
    }
#undef EMD_FINAL
    return emd::check_launch("bn_stats_final");
}

// emd_conv3x3_cout1_reflect_f32 on the rolling kernel (gan_ops.hip calls it where a wave never straddles an image row)
int emd::launch_conv3x3_cout1_reflect_roll(const float* x, int ldx, const float* w, float bias, float* y, int B, int H, int W,
                                           int Cin, hipStream_t st) {
    constexpr int TH = 8;
    const int LP = Cin / 4, nstrip = (H + TH - 1) / TH;
    const long nthreads = (long)B * nstrip * W * LP;
    unsigned nb;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL((conv3x3_cout1_roll<TH, true>), dim3(nb), dim3(256), 0, st, x, ldx, w, 1.f, bias, y, H, W, LP, nstrip,
                       nthreads, 0, 0.f, 0);
    return emd::check_launch("conv3x3_cout1_roll (reflect)");
}

// stride-1 depthwise over the REFLECT-padded input on the rolling kernel (called by emd_dw3x3_reflect*_f32, gan_ops.hip;
// arguments already validated there)
int emd::launch_dw3x3_reflect_roll(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C,
                                   bool split, hipStream_t st) {
    const int C4 = C / 4, C4t = split ? (C + 31) / 32 * 8 : C4;
    const int TH = H >= 64 ? 16 : 8;
    const int nstrip = (H + TH - 1) / TH;
    const long nthreads = (long)B * nstrip * ((W + 15) / 16) * ((C4t + 15) / 16) * 256;
    unsigned nb;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    if (TH == 16) {
        if (split) hipLaunchKernelGGL((dw3x3_s1_roll<16, true, true>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t);
        else hipLaunchKernelGGL((dw3x3_s1_roll<16, false, true>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t);
    } else {
        if (split) hipLaunchKernelGGL((dw3x3_s1_roll<8, true, true>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t);
        else hipLaunchKernelGGL((dw3x3_s1_roll<8, false, true>), dim3(nb), dim3(256), 0, st, x, ldx, w, y, ldy, H, W, C4, nthreads, nstrip, C4t);
    }
    return emd::check_launch("dw3x3_s1_roll (reflect)");
}

extern "C" int emd_dw3x3_f32(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W,
                             int C, int stride, int rate, emd_stream_t stream) {
    return dw3x3_launch<false>("emd_dw3x3_f32", x, ldx, w, y, ldy, B, H, W, C, stride, rate, stream);
}

extern "C" int emd_dw3x3_split32_f32(const float* x, int ldx, const float* w, void* y, int ldy, int B, int H, int W,
                                     int C, int stride, int rate, emd_stream_t stream) {
    return dw3x3_launch<true>("emd_dw3x3_split32_f32", x, ldx, w, static_cast<float*>(y), ldy, B, H, W, C, stride, rate, stream);
}

// y = depthwise3x3( relu(x * pre_scale + pre_shift) ): the previous block's batch-statistics norm + relu applied on the fly
extern "C" int emd_dw3x3_pre_f32(const float* x, int ldx, const float* pre_scale, const float* pre_shift, const float* w, float* y,
                                 int ldy, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream) {
    return dw3x3_launch<false, true>("emd_dw3x3_pre_f32", x, ldx, w, y, ldy, B, H, W, C, stride, rate, stream, pre_scale, pre_shift);
}

// The same with the activation and the statistics' granularity chosen (round 4: graph D' -- the training step's affine + relu6 of a
// separable conv whose only consumer is the next one's depthwise stage is applied in that stage's loads; slim.separable_convolution2d +
// _batch_norm_fn + relu6, machine_learning/denoiser.py:110-136 with phase = True): pre_images != 0: pre_scale / pre_shift are [B][C]
// (per-image statistics); act: EMD_ACT_RELU6 (1) or EMD_ACT_RELU (2).  Bits of emd_affine_act[_images]_f32 followed by emd_dw3x3_f32.
extern "C" int emd_dw3x3_pre_act_f32(const float* x, int ldx, const float* pre_scale, const float* pre_shift, int pre_images, int act,
                                     const float* w, float* y, int ldy, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream) {
    EMD_REQUIRE(act == 1 || act == 2, EMD_E_INVALID, "emd_dw3x3_pre_act_f32: act must be EMD_ACT_RELU6 or EMD_ACT_RELU");
    return dw3x3_launch<false, true>("emd_dw3x3_pre_act_f32", x, ldx, w, y, ldy, B, H, W, C, stride, rate, stream, pre_scale, pre_shift,
                                     pre_images ? (long)C : 0L, act == 1 ? 6.f : __builtin_inff());
}

extern "C" int emd_dw3x3_pre_split32_f32(const float* x, int ldx, const float* pre_scale, const float* pre_shift, const float* w,
                                         void* y, int ldy, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream) {
    return dw3x3_launch<true, true>("emd_dw3x3_pre_split32_f32", x, ldx, w, static_cast<float*>(y), ldy, B, H, W, C, stride, rate,
                                    stream, pre_scale, pre_shift);
}

extern "C" int emd_cin1_f32(const float* x, const float* w9, const float* a, const float* shift, float* y, int ldy,
                            int B, int H, int W, int Cout, int stride, int act, emd_stream_t stream) {
    EMD_REQUIRE(x && a && shift && y, EMD_E_INVALID, "emd_cin1_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_cin1_f32: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_cin1_f32: stride must be 1 or 2");
    EMD_REQUIRE(!(w9 && stride != 1), EMD_E_UNSUPPORTED, "emd_cin1_f32: the depthwise form is stride 1 only");
    const int N4 = Cout / 4;
    EMD_REQUIRE(Cout % 4 == 0 && N4 >= 1 && N4 <= 64 && (64 % N4) == 0, EMD_E_UNSUPPORTED,
                "emd_cin1_f32: Cout/4 must divide 64");
    EMD_REQUIRE(ldy % 4 == 0 && ldy >= Cout && emd::aligned16(y) && emd::aligned16(a) && emd::aligned16(shift),
                EMD_E_ALIGN, "emd_cin1_f32: alignment");
    if (B == 0) return EMD_OK;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const long npix = (long)B * Ho * Wo;
    unsigned nb;
    int rc = grid_for((npix + 63) / 64 * 64, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(cin1_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, w9, a, shift, y, ldy, H,
                       W, Ho, Wo, stride, N4, w9 ? 1 : 0, npix, act ? 1 : 0);
    return emd::check_launch("cin1_kernel");
}

extern "C" int emd_conv3x3_cin1_f32(const float* x, const float* w, const float* scale, const float* shift, void* y, int ldy, int B,
                                    int H, int W, int Cout, int stride, int act, int out_split, emd_stream_t stream) {
    EMD_REQUIRE(x && w && scale && shift && y, EMD_E_INVALID, "emd_conv3x3_cin1_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_cin1_f32: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_conv3x3_cin1_f32: stride must be 1 or 2");
    EMD_REQUIRE(Cout >= 8 && Cout % 8 == 0 && Cout <= 512, EMD_E_UNSUPPORTED, "emd_conv3x3_cin1_f32: Cout a multiple of 8, <= 512");
    EMD_REQUIRE(act == EMD_ACT_NONE || act == EMD_ACT_RELU6 || act == EMD_ACT_RELU, EMD_E_UNSUPPORTED, "emd_conv3x3_cin1_f32: act none / relu6 / relu");
    EMD_REQUIRE(emd::aligned16(w) && emd::aligned16(scale) && emd::aligned16(shift), EMD_E_ALIGN, "emd_conv3x3_cin1_f32: w, scale, shift 16-byte aligned");
    if (out_split)
        EMD_REQUIRE(ldy % 32 == 0 && ldy >= (Cout + 31) / 32 * 32 && (reinterpret_cast<uintptr_t>(y) & 127u) == 0, EMD_E_ALIGN,
                    "emd_conv3x3_cin1_f32: a split32 output needs y 128-byte aligned, ldy a multiple of 32, >= ceil32(Cout)");
    else
        EMD_REQUIRE(ldy % 4 == 0 && ldy >= Cout && emd::aligned16(y), EMD_E_ALIGN, "emd_conv3x3_cin1_f32: ldy a multiple of 4, >= Cout; y 16-byte aligned");
    if (B == 0) return EMD_OK;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    int pth = (Ho - 1) * stride + 3 - H, ptw = (Wo - 1) * stride + 3 - W;   // TF SAME: total padding, the smaller half first
    if (pth < 0) pth = 0;
    if (ptw < 0) ptw = 0;
    const long npix = (long)B * Ho * Wo;
    const int N8 = out_split ? (Cout + 31) / 32 * 4 : Cout / 8;
    unsigned nb;
    int rc = grid_for(npix * N8, &nb);
    if (rc != EMD_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_split)
        hipLaunchKernelGGL(conv3x3_cin1_kernel<true>, dim3(nb), dim3(256), 0, st, x, w, scale, shift, static_cast<float*>(y), ldy, H, W, Ho,
                           Wo, stride, pth / 2, ptw / 2, N8, Cout, npix, act);
    else
        hipLaunchKernelGGL(conv3x3_cin1_kernel<false>, dim3(nb), dim3(256), 0, st, x, w, scale, shift, static_cast<float*>(y), ldy, H, W, Ho,
                           Wo, stride, pth / 2, ptw / 2, N8, Cout, npix, act);
    return emd::check_launch("conv3x3_cin1_kernel");
}

extern "C" int emd_conv3x3_cout1_f32(const float* x, int ldx, const float* w, float scale, float shift, float* y,
                                     int B, int H, int W, int Cin, int act, float pre_bias, int pre_relu,
                                     emd_stream_t stream) {
    EMD_REQUIRE(x && w && y, EMD_E_INVALID, "emd_conv3x3_cout1_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_cout1_f32: bad shape");
    const int LP = Cin / 4;
    EMD_REQUIRE(Cin % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, EMD_E_UNSUPPORTED,
                "emd_conv3x3_cout1_f32: Cin/4 must be a power of two <= 64");
    EMD_REQUIRE(ldx % 4 == 0 && ldx >= Cin && emd::aligned16(x) && emd::aligned16(w), EMD_E_ALIGN,
                "emd_conv3x3_cout1_f32: alignment");
    if (B == 0) return EMD_OK;
    const long npix = (long)B * H * W;
    const int ppw = 64 / LP;
    unsigned nb;
    if (W % ppw == 0) {  // a wave never straddles an image row: rolling kernel (3 loads per output, not 9)
        constexpr int TH = 8;
        const int nstrip = (H + TH - 1) / TH;
        const long nthreads = (long)B * nstrip * W * LP;
        int rc = grid_for(nthreads, &nb);
        if (rc != EMD_OK) return rc;
        hipLaunchKernelGGL((conv3x3_cout1_roll<TH, false>), dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, w,
                           scale, shift, y, H, W, LP, nstrip, nthreads, act, pre_bias, pre_relu);
        return emd::check_launch("conv3x3_cout1_roll");
    }
    int rc = grid_for((npix + ppw - 1) / ppw * 64, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(conv3x3_cout1_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, w, scale,
                       shift, y, H, W, LP, npix, act, pre_bias, pre_relu);
    return emd::check_launch("conv3x3_cout1_kernel");
}

extern "C" int emd_resize_bilinear_f32(const float* x, int ldx, float* y, int ldy, int B, int Hi, int Wi, int Ho,
                                       int Wo, int C, emd_stream_t stream) {
    EMD_REQUIRE(x && y, EMD_E_INVALID, "emd_resize_bilinear_f32: null pointer");
    EMD_REQUIRE(B >= 0 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && C >= 4, EMD_E_INVALID,
                "emd_resize_bilinear_f32: bad shape");
    EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && emd::aligned16(x) &&
                    emd::aligned16(y), EMD_E_ALIGN, "emd_resize_bilinear_f32: alignment");
    if (B == 0) return EMD_OK;
    unsigned nb;
    if (Ho == 2 * Hi && Wo == 2 * Wi) {
        const long nsrc = (long)B * Hi * Wi * (C / 4);
        int rc2 = grid_for(nsrc, &nb);
        if (rc2 != EMD_OK) return rc2;
        hipLaunchKernelGGL(resize_up2_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, y, ldy, Hi, Wi,
                           C / 4, nsrc);
        return emd::check_launch("resize_up2_kernel");
    }
    const long nthreads = (long)B * Ho * Wo * (C / 4);
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, y, ldy,
                       Hi, Wi, Ho, Wo, C / 4, (float)Hi / (float)Ho, (float)Wi / (float)Wo, nthreads);
    return emd::check_launch("resize_bilinear_kernel");
}

// y = act(x*scale + shift) + res_act(res*res_scale + res_shift): emd_affine_act[_images]_f32 whose residual operand is given BEFORE its
// own affine + activation (round 4, graph D': the 1x1 residual projection's batch norm + relu6 -- conv_block_not_sep, machine_learning/
// denoiser.py:356-383 with phase = True -- applied where the block adds it, instead of in a pass of its own; bits of the two-pass route).
// images = 0: all vectors [C], npix = all pixels; images = B > 0: all vectors [B][C], npix = pixels per image.  res_act: RELU6 or RELU.
extern "C" int emd_affine_act_res_affine_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldres,
                                             const float* res_scale, const float* res_shift, int res_act, float* y, int ldy, int images,
                                             long npix, int C, int act, emd_stream_t stream) {
    EMD_REQUIRE(x && y && scale && shift && res && res_scale && res_shift, EMD_E_INVALID, "emd_affine_act_res_affine_f32: null pointer");
    EMD_REQUIRE(images >= 0 && npix >= 0 && C >= 4 && act >= 0 && act <= 4 && (res_act == 1 || res_act == 2), EMD_E_INVALID,
                "emd_affine_act_res_affine_f32: bad argument");
    EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldres % 4 == 0 && ldx >= C && ldy >= C && ldres >= C && emd::aligned16(x) &&
                    emd::aligned16(y) && emd::aligned16(res) && emd::aligned16(scale) && emd::aligned16(shift) && emd::aligned16(res_scale) &&
                    emd::aligned16(res_shift), EMD_E_ALIGN, "emd_affine_act_res_affine_f32: alignment");
    if (npix == 0) return EMD_OK;
    const long nthreads = (images ? (long)images : 1L) * npix * (C / 4);
    unsigned nb;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(affine_relu6_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, scale, shift, res, ldres, y, ldy,
                       C / 4, nthreads, act, images ? npix : 0L, res_scale, res_shift, res_act == 1 ? 6.f : __builtin_inff());
    return emd::check_launch("affine_relu6_kernel (residual affine)");
}

extern "C" int emd_affine_act_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res,
                                  int ldres, float* y, int ldy, long npix, int C, int act, emd_stream_t stream) {
    EMD_REQUIRE(x && y && scale && shift, EMD_E_INVALID, "emd_affine_act_f32: null pointer");
    EMD_REQUIRE(npix >= 0 && C >= 4 && act >= 0 && act <= 4, EMD_E_INVALID, "emd_affine_act_f32: bad argument");
    EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && emd::aligned16(x) &&
                    emd::aligned16(y) && emd::aligned16(scale) && emd::aligned16(shift) &&
                    (!res || (ldres % 4 == 0 && ldres >= C && emd::aligned16(res))), EMD_E_ALIGN,
                "emd_affine_act_f32: alignment");
    if (npix == 0) return EMD_OK;
    const long nthreads = npix * (C / 4);
    unsigned nb;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(affine_relu6_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, scale,
                       shift, res, ldres, y, ldy, C / 4, nthreads, act, 0L);
    return emd::check_launch("affine_relu6_kernel");
}

extern "C" int emd_affine_relu6_f32(const float* x, int ldx, const float* scale, const float* shift, float* y,
                                    int ldy, long npix, int C, int act, emd_stream_t stream) {
    return emd_affine_act_f32(x, ldx, scale, shift, nullptr, 0, y, ldy, npix, C, act ? 1 : 0, stream);
}

extern "C" size_t emd_bn_stats_workspace_bytes(long npix, int C) {
    if (npix <= 0 || C <= 0) return 0;
    return (size_t)emd::reduce_slabs(npix) * 2 * C * sizeof(double);
}

extern "C" int emd_bn_stats_f32(const float* x, int ldx, long npix, int C, float* mean, float* var, void* workspace,
                                emd_stream_t stream) {
    EMD_REQUIRE(x && mean && var && workspace, EMD_E_INVALID, "emd_bn_stats_f32: null pointer");
    EMD_REQUIRE(npix >= 1 && C >= 1 && ldx >= C, EMD_E_INVALID, "emd_bn_stats_f32: bad shape");
    EMD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EMD_E_ALIGN, "emd_bn_stats_f32: workspace alignment");
    const long nslab = emd::reduce_slabs(npix), rows_per_slab = emd::reduce_rows_per_slab(npix);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (C == 1 && ldx == 1)
        hipLaunchKernelGGL(bn_stats_partial_c1, dim3(1, (unsigned)nslab), dim3(256), 0, st, x, npix, rows_per_slab, ws);
    else if (C % 4 == 0 && ldx % 4 == 0 && emd::aligned16(x))
        hipLaunchKernelGGL(bn_stats_partial_v4, dim3((C + 63) / 64, (unsigned)nslab), dim3(256), 0, st, x, ldx, npix, C,
                           rows_per_slab, ws);
    else
        hipLaunchKernelGGL(bn_stats_partial, dim3((C + 63) / 64, (unsigned)nslab), dim3(256), 0, st, x, ldx, npix, C,
                           rows_per_slab, ws);
    return emd::launch_bn_stats_final(static_cast<const double*>(ws), (int)nslab, C, npix, mean, var, st);
}

// Per-image statistics of a batch [B][npix_img][C] in one pair of launches (instance norms; the batch-statistics norms of
// misc_py/apply_autoencoders.py:105-116, which the reference evaluates one crop per sess.run): mean / var are [B][C]; every
// image is reduced exactly as emd_bn_stats_f32 reduces it alone (same slabs, same order: identical bits).
extern "C" int emd_bn_stats_images_f32(const float* x, int ldx, int B, long npix_img, int C, float* mean, float* var,
                                       void* workspace, emd_stream_t stream) {
    EMD_REQUIRE(x && mean && var && workspace, EMD_E_INVALID, "emd_bn_stats_images_f32: null pointer");
    EMD_REQUIRE(B >= 1 && B <= 65535 && npix_img >= 1 && C >= 1 && ldx >= C, EMD_E_INVALID, "emd_bn_stats_images_f32: bad shape");
    EMD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EMD_E_ALIGN, "emd_bn_stats_images_f32: workspace alignment");
    const long nslab = emd::reduce_slabs(npix_img), rows_per_slab = emd::reduce_rows_per_slab(npix_img);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (C == 1 && ldx == 1)
        hipLaunchKernelGGL(bn_stats_partial_c1, dim3(1, (unsigned)nslab, B), dim3(256), 0, st, x, npix_img, rows_per_slab, ws);
    else if (C % 4 == 0 && ldx % 4 == 0 && emd::aligned16(x))
        hipLaunchKernelGGL(bn_stats_partial_v4, dim3((C + 63) / 64, (unsigned)nslab, B), dim3(256), 0, st, x, ldx, npix_img, C,
                           rows_per_slab, ws);
    else
        hipLaunchKernelGGL(bn_stats_partial, dim3((C + 63) / 64, (unsigned)nslab, B), dim3(256), 0, st, x, ldx, npix_img, C,
                           rows_per_slab, ws);
    return emd::launch_bn_stats_final(static_cast<const double*>(ws), (int)nslab, C, npix_img, mean, var, st, nullptr, nullptr, 0.f, nullptr, nullptr, B);
}

// y = act(x*scale[image] + shift[image]) [+ res]: emd_affine_act_f32 with per-image scale / shift vectors [B][C].
extern "C" int emd_affine_act_images_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res,
                                         int ldres, float* y, int ldy, int B, long npix_img, int C, int act,
                                         emd_stream_t stream) {
    EMD_REQUIRE(x && y && scale && shift, EMD_E_INVALID, "emd_affine_act_images_f32: null pointer");
    EMD_REQUIRE(B >= 0 && npix_img >= 1 && C >= 4 && act >= 0 && act <= 4, EMD_E_INVALID, "emd_affine_act_images_f32: bad argument");
    EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && emd::aligned16(x) &&
                    emd::aligned16(y) && emd::aligned16(scale) && emd::aligned16(shift) &&
                    (!res || (ldres % 4 == 0 && ldres >= C && emd::aligned16(res))), EMD_E_ALIGN,
                "emd_affine_act_images_f32: alignment");
    if (B == 0) return EMD_OK;
    const long nthreads = (long)B * npix_img * (C / 4);
    unsigned nb;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(affine_relu6_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, scale,
                       shift, res, ldres, y, ldy, C / 4, nthreads, act, npix_img);
    return emd::check_launch("affine_relu6_kernel (images)");
}

extern "C" int emd_bn_fold_f32(const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                               float* scale, float* shift, int C, emd_stream_t stream) {
    EMD_REQUIRE(mean && var && scale && shift && C >= 1, EMD_E_INVALID, "emd_bn_fold_f32: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), mean, var,
                       gamma, beta, eps, scale, shift, C);
    return emd::check_launch("bn_fold_kernel");
}

extern "C" int emd_avgpool2x2_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C,
                                  emd_stream_t stream) {
    EMD_REQUIRE(x && y, EMD_E_INVALID, "emd_avgpool2x2_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= 4, EMD_E_INVALID, "emd_avgpool2x2_f32: bad shape");
    EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && emd::aligned16(x) &&
                    emd::aligned16(y), EMD_E_ALIGN, "emd_avgpool2x2_f32: alignment");
    if (B == 0) return EMD_OK;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long nthreads = (long)B * Ho * Wo * (C / 4);
    unsigned nb;
    int rc = grid_for(nthreads, &nb);
    if (rc != EMD_OK) return rc;
    hipLaunchKernelGGL(avgpool2x2_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, y, ldy, H,
                       W, Ho, Wo, C / 4, nthreads);
    return emd::check_launch("avgpool2x2_kernel");
}
