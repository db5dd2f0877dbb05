// Graph K: learned symmetric-kernel denoiser (see include/emdenoise.h, emd_kernel_denoise_f32).
// replaces: misc_py/noise-removal-kernels.py:99-105, :378-399, :409-426.
//
// HBM-bound single-channel stencil: 8 algorithmic bytes per pixel (read 4, write 4).
//
// k3_tile<R,NW>    width 3, depth 2, D4-symmetric maps (the reference's configuration): see below.
// k3_rows<MODE,R>  width 3, any maps / depth 1.  One wavefront owns a strip of R output rows x 512 pixels:
//   each lane holds 8 consecutive pixels of a row (two 16-B loads, 2 KiB per wave-row, coalesced
//   along W), gets its two horizontal halo pixels from the neighbouring lanes by wave shuffles
//   (no LDS image), and rolls three rows through registers down the strip.  Every input row is
//   turned ONCE into its three row-contributions (as the top / middle / bottom row of a 3x3
//   window), so the sigmoid work is per INPUT pixel:
//     MODE_SYM  (D4-symmetric maps, depth 2): 3 sigmoids per input pixel (centre/edge/corner class)
//               instead of 9 per output pixel, because the weight a tap uses depends only on its
//               class and the value under it.
//     MODE_GEN  (any 3x3 maps, depth 2): 9 sigmoids per input pixel.
//     MODE_LIN  (depth 1): plain 3x3 correlation.
// k_generic        any odd width <= 15, depth <= 5, any H,W: one thread per pixel.
#include "emd_common.hpp"

namespace {

constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// a / (1 + 2^(p*ws + bs))  ==  a * sigmoid(w*p + b)  with ws = -w*log2(e), bs = -b*log2(e)
__device__ __forceinline__ float sig_term(float p, float ws, float bs, float a) {
    const float e = __builtin_amdgcn_exp2f(fmaf(p, ws, bs));
    return a * __builtin_amdgcn_rcpf(1.0f + e);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_generic(const float* __restrict__ x, float* __restrict__ y,
                                                 int B, int H, int W, int width, int depth,
                                                 const float* __restrict__ params) {
    const int ww = width * width;
    const float* wm = params;
    const float* bm = params + depth * ww;
    const float* sc = params + 2 * depth * ww;
    const int p = width >> 1;
    const long total = (long)B * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % W);
        const long t = idx / W;
        const int r = (int)(t % H);
        const float* img = x + (t / H) * (long)H * W;
        float acc = 0.f;
        for (int i = 0; i < width; ++i) {
            const int rr = reflect_idx(r + i - p, H);
            for (int j = 0; j < width; ++j) {
                const int cc = reflect_idx(c + j - p, W);
                const int k = i * width + j;
                float f = wm[k] * img[(long)rr * W + cc];
                for (int l = 1; l < depth; ++l) {
                    const float z = f + bm[l * ww + k];
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * z));
                    f = wm[l * ww + k] * (sc[l] * sg);
                }
                acc += f;
            }
        }
        y[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
enum { MODE_LIN = 0, MODE_SYM = 1, MODE_GEN = 2 };

constexpr int kPx = 8;  // pixels per lane

struct K3Params {  // pre-scaled on the device from the params block (wave-uniform -> SGPRs)
    float ws[9], bs[9], a[9];
};

template <int MODE>
__device__ __forceinline__ K3Params load_k3(const float* __restrict__ params, int depth) {
    K3Params q;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (MODE == MODE_LIN) {
            q.ws[k] = params[k];  // plain weights
            q.bs[k] = 0.f;
            q.a[k] = 0.f;
        } else {
            const float w0 = params[k];
            const float b1 = params[depth * 9 + 9 + k];
            const float w1 = params[9 + k];
            const float s1 = params[2 * depth * 9 + 1];
            q.ws[k] = -kLog2e * w0;
            q.bs[k] = -kLog2e * b1;
            q.a[k] = w1 * s1;
        }
    }
    return q;
}

// Row contributions of one input row: A[i][j] is what pixel column j of this row adds to the output
// row for which it is window-row i (0 top, 1 middle, 2 bottom).  p[0..9] = columns -1..8.
template <int MODE>
__device__ __forceinline__ void row_terms(const float (&p)[kPx + 2], const K3Params& q,
                                          float (&A0)[kPx], float (&A1)[kPx], float (&A2)[kPx]) {
    if (MODE == MODE_SYM) {
        // classes: centre k=4, edge k=1, corner k=0
        float E[kPx + 2], C[kPx + 2];
#pragma unroll
        for (int j = 0; j < kPx + 2; ++j) {
            E[j] = sig_term(p[j], q.ws[1], q.bs[1], q.a[1]);
            C[j] = sig_term(p[j], q.ws[0], q.bs[0], q.a[0]);
        }
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            const float Z = sig_term(p[j + 1], q.ws[4], q.bs[4], q.a[4]);
            A1[j] = Z + (E[j] + E[j + 2]);
            A0[j] = E[j + 1] + (C[j] + C[j + 2]);
            A2[j] = A0[j];
        }
    } else {
        float* const A[3] = {A0, A1, A2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < kPx; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int k = i * 3 + d;
                    if (MODE == MODE_LIN)
                        acc = fmaf(q.ws[k], p[j + d], acc);
                    else
                        acc += sig_term(p[j + d], q.ws[k], q.bs[k], q.a[k]);
                }
                A[i][j] = acc;
            }
        }
    }
}

template <int MODE, int R>
__global__ __launch_bounds__(256) void k3_rows(const float* __restrict__ x, float* __restrict__ y,
                                               int B, int H, int W, int depth,
                                               const float* __restrict__ params, int nseg,
                                               int nstrip) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nitems = (long)B * nstrip * nseg;
    if (item >= nitems) return;  // whole wave exits together
    const int seg = (int)(item % nseg);
    const long bs_ = item / nseg;
    const int strip = (int)(bs_ % nstrip);
    const int b = (int)(bs_ / nstrip);

    const K3Params q = load_k3<MODE>(params, depth);

    const int px0 = seg * (64 * kPx) + lane * kPx;
    const bool active = px0 < W;  // W % 8 == 0, so an active lane owns 8 valid pixels
    const bool first_px = px0 == 0;
    const bool last_px = px0 + kPx >= W;
    const float* img = x + (long)b * H * W;
    float* out = y + (long)b * H * W;
    const int r0 = strip * R;

    float s0[kPx], s1[kPx];
#pragma unroll
    for (int t = 0; t < R + 2; ++t) {
        int rr = reflect_idx(r0 - 1 + t, H);
        rr = rr < 0 ? 0 : (rr >= H ? H - 1 : rr);  // rows past a ragged last strip: any valid row
        const float* row = img + (long)rr * W;
        float p[kPx + 2];
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (active) {
            v0 = *reinterpret_cast<const float4*>(row + px0);
            v1 = *reinterpret_cast<const float4*>(row + px0 + 4);
        }
        p[1] = v0.x; p[2] = v0.y; p[3] = v0.z; p[4] = v0.w;
        p[5] = v1.x; p[6] = v1.y; p[7] = v1.z; p[8] = v1.w;
        // horizontal halo from the neighbouring lanes
        float left = __shfl_up(p[8], 1);
        float right = __shfl_down(p[1], 1);
        if (first_px) left = p[2];                     // REFLECT: column -1 -> column 1
        else if (lane == 0 && active) left = row[px0 - 1];
        if (last_px) right = p[7];                     // REFLECT: column W -> column W-2
        else if (lane == 63) right = row[px0 + kPx];
        p[0] = left;
        p[kPx + 1] = right;

        float A0[kPx], A1[kPx], A2[kPx];
        row_terms<MODE>(p, q, A0, A1, A2);

        if (t >= 2) {
            const int orow = r0 + t - 2;
            if (active && orow < H) {
                float4 o0, o1;
                o0.x = s0[0] + A2[0]; o0.y = s0[1] + A2[1]; o0.z = s0[2] + A2[2]; o0.w = s0[3] + A2[3];
                o1.x = s0[4] + A2[4]; o1.y = s0[5] + A2[5]; o1.z = s0[6] + A2[6]; o1.w = s0[7] + A2[7];
                float* orow_p = out + (long)orow * W + px0;
                *reinterpret_cast<float4*>(orow_p) = o0;
                *reinterpret_cast<float4*>(orow_p + 4) = o1;
            }
        }
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            s0[j] = (t >= 1) ? s1[j] + A1[j] : 0.f;
            s1[j] = A0[j];
        }
    }
}

template <int MODE, int R>
int launch_k3(const float* x, float* y, int B, int H, int W, int depth, const float* params,
              hipStream_t st) {
    const int nseg = (W + 64 * kPx - 1) / (64 * kPx);
    const int nstrip = (H + R - 1) / R;
    const long nitems = (long)B * nstrip * nseg;
    const long nblk = (nitems + 3) / 4;
    if (nblk > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "emd_kernel_denoise_f32: grid too large");
    hipLaunchKernelGGL((k3_rows<MODE, R>), dim3((unsigned)nblk), dim3(256), 0, st, x, y, B, H, W, depth,
                       params, nseg, nstrip);
    return emd::check_launch("k3_rows");
}


// ------------------------------------------------------------------------------------------------
// k3_tile<R,NW>  (D4-symmetric maps, depth 2 -- the reference's configuration).
// Built for thread-level parallelism: a wave owns only R rows x 512 pixels, so a batch is tens of
// thousands of short-lived waves whose loads overlap other waves' sigmoid work (a plain copy of the
// same bytes needs that many waves to reach the HBM rate on this chip).
//   * lane l holds pixels [4l,4l+4) and [256+4l,256+4l+4) of a row: both 16-B loads of a wave are
//     lane-contiguous 1-KiB requests;
//   * exactly 3 sigmoids per input pixel: a lane evaluates the centre/edge/corner class terms of its
//     own 8 pixels only and gets the two horizontal neighbours' TERMS (not pixels) by wave shuffles;
//     the two 256-pixel halves are stitched with v_readlane; REFLECT at the image border reuses the
//     lane's own terms.  Only when W > 512 do the two outermost lanes evaluate one extra pixel.
//   * the NW waves of a workgroup own consecutive R-row strips and pass the single row-contribution
//     that crosses a strip boundary through LDS; only the workgroup's two outer rows are evaluated
//     twice.
//   out[r] = V[r-1] + M[r] + V[r+1],  M = centre + horizontal edges,  V = vertical edge + corners.
template <int R, int NW>
__global__ __launch_bounds__(NW * 64) void k3_tile(const float* __restrict__ x, float* __restrict__ y,
                                                    int H, int W, int depth,
                                                    const float* __restrict__ params) {
    static_assert(R >= 2 && NW >= 2, "");
    constexpr int N = 8;
    __shared__ float4 xch_first[NW][2][64];  // V of a wave's first row
    __shared__ float4 xch_last[NW][2][64];   // V of a wave's last row
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // grid = (column segments of 512, row strips of NW*R, images): no integer division, and all
    // in-image offsets are 32-bit (H*W < 2^31 is checked on the host) -- the scalar unit is shared
    // by the whole CU, so 64-bit index arithmetic per wave is not free.
    const int seg = blockIdx.x, strip = blockIdx.y;
    const size_t img_off = (size_t)blockIdx.z * (size_t)(H * W);

    const K3Params q = load_k3<MODE_SYM>(params, depth);

    const int base = seg * 512;
    const int px[2] = {base + lane * 4, base + 256 + lane * 4};
    const bool act[2] = {px[0] < W, px[1] < W};
    const bool has_left = seg > 0;                 // wave-uniform: pixels exist left of this tile
    const bool has_right = base + 512 < W;         // wave-uniform
    const float* img = x + img_off;
    float* out = y + img_off;
    const int a = strip * (NW * R) + wv * R;

    struct Raw { float4 g[2]; float ext; };
    auto load_row = [&](int r, Raw& p) {
        int rr = reflect_idx(r, H);
        rr = rr < 0 ? 0 : (rr >= H ? H - 1 : rr);
        const float* row = img + rr * W;
        p.g[0] = p.g[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        p.ext = 0.f;
        if (act[0]) p.g[0] = *reinterpret_cast<const float4*>(row + px[0]);
        if (act[1]) p.g[1] = *reinterpret_cast<const float4*>(row + px[1]);
        if (has_left && lane == 0) p.ext = row[base - 1];
        if (has_right && lane == 63) p.ext = row[base + 512];
    };
    // V[j], M[j] for this lane's 8 pixels of one input row
    auto terms = [&](const Raw& p, float (&V)[N], float (&M)[N]) {
        const float pv[N] = {p.g[0].x, p.g[0].y, p.g[0].z, p.g[0].w, p.g[1].x, p.g[1].y, p.g[1].z, p.g[1].w};
        float E[N], C[N], Z[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            Z[j] = sig_term(pv[j], q.ws[4], q.bs[4], q.a[4]);
            E[j] = sig_term(pv[j], q.ws[1], q.bs[1], q.a[1]);
            C[j] = sig_term(pv[j], q.ws[0], q.bs[0], q.a[0]);
        }
        float Eext = 0.f, Cext = 0.f;
        if (has_left || has_right) {  // wave-uniform; never taken when W <= 512
            Eext = sig_term(p.ext, q.ws[1], q.bs[1], q.a[1]);
            Cext = sig_term(p.ext, q.ws[0], q.bs[0], q.a[0]);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int o = 4 * g;
            float El = __shfl_up(E[o + 3], 1), Cl = __shfl_up(C[o + 3], 1);
            float Er = __shfl_down(E[o], 1), Cr = __shfl_down(C[o], 1);
            if (g == 1) {  // stitch the halves: left neighbour of lane 0 is lane 63 of half 0
                const float e = __shfl(E[3], 63), c = __shfl(C[3], 63);
                if (lane == 0) { El = e; Cl = c; }
            } else {
                const float e = __shfl(E[4], 0), c = __shfl(C[4], 0);
                if (lane == 63) { Er = e; Cr = c; }
                if (lane == 0 && has_left) { El = Eext; Cl = Cext; }
            }
            if (g == 1 && lane == 63 && has_right) { Er = Eext; Cr = Cext; }
            if (px[g] == 0) { El = E[o + 1]; Cl = C[o + 1]; }          // REFLECT: column -1 -> column 1
            if (px[g] + 4 >= W) { Er = E[o + 2]; Cr = C[o + 2]; }      // REFLECT: column W -> column W-2
            M[o + 0] = Z[o + 0] + (El + E[o + 1]);
            M[o + 1] = Z[o + 1] + (E[o + 0] + E[o + 2]);
            M[o + 2] = Z[o + 2] + (E[o + 1] + E[o + 3]);
            M[o + 3] = Z[o + 3] + (E[o + 2] + Er);
            V[o + 0] = E[o + 0] + (Cl + C[o + 1]);
            V[o + 1] = E[o + 1] + (C[o + 0] + C[o + 2]);
            V[o + 2] = E[o + 2] + (C[o + 1] + C[o + 3]);
            V[o + 3] = E[o + 3] + (C[o + 2] + Cr);
        }
    };
    auto store_row = [&](int orow, const float (&v)[N]) {
        if (orow < H) {
            float* o = out + orow * W;
            if (act[0]) *reinterpret_cast<float4*>(o + px[0]) = make_float4(v[0], v[1], v[2], v[3]);
            if (act[1]) *reinterpret_cast<float4*>(o + px[1]) = make_float4(v[4], v[5], v[6], v[7]);
        }
    };

    Raw praw[R], pedge;
#pragma unroll
    for (int t = 0; t < R; ++t) load_row(a + t, praw[t]);
    const bool top_wave = wv == 0, bot_wave = wv == NW - 1;
    if (top_wave) load_row(a - 1, pedge);
    else if (bot_wave) load_row(a + R, pedge);

    float first_part[N], cur[N], carry[N];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        float V[N], M[N];
        terms(praw[t], V, M);
        if (t == 0) {
            xch_first[wv][0][lane] = make_float4(V[0], V[1], V[2], V[3]);
            xch_first[wv][1][lane] = make_float4(V[4], V[5], V[6], V[7]);
#pragma unroll
            for (int j = 0; j < N; ++j) { first_part[j] = M[j]; carry[j] = V[j]; }
        } else {
            if (t == 1) {
#pragma unroll
                for (int j = 0; j < N; ++j) first_part[j] += V[j];
            } else {
                float o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = cur[j] + V[j];
                store_row(a + t - 1, o);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) { cur[j] = carry[j] + M[j]; carry[j] = V[j]; }
        }
    }
    xch_last[wv][0][lane] = make_float4(carry[0], carry[1], carry[2], carry[3]);
    xch_last[wv][1][lane] = make_float4(carry[4], carry[5], carry[6], carry[7]);

    float above[N], below[N];
    if (top_wave || bot_wave) {  // wave-uniform
        float V[N], M[N];
        terms(pedge, V, M);
#pragma unroll
        for (int j = 0; j < N; ++j) { above[j] = V[j]; below[j] = V[j]; }
    }
    __syncthreads();
    if (!top_wave) {
        const float4 u0 = xch_last[wv - 1][0][lane], u1 = xch_last[wv - 1][1][lane];
        above[0] = u0.x; above[1] = u0.y; above[2] = u0.z; above[3] = u0.w;
        above[4] = u1.x; above[5] = u1.y; above[6] = u1.z; above[7] = u1.w;
    }
    if (!bot_wave) {
        const float4 d0 = xch_first[wv + 1][0][lane], d1 = xch_first[wv + 1][1][lane];
        below[0] = d0.x; below[1] = d0.y; below[2] = d0.z; below[3] = d0.w;
        below[4] = d1.x; below[5] = d1.y; below[6] = d1.z; below[7] = d1.w;
    }
    float o0[N], o1[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { o0[j] = first_part[j] + above[j]; o1[j] = cur[j] + below[j]; }
    store_row(a, o0);
    store_row(a + R - 1, o1);
}

// k3_roll<RW, NW>  (D4-symmetric maps, depth 2): the software-pipelined form of k3_tile.  A wave owns 512 columns
// (8 pixels per lane) x RW output rows and WALKS DOWN them: for every input row i it evaluates the separable terms
// V_i, M_i once (output row r = V_{r-1} + M_r + V_{r+1}), keeping acc_r = V_{r-1} + M_r and V_r in registers, while
// the loads of rows i+1 and i+2 are already in flight and finished rows are being stored.  No LDS exchange and no
// barrier: a wave's load -> compute -> store chain overlaps with itself instead of relying on other waves.  The
// arithmetic around the six transcendentals per pixel is packed (v_pk_fma/add/mul_f32, two pixels per instruction).
template <int RW, int NW>
__global__ __launch_bounds__(NW * 64) void k3_roll(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                   int depth, const float* __restrict__ params) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    constexpr int NP = 4;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.x;
    const size_t img_off = (size_t)blockIdx.z * (size_t)(H * W);
    const K3Params q = load_k3<MODE_SYM>(params, depth);
    const int base = seg * 512;
    const int px[2] = {base + lane * 4, base + 256 + lane * 4};
    const bool act[2] = {px[0] < W, px[1] < W};
    const bool has_left = seg > 0, has_right = base + 512 < W;   // wave-uniform
    const float* img = x + img_off;
    float* out = y + img_off;
    const int a = (blockIdx.y * NW + wv) * RW;                   // first output row of this wave
    if (a >= H) return;
    const int nrows = a + RW <= H ? RW : H - a;

    struct Raw { float4 g[2]; float ext; };
    auto load_row = [&](int r, Raw& p) {
        int rr = reflect_idx(r, H);
        rr = rr < 0 ? 0 : (rr >= H ? H - 1 : rr);
        const float* row = img + rr * W;
        p.g[0] = p.g[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        p.ext = 0.f;
        if (act[0]) p.g[0] = *reinterpret_cast<const float4*>(row + px[0]);
        if (act[1]) p.g[1] = *reinterpret_cast<const float4*>(row + px[1]);
        if (has_left && lane == 0) p.ext = row[base - 1];
        if (has_right && lane == 63) p.ext = row[base + 512];
    };
    auto sig2 = [](f2 p, float ws, float bs, float aa) -> f2 {
        const f2 arg = __builtin_elementwise_fma(p, (f2){ws, ws}, (f2){bs, bs});
        const f2 e = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
        const f2 d = e + 1.0f;
        const f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        return r * aa;
    };
    auto terms = [&](const Raw& p, f2 (&V)[NP], f2 (&M)[NP]) {
        const f2 pv[NP] = {{p.g[0].x, p.g[0].y}, {p.g[0].z, p.g[0].w}, {p.g[1].x, p.g[1].y}, {p.g[1].z, p.g[1].w}};
        f2 E[NP], C[NP], Z[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            Z[j] = sig2(pv[j], q.ws[4], q.bs[4], q.a[4]);
            E[j] = sig2(pv[j], q.ws[1], q.bs[1], q.a[1]);
            C[j] = sig2(pv[j], q.ws[0], q.bs[0], q.a[0]);
        }
        float Eext = 0.f, Cext = 0.f;
        if (has_left || has_right) {
            Eext = sig_term(p.ext, q.ws[1], q.bs[1], q.a[1]);
            Cext = sig_term(p.ext, q.ws[0], q.bs[0], q.a[0]);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int o = 2 * g;
            float El = __shfl_up(E[o + 1].y, 1), Cl = __shfl_up(C[o + 1].y, 1);
            float Er = __shfl_down(E[o].x, 1), Cr = __shfl_down(C[o].x, 1);
            if (g == 1) {
                const float e = __shfl(E[1].y, 63), c = __shfl(C[1].y, 63);
                if (lane == 0) { El = e; Cl = c; }
            } else {
                const float e = __shfl(E[2].x, 0), c = __shfl(C[2].x, 0);
                if (lane == 63) { Er = e; Cr = c; }
                if (lane == 0 && has_left) { El = Eext; Cl = Cext; }
            }
            if (g == 1 && lane == 63 && has_right) { Er = Eext; Cr = Cext; }
            if (px[g] == 0) { El = E[o].y; Cl = C[o].y; }
            if (px[g] + 4 >= W) { Er = E[o + 1].x; Cr = C[o + 1].x; }
            const f2 eL = {El, E[o].x}, eM = {E[o].y, E[o + 1].x}, eR = {E[o + 1].y, Er};
            const f2 cL = {Cl, C[o].x}, cM = {C[o].y, C[o + 1].x}, cR = {C[o + 1].y, Cr};
            M[o] = Z[o] + (eL + eM);
            M[o + 1] = Z[o + 1] + (eM + eR);
            V[o] = E[o] + (cL + cM);
            V[o + 1] = E[o + 1] + (cM + cR);
        }
    };
    auto store_row = [&](int orow, const f2 (&v)[NP]) {
        float* o = out + orow * W;
        if (act[0]) *reinterpret_cast<float4*>(o + px[0]) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
        if (act[1]) *reinterpret_cast<float4*>(o + px[1]) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
    };

    // input rows a-1 .. a+nrows; three raw-row buffers rotate, loads run two rows ahead of the arithmetic
    Raw r0, r1, r2;
    load_row(a - 1, r0);
    load_row(a, r1);
    load_row(a + 1, r2);
    f2 acc[NP], carry[NP], V[NP], M[NP];
    terms(r0, V, M);                                  // row a-1: only V is needed
#pragma unroll
    for (int j = 0; j < NP; ++j) carry[j] = V[j];
    load_row(a + 2, r0);
    terms(r1, V, M);                                  // row a
    load_row(a + 3, r1);
#pragma unroll
    for (int j = 0; j < NP; ++j) { acc[j] = carry[j] + M[j]; carry[j] = V[j]; }
    // steady state, three rows per trip so that the buffer rotation is static
    int i = a + 1;                                    // next input row to evaluate; it sits in r2, then r0, then r1
    const int last = a + nrows;                       // last input row (the halo below)
#define EMD_K3_STEP(RAWCUR, RAWNEXT3)                                              \
    if (i <= last) {                                                               \
        terms(RAWCUR, V, M);                                                       \
        if (i + 3 <= last) load_row(i + 3, RAWNEXT3);                              \
        f2 o[NP];                                                                  \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) o[j] = acc[j] + V[j];       \
        store_row(i - 1, o);                                                       \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) { acc[j] = carry[j] + M[j]; carry[j] = V[j]; } \
        ++i;                                                                       \
    }
    for (int trip = 0; trip < (RW + 3) / 3; ++trip) {
        EMD_K3_STEP(r2, r2)   // evaluates row i (in r2), then refills r2 with row i+3
        EMD_K3_STEP(r0, r0)
        EMD_K3_STEP(r1, r1)
    }
#undef EMD_K3_STEP
}

template <int RW, int NW>
int launch_k3_roll(const float* x, float* y, int B, int H, int W, int depth, const float* params, hipStream_t st) {
    const int nseg = (W + 511) / 512;
    const int nstrip = (H + NW * RW - 1) / (NW * RW);
    if ((long)H * W >= 0x7fffffffL || nstrip > 65535)
        return emd::fail(EMD_E_UNSUPPORTED, "emd_kernel_denoise_f32: image too large for the tiled kernel");
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        const size_t off = (size_t)b0 * H * W;
        hipLaunchKernelGGL((k3_roll<RW, NW>), dim3(nseg, nstrip, nb), dim3(NW * 64), 0, st, x + off, y + off, H, W, depth, params);
    }
    return emd::check_launch("k3_roll");
}

template <int R, int NW>
int launch_k3_tile(const float* x, float* y, int B, int H, int W, int depth, const float* params,
                   hipStream_t st) {
    const int nseg = (W + 511) / 512;
    const int nstrip = (H + NW * R - 1) / (NW * R);
    if ((long)H * W >= 0x7fffffffL || nstrip > 65535)
        return emd::fail(EMD_E_UNSUPPORTED, "emd_kernel_denoise_f32: image too large for the tiled kernel");
    for (int b0 = 0; b0 < B; b0 += 65535) {  // gridDim.z limit
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        const size_t off = (size_t)b0 * H * W;
        hipLaunchKernelGGL((k3_tile<R, NW>), dim3(nseg, nstrip, nb), dim3(NW * 64), 0, st, x + off, y + off, H, W,
                           depth, params);
    }
    return emd::check_launch("k3_tile");
}

}  // namespace

extern "C" size_t emd_kernel_params_count(int width, int depth) {
    if (width < 1 || depth < 1) return 0;
    return (size_t)2 * depth * width * width + depth;
}

extern "C" int emd_kernel_denoise_f32(const float* x, float* y, int B, int H, int W, int width,
                                      int depth, const float* params, unsigned flags,
                                      emd_stream_t stream) {
    EMD_REQUIRE(x && y && params, EMD_E_INVALID, "emd_kernel_denoise_f32: null pointer");
    EMD_REQUIRE(x != y, EMD_E_INVALID, "emd_kernel_denoise_f32: y must not alias x");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_kernel_denoise_f32: bad shape");
    EMD_REQUIRE(width >= 1 && (width & 1) && width <= EMD_K_MAX_WIDTH, EMD_E_INVALID,
                "emd_kernel_denoise_f32: width must be odd and <= 15");
    EMD_REQUIRE(depth >= 1 && depth <= EMD_K_MAX_DEPTH, EMD_E_INVALID,
                "emd_kernel_denoise_f32: depth must be 1..5");
    EMD_REQUIRE(width / 2 < H && width / 2 < W, EMD_E_INVALID,
                "emd_kernel_denoise_f32: REFLECT padding needs width/2 < min(H,W)");
    EMD_REQUIRE((flags & ~EMD_K_SYMMETRIC) == 0, EMD_E_INVALID, "emd_kernel_denoise_f32: unknown flag");
    if (B == 0) return EMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);

    const bool fast = width == 3 && depth <= 2 && (W % kPx) == 0 && W >= 16 && H >= 2 &&
                      emd::aligned16(x) && emd::aligned16(y);
    if (fast) {
        constexpr int R = 8;
        if (depth == 1) return launch_k3<MODE_LIN, R>(x, y, B, H, W, depth, params, st);
        if (flags & EMD_K_SYMMETRIC) {
#ifndef EMD_K_RW
#define EMD_K_RW 8
#endif
#ifndef EMD_K_NW
#define EMD_K_NW 2
#endif
#ifdef EMD_K_TILE
            return launch_k3_tile<2, 8>(x, y, B, H, W, depth, params, st);
#else
            if (H >= 3) return launch_k3_roll<EMD_K_RW, EMD_K_NW>(x, y, B, H, W, depth, params, st);
            return launch_k3_tile<2, 8>(x, y, B, H, W, depth, params, st);
#endif
        }
        return launch_k3<MODE_GEN, R>(x, y, B, H, W, depth, params, st);
    }
    const long total = (long)B * H * W;
    long nblk = (total + 255) / 256;
    if (nblk > 256L * 32) nblk = 256L * 32;  // grid-stride the rest
    hipLaunchKernelGGL(k_generic, dim3((unsigned)nblk), dim3(256), 0, st, x, y, B, H, W, width, depth,
                       params);
    return emd::check_launch("k_generic");
}
