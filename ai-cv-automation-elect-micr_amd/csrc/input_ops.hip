// input_ops.hip -- the reference's host-side training input functions on the device:
//   misc_py/denoiser-multi-gpu.py:783-784  get_scale      25 + Exp(mean 75)                -> emd_get_scale_f32
//   :787-799                               gen_lq         scale0to1(Poisson(img * scale))   -> emd_gen_lq_f32
//   :817-828                               scale0to1      min-max, constant image -> 0.5   -> emd_minmax_images_f32 / emd_scale0to1_images_f32
//   :830-851                               flip_rotate    one of the 8 elements of D4      -> emd_flip_rotate_f32 (+ emd_d4_choices_i32)
//   :853-858                               preprocess     NaN/Inf -> 0.5, D4, min-max      -> emd_flip_rotate_f32(fix_nonfinite=1) + scale0to1
//   :861-870                               record_parser  truth = (mean(lq)/mean(img))*img -> emd_gen_lq_f32 (truth output)
// The reference draws from numpy's global Mersenne-Twister re-seeded from itself (:791): its stream is not
// reproducible by construction, so what is kept is every DISTRIBUTION and every deterministic formula.  Random numbers
// here come from Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; counter-based, no
// state): key = (seed lo, seed hi), counter = (index lo, index hi, draw number, stream tag), so a value depends only on
// (seed, image, pixel, draw) and never on the launch geometry.
// All of it is bandwidth-trivial (8 x 512^2 pixels per training step); the kernels are written for exactness first:
// min / max / integer sums are order-independent, the Poisson sampler runs in double precision.
#include <cmath>

#include "emd_common.hpp"

namespace {

constexpr unsigned kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u, kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;
// stream tags (counter word 3): independent sequences under one seed
constexpr unsigned kTagRaw = 0u, kTagScale = 1u, kTagChoice = 2u, kTagPoisson = 3u;

struct U4 {
    unsigned x, y, z, w;
};

__host__ __device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)kPhiloxM0 * c.x, p1 = (unsigned long long)kPhiloxM1 * c.z;
        const U4 n = {(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    return c;
}

// uniform double in (0, 1): the top 52 bits of two words, (m + 1/2) / 2^52 -- exactly representable, never 0 and never 1
__device__ inline double u01(unsigned hi, unsigned lo) {
    const unsigned long long m = (((unsigned long long)hi << 32) | lo) >> 12;   // 52 bits
    return ((double)m + 0.5) * (1.0 / 4503599627370496.0);
}

__global__ void philox_raw_kernel(unsigned* __restrict__ out, long n4, unsigned long long seed, unsigned long long counter0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const unsigned long long c = counter0 + (unsigned long long)i;
    const U4 r = philox4x32_10(U4{(unsigned)c, (unsigned)(c >> 32), 0u, kTagRaw}, (unsigned)seed, (unsigned)(seed >> 32));
    out[4 * i + 0] = r.x;
    out[4 * i + 1] = r.y;
    out[4 * i + 2] = r.z;
    out[4 * i + 3] = r.w;
}

// get_scale (:783-784): 25 + Exp(mean 75) = 25 - 75 ln(u);  choice (:833): int(8 * u)
__global__ void scale_choice_kernel(float* __restrict__ scale, int* __restrict__ choice, int B, unsigned long long seed,
                                    unsigned long long first_image) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long img = first_image + (unsigned long long)b;
    if (scale) {
        const U4 r = philox4x32_10(U4{(unsigned)img, (unsigned)(img >> 32), 0u, kTagScale}, (unsigned)seed, (unsigned)(seed >> 32));
        scale[b] = (float)(25.0 - 75.0 * log(u01(r.x, r.y)));
    }
    if (choice) {
        const U4 r = philox4x32_10(U4{(unsigned)img, (unsigned)(img >> 32), 0u, kTagChoice}, (unsigned)seed, (unsigned)(seed >> 32));
        choice[b] = (int)(r.x >> 29);   // floor(8 u), u = r.x / 2^32
    }
}

// ---- flip_rotate (:830-851) on square images, choice per image from device memory.  With n = H = W:
//   0 identity            out[i][j] = in[i][j]            4 flip axis 0        out[i][j] = in[n-1-i][j]
//   1 rot90 (ccw)         out[i][j] = in[j][n-1-i]        5 flip axis 1        out[i][j] = in[i][n-1-j]
//   2 rot180              out[i][j] = in[n-1-i][n-1-j]    6 flip(rot90, 0)     out[i][j] = in[j][i]
//   3 rot270              out[i][j] = in[n-1-j][i]        7 flip(rot90, 1)     out[i][j] = in[n-1-j][n-1-i]
// Choices 1, 3, 6, 7 transpose: the 64 x 64 tile goes through LDS so that both the global reads (along the source's
// rows) and the global writes (along the destination's rows) are lane-contiguous.
constexpr int kT = 64;
__global__ void __launch_bounds__(256) flip_rotate_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                         const int* __restrict__ choice, int fix_nonfinite) {
    __shared__ float tile[kT][kT + 1];
    const int b = blockIdx.z;
    const int ch = choice ? (choice[b] & 7) : 0;
    const bool transposing = ch == 1 || ch == 3 || ch == 6 || ch == 7;
    const float* xi = x + (long)b * n * n;
    float* yo = y + (long)b * n * n;
    const int oi0 = blockIdx.y * kT, oj0 = blockIdx.x * kT;   // output tile origin
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;    // 64 x 4
    // source tile origin and orientation: out[oi][oj] = in[si][sj]
    //   non-transposing: si = fi(oi), sj = fj(oj);  transposing: si = fi(oj), sj = fj(oi)
    const bool flip_i = (ch == 2 || ch == 4 || ch == 3 || ch == 7);   // source row index runs backwards
    const bool flip_j = (ch == 2 || ch == 5 || ch == 1 || ch == 7);   // source column index runs backwards
    // rows of the source tile: r in [0,64) <-> (transposing ? output column oj0 + r : output row oi0 + r)
    const int ro0 = transposing ? oj0 : oi0, co0 = transposing ? oi0 : oj0;
#pragma unroll 4
    for (int r = ty; r < kT; r += 4) {
        const int ro = ro0 + r, co = co0 + tx;   // the output-side indices this source element maps to
        float v = 0.f;
        if (ro < n && co < n) {
            const int si = flip_i ? n - 1 - ro : ro, sj = flip_j ? n - 1 - co : co;
            v = xi[(long)si * n + sj];
            if (fix_nonfinite && !(fabsf(v) <= 3.402823466e38f)) v = 0.5f;   // NaN or +-Inf -> 0.5 (:855-856)
        }
        tile[r][tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < kT; r += 4) {
        const int oi = oi0 + r, oj = oj0 + tx;
        if (oi < n && oj < n) yo[(long)oi * n + oj] = transposing ? tile[tx][r] : tile[r][tx];
    }
}

// ---- per-image min / max (scale0to1, :817-828).  min and max are exact and order-independent.
__device__ inline float wave_min(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// grid (slabs, B); partial[b][slab] = {min, max}; NaNs are ignored by fminf/fmaxf (the reference replaces them first)
__global__ void __launch_bounds__(256) minmax_partial_kernel(const float* __restrict__ x, long npix, float2* __restrict__ part) {
    __shared__ float smn[4], smx[4];
    const int b = blockIdx.y, nslab = gridDim.x;
    const float* xi = x + (long)b * npix;
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)nslab * 256) {
        const float v = xi[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        part[(long)b * nslab + blockIdx.x] = make_float2(fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3])),
                                                         fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3])));
}
__global__ void minmax_final_kernel(const float2* __restrict__ part, int nslab, float* __restrict__ mn_out, float* __restrict__ mx_out) {
    const int b = blockIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int s = threadIdx.x; s < nslab; s += 64) {
        const float2 p = part[(long)b * nslab + s];
        mn = fminf(mn, p.x);
        mx = fmaxf(mx, p.y);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (threadIdx.x == 0) mn_out[b] = mn, mx_out[b] = mx;
}

// y = (x - min) / (max - min) in float32 (what numpy computes for a float32 image, correctly rounded subtract and divide);
// min == max -> 0.5 (:823-824)
__global__ void scale0to1_kernel(const float* __restrict__ x, float* __restrict__ y, long npix, const float* __restrict__ mn,
                                 const float* __restrict__ mx) {
    const int b = blockIdx.y;
    const float lo = mn[b], hi = mx[b];
    const float d = hi - lo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const long k = (long)b * npix + i;
        y[k] = (lo == hi) ? 0.5f : (x[k] - lo) / d;
    }
}

// ---- Poisson(lambda), exact samplers in double precision.
// lambda < 10: inversion by sequential search of the CDF with one uniform.  lambda >= 10: PTRS, the transformed
// rejection method of W. Hoermann, "The transformed rejection method for generating Poisson random variables",
// Insurance: Mathematics and Economics 12 (1993) 39-45 -- the algorithm numpy's Generator.poisson uses above 10 too.
__device__ inline long poisson_small(double lam, double u) {
    double p = exp(-lam), s = p;
    long k = 0;
    while (u > s && k < 200) {   // P(k >= 200 | lambda < 10) < 1e-180; the bound also ends the loop when s saturates below u
        ++k;
        p *= lam / (double)k;
        s += p;
    }
    return k;
}

__device__ inline long poisson_ptrs(double lam, unsigned long long pix, unsigned long long img, unsigned k0, unsigned k1) {
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (unsigned attempt = 0; attempt < 64; ++attempt) {   // acceptance > 0.7 per attempt: 64 fail with probability < 1e-33
        // counter = (pixel, image low word, draw | image high bits << 8, tag), draw = 1 + attempt (0 is the small-lambda draw); the key
        // is the seed alone.  (Until round 3 the image index was XORed into the key: seed 0 / image 1 and seed 1 / image 0 then drew
        // the same noise field -- a per-rank seed scheme would have given ranks identical Poisson noise.)
        const U4 r = philox4x32_10(U4{(unsigned)pix, (unsigned)img, (1u + attempt) | ((unsigned)(img >> 32) << 8), kTagPoisson}, k0, k1);
        const double U = u01(r.x, r.y) - 0.5, V = u01(r.z, r.w);
        const double us = 0.5 - fabs(U);
        const long k = (long)floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + (double)k * loglam - lgamma((double)k + 1.0)) return k;
    }
    return (long)floor(lam + 0.5);
}

// counts[b][i] = Poisson(img[b][i] * scale[b]) as int32; per-(image, slab) {min count, max count} and {sum counts (int64, exact),
// sum img (double)} partials.  A negative or non-finite rate draws 0 (numpy raises ValueError for lam < 0; preprocess() has
// already mapped the image into [0,1]).
__global__ void __launch_bounds__(256) poisson_counts_kernel(const float* __restrict__ img, const float* __restrict__ scale, long npix,
                                                            unsigned long long seed, unsigned long long first_image,
                                                            int* __restrict__ counts, int2* __restrict__ part_mm,
                                                            double2* __restrict__ part_sum) {
    __shared__ int smn[4], smx[4];
    __shared__ double ssc[4], ssi[4];
    const int b = blockIdx.y, nslab = gridDim.x;
    const unsigned long long gimg = first_image + (unsigned long long)b;
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    const double sc = (double)scale[b];
    int mn = 0x7fffffff, mx = -0x7fffffff - 1;
    long long sumc = 0;
    double sumi = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)nslab * 256) {
        const float v = img[(long)b * npix + i];
        const double lam = (double)v * sc;   // the rate in double (the reference's NumPy 1.x keeps float32 image * np.float64 scalar in float32:
                                             // the last bits of a rate of ~100 counts, statistically irrelevant -- the sampler is not bit-matched anyway)
        long k = 0;
        if (lam > 0.0 && lam < 1e9) {
            if (lam < 10.0) {
                const U4 r = philox4x32_10(U4{(unsigned)i, (unsigned)gimg, (unsigned)(gimg >> 32) << 8, kTagPoisson}, k0, k1);
                k = poisson_small(lam, u01(r.x, r.y));
            } else {
                k = poisson_ptrs(lam, (unsigned long long)i, gimg, k0, k1);
            }
        }
        const int c = (int)k;
        counts[(long)b * npix + i] = c;
        mn = min(mn, c);
        mx = max(mx, c);
        sumc += c;
        sumi += (double)v;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o));
        mx = max(mx, __shfl_xor(mx, o));
        sumc += __shfl_xor(sumc, o);
        sumi += __shfl_xor(sumi, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) smn[w] = mn, smx[w] = mx, ssc[w] = (double)sumc, ssi[w] = sumi;
    __syncthreads();
    if (threadIdx.x == 0) {
        part_mm[(long)b * nslab + blockIdx.x] = make_int2(min(min(smn[0], smn[1]), min(smn[2], smn[3])), max(max(smx[0], smx[1]), max(smx[2], smx[3])));
        part_sum[(long)b * nslab + blockIdx.x] = make_double2(ssc[0] + ssc[1] + ssc[2] + ssc[3], (ssi[0] + ssi[1]) + (ssi[2] + ssi[3]));
    }
}

// one wave per image: stats[b] = {min count, max count, mean(lq) / mean(img)} (fixed-order reduction: deterministic)
__global__ void gen_lq_final_kernel(const int2* __restrict__ part_mm, const double2* __restrict__ part_sum, int nslab, long npix,
                                    double* __restrict__ stats) {
    const int b = blockIdx.x;
    int mn = 0x7fffffff, mx = -0x7fffffff - 1;
    double sc = 0.0, si = 0.0;
    for (int s = threadIdx.x; s < nslab; s += 64) {
        const int2 p = part_mm[(long)b * nslab + s];
        const double2 q = part_sum[(long)b * nslab + s];
        mn = min(mn, p.x);
        mx = max(mx, p.y);
        sc += q.x;
        si += q.y;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o));
        mx = max(mx, __shfl_xor(mx, o));
        sc += __shfl_xor(sc, o);
        si += __shfl_xor(si, o);
    }
    if (threadIdx.x == 0) {
        // mean(lq) with lq = (c - mn) / (mx - mn):  (sum c / n - mn) / (mx - mn).  Constant counts: the reference's scale0to1 receives
        // the INT64 Poisson array there and ndarray.fill(0.5) on int64 stores 0 (denoiser-multi-gpu.py:797, :824): lq = 0, truth = 0 * img
        const double n = (double)npix;
        const double mean_lq = (mx == mn) ? 0.0 : (sc / n - (double)mn) / ((double)mx - (double)mn);
        const double mean_img = si / n;
        stats[3 * b + 0] = (double)mn;
        stats[3 * b + 1] = (double)mx;
        stats[3 * b + 2] = mean_lq / mean_img;
    }
}

// lq = float32((c - min) / (max - min)) computed in float64 (numpy: int64 counts -> float64 true division -> astype(float32));
// truth = float32(ratio) * img (:868: float32 scalar times float32 image)
__global__ void gen_lq_apply_kernel(const int* __restrict__ counts, const float* __restrict__ img, long npix,
                                    const double* __restrict__ stats, float* __restrict__ lq, float* __restrict__ truth) {
    const int b = blockIdx.y;
    const double mn = stats[3 * b], mx = stats[3 * b + 1];
    const float ratio = (float)stats[3 * b + 2];
    const double d = mx - mn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const long k = (long)b * npix + i;
        lq[k] = (mx == mn) ? 0.f : (float)(((double)counts[k] - mn) / d);   // constant counts: 0, not 0.5 (int64 fill, see above)
        if (truth) truth[k] = ratio * img[k];
    }
}

int slabs_for(long npix) {
    long s = (npix + 256 * 16 - 1) / (256 * 16);   // >= 16 pixels per thread
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

}  // namespace

extern "C" int emd_philox4x32_u32(unsigned* out, long n4, unsigned long long seed, unsigned long long counter0, emd_stream_t stream) {
    EMD_REQUIRE(n4 >= 0, EMD_E_INVALID, "emd_philox4x32_u32: negative count");
    if (n4 == 0) return EMD_OK;
    EMD_REQUIRE(out, EMD_E_INVALID, "emd_philox4x32_u32: null pointer");
    hipLaunchKernelGGL(philox_raw_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), out, n4,
                       seed, counter0);
    return emd::check_launch("philox_raw_kernel");
}

extern "C" int emd_get_scale_f32(float* scale, int B, unsigned long long seed, unsigned long long first_image, emd_stream_t stream) {
    EMD_REQUIRE(B >= 0, EMD_E_INVALID, "emd_get_scale_f32: negative batch");
    if (B == 0) return EMD_OK;
    EMD_REQUIRE(scale, EMD_E_INVALID, "emd_get_scale_f32: null pointer");
    hipLaunchKernelGGL(scale_choice_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), scale,
                       static_cast<int*>(nullptr), B, seed, first_image);
    return emd::check_launch("scale_choice_kernel");
}

extern "C" int emd_d4_choices_i32(int* choice, int B, unsigned long long seed, unsigned long long first_image, emd_stream_t stream) {
    EMD_REQUIRE(B >= 0, EMD_E_INVALID, "emd_d4_choices_i32: negative batch");
    if (B == 0) return EMD_OK;
    EMD_REQUIRE(choice, EMD_E_INVALID, "emd_d4_choices_i32: null pointer");
    hipLaunchKernelGGL(scale_choice_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<float*>(nullptr), choice, B, seed, first_image);
    return emd::check_launch("scale_choice_kernel");
}

extern "C" int emd_flip_rotate_f32(const float* x, float* y, int B, int H, int W, const int* choice_dev, int fix_nonfinite,
                                   emd_stream_t stream) {
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_flip_rotate_f32: bad shape");
    if (B == 0) return EMD_OK;
    EMD_REQUIRE(x && y && x != y, EMD_E_INVALID, "emd_flip_rotate_f32: null or aliased pointer");
    EMD_REQUIRE(H == W, EMD_E_UNSUPPORTED, "emd_flip_rotate_f32: square images only (a batch keeps one shape under rot90)");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_flip_rotate_f32: batch > 65535");
    const unsigned t = (unsigned)((H + kT - 1) / kT);
    hipLaunchKernelGGL(flip_rotate_kernel, dim3(t, t, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, H, choice_dev,
                       fix_nonfinite);
    return emd::check_launch("flip_rotate_kernel");
}

extern "C" size_t emd_input_workspace_bytes(int B, long npix) {
    if (B <= 0 || npix <= 0) return 0;
    const size_t nslab = (size_t)slabs_for(npix);
    // counts (int32) | {min,max} partials | {sum,sum} partials | stats (3 doubles per image); each block 16-byte aligned
    const size_t counts = ((size_t)B * (size_t)npix * 4 + 15) & ~(size_t)15;
    return counts + (size_t)B * nslab * (8 + 16) + (size_t)B * 3 * 8 + 64;
}

extern "C" int emd_minmax_images_f32(const float* x, int B, long npix, float* mn, float* mx, void* workspace, emd_stream_t stream) {
    EMD_REQUIRE(B >= 0 && npix >= 1, EMD_E_INVALID, "emd_minmax_images_f32: bad shape");
    if (B == 0) return EMD_OK;
    EMD_REQUIRE(x && mn && mx && workspace, EMD_E_INVALID, "emd_minmax_images_f32: null pointer");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_minmax_images_f32: batch > 65535");
    const int nslab = slabs_for(npix);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float2* part = static_cast<float2*>(workspace);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nslab, B), dim3(256), 0, st, x, npix, part);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(B), dim3(64), 0, st, static_cast<const float2*>(part), nslab, mn, mx);
    return emd::check_launch("minmax_images");
}

extern "C" int emd_scale0to1_images_f32(const float* x, float* y, int B, long npix, const float* mn, const float* mx,
                                        emd_stream_t stream) {
    EMD_REQUIRE(B >= 0 && npix >= 1, EMD_E_INVALID, "emd_scale0to1_images_f32: bad shape");
    if (B == 0) return EMD_OK;
    EMD_REQUIRE(x && y && mn && mx, EMD_E_INVALID, "emd_scale0to1_images_f32: null pointer");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_scale0to1_images_f32: batch > 65535");
    const unsigned gx = (unsigned)((npix + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(scale0to1_kernel, dim3(gx > 1024 ? 1024 : gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, npix, mn,
                       mx);
    return emd::check_launch("scale0to1_kernel");
}

extern "C" int emd_gen_lq_f32(const float* img, const float* scale, float* lq, float* truth, int* counts_out, int B, long npix,
                              unsigned long long seed, unsigned long long first_image, void* workspace, emd_stream_t stream) {
    EMD_REQUIRE(B >= 0 && npix >= 1, EMD_E_INVALID, "emd_gen_lq_f32: bad shape");
    if (B == 0) return EMD_OK;
    EMD_REQUIRE(img && scale && lq && workspace, EMD_E_INVALID, "emd_gen_lq_f32: null pointer");
    EMD_REQUIRE(lq != img && truth != img, EMD_E_INVALID, "emd_gen_lq_f32: outputs alias the input");
    EMD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, EMD_E_ALIGN, "emd_gen_lq_f32: workspace must be 16-byte aligned");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_gen_lq_f32: batch > 65535");
    const int nslab = slabs_for(npix);
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const size_t counts_bytes = ((size_t)B * (size_t)npix * 4 + 15) & ~(size_t)15;
    int* counts = counts_out ? counts_out : reinterpret_cast<int*>(ws);
    double2* part_sum = reinterpret_cast<double2*>(ws + counts_bytes);
    double* stats = reinterpret_cast<double*>(ws + counts_bytes + (size_t)B * nslab * 16);
    int2* part_mm = reinterpret_cast<int2*>(ws + counts_bytes + (size_t)B * nslab * 16 + (((size_t)B * 3 * 8 + 15) & ~(size_t)15));
    hipLaunchKernelGGL(poisson_counts_kernel, dim3(nslab, B), dim3(256), 0, st, img, scale, npix, seed, first_image, counts, part_mm,
                       part_sum);
    hipLaunchKernelGGL(gen_lq_final_kernel, dim3(B), dim3(64), 0, st, static_cast<const int2*>(part_mm),
                       static_cast<const double2*>(part_sum), nslab, npix, stats);
    const unsigned gx = (unsigned)((npix + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(gen_lq_apply_kernel, dim3(gx > 1024 ? 1024 : gx, B), dim3(256), 0, st, static_cast<const int*>(counts), img, npix,
                       static_cast<const double*>(stats), lq, truth);
    return emd::check_launch("gen_lq");
}
