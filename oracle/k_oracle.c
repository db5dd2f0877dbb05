/* Graph K oracle in plain C (oracle; test infrastructure only -- never linked into the product).
 *
 * Restates misc_py/noise-removal-kernels.py:96-431 of the reference:
 *   pad        (:99-105)   tf.pad(mode="REFLECT") by w/2 on H and W
 *   filter_fn  (:378-399)  f = W0*P; for i in 1..d-1: f = Wi*(s_i*sigmoid(f+Bi)); out = sum(f)
 *   pixel loop (:409-417)  one w x w patch of the padded image per output pixel
 * The output is returned un-transposed (see oracle/kernel_denoiser.py header).
 * PARITY UNPINNED except for the box-mean known-answer test (KAT #1).
 *
 * Also used as the timed CPU baseline ("port") by bench.py: OpenMP over image rows,
 * thread count chosen by the caller.
 *
 * Build: see oracle/Makefile  ->  oracle/_build/libk_oracle.so
 */
#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/* x, y: [B,H,W] float32 (the NHWC tensor with C == 1).
 * wmaps, bmaps: [depth][width*width] full maps (bmaps[0] unused); s: [depth] (s[0] unused).
 * Returns 0 on success, -1 on invalid arguments. */
int k_oracle_f32(const float* x, float* y, int B, int H, int W, int width, int depth,
                 const float* wmaps, const float* bmaps, const float* s, int nthreads) {
    if (!x || !y || B < 0 || H < 1 || W < 1 || width < 1 || (width & 1) == 0 || depth < 1) return -1;
    const int p = width / 2;
    if (p >= H || p >= W) return -1; /* REFLECT needs pad < dim */
    const int ww = width * width;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const long rows = (long)B * H;
#pragma omp parallel for schedule(static)
    for (long br = 0; br < rows; ++br) {
        const int b = (int)(br / H), r = (int)(br % H);
        const float* img = x + (size_t)b * H * W;
        float* out = y + ((size_t)b * H + r) * W;
        for (int c = 0; c < W; ++c) {
            float acc = 0.f;
            for (int i = 0; i < width; ++i) {
                const int rr = reflect_idx(r + i - p, H);
                for (int j = 0; j < width; ++j) {
                    const int cc = reflect_idx(c + j - p, W);
                    float f = wmaps[i * width + j] * img[(size_t)rr * W + cc];
                    for (int l = 1; l < depth; ++l) {
                        const float z = f + bmaps[l * ww + i * width + j];
                        const float sg = 1.0f / (1.0f + expf(-z));
                        f = wmaps[l * ww + i * width + j] * (s[l] * sg);
                    }
                    acc += f;
                }
            }
            out[c] = acc;
        }
    }
    return 0;
}

int k_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
