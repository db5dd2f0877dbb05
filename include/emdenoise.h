/* emdenoise.h -- C ABI of libemdenoise.so: MI355X (gfx950) kernels for the micrograph-denoising
 * hot path of Jeffrey-Ede/AI-CV-Automation-Elect-Micr.
 *
 * The reference has no FFI / plugin interface for this path: its arithmetic is a graph of stock
 * TensorFlow ops built by Python.  Each entry point below therefore replaces the
 * TensorFlow op call(s) cited next to it ("replaces: file:line"), and is bound from Python with
 * ctypes exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types.
 *   - Every data pointer is a DEVICE pointer unless its name ends in _host.
 *   - Activations are NHWC float32 (reference: data_format='NHWC', machine_learning/denoiser.py:120;
 *     placeholders tf.float32, :613).  A tensor may be a channel slice of a wider buffer: it is
 *     described by its channel count C and its pixel stride ld (elements between consecutive
 *     pixels, ld >= C), which is how tf.concat (denoiser.py:203, :353, :365) is made free.
 *   - The caller owns every buffer; the library allocates no device memory and keeps no global
 *     mutable state.  Every call takes the hipStream_t to launch on (as void*), is asynchronous
 *     with respect to the host and is safe to capture into a hipGraph.
 *   - Return value: EMD_OK (0) or a negative EMD_E_* code; emd_last_error() returns a
 *     thread-local description of the last failure on the calling thread.  Nothing throws
 *     across the ABI.
 */
#ifndef EMDENOISE_H
#define EMDENOISE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMD_VERSION 100 /* 0.1.0 */

#define EMD_OK 0
#define EMD_E_INVALID (-1)     /* bad argument (null pointer, non-positive size, bad enum) */
#define EMD_E_UNSUPPORTED (-2) /* valid request this build has no kernel for */
#define EMD_E_ALIGN (-3)       /* pointer / stride alignment requirement not met */
#define EMD_E_LAUNCH (-4)      /* HIP reported an error at launch */

typedef void* emd_stream_t; /* hipStream_t */

int emd_version(void);
const char* emd_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Graph K: the learned symmetric-kernel ("dedicated kernel") denoiser.
 * replaces: misc_py/noise-removal-kernels.py:99-105 (tf.pad REFLECT), :378-399 (filter_fn:
 *           W0*P -> [ +Bi -> sigmoid -> fully_connected scalar -> Wi* ] x (depth-1) -> reduce_sum),
 *           :409-426 (the per-pixel Python loop that instantiates filter_fn at every pixel), and the
 *           per-pixel sess.run loop of misc_py/apply_kernels+MLPs.py:669-698.
 *
 * x, y     : [B,H,W] float32 (NHWC with C == 1); y may not alias x.
 * width    : odd kernel width w, 3..EMD_K_MAX_WIDTH; REFLECT padding needs w/2 < min(H,W).
 * depth    : 1..EMD_K_MAX_DEPTH.
 * params   : device float array, emd_kernel_params_count(width, depth) elements:
 *              wmaps [depth][w*w]   full w x w weight maps W0..W(depth-1)
 *              bmaps [depth][w*w]   bias maps (bmaps[0] is ignored)
 *              s     [depth]        fully_connected scalars (s[0] is ignored)
 * flags    : EMD_K_SYMMETRIC asserts that every map is D4-symmetric (as make_layer,
 *            noise-removal-kernels.py:107-358, always builds them); it enables the kernel that
 *            evaluates 3 sigmoids per input pixel instead of 9 per output pixel.  Results are
 *            undefined if the flag is set for maps that are not symmetric.
 * The image is returned un-transposed (the trainer's transposed assembly at :421-424 is undone by
 * the reference itself at :712).
 */
#define EMD_K_MAX_WIDTH 15
#define EMD_K_MAX_DEPTH 5
#define EMD_K_SYMMETRIC 1u

size_t emd_kernel_params_count(int width, int depth);
int emd_kernel_denoise_f32(const float* x, float* y, int B, int H, int W, int width, int depth,
                           const float* params, unsigned flags, emd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EMDENOISE_H */
