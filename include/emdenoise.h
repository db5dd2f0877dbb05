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
#define EMD_E_ALLOC (-5)       /* a device allocation or upload inside the library failed (emd_graph_create only) */

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

/* ------------------------------------------------------------------------------------------------
 * Graph D: the depthwise-separable encoder-decoder (machine_learning/denoiser.py:58-398).
 *
 * Fused epilogue shared by the matrix-core entry points (per output channel n):
 *     v = acc * scale1[n] + shift1[n]          conv bias and the inference batch norm(s) folded
 *     if (act)    v = min(max(v,0),6)          tf.nn.relu6                        (denoiser.py:83)
 *     if (scale2) v = relu6(v*scale2[n]+shift2[n])   a second batch_then_activ  (:170,:176,:182)
 *     if (res)    v += res[pixel][n]           the "+=" residual that follows     (:264 ...)
 * scale*, shift*: device float[Cout]; scale2/shift2/res may be NULL.
 *
 * precision: EMD_PREC_BF16X3 (split-bf16, 3 MFMA passes, ~2^-16 relative: the parity mode) or
 *            EMD_PREC_BF16   (one bf16 MFMA pass, ~2^-9 relative per layer: the fast mode).
 */
#define EMD_PREC_BF16 1
#define EMD_PREC_BF16X3 3

/* activation codes of the `act` arguments (applied after the first affine, and after the second if present) */
#define EMD_ACT_NONE 0
#define EMD_ACT_RELU6 1 /* tf.nn.relu6: machine_learning/denoiser.py:83 */
#define EMD_ACT_RELU 2  /* tf.nn.relu:  misc_py/modified_Xception.py:209, :222, :312 */
#define EMD_ACT_LEAKY 4 /* tf.nn.leaky_relu, alpha 0.2: misc_py/gan-infilling-100.py:178 (matrix-core epilogues, emd_affine_act_f32) */
#define EMD_ACT_RELU6_CLIP01 3 /* relu6 then tf.clip_by_value(.,0,1), misc_py/denoiser-multi-gpu.py:534-538 (emd_affine_act_f32 only) */

/* Round 4: the two per-channel steps of the chain can run inside the kernel that finishes the reduction in front of them (one launch
 * less per layer and direction: ~700 launches of 4-5 us per training step, each a link in its stream's dependent chain).  The argument
 * blocks (host structs; every pointer a device pointer; [C], or [B][C] in the per-image forms, exactly as the separate calls take them):
 *   emd_bn_train_fold_t: emd_bn_train_fold[_images]_f32's parameters and outputs  -> emd_conv1x1_stats_fold_f32, emd_conv3x3_stats_fold_f32,
 *                        emd_deconv3x3s2_stats_fold_f32 (the conv, its output's statistics AND the fold: mean / var are still written)
 *   emd_bn_bwd_prep_t:   emd_bn_bwd_prep[_images]_f32's                            -> emd_bn_bwd_reduce_prep_f32, emd_dw3x3_bn_bwd_reduce_f32 */
typedef struct {
    const float *gamma1, *beta1, *gamma2, *beta2, *bias;   /* gamma1 / beta1 NULL: a single norm; bias NULL or the conv bias in front of it */
    float eps, pad_;
    float *scale, *shift, *rstd1, *rstd2;                  /* outputs (rstd2: the double norm only) */
    float *mm1, *mv1, *mm2, *mv2;                          /* moving statistics to update from image 0 / the batch, or all NULL */
    double decay;
} emd_bn_train_fold_t;
typedef struct {
    const float *gamma1, *gamma2, *rstd1, *rstd2;
    float eps, pad_;
    float *K, *m1, *m2;                                    /* outputs for the apply step */
    float *dgamma1, *dgamma2, *dbeta2;                     /* parameter gradients, ADDED into */
} emd_bn_bwd_prep_t;

/* Host-side weight packing for the matrix-core kernels (all pointers are HOST pointers).
 * w_host : taps x Cin x Cout float32 in TensorFlow order, [taps][Cin][Cout] (slim.conv2d /
 *          pointwise_weights, cout_major = 0) or [taps][Cout][Cin] (slim.conv2d_transpose, cout_major = 1).
 * hi/lo  : emd_packed_weight_elems(taps,Cin,Cout) bf16 words each: w = hi + lo (+2^-17), stored
 *          [Cout padded to 128][taps][Cin padded to 64], zero padded. */
size_t emd_packed_weight_elems(int taps, int Cin, int Cout);
int emd_pack_weights_bf16(const float* w_host, int taps, int Cin, int Cout, int cout_major,
                          uint16_t* hi_host, uint16_t* lo_host);

/* 1x1 convolution on the matrix cores, optional stride 2 (TF SAME for k=1: samples x[0::2]).
 * replaces: the pointwise half of slim.separable_convolution2d + normalizer BN + batch_then_activ
 *           (denoiser.py:113-134); slim.conv2d(kernel_size=1[,stride=2]) + bias + BN + relu6
 *           (:91-97 with :359/:371/:383, :159-164, :208-214, :220-227); the residual adds.
 * x [B,H,W,Cin] pixel stride ldx;  y [B,ceil(H/s),ceil(W/s),Cout] pixel stride ldy;  res like y, ldres.
 * Cin, Cout, ldx, ldy, ldres multiples of 4; x, y, res, whi, wlo, scale*, shift* 16-byte aligned.
 * wlo may be NULL with EMD_PREC_BF16. */
int emd_conv1x1_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo,
                    const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                    const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin,
                    int Cout, int stride, int act, int precision, emd_stream_t stream);

/* Dense 3x3 convolution on the matrix cores (9-tap implicit GEMM), TF SAME, stride 1 or 2, or stride 1 with
 * dilation `rate` <= 31.
 * replaces: tf.layers.conv2d(kernel_size=3[, dilation_rate=r]) + bias + BN + relu6 -- the ASPP rate branches
 *           of the training twin (misc_py/denoiser-multi-gpu.py:306-328).
 * whi/wlo: emd_pack_weights_bf16(taps = 9, w_host = [ky][kx][Cin][Cout]).  Alignment rules as emd_conv1x1_f32. */
int emd_conv3x3_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                    const float* shift1, const float* scale2, const float* shift2, const float* res, int ldres,
                    float* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride, int rate, int act,
                    int precision, emd_stream_t stream);

/* tf.nn.pool(window_shape=(2,2), "AVG", "SAME", strides=(2,2)): y [B,ceil(H/2),ceil(W/2),C].
 * replaces: the image-level branch of the training twin's ASPP (misc_py/denoiser-multi-gpu.py:331-335). */
int emd_avgpool2x2_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C,
                       emd_stream_t stream);

/* 3x3 stride-2 transposed convolution, output exactly 2H x 2W, as four output-phase GEMMs.
 * replaces: slim.conv2d_transpose(kernel_size=3, stride=2, padding='same') + bias + BN + relu6
 *           (denoiser.py:141-148):  y[2i+k] += x[i]*w[k], cropped at the end.
 * whi/wlo: HOST arrays of 4 DEVICE pointers, one packed block per phase (phase = 2*row_parity +
 *          col_parity) holding the taps emd_deconv_phase_taps lists, in that order, cout_major = 1. */
int emd_deconv_phase_taps(int phase, int* ky, int* kx);
int emd_deconv3x3s2_f32(const float* x, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4],
                        const float* scale1, const float* shift1, float* y, int ldy, int B, int H, int W,
                        int Cin, int Cout, int act, int precision, emd_stream_t stream);

/* The whole strided_conv_block (denoiser.py:110-136) for stride 1 in ONE kernel: depthwise 3x3 (SAME) ->
 * pointwise 1x1 on the matrix cores -> fused epilogue (above).  The depthwise result never reaches HBM.
 * replaces: slim.separable_convolution2d + normalizer BN + batch_then_activ (+ the residual "+=").
 * Supported when emd_sep3x3_fused_supported() returns 1: stride 1, rate 1, H%8==0, W%16==0, Cin%32==0,
 * Cout%4==0, Cout<=128 (one N tile), or Cout<=256 with Cin<=256 (one 256-column tile on 4 x 16 pixels; no generated-input form);
 * otherwise call emd_dw3x3_f32 + emd_conv1x1_f32.
 * x [B,H,W,Cin] ldx; dw [3][3][Cin]; whi/wlo packed pointwise weights (taps=1); y [B,H,W,Cout] ldy. */
int emd_sep3x3_fused_supported(int H, int W, int Cin, int Cout, int stride, int rate);
int emd_sep3x3_fused_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                         const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                         const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin,
                         int Cout, int act, int precision, emd_stream_t stream);
/* The same block with stride 2 (strided_conv_block(stride=2), machine_learning/denoiser.py:258, :273, :288), one launch (round 3): x
 * [B,H,W,Cin] with H%8==0, W%32==0 (TF SAME on even sizes: no padding before, one pixel after), y [B,H/2,W/2,Cout], Cout <= 256,
 * res (optional) in the output's shape.  Split-bf16.  emd_sep3x3_fused_supported(H, W, Cin, Cout, 2, 1) says where it applies. */
int emd_sep3x3_fused_s2_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                            const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                            const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                            emd_stream_t stream);
/* The same on the tf.pad(REFLECT, 1) image with VALID padding: graph G's down-sampling strided_conv_block(stride 2, pad_size = (1, 1))
 * (misc_py/gan-infilling-100.py:205-243, :345-352).  Same shape rules (emd_sep3x3_fused_supported(H, W, Cin, Cout, 2, 1)). */
int emd_sep3x3_fused_s2_reflect_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                                    const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                    const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                    emd_stream_t stream);

/* Depthwise 3x3, TF SAME padding, stride 1 or 2 (rate 1) or stride 1 with dilation `rate`.
 * replaces: the depthwise half of slim.separable_convolution2d (denoiser.py:113-131).
 * x [B,H,W,C] pixel stride ldx; w [3][3][C] (TF [3,3,C,1]); y [B,ceil(H/s),ceil(W/s),C] pixel stride ldy. */
int emd_dw3x3_f32(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C,
                  int stride, int rate, emd_stream_t stream);

/* ---- "split32" activations: the pointwise GEMM fed entirely by LDS-DMA (csrc/gemm_split.hip).
 * A split32 tensor [npix][C] holds every value as bf16 hi + bf16 lo (x = hi + lo + O(2^-17 x), both round-to-nearest):
 * pixel pitch ld in 4-byte units (ld % 32 == 0, ld >= emd_split32_ld(C) = C rounded up to 32); inside a pixel, channel
 * group g = c/32 occupies bytes [128 g, 128 g + 128): 32 x hi, then 32 x lo; channels C..ld are zero.  Same bytes and
 * pitch as the fp32 NHWC tensor it stands for; base address 128-byte aligned.
 *
 * emd_dw3x3_split32_f32     = emd_dw3x3_f32 whose result is written in split32 form (the depthwise half of
 *                             slim.separable_convolution2d, machine_learning/denoiser.py:113-131).
 * emd_to_split32_f32        converts an fp32 tensor (pitch ldx floats).
 * emd_conv1x1_split32_f32   = emd_conv1x1_f32 (stride 1, EMD_PREC_BF16X3) on a split32 input: the pointwise half + BN x2
 *                             + relu6 + residual (:123, :134, :246); results are bit-identical to emd_conv1x1_f32 on the
 *                             fp32 twin of xs.  M = number of pixels.  256 x 128 tiles, 512 threads, K step 32, both
 *                             operands by global_load_lds_dwordx4, XOR-swizzled LDS rows.
 * emd_conv1x1_split32_supported: 1 where this kernel is the better choice (Cin, Cout >= 128 and >= 256 tiles). */
int emd_split32_ld(int C);
int emd_to_split32_f32(const float* x, int ldx, void* y, int ldy, long npix, int C, emd_stream_t stream);
int emd_dw3x3_split32_f32(const float* x, int ldx, const float* w, void* y, int ldy, int B, int H, int W, int C,
                          int stride, int rate, emd_stream_t stream);
int emd_dw3x3_reflect_split32_f32(const float* x, int ldx, const float* w, void* y, int ldy, int B, int H, int W, int C,
                                  int stride, emd_stream_t stream); /* emd_dw3x3_reflect_f32 (graph G) with split32 output */
int emd_conv1x1_split32_supported(long M, int Cin, int Cout);
/* emd_conv1x1_split32_f32 (no second affine, no residual) that also returns the per-channel batch mean and BIASED variance
 * of its output y -- what emd_bn_stats_f32 computes in a second pass over y (the batch-statistics norms that follow the
 * pointwise convs of misc_py/modified_Xception.py:302-323).  The GEMM epilogue leaves one double partial per (256-row tile,
 * channel); a fixed-order final reduction follows (deterministic).  workspace: emd_conv1x1_split32_stats_workspace_bytes(M,
 * Cout) bytes, 8-byte aligned. */
size_t emd_conv1x1_split32_stats_workspace_bytes(long M, int Cout);
int emd_conv1x1_split32_stats_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                                  const float* shift1, float* y, int ldy, long M, int Cin, int Cout, int act, float* mean,
                                  float* var, void* workspace, emd_stream_t stream);
/* ... and folds the batch norm that uses those statistics in the same final-reduction launch (emd_bn_fold_f32's arithmetic on
 * the same float mean / var: scale = gamma / sqrt(var + eps), gamma NULL = 1; shift = beta - mean * scale, beta NULL = 0): one
 * launch instead of two between the GEMM and the kernel that applies the norm. */
int emd_conv1x1_split32_stats_fold_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                                       const float* shift1, float* y, int ldy, long M, int Cin, int Cout, int act, float* mean,
                                       float* var, void* workspace, const float* gamma, const float* beta, float eps,
                                       float* scale, float* shift, emd_stream_t stream);
/* Dense 3x3 convolution (emd_conv3x3_f32: TF SAME, stride 1/2, dilation) and the 3x3 stride-2 transposed convolution
 * (emd_deconv3x3s2_f32) on a split32 input, same packed weights, same arithmetic (bit-identical results); out_split != 0
 * writes y itself as a split32 tensor (pitch ldy 4-byte units, % 32; channels Cout..ceil32(Cout) zero) for a following
 * split32 convolution -- tf.layers.conv2d / conv2d_transpose chains (misc_py/modified_Xception.py:215-229, :538-621;
 * machine_learning/denoiser.py:141-148) then never write an fp32 activation.  Cin <= 2048. */
int emd_conv3x3_split32_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                            const float* shift1, const float* scale2, const float* shift2, const float* res, int ldres,
                            void* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride, int rate, int act,
                            int out_split, emd_stream_t stream);
int emd_deconv3x3s2_split32_f32(const void* xs, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4],
                                const float* scale1, const float* shift1, void* y, int ldy, int B, int H, int W, int Cin,
                                int Cout, int act, int out_split, emd_stream_t stream);
int emd_conv1x1_split32_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                            const float* shift1, const float* scale2, const float* shift2, const float* res,
                            int ldres, float* y, int ldy, long M, int Cin, int Cout, int act, emd_stream_t stream);
/* emd_conv1x1_split32_f32 / emd_sep3x3_fused_f32 writing y as a split32 tensor (pitch ldy in 4-byte units, a multiple of 32;
 * y 128-byte aligned; channels Cout..ceil32(Cout) zero; the fused separable form needs Cout % 32 == 0): the producer of a
 * split32 convolution's input writes no fp32 activation and needs no emd_to_split32_f32 pass (graph D: deconv2_b -> deconv2to1,
 * deconv1_b -> deconv1to0, machine_learning/denoiser.py:357-362, :369-374).  Same values as the fp32 form, then split. */
int emd_conv1x1_split32_out_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                                const float* shift1, const float* scale2, const float* shift2, const float* res,
                                int ldres, void* y, int ldy, long M, int Cin, int Cout, int act, emd_stream_t stream);
int emd_sep3x3_fused_out_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                             const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                             const float* res, int ldres, void* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                             emd_stream_t stream);
/* 1 where a graph host should take emd_deconv3x3s2_fused_split32_f32 for this layer ([B,H,W,Cin] input): always where its
 * patch-resident kernel applies (H % 8 == 0, W % 32 == 0, Cin % 32 == 0 -- independent of B, because that kernel sums in another order
 * than the GEMM forms and a result must not depend on the batch size), otherwise from B*H*W >= 49152 on (speed only). */
int emd_deconv3x3s2_fused_preferred(int B, int H, int W, int Cin, int Cout);

/* emd_deconv3x3s2_split32_f32 as ONE launch: a workgroup computes the four output phases of its 256 input pixels back to
 * back, so the input is read from HBM once instead of once per phase launch.  Same arguments, bit-identical results. */
int emd_deconv3x3s2_fused_split32_f32(const void* xs, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4],
                                      const float* scale1, const float* shift1, void* y, int ldy, int B, int H, int W, int Cin,
                                      int Cout, int act, int out_split, emd_stream_t stream);

/* emd_dw3x3_f32 / emd_dw3x3_split32_f32 on relu(x * pre_scale + pre_shift) (per channel, device float[C]): the
 * batch-statistics norm + relu that ends the previous separable block of misc_py/modified_Xception.py (:302-323) applied
 * while the depthwise kernel loads its input, instead of in a pass of its own; padding is applied after it (TF pads the
 * activated tensor). */
int emd_dw3x3_pre_f32(const float* x, int ldx, const float* pre_scale, const float* pre_shift, const float* w, float* y,
                      int ldy, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream);
int emd_dw3x3_pre_split32_f32(const float* x, int ldx, const float* pre_scale, const float* pre_shift, const float* w,
                              void* y, int ldy, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream);
/* The same for the training step (round 4; graph D', slim.separable_convolution2d + _batch_norm_fn(is_training) + relu6,
 * machine_learning/denoiser.py:110-136 with phase = True; misc_py/denoiser-multi-gpu.py:752-782): the affine + activation of a
 * separable conv whose only consumer is the next one's depthwise stage is applied in that stage's loads.  act: EMD_ACT_RELU6 or
 * EMD_ACT_RELU; pre_images != 0: pre_scale / pre_shift are [B][C] (per-image statistics: a batched pass of one-image towers), else
 * [C].  Bits of emd_affine_act_f32 / emd_affine_act_images_f32 followed by emd_dw3x3_f32. */
int emd_dw3x3_pre_act_f32(const float* x, int ldx, const float* pre_scale, const float* pre_shift, int pre_images, int act,
                          const float* w, float* y, int ldy, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream);

/* Dense 3x3 conv of a ONE-channel image + per-channel affine + activation (graph X's entry conv: tf.layers.conv2d(1 -> 32, k 3,
 * stride 2) + bias -> batch norm -> relu, misc_py/modified_Xception.py:356-364; the caller folds bias and norm into scale / shift):
 * x [B,H,W] fp32 contiguous, w [9][Cout] fp32 (tap-major), y [B,Ho,Wo,Cout] fp32 (pitch ldy floats) or, out_split != 0, a split32
 * tensor (pitch ldy 4-byte units, a multiple of 32; padding channels written as zero).  TF SAME, stride 1 or 2.  fp32 FMAs. */
int emd_conv3x3_cin1_f32(const float* x, const float* w, const float* scale, const float* shift, void* y, int ldy, int B, int H, int W,
                         int Cout, int stride, int act, int out_split, emd_stream_t stream);

/* Layers fed by the 1-channel image: y[pix][n] = act( d[pix]*a[n] + shift[n] ).
 * w9 != NULL: d = 3x3 SAME depthwise of x with the 9 weights w9 (stride 1)  -- cnn0 (denoiser.py:252),
 *             a[n] = pointwise_weights[0][n] * folded BN scale;
 * w9 == NULL: d = x sampled with `stride`                                   -- residual0 (:263),
 *             a[n] = weights[0][n] * folded BN scale, shift includes the bias.
 * x [B,H,W]; y [B,ceil(H/s),ceil(W/s),Cout] pixel stride ldy; Cout/4 must divide 64. */
int emd_cin1_f32(const float* x, const float* w9, const float* a, const float* shift, float* y, int ldy,
                 int B, int H, int W, int Cout, int stride, int act, emd_stream_t stream);

/* Dense 3x3 SAME convolution to ONE output channel + scalar affine + relu6.
 * replaces: the final slim.conv2d(num_outputs=1, kernel_size=3) + bias + BN + relu6 (denoiser.py:387).
 * x [B,H,W,Cin] pixel stride ldx; w [3][3][Cin]; y [B,H,W]; scale/shift: bias and BN folded.
 * act: 0 none, 1 relu6, 2 relu6 then tf.clip_by_value(.,0,1) (misc_py/denoiser-multi-gpu.py:534-538).
 * pre_relu != 0: the convolution is followed by "+pre_bias, relu" BEFORE the affine, i.e.
 *   tf.layers.conv2d(activation=relu) -> BN -> relu (conv_block, misc_py/modified_Xception.py:215-229, :621). */
int emd_conv3x3_cout1_f32(const float* x, int ldx, const float* w, float scale, float shift, float* y, int B,
                          int H, int W, int Cin, int act, float pre_bias, int pre_relu, emd_stream_t stream);

/* tf.image.resize_images(x,[Ho,Wo]): bilinear, align_corners=False, legacy (no half-pixel) sampling.
 * replaces: denoiser.py:199 (identity size) and :350 (32 -> 128). */
int emd_resize_bilinear_f32(const float* x, int ldx, float* y, int ldy, int B, int Hi, int Wi, int Ho, int Wo,
                            int C, emd_stream_t stream);

/* A lone inference batch norm (+ relu6): y = act(x*scale + shift) per channel.
 * replaces: batch_then_activ on the ASPP image-level branch (denoiser.py:200). */
int emd_affine_relu6_f32(const float* x, int ldx, const float* scale, const float* shift, float* y, int ldy,
                         long npix, int C, int act, emd_stream_t stream);

/* y = act(x*scale + shift) [+ res], act = EMD_ACT_*; y may be x (in place).  The normalise+activate step that
 * follows a batch-statistics batch norm, and the "+ residual" after it (misc_py/modified_Xception.py:397, :533). */
int emd_affine_act_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldres,
                       float* y, int ldy, long npix, int C, int act, emd_stream_t stream);

/* Batch statistics for tf.contrib.layers.batch_norm called with its defaults (is_training=True) -- what the
 * separable convs of misc_py/modified_Xception.py:302-323 do even at inference: per-channel mean and BIASED
 * variance of x [npix, C] (pixel stride ldx), accumulated in double.  workspace: device buffer of
 * emd_bn_stats_workspace_bytes(npix, C) bytes, 8-byte aligned.  emd_bn_fold_f32 turns (mean, var, gamma|NULL,
 * beta|NULL, eps) into the (scale, shift) of one affine, on the device (no host round trip). */
/* Per-image forms (instance norms; the batch-statistics norms of misc_py/apply_autoencoders.py:105-116, which the reference
 * evaluates one crop per sess.run): x is [B][npix_img][C]; mean / var / scale / shift are [B][C]; workspace:
 * B x emd_bn_stats_workspace_bytes(npix_img, C) bytes.  Image b gets exactly the bits emd_bn_stats_f32 gives it alone. */
int emd_bn_stats_images_f32(const float* x, int ldx, int B, long npix_img, int C, float* mean, float* var, void* workspace,
                            emd_stream_t stream);

/* The convolutions of a TRAINING forward pass (misc_py/denoiser-multi-gpu.py:200-540 with phase = True, :752-782: every convolution is
 * followed by a batch norm on BATCH statistics): y = conv(x), no affine, no activation (ones / zeros: device vectors of Cout ones and
 * zeros, 16-byte aligned), plus the per-channel mean and biased variance of y, gathered in the GEMM's epilogue -- what
 * emd_bn_stats_f32 (images = 0: over all B * Ho * Wo pixels, mean / var [Cout]) or emd_bn_stats_images_f32 (images = 1: per image,
 * [B][Cout]) would return for y, without their pass over it (one partial per 128-row tile, reduced in a fixed order: deterministic;
 * the last bits differ from the two-launch form's, whose partials are cut differently).  images = 1 needs Ho * Wo % 128 == 0
 * (EMD_E_UNSUPPORTED otherwise: call the convolution and the statistics separately).
 * workspace: emd_conv_stats_workspace_bytes(B * Ho * Wo, Cout) bytes, 8-byte aligned.  emd_conv1x1_stats_f32: slim.conv2d 1x1 /
 * the pointwise half of slim.separable_convolution2d, stride 1 or 2 (TF SAME: samples x[0::2]); emd_conv3x3_stats_f32: dense 3x3,
 * stride 1, dilation `rate` (the ASPP branches of graph D', :330-353). */
size_t emd_conv_stats_workspace_bytes(long M, int Cout);
int emd_conv1x1_stats_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones, const float* zeros,
                          float* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride, int precision, int images,
                          float* mean, float* var, void* workspace, emd_stream_t stream);
int emd_conv3x3_stats_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones, const float* zeros,
                          float* y, int ldy, int B, int H, int W, int Cin, int Cout, int rate, int precision, int images,
                          float* mean, float* var, void* workspace, emd_stream_t stream);
/* The transposed 3x3 stride-2 conv of a training forward pass (emd_deconv3x3s2_f32, no affine, no activation; machine_learning/
 * denoiser.py:141-148 under phase = True) + the batch statistics of its output from the four phase GEMMs' epilogues: mean / var [Cout]
 * over all B * 2H * 2W output pixels, or images != 0: [B][Cout] per image (needs H * W % 128 == 0: EMD_E_UNSUPPORTED otherwise).
 * workspace: emd_conv_stats_workspace_bytes(4 * B * H * W, Cout) bytes. */
int emd_deconv3x3s2_stats_f32(const float* x, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4], const float* ones,
                              const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int precision, int images,
                              float* mean, float* var, void* workspace, emd_stream_t stream);
/* The three above with the training-mode fold of the norm behind the conv in the statistics' final kernel (emd_bn_train_fold[_images]_f32's
 * step: scale, shift, rstd1, rstd2, moving-average updates; mean / var are written as well): one launch less per layer. */
int emd_conv1x1_stats_fold_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones, const float* zeros, float* y,
                               int ldy, int B, int H, int W, int Cin, int Cout, int stride, int precision, int images, float* mean, float* var,
                               void* workspace, const emd_bn_train_fold_t* fold, emd_stream_t stream);
int emd_conv3x3_stats_fold_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones, const float* zeros, float* y,
                               int ldy, int B, int H, int W, int Cin, int Cout, int rate, int precision, int images, float* mean, float* var,
                               void* workspace, const emd_bn_train_fold_t* fold, emd_stream_t stream);
int emd_deconv3x3s2_stats_fold_f32(const float* x, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4], const float* ones,
                                   const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int precision, int images,
                                   float* mean, float* var, void* workspace, const emd_bn_train_fold_t* fold, emd_stream_t stream);
int emd_affine_act_images_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldres,
                              float* y, int ldy, int B, long npix_img, int C, int act, emd_stream_t stream);
/* y = act(x*scale + shift) + res_act(res*res_scale + res_shift): the residual operand given BEFORE its own affine + activation (round 4,
 * graph D': the 1x1 residual projection's batch norm + relu6 -- conv_block_not_sep, machine_learning/denoiser.py:356-383 with phase =
 * True -- applied where the block adds it instead of in a pass of its own; bits of the two-pass route).  images = 0: vectors [C], npix =
 * all pixels; images = B > 0: vectors [B][C], npix = pixels per image.  res_act: EMD_ACT_RELU6 or EMD_ACT_RELU. */
int emd_affine_act_res_affine_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldres,
                                  const float* res_scale, const float* res_shift, int res_act, float* y, int ldy, int images, long npix,
                                  int C, int act, emd_stream_t stream);
size_t emd_bn_stats_workspace_bytes(long npix, int C);
int emd_bn_stats_f32(const float* x, int ldx, long npix, int C, float* mean, float* var, void* workspace,
                     emd_stream_t stream);
int emd_bn_fold_f32(const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                    float* scale, float* shift, int C, emd_stream_t stream);

/* ================================================================================================
 * Training path of graph D' (misc_py/denoiser-multi-gpu.py): tf.gradients of the tower loss (:752-782) and the
 * Nesterov train op (:1011-1077).  Data gradients of the 1x1 / 3x3 / transposed convolutions are the forward
 * entry points above run with weights packed transposed (emd_pack_weights_dev); the rest follows.
 * Convention: PARAMETER gradients are ADDED into their destination (towers / micro-batches accumulate one
 * gradient set, :1040; zero it once per step), activation gradients are written.
 * ================================================================================================ */

/* Weight gradient of any convolution above: dw[t][k][n] += sum_m a[src_t(m)][k] * dy[m][n].
 * replaces: the Conv2DBackpropFilter nodes tf.gradients (:779) creates for tf.layers.conv2d / the pointwise half of
 * slim.separable_convolution2d (:225-276) / tf.layers.conv2d_transpose (:278-289).
 * m runs over the [B,Hg,Wg] grid of dy (pixel stride ldd, N channels); a is [B,Ha,Wa,K] (pixel stride lda) read at
 * (i*sa + tap_dy[t], j*sa + tap_dx[t]), zero outside.  conv (stride s, rate r, SAME pad pt): a = layer input,
 * sa = s, tap = k*r - pt, dw in TF layout [taps][Cin][Cout].  Transposed conv: a = the gradient w.r.t. its OUTPUT,
 * dy := its INPUT, sa = 2, tap = k, dw in TF layout [taps][Cout][Cin].  K, N multiples of 4. */
int emd_conv_wgrad_f32(const float* a, int lda, const float* dy, int ldd, float* dw, int B, int Hg, int Wg, int Ha, int Wa,
                       int K, int N, int ntaps, const int* tap_dy, const int* tap_dx, int sa, emd_stream_t stream);

/* emd_pack_weights_bf16 for weights that live on the DEVICE (re-packed after every optimizer step).
 * w [src_taps][Cin][Cout] (cout_major 0) or [src_taps][Cout][Cin] (cout_major 1); packed tap t is source tap
 * tap_sel[t] (NULL: identity, needs ntaps == src_taps) -- reversed order for the data gradient of a 3x3 conv,
 * emd_deconv_phase_taps subsets for the transposed conv.  hi/lo: emd_packed_weight_elems(ntaps,Cin,Cout) elements. */
int emd_pack_weights_dev(const float* w, int src_taps, int ntaps, const int* tap_sel, int Cin, int Cout, int cout_major,
                         uint16_t* hi, uint16_t* lo, emd_stream_t stream);

/* All of a model's packs in ONE launch (the re-pack after an optimizer step is ~860 packs of a few KB..MB each: launch-bound one
 * by one).  A job = one emd_pack_weights_dev call; emd_pack_job_fill validates the arguments on the host and fills the derived
 * fields; the caller numbers the jobs' blocks consecutively (first_block = sum of the earlier jobs' n_blocks), copies the table
 * to the device once (the pointers in it do not change between steps) and passes the total block count. */
typedef struct emd_pack_job {
    const float* w;
    uint16_t* hi;
    uint16_t* lo;
    unsigned long long tap_sel;   /* 4 bits per packed tap: its source tap */
    long total;                   /* packed elements of one plane */
    long first_block;             /* caller-assigned */
    long n_blocks;                /* workgroups of this job in the batch launch (set by emd_pack_job_fill: ceil(total / 256), or the 16 x 16
                                   * tile count of the transposing form) */
    int ntaps, cin, cout, cout_major, cpad, pad_;
} emd_pack_job_t;
int emd_pack_job_fill(emd_pack_job_t* job, const float* w, int src_taps, int ntaps, const int* tap_sel, int Cin, int Cout,
                      int cout_major, uint16_t* hi, uint16_t* lo);
int emd_pack_weights_batch_dev(const emd_pack_job_t* jobs_dev, int n_jobs, long n_blocks, emd_stream_t stream);

/* Data gradient of the stride-2 1x1 conv (residual branches, :225-238 with strides=2):
 * dx[b,2i,2j,:] = dy[b,i,j,:] * W^T (+ res at the same pixels); other pixels of dx are left as they are.
 * dy [B,ceil(H/2),ceil(W/2),Cout]; dx [B,H,W,Cin]; whi/wlo packed with (Cin:=Cout, Cout:=Cin, cout_major 1). */
int emd_conv1x1_s2_bwd_data_f32(const float* dy, int ldd, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                                const float* shift1, const float* res, int ldres, float* dx, int ldx, int B, int H, int W,
                                int Cout, int Cin, int precision, emd_stream_t stream);

/* Training-mode batch norm chain  r -> [BN1] -> BN2 -> relu6 [-> clip]  (:210-223; contrib batch_norm, fused,
 * decay 0.999, eps 1e-3), see csrc/bn_train.hip for the algebra.
 * emd_bn_train_fold_f32: batch (mean,var) of r (emd_bn_stats_f32) -> forward affine (scale, shift), rstd1 (and
 *   rstd2 for the double norm: gamma1/beta1 non-NULL), and, if mm2 != NULL, the moving-average updates
 *   (mm1/mv1: BN1's, double norm only; bias: the conv bias that precedes a single BN, may be NULL).
 * emd_bn_bwd_reduce_f32: s1[c] = sum g, s2[c] = sum g*(x-mean)*rstd, g = dy*mask(x*mscale+mshift);
 *   mask 0 none, 1 relu6 (0<z<6), 2 relu6 then clip [0,1] (0<z<=1), 3 leaky_relu 0.2 (graph G).  x == NULL: s1 only.
 *   accumulate_s1 != 0: s1 += (bias gradients).  workspace: emd_chan_reduce_workspace_bytes(npix, C) bytes.
 * emd_bn_bwd_prep_f32: (s1, t=s2) -> K, m1, m2 for the apply step; dgamma1, dgamma2, dbeta2 += .
 * emd_bn_bwd_apply_f32: dx = K*(g - m1 - (x-mean)*m2); dx may be dy.  C = 1 is allowed (the final layer). */
int emd_bn_bwd_reduce_prep_f32(const float* dy, int ldd, const float* x, int ldx, const float* mean, const float* rstd, const float* mscale,
                               const float* mshift, int mask, int images, long npix, int C, float* s1, float* s2, void* workspace,
                               const emd_bn_bwd_prep_t* prep, emd_stream_t stream);
/* The reduction (+ per-channel step) and the apply pass for a gradient that is the data gradient of a 3x3 conv to ONE output channel (the
 * network's final conv, :528-532): dy[p][c] = sum_taps g1[p + (1-ky, 1-kx)] * w9[3ky+kx][c] is formed from the 1-channel image g1 [B][H][W]
 * in both passes and never written (emd_conv3x3_cout1_bwd_data_f32's arithmetic).  images != 0: per-image vectors [B][C]. */
int emd_bn_bwd_reduce_prep_cout1_f32(const float* g1, const float* w9, int B, int H, int W, const float* x, int ldx, const float* mean,
                                     const float* rstd, const float* mscale, const float* mshift, int mask, int images, int C, float* s1,
                                     float* s2, void* workspace, const emd_bn_bwd_prep_t* prep, emd_stream_t stream);
int emd_bn_bwd_apply_cout1_f32(const float* g1, const float* w9, int B, int H, int W, const float* x, int ldx, const float* K, const float* m1,
                               const float* mean, const float* m2, const float* mscale, const float* mshift, int mask, int images, float* dx,
                               int ldo, int C, emd_stream_t stream);
size_t emd_chan_reduce_workspace_bytes(long npix, int C);
int emd_bn_train_fold_f32(const float* mean, const float* var, const float* gamma1, const float* beta1, const float* gamma2,
                          const float* beta2, const float* bias, float eps, long npix, int C, float* scale, float* shift,
                          float* rstd1, float* rstd2, float* mm1, float* mv1, float* mm2, float* mv2, double decay,
                          emd_stream_t stream);
int emd_bn_bwd_reduce_f32(const float* dy, int ldd, const float* x, int ldx, const float* mean, const float* rstd,
                          const float* mscale, const float* mshift, int mask, long npix, int C, float* s1, float* s2,
                          int accumulate_s1, void* workspace, emd_stream_t stream);
int emd_bn_bwd_prep_f32(const float* s1, const float* t, const float* gamma1, const float* gamma2, const float* rstd1,
                        const float* rstd2, float eps, long npix, int C, float* K, float* m1, float* m2, float* dgamma1,
                        float* dgamma2, float* dbeta2, emd_stream_t stream);
int emd_bn_bwd_apply_f32(const float* dy, int ldd, const float* x, int ldx, const float* K, const float* m1,
                         const float* mean, const float* m2, const float* mscale, const float* mshift, int mask, float* dx,
                         int ldo, long npix, int C, emd_stream_t stream);
/* Per-image forms of the four (B images of npix pixels each; statistics / coefficient vectors [B][C]; the parameter
 * vectors gamma / beta / bias and their gradients stay [C]; the moving statistics follow image 0, the first tower,
 * misc_py/denoiser-multi-gpu.py:701-707).  Image b is reduced exactly as it would be alone, so the one-image towers of a
 * rank (:763) run as ONE batched pass with per-image statistics from emd_bn_stats_images_f32 / emd_affine_act_images_f32:
 * identical arithmetic per image, B times the GEMM M, B times fewer launches.  Workspace of the reduce: B x
 * emd_chan_reduce_workspace_bytes(npix, C). */
int emd_bn_train_fold_images_f32(const float* mean, const float* var, const float* gamma1, const float* beta1,
                                 const float* gamma2, const float* beta2, const float* bias, float eps, long npix, int B, int C,
                                 float* scale, float* shift, float* rstd1, float* rstd2, float* mm1, float* mv1, float* mm2,
                                 float* mv2, double decay, emd_stream_t stream);
int emd_bn_bwd_reduce_images_f32(const float* dy, int ldd, const float* x, int ldx, const float* mean, const float* rstd,
                                 const float* mscale, const float* mshift, int mask, int B, long npix, int C, float* s1,
                                 float* s2, void* workspace, emd_stream_t stream);
int emd_bn_bwd_prep_images_f32(const float* s1, const float* t, const float* gamma1, const float* gamma2, const float* rstd1,
                               const float* rstd2, float eps, long npix, int B, int C, float* K, float* m1, float* m2,
                               float* dgamma1, float* dgamma2, float* dbeta2, emd_stream_t stream);
int emd_bn_bwd_apply_images_f32(const float* dy, int ldd, const float* x, int ldx, const float* K, const float* m1,
                                const float* mean, const float* m2, const float* mscale, const float* mshift, int mask,
                                float* dx, int ldo, int B, long npix, int C, emd_stream_t stream);

/* The training-mode batch norm of a SMALL map as ONE launch per direction (round 4): per-image statistics (a tower of one image,
 * misc_py/denoiser-multi-gpu.py:763, or B of them as one batched pass; vectors [B][C]), npix <= 4096 pixels per image, C % 4 == 0
 * (emd_bn_train_small_supported; EMD_E_UNSUPPORTED otherwise: use the slab forms above).
 * emd_bn_train_fwd_small_f32 == emd_bn_stats_images_f32 + emd_bn_train_fold_images_f32 + emd_affine_act_images_f32:
 *   out = act(r * scale + shift) [+ res], with scale / shift / rstd1 / rstd2 / mean returned for the reverse pass and the moving
 *   statistics (NULL = leave them) updated from image 0.  gamma1 / beta1 NULL: a single norm (gamma2, beta2) behind conv + bias.
 * emd_bn_train_bwd_small_f32 == emd_bn_bwd_reduce_images_f32 + emd_bn_bwd_prep_images_f32 + emd_bn_bwd_apply_images_f32:
 *   dx = d loss / d r (dx may be dy or x), the norms' parameter gradients ADDED (float atomics) into dgamma1 / dgamma2 / dbeta2.
 * Same formulas as the slab forms; their sums are cut differently, so the two agree to rounding, not bit for bit. */
int emd_bn_train_small_supported(long npix, int C);
int emd_bn_train_fwd_small_f32(const float* r, int ldr, int B, long npix, int C, const float* gamma1, const float* beta1,
                               const float* gamma2, const float* beta2, const float* bias, float eps, float* scale, float* shift,
                               float* rstd1, float* rstd2, float* mean, float* mm1, float* mv1, float* mm2, float* mv2, double decay,
                               const float* res, int ldres, float* out, int ldo, int act, emd_stream_t stream);
int emd_bn_train_bwd_small_f32(const float* dy, int ldd, const float* x, int ldx, int B, long npix, int C, const float* mean,
                               const float* rstd1, const float* rstd2, const float* mscale, const float* mshift, int mask,
                               const float* gamma1, const float* gamma2, float eps, float* dgamma1, float* dgamma2, float* dbeta2,
                               float* dx, int ldo, emd_stream_t stream);

/* Depthwise 3x3 backward (the depthwise half of slim.separable_convolution2d, :253-273); shapes as emd_dw3x3_f32
 * (x, dx [B,H,W,C]; dy [B,ceil(H/s),ceil(W/s),C]); dw [3][3][C] +=. */
int emd_dw3x3_wgrad_f32(const float* x, int ldx, const float* dy, int ldd, float* dw, int B, int H, int W, int C, int stride,
                        int rate, emd_stream_t stream);
int emd_dw3x3_bwd_data_f32(const float* dy, int ldd, const float* w, float* dx, int ldx, int B, int H, int W, int C,
                           int stride, int rate, emd_stream_t stream);
/* Round 4: the data gradient of a stride-1 depthwise 3x3 fused with the batch-norm backward of the layer BEFORE it, for a gradient that
 * has no other contribution (the output of a separable conv consumed by exactly one separable conv): dy = emd_dw3x3_f32(dd, w_flipped)
 * is formed on the fly in both passes and never written.
 *   emd_dw3x3_bn_bwd_reduce_f32 = emd_dw3x3_f32 + emd_bn_bwd_reduce[_images]_f32 (s1 = sum g, s2 = sum g * (r - mean) * rstd,
 *                                 g = dy * mask(r * mscale + mshift)); workspace: emd_dw3x3_bn_bwd_workspace_bytes(B, H, W, C)
 *   emd_dw3x3_bn_bwd_apply_f32  = emd_dw3x3_f32 + emd_bn_bwd_apply[_images]_f32 (dr = K * (g - m1 - (r - mean) * m2); dr may be r)
 * dw_consumer (or NULL; needs mask = relu6): [9][C] += the CONSUMER's depthwise weight gradient = emd_dw3x3_wgrad_pre_f32(r, mscale,
 * mshift, relu6, dd): the reduction pass streams exactly its operands.
 * r, dr [B,H,W,C] (the consumer's INPUT grid); dd [B,ceil(H/stride),ceil(W/stride),C]; stride 1 or 2, rate (dilation, stride 1 only) as
 * emd_dw3x3_f32 (stride 1, rate 1: the rolling-window form; else a gather form = emd_dw3x3_bwd_data_f32's arithmetic);
 * w_flipped [9][C] = the consumer's depthwise taps reversed (tap t = original tap 8 - t); images != 0: every
 * per-channel vector is [B][C] (per-image statistics).  (tf.gradients of machine_learning/denoiser.py:110-136 with phase = True.) */
size_t emd_dw3x3_bn_bwd_workspace_bytes(int B, int H, int W, int C);
int emd_dw3x3_bn_bwd_reduce_f32(const float* dd, int ldd, const float* w_flipped, const float* r, int ldr, const float* mean,
                                const float* rstd, const float* mscale, const float* mshift, int mask, int images, int B, int H, int W,
                                int C, int stride, int rate, float* s1, float* s2, float* dw_consumer, void* workspace,
                                const emd_bn_bwd_prep_t* prep /* or NULL */, emd_stream_t stream);
int emd_dw3x3_bn_bwd_apply_f32(const float* dd, int ldd, const float* w_flipped, const float* r, int ldr, const float* K, const float* m1,
                               const float* mean, const float* m2, const float* mscale, const float* mshift, int mask, int images,
                               float* dr, int ldo, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream);
/* Both gradients of a stride-1 depthwise 3x3 in one pass (round 4): dx = emd_dw3x3_f32(dd, w_flipped) -- the data gradient, its bits -- and
 * dw[9][C] += emd_dw3x3_wgrad_f32(x, dd); dd is read once instead of twice.  x, dx, dd [B,H,W,C]; w_flipped = the taps reversed. */
int emd_dw3x3_bwd_both_f32(const float* dd, int ldd, const float* w_flipped, const float* x, int ldx, float* dx, int ldo, float* dw, int B,
                           int H, int W, int C, emd_stream_t stream);
/* emd_dw3x3_wgrad_f32 with the layer's input given as the pre-activation tensor r of the layer before it (the forward pass ran
 * emd_dw3x3_pre_act_f32 on it and never wrote x = act(r * pre_scale + pre_shift)): x is rebuilt in the loads.  Arguments as there. */
int emd_dw3x3_wgrad_pre_f32(const float* r, int ldx, const float* pre_scale, const float* pre_shift, int pre_images, int act,
                            const float* dy, int ldd, float* dw, int B, int H, int W, int C, int stride, int rate, emd_stream_t stream);

/* Backward of the final 3x3 conv to one channel (:528-532): dy [B,H,W]; dw [3][3][Cin] +=; dx [B,H,W,Cin]. */
int emd_conv3x3_cout1_wgrad_f32(const float* x, int ldx, const float* dy, float* dw, int B, int H, int W, int Cin,
                                emd_stream_t stream);
int emd_conv3x3_cout1_bwd_data_f32(const float* dy, const float* w, float* dx, int ldx, int B, int H, int W, int Cin,
                                   emd_stream_t stream);

/* Gradients of emd_resize_bilinear_f32 (dx [B,Hi,Wi,C] from dy [B,Ho,Wo,C]) and emd_avgpool2x2_f32. */
int emd_resize_bilinear_bwd_f32(const float* dy, int ldd, float* dx, int ldx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                emd_stream_t stream);
int emd_avgpool2x2_bwd_f32(const float* dy, int ldd, float* dx, int ldx, int B, int H, int W, int C, emd_stream_t stream);

/* y += alpha*x over [npix, C] (gradient fan-in where a tensor feeds several layers). */
int emd_axpy_f32(const float* x, int ldx, float* y, int ldy, long npix, int C, float alpha, emd_stream_t stream);

/* _tower_fn's loss (:768-775): mse = mean((out-truth)^2); loss = 1000*mse if mse < 1e-3 else sqrt(1000*mse)
 * (weight_decay = 0, :117).  result3 (device) = {mse, loss, f}; dout (may be NULL) = grad_scale * dloss/dout.
 * workspace: emd_denoise_loss_workspace_bytes() bytes.  No host synchronisation. */
size_t emd_denoise_loss_workspace_bytes(void);
int emd_denoise_loss_f32(const float* out, const float* truth, long n, float grad_scale, float* result3, float* dout,
                         void* workspace, emd_stream_t stream);

/* tf.train.MomentumOptimizer(lr, momentum, use_nesterov=True) (:1064-1066) on a flat parameter vector:
 * g = grad*grad_scale (1/number of gradient sets, :1040); accum = momentum*accum + g; param -= lr*(g + momentum*accum). */
int emd_nesterov_step_f32(float* param, const float* grad, float* accum, long n, float lr, float momentum, float grad_scale,
                          emd_stream_t stream);

/* ================================================================================================
 * Graph G: the in-filling GAN's generator (misc_py/gan-infilling-100.py:133-374), inference.  Its pointwise halves,
 * SAME-padded separable convs and resizes are the graph-D entry points with act = EMD_ACT_LEAKY; what follows is what
 * only G has.
 * ================================================================================================ */

/* Depthwise 3x3 over the tf.pad(REFLECT, 1) input, VALID, stride 1 or 2 -- the depthwise half of
 * strided_conv_block(pad_size=(1,1)) (:205-243): output (oy,ox) reads input rows oy*s-1..oy*s+1, index -1 -> 1,
 * H -> H-2.  x [B,H,W,C]; y [B,(H-1)/s+1,(W-1)/s+1,C]; w [3][3][C]. */
int emd_dw3x3_reflect_f32(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C, int stride,
                          emd_stream_t stream);

/* The first layer (:343-347): 7x7 separable conv on the 1-channel image, reflect-padded by 3, VALID:
 * y[pix][n] = act(d[pix]*a[n] + shift[n]), d = 7x7 depthwise (w49), a = pointwise weight * folded BN scale;
 * act != 0: leaky_relu(0.2).  x [B,H,W]; y [B,H,W,Cout] pixel stride ldy; Cout/4 must divide 64. */
int emd_cin1_k7_reflect_f32(const float* x, const float* w49, const float* a, const float* shift, float* y, int ldy, int B,
                            int H, int W, int Cout, int act, emd_stream_t stream);

/* The last conv (:362-369): tf.pad(REFLECT,1) + slim.conv2d(1, 3, VALID) + bias, no activation.
 * x [B,H,W,Cin]; w [3][3][Cin]; y [B,H,W]. */
int emd_conv3x3_cout1_reflect_f32(const float* x, int ldx, const float* w, float bias, float* y, int B, int H, int W, int Cin,
                                  emd_stream_t stream);

/* _instance_norm with its fixed unit affine (:140-148) + tf.tanh (:372) on a 1-channel batch:
 * y = tanh((x - mean[b]) * rsqrt(var[b] + eps)); mean/var per image (emd_bn_stats_f32 with C = 1 on each image). */
int emd_instnorm_tanh_f32(const float* x, const float* mean, const float* var, float* y, int B, long npix_img, float eps,
                          emd_stream_t stream);

/* The decoder pair "separable conv + 1x1 residual projection of the SAME input" in one launch
 * (machine_learning/denoiser.py:356-359, :368-371, :380-383: deconv*_a = strided_conv_block(concat) and
 * residual*_d = conv_block_not_sep(concat, kernel_size=1)):
 *   y  = act(pointwise(depthwise3x3(x)) * scale1 + shift1)     as emd_sep3x3_fused_f32 (stride 1, TF SAME)
 *   y2 = relu6((x . W2) * scale_b + shift_b)                    W2 packed as for emd_conv1x1_f32; bias and BN folded into scale_b / shift_b
 * The 384- / 128-channel input -- the largest tensors of the decoder -- is read from HBM once instead of twice.
 * Split-bf16 precision.  Supported (emd_sep3x3_dual_supported): W%16==0, Cin%32==0, both Cout%4==0 and <= 128,
 * H%8==0 (H%4==0 when either output has more than 64 channels).  emd_sep3x3_dual_preferred: 1 where the one-launch form is also the
 * faster route (always up to 64 | 64 channels; wider only for H%8==0, W%32==0) -- what a graph executor should ask. */
int emd_sep3x3_dual_supported(int H, int W, int Cin, int Cout, int Cout2);
int emd_sep3x3_dual_preferred(int H, int W, int Cin, int Cout, int Cout2);
int emd_sep3x3_dual_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                        const float* scale1, const float* shift1, float* y, int ldy, const uint16_t* w2hi,
                        const uint16_t* w2lo, const float* scale_b, const float* shift_b, float* y2, int ldy2, int B, int H,
                        int W, int Cin, int Cout, int Cout2, int act, emd_stream_t stream);

/* The separable conv of the 728-channel flow as ONE kernel (csrc/sep_gemm.hip): the depthwise 3x3 stage (stride 1, TF SAME) is
 * computed per 32-channel K step inside the pointwise GEMM -- no depthwise launch, no intermediate tensor, 128-pixel x
 * 384-channel workgroup tiles (machine_learning/denoiser.py:110-136 as used by :297-302, :312-325).  Arguments as
 * emd_sep3x3_fused_f32 (split-bf16 precision only).  Supported (emd_sep3x3_gemm_supported): H%4==0, W%32==0,
 * 256 <= Cin <= 4096, 384 < Cout <= 768, Cin%4==0, Cout%4==0. */
int emd_sep3x3_gemm_supported(int H, int W, int Cin, int Cout);
int emd_sep3x3_gemm_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                        const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                        const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                        emd_stream_t stream);

/* emd_sep3x3_fused_f32 with the depthwise stage reading the tf.pad(REFLECT, 1) border instead of zeros: the
 * stride-1 strided_conv_block(pad_size=(1,1)) of graph G (:205-243).  Same arguments and support rule. */
int emd_sep3x3_fused_reflect_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                                 const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                 const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout,
                                 int act, int precision, emd_stream_t stream);

/* emd_dw3x3_reflect_f32 on a GENERATED input: the C-channel tensor the depthwise conv reads is act(d[pixel] * gen_a[c] + gen_t[c]),
 * d one value per pixel with pitch ldd floats (channel 0 of a 4-channel emd_cin1_k7_reflect_f32 output with a = (1,0,0,0), no
 * activation); leaky_act != 0: tf.nn.leaky_relu(alpha 0.2).  The generator's first layer feeding its second
 * (misc_py/gan-infilling-100.py:343-349) without the [B,H,W,C] tensor in memory; bit-identical to the two calls it replaces. */
int emd_dw3x3_reflect_gen_f32(const float* d, int ldd, const float* gen_a, const float* gen_t, int leaky_act, const float* w,
                              float* y, int ldy, int B, int H, int W, int C, int stride, emd_stream_t stream);

/* emd_sep3x3_fused_f32 on a GENERATED input: the Cin-channel tensor the depthwise stage reads is
 * act_gen(d[pixel] * gen_a[c] + gen_t[c]) with d a one-value-per-pixel tensor of pitch ldd floats (e.g. channel 0 of a
 * 4-channel emd_cin1_f32 output with a = (1,0,0,0), no activation).  It is the layer after the one fed by the 1-channel
 * micrograph (cnn0 -> cnn0_last, machine_learning/denoiser.py:252-255): cnn0 = relu6(BN(depthwise(img) (x) pointwise)) is
 * rebuilt in registers and never written to memory.  gen_act is an EMD_ACT_* code (0 none, 1 relu6, 2 relu, 4 leaky 0.2);
 * reflect as in emd_sep3x3_fused_reflect_f32.  Bit-identical to emd_cin1_f32 followed by emd_sep3x3_fused_f32. */
int emd_sep3x3_fused_gen_f32(const float* d, int ldd, const float* gen_a, const float* gen_t, int gen_act, const float* dw,
                             const uint16_t* whi, const uint16_t* wlo, const float* scale1, const float* shift1,
                             const float* scale2, const float* shift2, const float* res, int ldres, float* y, int ldy, int B,
                             int H, int W, int Cin, int Cout, int act, int precision, int reflect, emd_stream_t stream);

/* Discriminator head (misc_py/gan-infilling-100.py:560-567, :708): a fully connected layer to ONE output per row,
 * y[b] = x[b,:K].w + bias (x row stride ldx), and output = sigmoid(max(small, medium, large)). */
int emd_fc_rows_f32(const float* x, int ldx, const float* w, float bias, float* y, int B, int K, emd_stream_t stream);
int emd_max3_sigmoid_f32(const float* a, const float* b, const float* c, float* y, int n, emd_stream_t stream);

/* ================================================================================================
 * Training side of graph G (misc_py/gan-infilling-100.py:982-1088 towers, :1378-1379 / :1429-1431 optimizers).  The
 * convolution gradients are the graph-D' entry points with mask 3 (leaky_relu) in emd_bn_bwd_{reduce,apply}_f32.
 * ================================================================================================ */

/* Head of one tower (batch_size 1, :74): out = sigmoid(max(logit3)); mode 0: discriminator loss
 * -log(clip(1-|label-out|, 1e-8, 1-1e-8)) (:1080); mode 1: generator loss -log(clip(out, 1e-8, 1)) (:1037).
 * result2 = {out, loss}; dlogit3 = grad_scale * dloss/dlogit (arg-max branch only).  All device pointers. */
int emd_gan_head_f32(const float* logit3, float label, int mode, float grad_scale, float* result2, float* dlogit3,
                     emd_stream_t stream);
/* Backward of emd_fc_rows_f32 for one row: dw[k] += x[k]*g, *db += g, dx[k] = w[k]*g with g = *dlogit (device). */
int emd_fc_row_bwd_f32(const float* x, const float* w, const float* dlogit, float* dw, float* db, float* dx, int K,
                       emd_stream_t stream);
/* Gradient of tf.reduce_mean(x, [1,2]) (:578): y[p][c] = v[c]*alpha for every pixel p of [npix, C]. */
int emd_bcast_rows_f32(const float* v, float* y, int ldy, long npix, int C, float alpha, emd_stream_t stream);
/* out[0] = |scale*x|^2 of a flat vector (double accumulation): the global norm of clip_gradients_by_norm.
 * workspace: emd_sumsq_workspace_bytes() bytes. */
size_t emd_sumsq_workspace_bytes(void);
int emd_sumsq_f32(const float* x, long n, float scale, float* out, void* workspace, emd_stream_t stream);
/* tf.train.AdamOptimizer(lr, beta1 = 0.5) inside tf.contrib.estimator.clip_gradients_by_norm(., clip_norm):
 * g = grad*grad_scale*clip_norm/max(sqrt(*gnorm_sq), clip_norm) (gnorm_sq NULL: no clipping); m, v moment updates;
 * param -= lr_t*m/(sqrt(v)+eps), lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) computed by the caller. */
int emd_adam_step_f32(float* param, const float* grad, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                      float eps, float grad_scale, const float* gnorm_sq, float clip_norm, emd_stream_t stream);
/* The same with lr_t (which changes every step through the bias correction) read from DEVICE memory, so that the
 * optimizer step can live inside a replayed hipGraph. */
int emd_adam_step_dev_f32(float* param, const float* grad, float* m, float* v, long n, const float* lr_t_dev, float beta1,
                          float beta2, float eps, float grad_scale, const float* gnorm_sq, float clip_norm,
                          emd_stream_t stream);

/* Generator-side training (the generator tower, :982-1046; its batch norms stay on MOVING statistics while the tower
 * gradients are evaluated, :1667).
 * Reflect-padded depthwise 3x3 backward (shapes as emd_dw3x3_reflect_f32; dw += ; dx written) and the same for the last
 * 3x3 conv to one channel (dy [B,H,W]). */
int emd_dw3x3_reflect_wgrad_f32(const float* x, int ldx, const float* dy, int ldd, float* dw, int B, int H, int W, int C,
                                int stride, emd_stream_t stream);
int emd_dw3x3_reflect_bwd_data_f32(const float* dy, int ldd, const float* w, float* dx, int ldx, int B, int H, int W, int C,
                                   int stride, emd_stream_t stream);
int emd_conv3x3_cout1_reflect_wgrad_f32(const float* x, int ldx, const float* dy, float* dw, int B, int H, int W, int Cin,
                                        emd_stream_t stream);
int emd_conv3x3_cout1_reflect_bwd_data_f32(const float* dy, const float* w, float* dx, int ldx, int B, int H, int W, int Cin,
                                           emd_stream_t stream);
/* First layer for training: d4[pix] = (7x7 reflect depthwise of the 1-channel image, 0, 0, 0) -- the pointwise half
 * then runs as a K = 4 GEMM -- and dw49[t] += sum x[reflect(p + t)] * dd4[p][0]. */
int emd_dw7_c1_reflect_f32(const float* x, const float* w49, float* d4, int B, int H, int W, emd_stream_t stream);
int emd_dw7_c1_reflect_wgrad_f32(const float* x, const float* dd4, float* dw49, int B, int H, int W, emd_stream_t stream);
/* g = dy * (1 - y^2): tf.tanh (:372). */
int emd_tanh_bwd_f32(const float* dy, const float* y, float* g, long n, emd_stream_t stream);
/* One feature-matching term (:1027-1035): *loss_acc += weight*mean|a-b|; dy (accumulate ? += : =) weight*sign(a-b)/n. */
int emd_l1_feature_f32(const float* a, const float* b, long n, float weight, float* dy, int accumulate, float* loss_acc,
                       emd_stream_t stream);
/* Gradient of one crop of get_multiscale_crops (:957-980): channel 0 of dcrop [n,n,ldc] is added into dimg [S,S] at
 * the mirror image of padded position (y0+i, x0+j) (padding 3S/4, REFLECT). */
int emd_crop_scatter_f32(const float* dcrop, int ldc, float* dimg, int y0, int x0, int n, int S, emd_stream_t stream);
/* The same with the offset pair (y0, x0) read from DEVICE memory: a captured hipGraph is replayed with new crops. */
int emd_crop_scatter_dev_f32(const float* dcrop, int ldc, float* dimg, const int* yx_dev, int n, int S, emd_stream_t stream);
/* Inference-mode double batch norm of a generator separable conv: (scale, shift) of the forward affine and the vectors
 * its parameter gradients need (see csrc/gan_train.hip); emd_bn_infer_grads_f32 adds them (s1 = sum g,
 * t1 = sum g*(r-mu1)/s1, t2 = sum g*(z1-mu2)/s2 from emd_bn_bwd_reduce_f32). */
int emd_bn_infer_fold2_f32(const float* g1, const float* b1, const float* m1, const float* v1, const float* g2,
                           const float* b2, const float* m2, const float* v2, float eps, int C, float* scale, float* shift,
                           float* mprime, float* rprime, float* rstd1, float* a2, emd_stream_t stream);
int emd_bn_infer_grads_f32(const float* s1, const float* t1, const float* t2, const float* a2, int C, float* dg1, float* db1,
                           float* dg2, float* db2, emd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training input functions on the device (csrc/input_ops.hip).  They replace the numpy bodies of
 * misc_py/denoiser-multi-gpu.py:783-870 (get_scale, gen_lq, scale0to1, flip_rotate, preprocess, record_parser), which the
 * reference runs in tf.py_func threads.  Images are dense float32 [B][npix] (or [B][H][W]); all pointers are device
 * pointers.  Random numbers: Philox4x32-10 keyed by `seed`, indexed by (first_image + b, pixel, draw): results do not depend
 * on the launch geometry or on how a data set is cut into batches.  The reference's own stream (numpy Mersenne-Twister
 * re-seeded from itself, :791) is irreproducible by construction; distributions and deterministic formulas are kept. */
/* Raw generator output, for known-answer tests: out[4*i..4*i+3] = Philox4x32-10(counter = (counter0 + i, 0, 0), key = seed). */
int emd_philox4x32_u32(unsigned* out, long n4, unsigned long long seed, unsigned long long counter0, emd_stream_t stream);
/* get_scale (:783-784): scale[b] = 25 + Exp(mean 75). */
int emd_get_scale_f32(float* scale, int B, unsigned long long seed, unsigned long long first_image, emd_stream_t stream);
/* The draw of flip_rotate (:833): choice[b] = int(8 * U[0,1)). */
int emd_d4_choices_i32(int* choice, int B, unsigned long long seed, unsigned long long first_image, emd_stream_t stream);
/* flip_rotate (:830-851) with the element of D4 given per image (choice_dev[b] in 0..7, device memory; NULL = identity):
 * 0 identity, 1-3 np.rot90(img, k), 4 np.flip(img, 0), 5 np.flip(img, 1), 6 / 7 np.flip(np.rot90(img, 1), 0 / 1).
 * Square images only (H == W).  fix_nonfinite != 0 also applies preprocess()'s NaN / Inf -> 0.5 (:855-856).  x != y. */
int emd_flip_rotate_f32(const float* x, float* y, int B, int H, int W, const int* choice_dev, int fix_nonfinite,
                        emd_stream_t stream);
/* Bytes of scratch emd_gen_lq_f32 / emd_minmax_images_f32 need for a batch of B images of npix pixels (16-byte aligned). */
size_t emd_input_workspace_bytes(int B, long npix);
/* Per-image minimum and maximum (the reductions of scale0to1, :817-828); exact. */
int emd_minmax_images_f32(const float* x, int B, long npix, float* mn, float* mx, void* workspace, emd_stream_t stream);
/* scale0to1 (:817-828): y = (x - min) / (max - min) in float32, a constant image becomes 0.5; y may alias x. */
int emd_scale0to1_images_f32(const float* x, float* y, int B, long npix, const float* mn, const float* mx,
                             emd_stream_t stream);
/* gen_lq + the truth rescale of record_parser (:787-799, :861-870): counts = Poisson(img * scale[b]) (exact samplers in
 * double precision: CDF inversion below a rate of 10, Hoermann's PTRS rejection method above), lq = scale0to1(counts)
 * evaluated in float64 and rounded to float32 as numpy does for integer counts, truth = float32(mean(lq) / mean(img)) * img
 * (truth may be NULL).  counts_out (int32 [B][npix]) may be NULL; when given it receives the raw counts. */
int emd_gen_lq_f32(const float* img, const float* scale, float* lq, float* truth, int* counts_out, int B, long npix,
                   unsigned long long seed, unsigned long long first_image, void* workspace, emd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Graph D as a native executor (csrc/graph_exec.hip): architecture() of machine_learning/denoiser.py:58-398 for
 * a host that is not Python.  emd_graph_create takes the weights as HOST float32 arrays keyed by TensorFlow variable name
 * (the names tf.train.Saver stores under scope 'nn', :514: nn/SeparableConv2d[_k]/{depthwise_weights,pointwise_weights},
 * nn/SeparableConv2d[_k]/BatchNorm/{beta,gamma,moving_mean,moving_variance}, nn/BatchNorm[_k]/..., nn/Conv[_k]/{weights,biases},
 * nn/Conv2d_transpose[_k]/{weights,biases}; 658 variables), folds the inference batch norms (float64), packs the matrix-core
 * weights and uploads them into device memory owned by the handle.  emd_graph_run launches the whole forward pass on `stream`:
 * x, y device float32 [B,S,S,1] (S a multiple of 16; no output clip, :396), activations in the caller's `workspace` (device
 * memory, emd_graph_workspace_bytes(g, B, S) bytes).  Same kernels in the same order as emdenoise.denoiser.DenoiserEngine:
 * bit-identical results.  variant: 0 = graph D; 1 = graph D', the inference graph of the training twin
 * misc_py/denoiser-multi-gpu.py:200-540 (phase=False): tf.layers variable names (nn/conv2d[_k]/{kernel,bias}, nn/conv2d_transpose[_k]/...,
 * the ASPP convs nn/{1x1,lowRate,mediumRate,highRate,imageLevel,pellet}), dense dilated 3x3 ASPP branches, a real image-level
 * branch, output clipped to [0,1] (:534-538); 2 = graph X, the Xception autoencoder misc_py/modified_Xception.py:194-654 (variables under
 * scope "pellet": pellet/conv2d[_k]/{kernel,bias}, pellet/SeparableConv2d[_k]/{depthwise_weights,pointwise_weights,BatchNorm/...},
 * pellet/conv2d_transpose[_k]/..., pellet/{1x1,lowRate,mediumRate,highRate,imageLevel}/..., pellet/BatchNorm[_k]/...; S a multiple of 64;
 * the separable convs' norms run on the statistics of the batch handed to emd_graph_run; output clipped to [0,1]), same kernels as
 * emdenoise.xception.XceptionEngine, bit-identical; 3 = graph G's generator misc_py/gan-infilling-100.py:133-374 (variables under
 * "GAN/Gen" and "GAN/Gen/reg"; x = the 1/64-sampled image with missing pixels -1, S a multiple of 16, >= 32; y in (-1,1)), same kernels
 * as emdenoise.gan.GeneratorEngine, bit-identical.
 * A handle is NOT re-entrant: emd_graph_run calls on one handle must be serialised by the caller (one stream, one thread at a
 * time) -- the fork / join events and the side streams of the two-streams form belong to the handle; use one handle per
 * concurrent stream (the weights are ~100 MB).  emd_graph_workspace_bytes returns the larger of the two launch forms' needs, so a
 * size asked for before emd_graph_set_two_streams stays valid after it.  emd_graph_create returns EMD_E_ALLOC when a device
 * allocation or upload fails (EMD_E_INVALID: a missing / mis-sized variable). */
typedef struct emd_graph emd_graph_t;
int emd_graph_create(emd_graph_t** graph, int variant, int n_vars, const char* const* names, const float* const* host_data,
                     const long* counts);
size_t emd_graph_workspace_bytes(emd_graph_t* graph, int B, int S);
int emd_graph_run(emd_graph_t* graph, const float* x, float* y, int B, int S, void* workspace, size_t workspace_bytes,
                  emd_stream_t stream);
void emd_graph_destroy(emd_graph_t* graph);
/* Launch-order option (speed only, same bits): on != 0 runs the 1/16-resolution flow (denoiser.py:312-325) of an even batch as two
 * halves on two internal streams, forked from and joined to `stream` (capturable), as the Python engine does.  If the side
 * streams cannot be created the run falls back to the single-stream sequence (same results).  Default off: measured slower from a host that enqueues as fast as C
 * does. */
int emd_graph_set_two_streams(emd_graph_t* graph, int on);

/* ------------------------------------------------------------------------------------------------
 * Host utility (no GPU): CRC-32C (Castagnoli) of a HOST buffer, continuing from `crc` (0 to start).
 * Used by the TFRecord reader (emdenoise.input_pipeline) for the container that
 * misc_py/TFRecord_creator.py:57-85 writes with tf.python_io.TFRecordWriter. */
uint32_t emd_crc32c(const void* data_host, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* EMDENOISE_H */
