/* emdenoise_dev.h -- DEVELOPMENT hooks of libemdenoise.so.  NOT part of the drop-in boundary (include/emdenoise.h).
 *
 * These are the only places where the library keeps process-global mutable state.  They default to "off", change
 * speed only (kernel variant selection, in-kernel time stamps), never results beyond what DESIGN.md states for the
 * variant, and exist for the A/B tools under tools/ and for bench.py's isolated-GEMM comparison.  A product host must
 * not call them.  Every symbol the shared library exports with an emd_ prefix is declared either in emdenoise.h or
 * here (tests/test_abi.py checks both directions).
 *
 * Nothing in the library reads the environment.  The kernel-selection knobs are fields of one struct (csrc/emd_common.hpp, emd::Knobs),
 * set by name through emd_debug_knob; defaults = the measured-best path:
 *   sep_pipe (1)        1: the LDS-DMA pipelined fused separable conv (csrc/sep_pipe.hip) where it covers the shape, 0: csrc/sep_fused.hip
 *   sep_mode (-1)       sep_pipe schedule of the one-output instances: -1 = rule, 0 / 1 = the patch requested two / one steps ahead
 *   sep_nw (0)          sep_pipe waves per workgroup: 0 = rule (4 wherever there is an instance), 8 (8 x 32 pixel tiles, one workgroup per CU)
 *                       or 4 (8 x 16 tiles, two per CU: up to 128 output columns, 64 | 64 for two outputs)
 *   sep_ablate (0)      sep_pipe timing experiments: bit 0 no depthwise stage, 1 no MFMA stage, 2 no epilogue, 3 no patch DMA after the
 *                       prologue, 4 no weight DMA after it, 5 no residual loads -- RESULTS ARE WRONG when non-zero
 *   sep_tpw (0)         tiles per workgroup of the fused separable convs (0 = rule)
 *   sep_xcd (1)         0 = launch-order tiles instead of one contiguous run of tiles per XCD
 *   sep_wide (1)        sep_fused 256-column single-output form: 0 never, 1 Cin <= 256, 2 whenever it fits
 *   sep_wres (1)        0 = per-chunk pointwise weight loads in sep_fused's 64-column instances (default: resident in LDS)
 *   epi_width (0)       epilogue of sep_pipe / conv3_pipe / deconv_pipe: 1 = a lane keeps its channel and stores one dword per pixel, 4 = 4 x 4 transpose
 *                       inside lane quads, then 16 bytes per lane; 0 = the kernel's rule (same values either way)
 *   deconv_direct (3)   one-launch transposed conv: 3 = the patch-resident kernel (csrc/deconv_pipe.hip; sums chunk-major: last-bit
 *                       differences to the GEMM forms) where H % 8 == 0 and W % 32 == 0, else as 1; 1 = GEMM form with the epilogue straight
 *                       from the accumulators; 2 = the same on 128-row tiles, two workgroups per CU; 0 = LDS-staged epilogue
 *   nt_mask (7)         non-temporal output stores: bit 0 split32 convolutions, bit 1 sep_fused, bit 2 pointwise split32 GEMM
 *   dw_xcd (1)          depthwise kernels: 0 = launch-order tiles, 1 = XCD-contiguous up to 128 x 128 maps, 2 = always
 *   dw_th (0)           strip height of the rolling depthwise kernel (0 = rule)
 *   split_narrow (1)    pointwise split32 GEMM: 128 x 64 tiles for the small batches whose 128 x 128 tiles leave CUs idle (0 = never)
 *   conv3_pipe (1)      dense 3x3 conv, stride 1, rate 1, H % 8 == 0, W % 32 == 0, <= 256 output channels (2: any width): the patch-resident kernel
 *                       (conv3_pipe.hip; sums chunk-major, taps inside: last-bit differences to the tap-major GEMM), 0 = gemm_split_conv_kernel
 *   split_wide (0)      pointwise split32 GEMM: 256 x 192 tiles where they fill the chip (N = 728: four column tiles, no half round): 0 never
 *                       (default: same bits, slower inside graph D), 1 = 8 waves of 64 x 96, 2 = 4 waves of 128 x 96
 *   split_variant (-1)  pointwise split32 GEMM pipeline variant (-1 = dispatch rule)
 *   split_lead (2)      pointwise split32 GEMM, 16x16x32 form: the DMA of tile kt + 3 issued in step kt into the stage of tile kt (whose fragments
 *                       are in registers): two tiles in flight on three stages; 1 = one step ahead (round 2's schedule; same bits)
 */
#ifndef EMDENOISE_DEV_H
#define EMDENOISE_DEV_H
#ifdef __cplusplus
extern "C" {
#endif

/* Set one of the knobs above by name.  Returns 0, or -1 for an unknown name. */
int emd_debug_knob(const char* name, long value);
/* Force the pipeline variant of emd_conv1x1_split32_f32 (csrc/gemm_split.hip); -1 restores the dispatch rule. */
void emd_debug_split_variant(int v);
/* Device buffer that the split32 GEMM writes s_memtime phase stamps into (NULL = off). */
void emd_debug_split_stamps(void* device_buf);
/* Device buffer that the fused separable conv writes s_memtime phase stamps into (NULL = off). */
void emd_debug_sep_stamps(void* device_buf);
/* The same for emd_sep3x3_gemm_f32 (csrc/sep_gemm.hip): 8 x int64 per workgroup = cycle sums of {own DMA wait, barrier a,
 * depthwise stage, barrier b, fragment reads + DMA issue + MFMAs, -, end stamp, -}. */
void emd_debug_sepgemm_stamps(void* device_buf);


/* On-box peak micro-benchmarks (csrc/dev_bench.hip), timed by bench.py with HIP events (SURVEY.md 8d asks for the measured
 * peaks next to the nominal ones).  emd_stream_t is hipStream_t; return codes as in emdenoise.h. */
/* dst[i] = src[i], n floats (multiple of 4), 16-byte aligned: the read + write stream-copy roof. */
int emd_debug_stream_copy_f32(const float* src, float* dst, long n, void* stream);
/* `workgroups` x 4 waves each issue iters x 32 back-to-back v_mfma_f32_32x32x16_bf16 (32768 flop each) on the 4 KiB of
 * bf16 operands at `ops`; `out` (workgroups x 256 floats) exists to keep the chain alive. */
int emd_debug_mfma_peak_bf16(const void* ops, float* out, int workgroups, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMDENOISE_DEV_H */
