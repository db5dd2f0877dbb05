/* emdenoise_dev.h -- DEVELOPMENT hooks of libemdenoise.so.  NOT part of the drop-in boundary (include/emdenoise.h).
 *
 * These are the only places where the library keeps process-global mutable state.  They default to "off", change
 * speed only (kernel variant selection, in-kernel time stamps), never results beyond what DESIGN.md states for the
 * variant, and exist for the A/B tools under tools/ and for bench.py's isolated-GEMM comparison.  A product host must
 * not call them.  Every symbol the shared library exports with an emd_ prefix is declared either in emdenoise.h or
 * here (tests/test_abi.py checks both directions).
 *
 * Environment knobs read once per process by the same translation units (same rule: speed only, default = the measured-best path):
 *   EMD_SPLIT_VARIANT  csrc/gemm_split.hip  pointwise split32 GEMM pipeline variant (-1 = dispatch rule)
 *   EMD_SEP_TPW        csrc/sep_fused.hip   tiles per workgroup of the fused separable conv (0 = rule)
 *   EMD_DW_TH          csrc/dw_misc.hip     strip height of the rolling depthwise kernel (0 = rule)
 *   EMD_NT             gemm_split / sep_fused   mask of the non-temporal output stores (default 7: bit 0 split32 convolutions,
 *                                           bit 1 fused separable conv, bit 2 pointwise split32 GEMM)
 *   EMD_SEP_XCD        csrc/sep_fused.hip   0 = launch-order tiles instead of one contiguous run of tiles per XCD
 *   EMD_SEP_WIDE       csrc/sep_fused.hip   256-column single-output form: 0 never, 1 (default) Cin <= 256, 2 whenever it fits
 *   EMD_SEP_WRES       csrc/sep_fused.hip   0 = per-chunk pointwise weight loads in the 64-column instances (default: resident in LDS)
 *   EMD_DW_XCD         csrc/dw_misc.hip     0 = launch-order tiles in the depthwise kernels
 */
#ifndef EMDENOISE_DEV_H
#define EMDENOISE_DEV_H
#ifdef __cplusplus
extern "C" {
#endif

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
