// Shared host-side helpers for libemdenoise.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/emdenoise.h"

namespace emd {

// thread-local last-error text behind emd_last_error()
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

inline int fail(int code, const char* what) {
    set_error("%s", what);
    return code;
}

// Call after every kernel launch: turns a launch-time HIP error into EMD_E_LAUNCH.
inline int check_launch(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", kernel, hipGetErrorString(e));
        return EMD_E_LAUNCH;
    }
    return EMD_OK;
}

// dw_misc.hip: the rolling depthwise kernel with REFLECT borders (graph G), used by gan_ops.hip
int launch_dw3x3_reflect_roll(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C, bool split,
                              hipStream_t st);

// dw_misc.hip: the rolling 3x3 conv to one channel over the REFLECT-padded input, y = conv + bias (W % (256 / Cin) == 0)
int launch_conv3x3_cout1_reflect_roll(const float* x, int ldx, const float* w, float bias, float* y, int B, int H, int W, int Cin,
                                      hipStream_t st);

struct BnFoldArgs;   // bn_chain_dev.hpp
struct BnPrepArgs;
int bn_prep_args(const emd_bn_bwd_prep_t* p, BnPrepArgs* out);        // bn_train.hip: the public structs -> device argument blocks, checked
int bn_fold_args(const emd_bn_train_fold_t* p, BnFoldArgs* out);
// bn_train.hip; prep != NULL: bn_bwd_prep_kernel's per-channel step in the same launch (npix: pixels per reduction)
int launch_chan_reduce_final(const double* part, int nslab, int C, int B, float* s1, float* s2, hipStream_t st, const BnPrepArgs* prep = nullptr,
                             long npix = 0);
int launch_bn_stats_final(const double* part, int nslab, int C, long npix, float* mean, float* var, hipStream_t st,
                          const float* gamma = nullptr, const float* beta = nullptr, float eps = 0.f, float* scale = nullptr,
                          float* shift = nullptr, int images = 1, const BnFoldArgs* train_fold = nullptr);   // scale != NULL: the norm is folded in the same launch; images > 1 (no fold): per-image statistics

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;  // CDNA4 wavefront

// Development knobs: the ONLY process-global mutable state of the library, set through emd_debug_knob (include/emdenoise_dev.h),
// never read from the environment.  Defaults = the measured-best path; they change speed, not results (DESIGN.md lists the exceptions).
struct Knobs {
    int sep_pipe = 1;        // 1: the LDS-DMA pipelined fused separable conv (sep_pipe.hip) where it covers the shape; 0: sep_fused.hip
    int sep_pipe2 = 0;       // the software-pipelined kernel (sep_pipe2.hip, round 4): 0 never (default: measured at parity standalone on its best shape and slower inside graph D, profiles/r04_experiments.txt 3), 1 the two-output launches of more than 64 columns, 2 wherever it has an instance (the parity tests)
    int sep_gen_pipe = 0;    // generated-input fused separable conv (cnn0_last): 1 = sep_pipe.hip's 4-wave instance (round 4; same bits), 0 (default) = sep_fused.hip's register-staged kernel -- measured 814-841 us (0) against 833-849 (1) at [32,512,512,64 -> 64], profiles/r04_experiments.txt 9
    int sep_mode = -1;       // sep_pipe schedule of the one-output instances: -1 = rule, 0 / 1 = the patch two / one steps ahead
    int sep_nw = 0;          // sep_pipe waves per workgroup: 0 = rule (4 wherever there is an instance), 8 (8 x 32 tiles, one workgroup per CU) or 4 (8 x 16 tiles, two per CU; <= 128 output channels, 64 | 64 for two outputs)
    int sep_ablate = 0;      // sep_pipe phase ablation bits for timing experiments (results are wrong when non-zero)
    int sep_tpw = 0;         // tiles per workgroup of the fused separable convs (0 = rule)
    int sep_xcd = 1;         // one contiguous run of tiles per XCD
    int sep_wide = 1;        // sep_fused 256-column single-output form: 0 never, 1 Cin <= 256, 2 whenever it fits
    int sep_wres = 1;        // sep_fused 64-column instances keep the pointwise weights resident in LDS
    int epi_width = 0;       // dev: dwords a lane stores at a time in the patch-resident kernels' epilogue: 1 (a lane = one channel, as the MFMA leaves it) or 4 (after a 4 x 4 transpose inside lane quads); 0 = the kernel's rule (1, except sep_pipe with a residual on > 128 columns)
    int deconv_direct = 3;   // one-launch transposed conv: 3 = the patch-resident kernel (deconv_pipe.hip) where it covers the shape (H % 8 == 0, W % 32 == 0), else as 1; 1 = GEMM form, epilogue from the accumulators (deconv4_split_kernel); 2 = the same on 128-row tiles, two workgroups per CU; 0 = LDS-staged epilogue
    int nt_mask = 7;         // non-temporal output stores: bit 0 split32 convolutions, bit 1 fused separable conv, bit 2 pointwise GEMM
    int dw_xcd = 1;          // XCD-contiguous tile order in the depthwise kernels
    int dw_th = 0;           // strip height of the rolling depthwise kernel (0 = rule)
    int split_narrow = 1;    // pointwise split32 GEMM: 128 x 64 tiles where 128 x 128 tiles leave CUs idle (0 = never)
    int split_lead = 2;      // pointwise split32 GEMM (16x16x32 form): DMA issued 2 (default) or 1 K steps ahead on the same three stages (same bits)
    int split_wide = 0;      // pointwise split32 GEMM: 256 x 192 tiles (gemm_split16_wide_kernel) where they fill the chip: 0 never (default: slower in graph D), 1 = 8 waves of 64 x 96, 2 = 4 waves of 128 x 96
    int conv3_pipe = 1;      // dense 3x3 conv (stride 1, rate 1, H % 8 == 0, W % 32 == 0) on the patch-resident kernel (conv3_pipe.hip): 1 = up to 192 output channels, 2 = any width, 0 = never
    int split_variant = -1;  // pointwise split32 GEMM pipeline variant (-1 = dispatch rule)
    int sep_stamp_wave = 0;  // sep_pipe2 in-kernel stamps: the wave (0-7) that reports
    int wgrad_msplit = 0;    // weight-gradient GEMM: slices of the pixel dimension (each adds its partial sums atomically): 0 = rule, n = at most n
    int wgrad_tile = 0;      // weight-gradient GEMM: 0 = rule (128 where a dimension exceeds 64), 64 = 64 x 64 tiles everywhere
    long long* sep_stamps = nullptr;   // device buffer for the in-kernel phase stamps of the fused separable convs
};
extern Knobs g_knobs;

// Per-channel reductions over [npix, C] (batch statistics, batch-norm backward sums) run as "slabs" of rows, one
// workgroup per (64 channels, slab), combined by a second small kernel.  At most 512 slabs, at least 64 rows each:
// small maps (a 32x32 tower) still spread over tens of workgroups, large ones give every thread a few hundred rows.
inline long reduce_rows_per_slab(long npix) {
    long r = (npix + 511) / 512;
    return r < 64 ? 64 : r;
}
inline long reduce_slabs(long npix) {
    const long r = reduce_rows_per_slab(npix);
    return (npix + r - 1) / r;
}

}  // namespace emd

#define EMD_REQUIRE(cond, code, msg)          \
    do {                                      \
        if (!(cond)) return emd::fail(code, msg); \
    } while (0)
