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

int launch_bn_stats_final(const double* part, int nslab, int C, long npix, float* mean, float* var, hipStream_t st,
                          const float* gamma = nullptr, const float* beta = nullptr, float eps = 0.f, float* scale = nullptr,
                          float* shift = nullptr);   // scale != NULL: the norm is folded in the same launch

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;  // CDNA4 wavefront

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
