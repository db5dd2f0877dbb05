// Dev microbenchmark: issue rate of v_exp_f32 / v_rcp_f32 vs v_fma_f32 / v_pk_fma_f32 on gfx950, and whether they overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    f2 p0 = {a0, a1}, p1 = {a2, a3};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a0 = fmaf(a0, s, 0.5f); a1 = fmaf(a1, s, 0.5f); a2 = fmaf(a2, s, 0.5f); a3 = fmaf(a3, s, 0.5f); }
        if (MODE == 1) { a0 = __builtin_amdgcn_exp2f(a0 * s); a1 = __builtin_amdgcn_exp2f(a1 * s); a2 = __builtin_amdgcn_exp2f(a2 * s); a3 = __builtin_amdgcn_exp2f(a3 * s); }
        if (MODE == 2) { a0 = __builtin_amdgcn_rcpf(a0 + s); a1 = __builtin_amdgcn_rcpf(a1 + s); a2 = __builtin_amdgcn_rcpf(a2 + s); a3 = __builtin_amdgcn_rcpf(a3 + s); }
        if (MODE == 3) { p0 = __builtin_elementwise_fma(p0, (f2){s, s}, (f2){0.5f, 0.5f}); p1 = __builtin_elementwise_fma(p1, (f2){s, s}, (f2){0.5f, 0.5f}); }
        if (MODE == 4) {  // 2 trans + 4 fma per iteration, independent chains
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_rcpf(a1);
            a2 = fmaf(a2, s, 0.5f); a3 = fmaf(a3, s, 0.5f); p0 = __builtin_elementwise_fma(p0, (f2){s, s}, (f2){0.5f, 0.5f});
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y;
}
template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 16), dim3(256), 0, 0, d, 0.999f, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 16), dim3(256), 0, 0, d, 0.999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 16 * 256 * 4);
    const int iters = 4096;
    const double waves = 256.0 * 16 * 4;   // waves launched
    const char* names[5] = {"4 x v_fma_f32", "4 x (mul+v_exp_f32)", "4 x (add+v_rcp_f32)", "2 x v_pk_fma_f32", "exp+rcp+2fma+1pk"};
    float ms[5] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters)};
    for (int m = 0; m < 5; ++m) {
        // cycles per wave-iteration per SIMD at 2.1 GHz: waves/1024 SIMDs share a SIMD
        const double cyc = ms[m] * 1e-3 * 2.1e9 / (waves / 1024.0) / iters;
        printf("%-22s %8.3f ms   %6.1f SIMD cycles per wave-iteration\n", names[m], ms[m], cyc);
    }
    return 0;
}
