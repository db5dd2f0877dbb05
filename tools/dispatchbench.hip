// Dev tool: how long does the chip take just to dispatch N workgroups of a given shape (empty kernels)?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void empty_k(int* p) {
    __shared__ int s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (LDS > 0) { s[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)] = threadIdx.x; __syncthreads(); if (p && s[0] == -12345) p[0] = 1; }
    else if (p && threadIdx.x == 99999) p[0] = 1;
}
template <int LDS>
void run(const char* name, dim3 grid, int threads) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_k<LDS>, grid, dim3(threads), 0, 0, nullptr);
    (void)hipDeviceSynchronize();
    const int reps = 100;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(empty_k<LDS>, grid, dim3(threads), 0, 0, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long waves = (long)grid.x * grid.y * grid.z * (threads / 64);
    printf("%-34s blocks %7ld x %4d thr lds %6d: %7.2f us/launch  (%.2f waves/ns)\n", name, (long)grid.x * grid.y * grid.z, threads, LDS, ms * 1e3 / reps, waves / (ms * 1e6 / reps));
}
int main() {
    run<0>("1024 x 512thr no LDS", dim3(1, 32, 32), 512);
    run<32768>("1024 x 512thr 32KB LDS (k3_tile)", dim3(1, 32, 32), 512);
    run<0>("2048 x 256thr no LDS", dim3(2048), 256);
    run<16384>("2048 x 256thr 16KB LDS", dim3(2048), 256);
    run<0>("8192 x 64thr", dim3(8192), 64);
    run<0>("512 x 256thr (k3_rows)", dim3(512), 256);
    run<0>("32768 x 256thr (copy1)", dim3(32768), 256);
    run<0>("65536 x 256thr", dim3(65536), 256);
    run<0>("256 x 256thr", dim3(256), 256);
    return 0;
}
