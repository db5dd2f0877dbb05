// Dev tool (not part of the library): what a plain copy reaches on this box at the K workload's
// size, for several access shapes -- the practical roof the stencil kernel is compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// (a) one float4 per thread, lane-contiguous
__global__ void copy1(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// (b) two float4 per thread at 32-B lane stride (the k3 kernels' shape)
__global__ void copy2_strided(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) { float4 u = a[i], v = a[i + 1]; b[i] = u; b[i + 1] = v; }
}
// (c) two float4 per thread, each instruction lane-contiguous (second is +64 lanes)
__global__ void copy2_split(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int l = threadIdx.x & 63;
    long i = w * 128 + l;
    if (i + 64 < n) { float4 u = a[i], v = a[i + 64]; b[i] = u; b[i + 64] = v; }
}
// (d) grid-stride, R float4 per thread
template <int R>
__global__ void copy_gs(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) if (i + r * stride < n) v[r] = a[i + r * stride];
#pragma unroll
    for (int r = 0; r < R; ++r) if (i + r * stride < n) b[i + r * stride] = v[r];
}

int main() {
    const long sizes[] = {33554432L, 268435456L, 2147483648L};
    for (long bytes : sizes) {
        long n = bytes / 16;
        float4 *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 5; ++i) launch();
            hipDeviceSynchronize();
            const int reps = 50;
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 1e3 / reps;
            printf("%-28s %6ld MB each way: %8.2f us/launch  %7.1f GB/s (r+w)\n", name, bytes >> 20, us, 2.0 * bytes / us / 1e3);
        };
        timeit("copy1 (1xfloat4/thread)", [&] { hipLaunchKernelGGL(copy1, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); });
        timeit("copy2 lane-stride 32B", [&] { hipLaunchKernelGGL(copy2_strided, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, a, b, n); });
        timeit("copy2 split contiguous", [&] { hipLaunchKernelGGL(copy2_split, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, a, b, n); });
        timeit("grid-stride R=4 2048 blk", [&] { hipLaunchKernelGGL(copy_gs<4>, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, a, b, n); });
        timeit("grid-stride R=8", [&] { hipLaunchKernelGGL(copy_gs<8>, dim3((n / 8 + 255) / 256), dim3(256), 0, 0, a, b, n); });
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
