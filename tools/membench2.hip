// Dev tool: which part of the depthwise kernels' access shape costs bandwidth against a plain float4 copy?  Tensor [B,H,W,C] fp32 NHWC.
//   copy1      one float4 per thread, linear
//   seg        workgroup = 16 pixels x 16 channel quads (256-B runs per pixel), one row per workgroup, cblk fastest (the depthwise kernels' tile)
//   walk<TH>   the same tile walking TH rows (loads 3 rows ahead), one load + one store per row
//   walk3<TH>  ... with the left / right neighbour loads as well (3 loads per row)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void copy1(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__device__ inline int xcd_run(int bid, int nb) { const int q = nb >> 3, r = nb & 7, x = bid & 7, l = bid >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l; }

template <int TH, bool NB3, bool XCD, int PF = 3>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4) {
    const int ncb = (C4 + 15) >> 4, npb = (W + 15) >> 4, nstrip = (H + TH - 1) / TH;
    int bidx = XCD ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
    const int cblk = bidx % ncb; bidx /= ncb;
    const int pblk = bidx % npb; bidx /= npb;
    const int strip = bidx % nstrip;
    const long b = bidx / nstrip;
    const int c4 = cblk * 16 + (threadIdx.x & 15), ox = pblk * 16 + (threadIdx.x >> 4);
    if (c4 >= C4 || ox >= W) return;
    const int C = C4 * 4;
    const float* xb = x + (b * H) * (long)W * C + c4 * 4;
    float* yb = y + (b * H) * (long)W * C + c4 * 4;
    const long xl = ox > 0 ? -(long)C : 0, xr = ox + 1 < W ? (long)C : 0;
    float4 rc[PF], rl[PF], rr[PF];
    const int oy0 = strip * TH;
    auto rowp = [&](int t) { int iy = oy0 + t; iy = iy >= H ? H - 1 : iy; return xb + ((long)iy * W + ox) * C; };
#pragma unroll
    for (int t = 0; t < PF; ++t) {
        const float* r = rowp(t);
        rc[t] = *reinterpret_cast<const float4*>(r);
        if (NB3) { rl[t] = *reinterpret_cast<const float4*>(r + xl); rr[t] = *reinterpret_cast<const float4*>(r + xr); }
    }
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        float4 v = rc[t % PF];
        if (NB3) { const float4 l = rl[t % PF], r = rr[t % PF]; v.x += l.x + r.x; v.y += l.y + r.y; v.z += l.z + r.z; v.w += l.w + r.w; }
        if (t + PF < TH) {
            const float* r = rowp(t + PF);
            rc[t % PF] = *reinterpret_cast<const float4*>(r);
            if (NB3) { rl[t % PF] = *reinterpret_cast<const float4*>(r + xl); rr[t % PF] = *reinterpret_cast<const float4*>(r + xr); }
        }
        const int oy = oy0 + t;
        if (oy < H) *reinterpret_cast<float4*>(yb + ((long)oy * W + ox) * C) = v;
    }
}

// the depthwise kernels' exact traffic: TH + 2 input rows (the halo rows above / below belong to the neighbouring strips), three loads per
// row, TH output rows
template <int TH, bool XCD>
__global__ __launch_bounds__(256) void walkh(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4) {
    const int ncb = (C4 + 15) >> 4, npb = (W + 15) >> 4, nstrip = (H + TH - 1) / TH;
    int bidx = XCD ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
    const int cblk = bidx % ncb; bidx /= ncb;
    const int pblk = bidx % npb; bidx /= npb;
    const int strip = bidx % nstrip;
    const long b = bidx / nstrip;
    const int c4 = cblk * 16 + (threadIdx.x & 15), ox = pblk * 16 + (threadIdx.x >> 4);
    if (c4 >= C4 || ox >= W) return;
    const int C = C4 * 4;
    const float* xb = x + (b * H) * (long)W * C + c4 * 4;
    float* yb = y + (b * H) * (long)W * C + c4 * 4;
    const long xl = ox > 0 ? -(long)C : 0, xr = ox + 1 < W ? (long)C : 0;
    constexpr int PF = 3, NR = TH + 2;
    float4 rc[PF], rl[PF], rr[PF];
    const int oy0 = strip * TH;
    auto rowp = [&](int t) { int iy = oy0 - 1 + t; iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy); return xb + ((long)iy * W + ox) * C; };
#pragma unroll
    for (int t = 0; t < PF && t < NR; ++t) {
        const float* r = rowp(t);
        rc[t] = *reinterpret_cast<const float4*>(r); rl[t] = *reinterpret_cast<const float4*>(r + xl); rr[t] = *reinterpret_cast<const float4*>(r + xr);
    }
    float4 s0 = {0, 0, 0, 0}, s1 = s0;
#pragma unroll
    for (int t = 0; t < NR; ++t) {
        const float4 c = rc[t % PF], l = rl[t % PF], r = rr[t % PF];
        if (t + PF < NR) {
            const float* q = rowp(t + PF);
            rc[t % PF] = *reinterpret_cast<const float4*>(q); rl[t % PF] = *reinterpret_cast<const float4*>(q + xl); rr[t % PF] = *reinterpret_cast<const float4*>(q + xr);
        }
        float4 h = {c.x + l.x + r.x, c.y + l.y + r.y, c.z + l.z + r.z, c.w + l.w + r.w};
        if (t >= 2) {
            const int oy = oy0 + t - 2;
            if (oy < H) *reinterpret_cast<float4*>(yb + ((long)oy * W + ox) * C) = float4{s0.x + h.x, s0.y + h.y, s0.z + h.z, s0.w + h.w};
        }
        s0 = float4{s1.x + h.x, s1.y + h.y, s1.z + h.z, s1.w + h.w};
        s1 = h;
    }
}

// walking ALONG a row instead of down the rows: workgroup = 16 rows x 16 channel quads, each thread streams XW consecutive pixels of its row
template <int XW>
__global__ __launch_bounds__(256) void walkx(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4) {
    const int ncb = (C4 + 15) >> 4, nxb = (W + XW - 1) / XW, nyb = (H + 15) >> 4;
    int bidx = blockIdx.x;
    const int cblk = bidx % ncb; bidx /= ncb;
    const int xb_ = bidx % nxb; bidx /= nxb;
    const int yb_ = bidx % nyb;
    const long b = bidx / nyb;
    const int c4 = cblk * 16 + (threadIdx.x & 15), oy = yb_ * 16 + (threadIdx.x >> 4);
    if (c4 >= C4 || oy >= H) return;
    const int C = C4 * 4;
    const float* src = x + ((b * H + oy) * (long)W + (long)xb_ * XW) * C + c4 * 4;
    float* dst = y + ((b * H + oy) * (long)W + (long)xb_ * XW) * C + c4 * 4;
    constexpr int PF = 3;
    float4 r[PF];
#pragma unroll
    for (int t = 0; t < PF; ++t) r[t] = *reinterpret_cast<const float4*>(src + (long)t * C);
#pragma unroll 4
    for (int t = 0; t < XW; ++t) {
        const float4 v = r[t % PF];
        if (t + PF < XW) r[t % PF] = *reinterpret_cast<const float4*>(src + (long)(t + PF) * C);
        *reinterpret_cast<float4*>(dst + (long)t * C) = v;
    }
}

// one output row per workgroup, the three input rows shared through LDS: workgroup = 16 pixels x 16 channel quads, a thread loads its own
// pixel of rows y-1, y, y+1 (the two edge columns load the halo pixel too), then reads the 3 x 3 neighbourhood from LDS
template <bool XCD>
__global__ __launch_bounds__(256) void rowlds(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4) {
    __shared__ float4 tile[3][18][16];
    const int ncb = (C4 + 15) >> 4, npb = (W + 15) >> 4;
    int bidx = XCD ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
    const int cblk = bidx % ncb; bidx /= ncb;
    const int pblk = bidx % npb; bidx /= npb;
    const int oy = bidx % H;
    const long b = bidx / H;
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;
    const int c4 = cblk * 16 + q, ox = pblk * 16 + p;
    const bool live = c4 < C4 && ox < W;
    const int C = C4 * 4;
    const float* xb = x + (b * H) * (long)W * C + (live ? c4 : 0) * 4;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int iy = oy - 1 + r; iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        const int cx = ox < W ? ox : W - 1;
        const float* row = xb + ((long)iy * W) * C;
        tile[r][p + 1][q] = *reinterpret_cast<const float4*>(row + (long)cx * C);
        if (p == 0) tile[r][0][q] = *reinterpret_cast<const float4*>(row + (long)(cx > 0 ? cx - 1 : 0) * C);
        if (p == 15) tile[r][17][q] = *reinterpret_cast<const float4*>(row + (long)(cx + 1 < W ? cx + 1 : W - 1) * C);
    }
    __syncthreads();
    if (!live) return;
    float4 s = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d) { const float4 v = tile[r][p + d][q]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    *reinterpret_cast<float4*>(y + ((b * H + oy) * (long)W + ox) * C + c4 * 4) = s;
}

int main() {
    const int shapes[][4] = {{32, 128, 128, 256}, {32, 32, 32, 728}, {32, 256, 256, 128}, {32, 512, 512, 64}};
    for (auto& sh : shapes) {
        const int B = sh[0], H = sh[1], W = sh[2], C = sh[3], C4 = C / 4;
        const long n = (long)B * H * W * C4, bytes = n * 16;
        float *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            hipDeviceSynchronize();
            const int reps = 20;
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 1e3 / reps;
            printf("[%d,%d,%d,%d] %-22s %8.1f us  %7.0f GB/s (r+w)\n", B, H, W, C, name, us, 2.0 * bytes / us / 1e3);
        };
        timeit("copy1", [&] { hipLaunchKernelGGL(copy1, dim3((n + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n); });
        const int ncb = (C4 + 15) / 16, npb = (W + 15) / 16;
#define RUN(TH, NB3, XCD, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((walk<TH, NB3, XCD>), dim3((unsigned)((long)B * ((H + TH - 1) / TH) * npb * ncb)), dim3(256), 0, 0, a, b, H, W, C4); });
        RUN(1, false, false, "seg (1 row)")
        RUN(1, false, true, "seg (1 row) xcd")
        RUN(8, false, false, "walk 8")
        RUN(8, false, true, "walk 8 xcd")
        RUN(16, false, false, "walk 16")
        RUN(16, false, true, "walk 16 xcd")
        RUN(16, true, false, "walk 16 +nb")
        RUN(16, true, true, "walk 16 +nb xcd")
        RUN(32, false, true, "walk 32 xcd")
#define RUNP(TH, PF, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((walk<TH, false, false, PF>), dim3((unsigned)((long)B * ((H + TH - 1) / TH) * npb * ncb)), dim3(256), 0, 0, a, b, H, W, C4); });
        RUNP(16, 6, "walk 16, 6 rows ahead")
        RUNP(16, 8, "walk 16, 8 rows ahead")
        RUNP(16, 16, "walk 16, all 16 first")
        RUNP(8, 8, "walk 8, all 8 first")
        RUNP(4, 4, "walk 4, all 4 first")
#define RUNL(TH, PF, LDS, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((walk<TH, false, false, PF>), dim3((unsigned)((long)B * ((H + TH - 1) / TH) * npb * ncb)), dim3(256), LDS, 0, a, b, H, W, C4); });
        RUNL(16, 16, 20 * 1024, "walk 16 all first, 8 WG/CU")
        RUNL(16, 16, 40 * 1024, "walk 16 all first, 4 WG/CU")
        RUNL(16, 16, 80 * 1024, "walk 16 all first, 2 WG/CU")
        RUNL(16, 3, 40 * 1024, "walk 16, 4 WG/CU")
        timeit("row + LDS", [&] { hipLaunchKernelGGL((rowlds<false>), dim3((unsigned)((long)B * H * npb * ncb)), dim3(256), 0, 0, a, b, H, W, C4); });
        timeit("row + LDS xcd", [&] { hipLaunchKernelGGL((rowlds<true>), dim3((unsigned)((long)B * H * npb * ncb)), dim3(256), 0, 0, a, b, H, W, C4); });
#define RUNX(XW, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((walkx<XW>), dim3((unsigned)((long)B * ((H + 15) / 16) * ((W + XW - 1) / XW) * ncb)), dim3(256), 0, 0, a, b, H, W, C4); });
        RUNX(16, "walk along x, 16 px")
        RUNX(32, "walk along x, 32 px")
#define RUNH(TH, XCD, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((walkh<TH, XCD>), dim3((unsigned)((long)B * ((H + TH - 1) / TH) * npb * ncb)), dim3(256), 0, 0, a, b, H, W, C4); });
        RUNH(1, true, "dw-shape TH=1 xcd")
        RUNH(2, true, "dw-shape TH=2 xcd")
        RUNH(4, true, "dw-shape TH=4 xcd")
        RUNH(8, true, "dw-shape TH=8 xcd")
        RUNH(16, true, "dw-shape TH=16 xcd")
        RUNH(4, false, "dw-shape TH=4")
        RUNH(16, false, "dw-shape TH=16")
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
