// Dev tool: what the memory system delivers for the fused separable conv's access shape, without any arithmetic.
// A workgroup (512 threads, one per CU) streams the (TH+2) x (TW+2) pixel halo patch of its tiles into a two-stage LDS ring by
// LDS-DMA, CB bytes per pixel and step (CB = 128: one 32-channel chunk, what sep_pipe.hip does; 256 / 512: two / four chunks per
// step, i.e. longer contiguous runs per pixel), optionally storing the tile's output share per step (non-temporal dword stores as in
// sep_pipe's epilogue).  Varies: chunk bytes, tile shape, lead (steps ahead), input channels.  Reports algorithmic GB/s
// (tile pixels x channels x 4 in + stores) -- the number to compare with sep_pipe's own.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __attribute__((aligned(16))) float g_zero[8192];

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); }

// TH x TW tile, CB bytes per pixel and step, LEAD steps ahead (1 or 2: two stages)
// SM: store form: 0 none, 1 dword nt, 2 dword plain, 3 dwordx4 nt, 4 dwordx4 plain; RD: 0 = no input stream at all (stores only)
// WORK: synthetic compute between the barriers of a step, `work` iterations: 1 VALU (fma chain), 2 LDS reads (ds_read_b128 of the
// current stage), 3 MFMA (32x32x16 bf16 chain) -- does the stream overlap with it or add to it?
template <int TH, int TW, int CB, int LEAD, int SM = 1, int RD = 1, int WORK = 0>
__global__ __launch_bounds__(512, 2) void stream(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Cin, int Cout, int tpw,
                                                 int xcd, int work = 0) {
    constexpr int NW = 8, PW = TW + 2, PH = TH + 2, NPX = PH * PW;
    constexpr int SPP = CB / 16;                         // 16-byte lanes per pixel
    constexpr int PPI = 64 / SPP;                        // pixels per DMA piece
    constexpr int NPIECE = (NPX + PPI - 1) / PPI;
    constexpr int PP = (NPIECE + NW - 1) / NW;
    constexpr int STAGE = NPIECE * 1024;
    static_assert(2 * STAGE <= 150 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z, id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t = (id & 7) * (total >> 3) + (id >> 3);
        bx = t % gridDim.x; by = (t / gridDim.x) % gridDim.y; bz = t / (gridDim.x * gridDim.y);
    }
    const int xbase = bx * tpw * TW, y0 = by * TH;
    const long img = (long)bz * H * W;
    const float* psrc[PP];
    auto set_tile = [&](int xt) {
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            const int px_i = q * PPI + lane / SPP, k = lane % SPP;
            const int py = px_i / PW, px = px_i - py * PW;
            const int gy = y0 - 1 + py, gx = xt - 1 + px;
            const bool real = px_i < NPX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            psrc[j] = real ? x + (img + (long)gy * W + gx) * Cin + k * 4 : g_zero + k * 4;
        }
    };
    auto issue = [&](int stage, int coff) {
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[j] + coff), (lptr_t)(smem + stage * STAGE + q * 1024), 16, 0, 0);
        }
    };
    const int nchunks = Cin * 4 / CB, total = tpw * nchunks;
    int istep = 0, ic = 0, ixt = xbase;
    set_tile(xbase);
    auto advance = [&]() {
        if (istep + 1 >= total) return;
        ++istep;
        if (++ic == nchunks) { ic = 0; ixt += TW; set_tile(ixt); }
    };
    if (RD) issue(0, 0);
    if (RD && LEAD == 2) { advance(); issue(1, ic * (CB / 4)); }
    // output share per step: the tile's TH*TW*Cout floats over nchunks steps, one dword per lane and store
    const int out_per_step = TH * TW * Cout / nchunks;          // floats
    int ct = 0, x0 = xbase;
    float acc = 0.f;
    for (int t = 0; t < total; ++t) {
        if (LEAD == 2) wait_vm<PP>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (RD && LEAD == 1) { advance(); issue((t + 1) & 1, ic * (CB / 4)); }
        // touch the stage (one LDS read per thread) so that the data dependency is real
        acc += *reinterpret_cast<const float*>(smem + (t & 1) * STAGE + ((tid * 16) % STAGE));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (RD && LEAD == 2) { advance(); issue(t & 1, ic * (CB / 4)); }
        if (WORK == 1) {
            float a0 = acc, a1 = acc + 1.f, a2 = acc + 2.f, a3 = acc + 3.f;
            for (int i = 0; i < work; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f); }
            }
            acc += a0 + a1 + a2 + a3;
        } else if (WORK == 2) {
            typedef __attribute__((ext_vector_type(4))) float f4;
            f4 s4 = {0.f, 0.f, 0.f, 0.f};
            const unsigned char* sb = smem + (t & 1) * STAGE;
            for (int i = 0; i < work; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s4 += *reinterpret_cast<const f4*>(sb + ((tid * 16 + (i * 8 + k) * 8192) % (STAGE - 16)) / 16 * 16);
            }
            acc += s4[0] + s4[1] + s4[2] + s4[3];
        } else if (WORK == 3) {
            typedef __attribute__((ext_vector_type(8))) __bf16 b8;
            typedef __attribute__((ext_vector_type(16))) float f16v;
            f16v c = {};
            b8 av, bv;
#pragma unroll
            for (int k = 0; k < 8; ++k) { av[k] = (__bf16)(acc + k); bv[k] = (__bf16)(1.f + k); }
            for (int i = 0; i < work; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
            }
            acc += c[0] + c[5];
        }
        if (SM >= 3) {
            for (int i = tid * 4; i < out_per_step; i += 2048) {
                const int f = ct * out_per_step + i;
                const int pix = f / Cout, c = f - pix * Cout;
                const int py = pix / TW, px = pix - py * TW;
                float* dst = y + (img + (long)(y0 + py) * W + x0 + px) * Cout + c;
                typedef __attribute__((ext_vector_type(4))) float f4;
                const f4 v = {acc, acc, acc, acc};
                if (SM == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
                else *reinterpret_cast<f4*>(dst) = v;
            }
        } else if (SM == 2) {
            for (int i = tid; i < out_per_step; i += 512) {
                const int f = ct * out_per_step + i;
                const int pix = f / Cout, c = f - pix * Cout;
                const int py = pix / TW, px = pix - py * TW;
                y[(img + (long)(y0 + py) * W + x0 + px) * Cout + c] = acc;
            }
        } else if (SM == 1) {
            // stores: out_per_step floats, as rows of Cout contiguous floats per pixel (dword per lane, like sep_pipe's epilogue)
            for (int i = tid; i < out_per_step; i += 512) {
                const int f = ct * out_per_step + i;           // float index inside the tile's output [TH*TW][Cout]
                const int pix = f / Cout, c = f - pix * Cout;
                const int py = pix / TW, px = pix - py * TW;
                float* dst = y + (img + (long)(y0 + py) * W + x0 + px) * Cout + c;
                asm volatile("global_store_dword %0, %1, off nt" ::"v"(dst), "v"(acc) : "memory");
            }
        }
        if (++ct == nchunks) { ct = 0; x0 += TW; }
    }
    wait_vm<0>();
    if (acc == 123.456f) y[0] = acc;
}

int main(int argc, char** argv) {
    const int B = 32;
    // (S, Cin, Cout): deconv0_b-like, deconv0_a, cnn1, deconv1_a
    const int shapes[][3] = {{512, 128, 64}, {256, 384, 128}};
    for (auto& sh : shapes) {
        const int S = sh[0], Cin = sh[1], Cout = sh[2];
        const long nin = (long)B * S * S * Cin, nout = (long)B * S * S * Cout;
        float *a, *b;
        CK(hipMalloc(&a, nin * 4)); CK(hipMalloc(&b, nout * 4));
        CK(hipMemset(a, 1, nin * 4)); CK(hipMemset(b, 0, nout * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, bool stores, auto launch) {
            for (int i = 0; i < 2; ++i) launch();
            hipDeviceSynchronize();
            const int reps = 5;
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, bytes = 4.0 * (nin + (stores ? nout : 0));
            printf("[%d^2 %3d->%3d] %-44s %8.1f us  %7.0f GB/s algorithmic\n", S, Cin, Cout, name, us, bytes / us / 1e3);
            fflush(stdout);
        };
#define RUN(TH, TW, CB, LEAD, ST, XCD, TPW, NAME)                                                                                  \
    if (Cin * 4 % CB == 0 && S % TW == 0 && (S / TW) % TPW == 0)                                                                   \
        timeit(NAME, ST, [&] {                                                                                                     \
            hipLaunchKernelGGL((stream<TH, TW, CB, LEAD, (ST) ? 1 : 0, 1>), dim3(S / TW / TPW, S / TH, B), dim3(512), 0, 0, a, b, S, S, Cin, Cout, \
                               TPW, XCD);                                                                                          \
        });
#define RUNW(WORK, SM, RD, WK, NAME)                                                                                               \
    if (S % 32 == 0 && (S / 32) % 8 == 0)                                                                                          \
        timeit(NAME, SM != 0, [&] {                                                                                                \
            hipLaunchKernelGGL((stream<8, 32, 128, 2, SM, RD, WORK>), dim3(S / 32 / 8, S / 8, B), dim3(512), 0, 0, a, b, S, S, Cin, Cout, 8, 1, \
                               WK);                                                                                                \
        });
        RUNW(0, 0, 1, 0, "stream only (reads)")
        RUNW(0, 3, 1, 0, "stream + x4 nt stores")
        RUNW(1, 0, 0, 40, "VALU work 40 alone")
        RUNW(1, 0, 1, 40, "VALU work 40 + stream")
        RUNW(1, 0, 0, 80, "VALU work 80 alone")
        RUNW(1, 0, 1, 80, "VALU work 80 + stream")
        RUNW(2, 0, 0, 12, "LDS-read work 12 alone")
        RUNW(2, 0, 1, 12, "LDS-read work 12 + stream")
        RUNW(2, 0, 0, 24, "LDS-read work 24 alone")
        RUNW(2, 0, 1, 24, "LDS-read work 24 + stream")
        RUNW(3, 0, 0, 8, "MFMA work 8 alone")
        RUNW(3, 0, 1, 8, "MFMA work 8 + stream")
        RUNW(3, 0, 0, 16, "MFMA work 16 alone")
        RUNW(3, 0, 1, 16, "MFMA work 16 + stream")
        RUNW(3, 3, 1, 16, "MFMA work 16 + stream + x4 stores")
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
