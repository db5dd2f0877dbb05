// slim.conv2d_transpose(k = 3, s = 2, SAME) (machine_learning/denoiser.py:138-150; misc_py/modified_Xception.py:551-605) from a split32
// input with the input PATCH resident in LDS: the four output phases of a tile's 8 x 32 input pixels read their A fragments from one
// (8+1) x (32+1) pixel patch per 32-channel chunk (the taps reach one pixel up and one to the left), nine (phase, tap) weight tiles
// stream past it three per step, four accumulator sets (one per phase) of 32 pixels x 64 columns per wave.  Structure, swizzles, ring
// and vmcnt bookkeeping: conv3_pipe.hip.  Sums chunk-major (the one-launch GEMM form sums tap-major per phase): same error class,
// other last bits.  Reached through emd_deconv3x3s2_fused_split32_f32 (dev knob deconv_direct = 3).
#include <type_traits>

#include "conv3_params.hpp"

namespace {

using namespace emd;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(128))) unsigned char g_zero_dc[16384];   // padding pixels: "+ chunk offset" stays inside for Cin <= 4064

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ float dpp_f(float v, int xor2) {
    return xor2 ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true))
                : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
// 4 x 4 transpose inside a lane quad: in, lane i holds column i of the block; out, row i (sep_pipe.hip has the same helper)
__device__ __forceinline__ void quad_transpose(float (&r)[4], int li) {
    const bool b0 = li & 1, b1 = li & 2;
    float s0 = b0 ? r[0] : r[1], s1 = b0 ? r[2] : r[3];
    s0 = dpp_f(s0, 0);
    s1 = dpp_f(s1, 0);
    r[0] = b0 ? s0 : r[0]; r[1] = b0 ? r[1] : s0;
    r[2] = b0 ? s1 : r[2]; r[3] = b0 ? r[3] : s1;
    float t0 = b1 ? r[0] : r[2], t1 = b1 ? r[1] : r[3];
    t0 = dpp_f(t0, 1);
    t1 = dpp_f(t1, 1);
    r[0] = b1 ? t0 : r[0]; r[2] = b1 ? r[2] : t0;
    r[1] = b1 ? t1 : r[1]; r[3] = b1 ? r[3] : t1;
}
__device__ __forceinline__ unsigned xchg4(unsigned v, bool oddq) {
    const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true);   // row_shl:4
    const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    return oddq ? dn : up;
}

// Tap table of slim.conv2d_transpose(k = 3, s = 2, SAME) in the order the nine (phase, tap) weight tiles are brought in, three per step.
// Phase = 2 py + px of the output pixel (2 i + py, 2 j + px); emd_deconv_phase_taps (gemm_conv.hip) fixes the tap order inside a phase:
// phase 0: kernel (0,0) (0,2) (2,0) (2,2) reading input (i,j) (i,j-1) (i-1,j) (i-1,j-1); phase 1: (0,1) (2,1) reading (i,j) (i-1,j);
// phase 2: (1,0) (1,2) reading (i,j) (i,j-1); phase 3: (1,1) reading (i,j).
struct TapE { int ph, t, dy, dx; };
__device__ constexpr TapE kTaps[9] = {{0, 0, 0, 0}, {0, 1, 0, -1}, {0, 2, -1, 0}, {0, 3, -1, -1}, {1, 0, 0, 0}, {1, 1, -1, 0},
                                      {2, 0, 0, 0}, {2, 1, 0, -1}, {3, 0, 0, 0}};

// NW = 8: 8 x 32 input pixels per tile, three taps per step, one workgroup per CU (124 KB of LDS).  NW = 4: 4 x 32 pixels, two taps per
// step (five steps per chunk, the last with one tap), 75 KB: TWO workgroups per CU -- a transposed conv writes four times what it reads
// (deconv1to0: 262 KB of stores per 256 input pixels against 27.6 k clocks of MFMA issue), the stores retire in order with the loads
// behind them, and a workgroup waiting for its stores to drain leaves the matrix cores to its neighbour.
template <bool OSPLIT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void deconv_pipe_kernel(const DeconvPipeParams p) {
    constexpr int BN = 64, TW = 32, TH = NW;
    constexpr int TPS = NW == 8 ? 3 : 2, NSTEP = (9 + TPS - 1) / TPS;       // taps per step, steps per chunk
    constexpr int H1 = NW == 8 ? 1 : 2;                                     // phases 0 and 1 are complete after this step of the last chunk
    constexpr int PH = TH + 1, PWS = TW + 1, NPATCH = PH * PWS;            // one halo row above, one halo column to the left (33 slots per patch row)
    constexpr int NPIECE = (NPATCH + 7) / 8, PP = (NPIECE + NW - 1) / NW;  // 1 KiB pieces; surplus pieces of the last round repeat one
    constexpr int STAGE = NPIECE * 1024;
    constexpr int B_ONE = TPS * BN * 128, PB = TPS * BN / 8 / NW;           // weight tiles of a step: 8 pieces per tap
    constexpr int B_OFF = 2 * STAGE;
    constexpr int TN = 2, E = 4 * 4 * TN;                                  // stores per wave and tile: 4 phases x 4 row groups x TN
    __shared__ __attribute__((aligned(1024))) unsigned char smem[B_OFF + 2 * B_ONE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD k takes the k-th contiguous eighth of the tile list
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t = (id & 7) * (total >> 3) + (id >> 3);
        if ((total & 7) == 0) {
            bx = t % gridDim.x;
            by = (t / gridDim.x) % gridDim.y;
            bz = t / (gridDim.x * gridDim.y);
        }
    }
    const int n0 = (bx % p.n_ntiles) * BN;
    bx /= p.n_ntiles;
    const int xbase = bx * p.tpw * TW, y0 = by * TH;
    const long img = (long)bz * p.H * p.W;

    const int drow = lane >> 3, dk = lane & 7;
    const unsigned char* psrc[PP];
    unsigned pmove = 0;
    auto set_tile = [&](int xt) {
        pmove = 0;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            const int slot = q * 8 + drow;
            const int py = slot / PWS, px = slot - py * PWS;
            const int gy = y0 - 1 + py, gx = xt - 1 + px;
            const bool real = slot < NPATCH && gy >= 0 && gx >= 0;      // (gy < H, gx < W by construction: H % 8 == 0, W % 32 == 0)
            const int kk = dk ^ ((slot >> 1) & 7);
            const unsigned char* o = g_zero_dc + kk * 16;
            const unsigned char* o_px = p.x + (img + (long)gy * p.W + gx) * p.ldx_bytes + kk * 16;
            psrc[j] = real ? o_px : o;
            pmove |= real ? 1u << j : 0u;
        }
    };
    auto issue_patch = [&](int stage, int c) {
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[j] + c * 128), (lptr_t)(smem + stage * STAGE + q * 1024), 16, 0, 0);
        }
    };
    // weight rows of step s: row = k * 64 + output channel for the step's three table entries k; pieces XOR-swizzled by (row >> 1) & 7.
    // The sources are rebuilt at issue time from lane constants (a dozen VALU operations per step) rather than kept in 18 registers.
    int brow_n[PB], bcol[PB];
    bool blo[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = (wv * PB + j) * 8 + drow;     // (wv * PB + j) / 8 = the table entry inside the step: wave-uniform
        const int c = dk ^ ((row >> 1) & 7);
        brow_n[j] = n0 + (row & 63);
        bcol[j] = (c & 3) * 8;
        blo[j] = (c & 4) != 0;
    }
    auto issue_B = [&](int buf, int c, int s) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            int e = s * TPS + (wv * PB + j) / 8;              // scalar
            if (e > 8) e = 8;                                 // (NW = 4: the fifth step has one tap; its second tile is a copy nobody reads)
            const int ph = e < 4 ? 0 : (e < 6 ? 1 : (e < 8 ? 2 : 3)), t = e < 4 ? e : (e < 6 ? e - 4 : (e < 8 ? e - 6 : 0));
            const int nt = ph == 0 ? 4 : (ph == 3 ? 1 : 2);
            const uint16_t* hi_p = ph == 0 ? p.Whi[0] : (ph == 1 ? p.Whi[1] : (ph == 2 ? p.Whi[2] : p.Whi[3]));
            const uint16_t* lo_p = ph == 0 ? p.Wlo[0] : (ph == 1 ? p.Wlo[1] : (ph == 2 ? p.Wlo[2] : p.Wlo[3]));
            const uint16_t* src = (blo[j] ? lo_p : hi_p) + (long)brow_n[j] * (nt * p.Cpad) + t * p.Cpad + bcol[j] + c * 32;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + B_OFF + buf * B_ONE + (wv * PB + j) * 1024), 16, 0, 0);
        }
    };

    // A fragments: lane fr = input pixel fr of the wave's tile row; the four input offsets (dy, dx) in {0, -1}^2
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[4];   // index 2 * (dy == -1) + (dx == -1)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int slot = (wv + 1 - (o >> 1)) * PWS + fr + 1 - (o & 1);
        a_off[o] = slot * 128 + ((fh ^ ((slot >> 1) & 7)) << 4);
    }
    const int sw = (fr >> 1) & 7;
    const int b_off = B_OFF + fr * 128 + ((fh ^ sw) << 4);

    float es1[TN], et1[TN];
    const bool full = n0 + BN <= p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + fr;
        const bool valid = n < p.N;
        es1[j] = valid ? p.scale1[n] : 0.f;
        et1[j] = valid ? p.shift1[n] : 0.f;
        asm volatile("" ::"v"(es1[j]), "v"(et1[j]));
    }
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();

    f32x16 acc[4][TN];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ph][j][e] = 0.f;

    const int nchunks = p.Cin / 32;
    int ic = 0, ixt = xbase, ichunk = 0;
    const int tchunks = p.tpw * nchunks;
    set_tile(xbase);
    auto advance_patch = [&]() {
        if (ichunk + 1 >= tchunks) return;
        ++ichunk;
        if (++ic == nchunks) {
            ic = 0;
            const int xn = ixt + TW;
            if (ixt >= 1) {     // both tiles clear of the left image edge (there is no right halo): every real pixel moves one tile on
                const long step = (long)TW * p.ldx_bytes;
#pragma unroll
                for (int j = 0; j < PP; ++j) psrc[j] += ((pmove >> j) & 1) ? step : 0;
            } else {
                set_tile(xn);
            }
            ixt = xn;
        }
    };

    issue_patch(0, 0);
    issue_B(0, 0, 0);

    int x0 = xbase;
    int c = 0;
    bool epi = false;
    const int ngroups = p.tpw * nchunks;      // (chunk, tile) pairs of this workgroup: three steps each
    auto tap = [&](const unsigned char* stg, int bbase, auto E_, int k) {
        constexpr int e = decltype(E_)::value;
        constexpr TapE te = kTaps[e];
        const int ao = a_off[2 * (te.dy < 0) + (te.dx < 0)];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(stg + (ao ^ (ks << 5)));
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(stg + (ao ^ (ks << 5) ^ 64));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int bo = bbase + k * (BN * 128) + j * 4096;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(smem + (bo ^ (ks << 5)));
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(smem + (bo ^ (ks << 5) ^ 64));
                acc[te.ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[te.ph][j], 0, 0, 0);
                acc[te.ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[te.ph][j], 0, 0, 0);
                acc[te.ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[te.ph][j], 0, 0, 0);
            }
        }
    };
    // ---- epilogue of two phases, from the accumulators: phase (py, px) of input pixel (i, j) is output pixel (2 i + py, 2 j + px).
    // The tile's stores leave in two halves -- phases 0, 1 are complete after step 1 of the last chunk, phases 2, 3 after step 2 -- so that
    // each half has a whole step to drain before a wait has to include it (vmcnt retires in order).
    auto epilogue_pair = [&](int ph0) {
        int ldo = p.ldy;
        asm volatile("" : "+s"(ldo));
        const int li = fr & 3, cq = fr >> 2;
        const int Wo = 2 * p.W;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int ph = ph0 + pp;
            const long pixr = 4 * img + (long)(2 * (y0 + wv) + (ph >> 1)) * Wo + 2 * x0 + (ph & 1);
            float* obase = p.y + pixr * ldo;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float s1 = es1[j], t1 = et1[j];
                const int n4 = n0 + j * 32 + 4 * cq;
                const bool valid = n4 < p.N;
                unsigned voff;
                if constexpr (OSPLIT) voff = (unsigned)(2 * (4 * fh + li) * ldo) * 4u + (n4 >> 5) * 128u + ((cq & 1) ? 64u : 0u) + ((n4 & 31) >> 3) * 16u;
                else voff = (unsigned)(2 * (4 * fh + li) * ldo + n4) * 4u;
                f32x16& a16 = ph0 == 0 ? acc[pp][j] : acc[2 + pp][j];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float v = fmaf(a16[4 * q + k], s1, t1);
                        r[k] = fminf(fmaxf(fmaxf(v, lo), slope * v), hi);
                    }
                    quad_transpose(r, li);
                    f32x4 v = f32x4{r[0], r[1], r[2], r[3]};
                    float* ob = obase + (16 * q) * ldo;
                    if constexpr (!OSPLIT) {
                        if (valid) store_nt_s(ob, voff, v);
                    } else {
                        if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        unsigned h0, l0, h1, l1;
                        split2(v[0], v[1], h0, l0);
                        split2(v[2], v[3], h1, l1);
                        const bool oddq = cq & 1;
                        const unsigned r0 = xchg4(oddq ? h0 : l0, oddq), r1 = xchg4(oddq ? h1 : l1, oddq);
                        if (n4 < ((p.N + 31) & ~31)) store_nt_s(ob, voff, oddq ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1});
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) a16[e] = 0.f;
            }
        }
    };
    using std::integral_constant;
    // one step (compile-time S): B(g, S) -- and everything older -- has landed for this wave; younger groups that may stay in flight:
    // the next chunk's patch (issued in step 0, after B(g, 1)) at S = 1; the E / 2 stores of a half epilogue at the step after it
    // (the first half is issued after step H1 of a tile's last chunk, i.e. after B(g, H1 + 1); the second after the last step, i.e.
    // after B(g + 1, 0))
    auto step = [&](auto S_, int g) {
        constexpr int S = decltype(S_)::value;
        if constexpr (S == 1) wait_vm<PP>();
        else if (S == H1 + 1) { if (c + 1 == nchunks && full) wait_vm<E / 2>(); else wait_vm<0>(); }
        else if (S == 0) { if (epi && full) wait_vm<E / 2>(); else wait_vm<0>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const int u = g * NSTEP + S;             // weight buffer u & 1
        {
            int cn = c, sn = S + 1;
            if (sn == NSTEP) { sn = 0; cn = c + 1 == nchunks ? 0 : c + 1; }
            if (S == NSTEP - 1 && g + 1 >= ngroups) { cn = c; sn = S; }
            issue_B((u + 1) & 1, cn, sn);
        }
        if constexpr (S == 0) {
            const bool more = ichunk + 1 < tchunks;
            advance_patch();
            issue_patch(more ? (ichunk & 1) : ((ichunk + 1) & 1), ic);
        }
        const unsigned char* stg = smem + (g & 1) * STAGE;
        const int bbase = (u & 1) * B_ONE + b_off;
        tap(stg, bbase, integral_constant<int, TPS * S>{}, 0);
        if constexpr (TPS * S + 1 < 9) tap(stg, bbase, integral_constant<int, TPS * S + 1>{}, 1);
        if constexpr (TPS == 3 && TPS * S + 2 < 9) tap(stg, bbase, integral_constant<int, (TPS * S + 2 < 9 ? TPS * S + 2 : 8)>{}, 2);
    };
    for (int g = 0; g < ngroups; ++g) {
        step(integral_constant<int, 0>{}, g);
        epi = false;
        step(integral_constant<int, 1>{}, g);
        if constexpr (H1 == 1) { if (c + 1 == nchunks) epilogue_pair(0); }     // phases (0, 0) and (0, 1) are complete
        step(integral_constant<int, 2>{}, g);
        if constexpr (NSTEP == 5) {
            if (c + 1 == nchunks) epilogue_pair(0);
            step(integral_constant<int, 3>{}, g);
            step(integral_constant<int, 4>{}, g);
        }
        if (++c == nchunks) {
            c = 0;
            epi = true;
            epilogue_pair(2);     // phases (1, 0) and (1, 1)
            x0 += TW;
        }
    }
    wait_vm<0>();
}

}  // namespace

namespace emd {

bool deconv_pipe_covers(const DeconvPipeParams& p) {
    return g_knobs.deconv_direct == 3 && p.H % 8 == 0 && p.W % 32 == 0 && p.Cin % 32 == 0 && p.Cin >= 32 && p.Cin <= 4064 && p.N % 4 == 0 &&
           p.N <= 1024;
}

int deconv_pipe_launch(const DeconvPipeParams& p, int B, int out_split, hipStream_t st) {
    DeconvPipeParams q = p;
    const int tiles_w = p.W / 32;
    q.n_ntiles = (p.N + 63) / 64;
    const long wgs1 = (long)tiles_w * (p.H / 8) * B * q.n_ntiles;
    int tpw = 1;
    for (int t : {8, 4, 2})
        if (tiles_w % t == 0 && wgs1 / t >= 1024) { tpw = t; break; }
    if (g_knobs.sep_tpw > 0 && tiles_w % g_knobs.sep_tpw == 0) tpw = g_knobs.sep_tpw;
    q.tpw = tpw;
    const dim3 grid(tiles_w / tpw * q.n_ntiles, p.H / 8, B);
    if (g_knobs.deconv_nw == 4 && p.H % 4 == 0) {     // 4 x 32 tiles, two workgroups per CU
        const long wgs4 = (long)tiles_w * (p.H / 4) * B * q.n_ntiles;
        int t4 = 1;
        for (int t : {8, 4, 2})
            if (tiles_w % t == 0 && wgs4 / t >= 2048) { t4 = t; break; }
        if (g_knobs.sep_tpw > 0 && tiles_w % g_knobs.sep_tpw == 0) t4 = g_knobs.sep_tpw;
        q.tpw = t4;
        const dim3 grid4(tiles_w / t4 * q.n_ntiles, p.H / 4, B);
        if (out_split) hipLaunchKernelGGL((deconv_pipe_kernel<true, 4>), grid4, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((deconv_pipe_kernel<false, 4>), grid4, dim3(256), 0, st, q);
        return emd::check_launch("deconv_pipe_kernel<4 waves>");
    }
    if (out_split) hipLaunchKernelGGL((deconv_pipe_kernel<true, 8>), grid, dim3(512), 0, st, q);
    else hipLaunchKernelGGL((deconv_pipe_kernel<false, 8>), grid, dim3(512), 0, st, q);
    return emd::check_launch("deconv_pipe_kernel");
}

}  // namespace emd
