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
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 3" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 3" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// Fragment reads by hand: the compiler neither sees them nor waits for them (it would wait with lgkmcnt(0), i.e. also for the NEXT
// unit's reads issued behind them); wait_frags<N> lets the N youngest LDS reads stay in flight and ties the registers to the wait.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(const unsigned char* p) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((lptr_t)p), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void store_nt_d(const void* sbase, unsigned voff, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
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
// phase 2: (1,0) (1,2) reading (i,j) (i,j-1); phase 3: (1,1) reading (i,j).  The nine entries are ordered BY INPUT OFFSET -- (0,0) x 4,
// (0,-1) x 2, (-1,0) x 2, (-1,-1) -- so that the taps of one offset share their A fragments (the four phases read the same input
// pixels: 10 fragment loads per chunk instead of 18), every phase still taking its taps in its own order.  Phase 3 is complete after
// step 0 of a tile's last chunk, phase 2 after step 1, phases 1 and 0 after step 2: the tile's stores (four times what it read) leave
// in three instalments.
struct TapE { int ph, t, o; };      // o = 2 (dy == -1) + (dx == -1)
__device__ constexpr TapE kTaps[9] = {{3, 0, 0}, {1, 0, 0}, {2, 0, 0}, {0, 0, 0}, {2, 1, 1}, {0, 1, 1}, {1, 1, 2}, {0, 2, 2}, {0, 3, 3}};
// A step's 36 MFMAs per wave as six units of six: (tap entry, K half, A fragments loaded with it or kept from the unit before)
struct UnitE { int e, ks, la; };
__device__ constexpr UnitE kUnits[3][6] = {{{0, 0, 1}, {1, 0, 0}, {2, 0, 0}, {0, 1, 1}, {1, 1, 0}, {2, 1, 0}},
                                           {{3, 0, 1}, {3, 1, 1}, {4, 0, 1}, {5, 0, 0}, {4, 1, 1}, {5, 1, 0}},
                                           {{6, 0, 1}, {7, 0, 0}, {6, 1, 1}, {7, 1, 0}, {8, 0, 1}, {8, 1, 1}}};
__device__ constexpr int unit_abuf(int S, int I) {      // which of the two A register sets unit I of step S reads
    int n = 0;
    for (int i = 0; i <= I; ++i) n += kUnits[S][i].la;
    return (n - 1) & 1;
}
__device__ constexpr int unit_reads(int S, int I) { return I > 5 ? 0 : 2 + 4 * kUnits[S][I].la; }

// 8 x 32 input pixels x 64 columns per tile, three taps per step, one workgroup of eight waves per CU, a wave = two tile rows (64
// pixels) x 32 columns per phase (with the shared A fragments: 0.7 KB of LDS reads per MFMA; one row x 64 columns moved 1.0): 76 KB of patch ring (two chunks) + 72 KB of
// weight ring (THREE steps: the tile of step u + 2 is issued in step u, so that a store instalment has two whole steps to drain before
// a wait has to include it -- vmcnt retires loads and stores in order).
template <bool OSPLIT, bool ABL, int EPI>     // EPI: dwords a lane stores at a time (1, or 4 behind dev knob epi_width: see the epilogue).  ABL: the dev build with the ablation switches (knob sep_ablate; tools/deconv_ablate.py)
__global__ __launch_bounds__(512, 1) void deconv_pipe_kernel(const DeconvPipeParams p) {
    constexpr int NW = 8, BN = 64, TW = 32, TH = NW;
    constexpr int TPS = 3, NSTEP = 3;                                       // taps per step, steps per chunk
    constexpr int PH = TH + 1, PWS = TW + 1, NPATCH = PH * PWS;            // one halo row above, one halo column to the left (33 slots per patch row)
    constexpr int NPIECE = (NPATCH + 7) / 8, PP = (NPIECE + NW - 1) / NW;  // 1 KiB pieces; surplus pieces of the last round repeat one
    constexpr int STAGE = NPIECE * 1024;
    constexpr int B_ONE = TPS * BN * 128, PB = TPS * BN / 8 / NW;           // weight tiles of a step: 8 pieces per tap
    constexpr int B_OFF = 2 * STAGE;
    constexpr int TM = 2, EP = 16 * TM / EPI;                              // 32-pixel MFMA tiles per wave / stores per wave, tile and phase
    __shared__ __attribute__((aligned(1024))) unsigned char smem[B_OFF + 3 * B_ONE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;      // tile rows 2 wm, 2 wm + 1; columns 32 wn ..
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
    // patch sources: 32-bit offsets from the image (real pixels) or from the zero page (padding); bit j of pmove tells which
    const unsigned char* const ximg = p.x + img * p.ldx_bytes;
    const unsigned char* zpage = g_zero_dc;
    asm volatile("" : "+s"(zpage));      // (taken once: otherwise a GOT load per group)
    unsigned psrc[PP];
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
            psrc[j] = (real ? (unsigned)(gy * p.W + gx) * (unsigned)p.ldx_bytes : 0u) + kk * 16;
            pmove |= real ? 1u << j : 0u;
        }
    };
    const int abl = ABL ? p.ablate : 0;
    bool primed = false;
    auto issue_patch_piece = [&](int stage, int c, auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (ABL) { if ((abl & 8) && primed) return; }
        int q = wv + NW * j;
        if (q >= NPIECE) q -= NW;
        const unsigned char* base = ((pmove >> j) & 1) ? ximg : zpage;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (psrc[j] + c * 128)), (lptr_t)(smem + stage * STAGE + q * 1024), 16, 0, 0);
    };
    auto issue_patch = [&](int stage, int c) {
        static_assert(PP == 5, "pieces per wave");
        issue_patch_piece(stage, c, std::integral_constant<int, 0>{});
        issue_patch_piece(stage, c, std::integral_constant<int, 1>{});
        issue_patch_piece(stage, c, std::integral_constant<int, 2>{});
        issue_patch_piece(stage, c, std::integral_constant<int, 3>{});
        issue_patch_piece(stage, c, std::integral_constant<int, 4>{});
    };
    // weight rows of step s: row = k * 64 + output channel for the step's three table entries k; pieces XOR-swizzled by (row >> 1) & 7.
    // The sources are rebuilt at issue time from lane constants (a dozen VALU operations per step) rather than kept in 18 registers.
    // piece q = wv * PB + j (scalar) holds rows q * 8 + drow: entry q / 8 of the step, output channel n0 + (q & 7) * 8 + drow, 16-byte
    // column dk ^ (4 (q & 1) + (drow >> 1))
    const int bc0 = dk ^ (drow >> 1);
    auto issue_B_piece = [&](int buf, int c, int s, auto J_) {
        constexpr int j = decltype(J_)::value;
        if constexpr (ABL) { if ((abl & 16) && primed) return; }
        {
            const int e = s * TPS + (wv * PB + j) / 8;        // scalar: entry of kTaps
            const int ph = (0x001020213ull >> (4 * e)) & 15, t = (0x321110000ull >> (4 * e)) & 15;      // kTaps[e] as nibble tables
            const int nt = ph == 0 ? 4 : (ph == 3 ? 1 : 2);
            const uint16_t* hi_p = ph == 0 ? p.Whi[0] : (ph == 1 ? p.Whi[1] : (ph == 2 ? p.Whi[2] : p.Whi[3]));
            const uint16_t* lo_p = ph == 0 ? p.Wlo[0] : (ph == 1 ? p.Wlo[1] : (ph == 2 ? p.Wlo[2] : p.Wlo[3]));
            const int q = wv * PB + j, cc = bc0 ^ ((q & 1) << 2);
            const uint16_t* src = ((cc & 4) ? lo_p : hi_p) + (long)(n0 + (q & 7) * 8 + drow) * (nt * p.Cpad) + t * p.Cpad + (cc & 3) * 8 + c * 32;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + B_OFF + buf * B_ONE + (wv * PB + j) * 1024), 16, 0, 0);
        }
    };
    auto issue_B = [&](int buf, int c, int s) {
        static_assert(PB == 3, "pieces per wave");
        issue_B_piece(buf, c, s, std::integral_constant<int, 0>{});
        issue_B_piece(buf, c, s, std::integral_constant<int, 1>{});
        issue_B_piece(buf, c, s, std::integral_constant<int, 2>{});
    };

    // A fragments: lane fr = input pixel fr of the wave's tile rows 2 wm + i; the four input offsets (dy, dx) in {0, -1}^2
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[TM][4];   // second index 2 * (dy == -1) + (dx == -1)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int slot = (2 * wm + i + 1 - (o >> 1)) * PWS + fr + 1 - (o & 1);
            a_off[i][o] = slot * 128 + ((fh ^ ((slot >> 1) & 7)) << 4);
        }
    const int sw = (fr >> 1) & 7;
    // the four (ks, hi / lo) variants of the lane's weight-fragment address (bits 5 and 6; row 32 wn + fr of a tap's 64): everything else
    // a step adds is a multiple of 8 KiB known at compile time and rides in the read's offset field (two bases: the field holds 16 bits)
    const unsigned char* b_var[2][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        b_var[0][v] = smem + ((B_OFF + (wn * 32 + fr) * 128 + ((fh ^ sw) << 4)) ^ (v << 5));
        b_var[1][v] = b_var[0][v] + 2 * B_ONE;
    }

    const int nch = n0 + wn * 32 + fr;          // this lane's output channel
    const bool full = n0 + BN <= p.N && !(ABL && (abl & 28));     // (the counts hold while every load and store is issued)
    float es1 = nch < p.N ? p.scale1[nch] : 0.f, et1 = nch < p.N ? p.shift1[nch] : 0.f;
    asm volatile("" ::"v"(es1), "v"(et1));
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();

    f32x16 acc[4][TM];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ph][i][e] = 0.f;

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
                const unsigned step = (unsigned)(TW * p.ldx_bytes);
#pragma unroll
                for (int j = 0; j < PP; ++j) psrc[j] += ((pmove >> j) & 1) ? step : 0u;
            } else {
                set_tile(xn);
            }
            ixt = xn;
        }
    };

    const int ngroups = p.tpw * nchunks;      // (chunk, tile) pairs of this workgroup: three steps each
    issue_patch(0, 0);
    issue_B(0, 0, 0);
    issue_B(1, 0, 1);
    primed = true;

    int x0 = xbase;
    int c = 0;
    bool prev_last = false;                   // the group before this one closed a tile (its stores may still be in flight)
    // A step's 36 MFMAs run as six units (tap, K half) of six; the fragment registers of unit i + 1 are read from LDS BEFORE the MFMAs
    // of unit i are issued (two register sets each for A and B), so that a read's latency passes under matrix work.  Only a step's first
    // unit waits in the open (its weights land with the barrier).
    struct AFrag { bf16x8 h[TM], l[TM]; };
    struct BFrag { bf16x8 h, l; };
    AFrag fa[2];
    BFrag fb[2];
    auto load_unit = [&](const unsigned char* stg, auto S_, auto I_) {
        constexpr int S = decltype(S_)::value, I = decltype(I_)::value;
        constexpr UnitE u = kUnits[S][I];
        constexpr int ks = u.ks, k = u.e % TPS, o = kTaps[u.e].o, ab = unit_abuf(S, I), bb = I & 1;
        static_assert(u.e / TPS == S && TM == 2, "unit table");
        if constexpr (u.la) {
            fa[ab].h[0] = lds_read16<0>(stg + (a_off[0][o] ^ (ks << 5)));
            fa[ab].l[0] = lds_read16<0>(stg + (a_off[0][o] ^ (ks << 5) ^ 64));
            fa[ab].h[1] = lds_read16<0>(stg + (a_off[1][o] ^ (ks << 5)));
            fa[ab].l[1] = lds_read16<0>(stg + (a_off[1][o] ^ (ks << 5) ^ 64));
        }
        constexpr int bo = (S & 1) * B_ONE + k * (BN * 128);
        fb[bb].h = lds_read16<bo>(b_var[S >> 1][ks]);
        fb[bb].l = lds_read16<bo>(b_var[S >> 1][ks + 2]);
    };
    auto mma_unit = [&](auto S_, auto I_) {
        constexpr int S = decltype(S_)::value, I = decltype(I_)::value;
        constexpr int ph = kTaps[kUnits[S][I].e].ph, ab = unit_abuf(S, I), bb = I & 1;
        AFrag& a = fa[ab];
        BFrag& b = fb[bb];
        // the reads of the next unit (issued after this one's) may stay in flight; the registers are tied to the wait
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a.h[0]), "+v"(a.l[0]), "+v"(a.h[1]), "+v"(a.l[1]), "+v"(b.h), "+v"(b.l) : "n"(unit_reads(S, I + 1)));
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[ph][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[i], b.h, acc[ph][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[ph][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[i], b.l, acc[ph][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[ph][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[i], b.h, acc[ph][i], 0, 0, 0);
    };
    // between(slot) runs after unit `slot`'s MFMAs are issued: the step's DMA pieces go there, one or two per slot, instead of in one
    // block behind the barrier -- an LDS-DMA piece costs its wave 60-180 issue clocks (MI355X_MICROARCH.md), and with every wave of the
    // workgroup at the same point behind a barrier nobody feeds the matrix cores meanwhile.
    auto taps_of_step = [&](const unsigned char* stg, auto S_, auto&& between) {
        using std::integral_constant;
        if constexpr (ABL) {
            if (abl & 2) {
                between(integral_constant<int, 0>{}); between(integral_constant<int, 1>{}); between(integral_constant<int, 2>{});
                between(integral_constant<int, 3>{}); between(integral_constant<int, 4>{}); between(integral_constant<int, 5>{});
                return;
            }
        }
        load_unit(stg, S_, integral_constant<int, 0>{});
        load_unit(stg, S_, integral_constant<int, 1>{});
        mma_unit(S_, integral_constant<int, 0>{});
        between(integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        load_unit(stg, S_, integral_constant<int, 2>{});
        mma_unit(S_, integral_constant<int, 1>{});
        between(integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        load_unit(stg, S_, integral_constant<int, 3>{});
        mma_unit(S_, integral_constant<int, 2>{});
        between(integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        load_unit(stg, S_, integral_constant<int, 4>{});
        mma_unit(S_, integral_constant<int, 3>{});
        between(integral_constant<int, 3>{});
        __builtin_amdgcn_sched_barrier(0);
        load_unit(stg, S_, integral_constant<int, 5>{});
        mma_unit(S_, integral_constant<int, 4>{});
        between(integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_unit(S_, integral_constant<int, 5>{});
        between(integral_constant<int, 5>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- epilogue of one phase, from the accumulators: phase (py, px) of input pixel (i, j) is output pixel (2 i + py, 2 j + px).
    // A lane holds ONE output channel (n0 + 32 j + fr) of sixteen pixels (8 q + 4 fh + k): it stores them as they are, one dword per
    // pixel -- the 32 lanes of a half wave write the 128 contiguous bytes of a pixel's channel group, every store instruction two whole
    // lines -- instead of gathering four channels per lane first (EPI = 4: a 4 x 4 transpose inside lane quads, then 16-byte stores;
    // measured 6-8 % slower here although a pure store stream prefers 16 bytes per lane, profiles/r03_experiments.txt item 18).
    // split32 output: two pixels are split together (packed
    // conversions), the (even, odd) channel pair trades halves, the even lane stores hi (c, c + 1), the odd lane lo (c - 1, c).
    auto epilogue = [&](auto PH_) {
        constexpr int ph = decltype(PH_)::value;
        if constexpr (ABL) { if (abl & 4) return; }
        int ldo = p.ldy;
        asm volatile("" : "+s"(ldo));
        const int Wo = 2 * p.W;
        const bool odd = fr & 1;
        auto body = [&](auto LEAKY_) {
            constexpr bool LEAKY = decltype(LEAKY_)::value;
#pragma unroll
            for (int j = 0; j < TM; ++j) {      // the wave's two tile rows
                const long pixr = 4 * img + (long)(2 * (y0 + 2 * wm + j) + (ph >> 1)) * Wo + 2 * x0 + (ph & 1);
                float* obase = p.y + pixr * ldo;
                const float s1 = es1, t1 = et1;
                const int n = nch;
                f32x16& a16 = acc[ph][j];
                auto act = [&](float a) {
                    float v = fmaf(a, s1, t1);
                    if constexpr (LEAKY) v = fmaxf(v, slope * v);
                    return __builtin_amdgcn_fmed3f(v, lo, hi);
                };
                if constexpr (EPI == 1) {
                    unsigned voff;
                    bool live;
                    if constexpr (OSPLIT) {
                        voff = (unsigned)(8 * fh * ldo) * 4u + (n >> 5) * 128u + (odd ? 64u + 2u * ((n & 31) - 1) : 2u * (n & 31));
                        live = n < ((p.N + 31) & ~31);     // (es1 = et1 = 0 past N: the padding of the last channel group is written as zeros)
                    } else {
                        voff = (unsigned)(8 * fh * ldo + n) * 4u;
                        live = n < p.N;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            const float r0 = act(a16[4 * q + k]), r1 = act(a16[4 * q + k + 1]);
                            float* ob0 = obase + (16 * q + 2 * k) * ldo;
                            float* ob1 = ob0 + 2 * ldo;
                            if constexpr (!OSPLIT) {
                                if (live) {
                                    store_nt_d(ob0, voff, __builtin_bit_cast(unsigned, r0));
                                    store_nt_d(ob1, voff, __builtin_bit_cast(unsigned, r1));
                                }
                            } else {
                                unsigned h, l;                                 // (pixel k | pixel k + 1) halves of this channel
                                split2(r0, r1, h, l);
                                const unsigned got = swap_pair(odd ? h : l);   // even lane: the odd channel's hi pair; odd lane: the even channel's lo pair
                                const unsigned first = odd ? got : h, second = odd ? l : got;      // channel order inside the stored dword
                                if (live) {
                                    store_nt_d(ob0, voff, __builtin_amdgcn_perm(second, first, 0x05040100u));
                                    store_nt_d(ob1, voff, __builtin_amdgcn_perm(second, first, 0x07060302u));
                                }
                            }
                        }
                    }
                } else {
                    const int li = fr & 3, cq = fr >> 2;
                    const int n4 = n0 + wn * 32 + 4 * cq;
                    const bool valid = n4 < p.N;
                    unsigned voff;
                    if constexpr (OSPLIT) voff = (unsigned)(2 * (4 * fh + li) * ldo) * 4u + (n4 >> 5) * 128u + ((cq & 1) ? 64u : 0u) + ((n4 & 31) >> 3) * 16u;
                    else voff = (unsigned)(2 * (4 * fh + li) * ldo + n4) * 4u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) r[k] = act(a16[4 * q + k]);
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
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) a16[e] = 0.f;
            }
        };
        if (p.act == 4) body(std::true_type{});
        else body(std::false_type{});
    };
    using std::integral_constant;
    // One step (compile-time S) of group g, global step u = 3 g + S.  Issue order of a step: [barrier] B(u + 2); at S = 0 the next
    // group's patch (PP pieces); the taps; in a tile's last chunk the store instalment (S = 0: phase 3, EP stores; S = 1: phase 2, EP;
    // S = 2: phases 1 and 0, 2 EP).  The wait before the barrier needs B(u) -- issued in step u - 2 -- and everything older; what may stay in
    // flight is what was issued after it: the rest of step u - 2 and all of step u - 1.  (A tile whose columns are partly masked issues an
    // unknown number of stores: it counts none, which only waits longer.)
    auto step = [&](auto S_, int g) {
        constexpr int S = decltype(S_)::value;
        const bool this_last = c + 1 == nchunks;
        if constexpr (S == 0) {            // step u - 2 = S 1 (EP stores), u - 1 = S 2 (2 EP) of the previous group
            if (prev_last && full) wait_vm<PB + 3 * EP>(); else wait_vm<PB>();
        } else if constexpr (S == 1) {     // u - 2 = S 2 of the previous group (2 EP), u - 1 = S 0 of this one (patch, EP)
            if (!full || (!prev_last && !this_last)) wait_vm<PB + PP>();
            else if (prev_last && this_last) wait_vm<PB + PP + 3 * EP>();
            else if (this_last) wait_vm<PB + PP + EP>();
            else wait_vm<PB + PP + 2 * EP>();
        } else {                           // u - 2 = S 0 (patch, EP), u - 1 = S 1 (EP) of this group
            if (this_last && full) wait_vm<PB + PP + 2 * EP>(); else wait_vm<PB + PP>();
        }
        __builtin_amdgcn_s_barrier();
        int cn = c, sn = S + 2;
        if (sn >= NSTEP) { sn -= NSTEP; cn = this_last ? 0 : c + 1; }
        if (g * NSTEP + S + 2 >= ngroups * NSTEP) { cn = c; sn = S; }     // past the end: a copy nobody reads keeps the counts uniform
        int pstage = 0;
        if constexpr (S == 0) {
            const bool more = ichunk + 1 < tchunks;
            advance_patch();
            pstage = more ? (ichunk & 1) : ((ichunk + 1) & 1);
        }
        const unsigned char* stg = smem + (g & 1) * STAGE;
        // issue order of the step (what the waits above count on): B(u + 2)'s three pieces, then the patch's five, then the stores
        taps_of_step(stg, S_, [&](auto SLOT_) {
            constexpr int slot = decltype(SLOT_)::value;
            if constexpr (slot < PB) issue_B_piece((S + 2) % 3, cn, sn, SLOT_);
            if constexpr (S == 0 && slot >= PB - 1) {      // slots 2, 3, 4: one piece each; slot 5: the last two
                constexpr int j = slot - (PB - 1);
                if constexpr (j < 3) issue_patch_piece(pstage, ic, integral_constant<int, j>{});
                if constexpr (j == 3) { issue_patch_piece(pstage, ic, integral_constant<int, 3>{}); issue_patch_piece(pstage, ic, integral_constant<int, 4>{}); }
            }
        });
        if (this_last) {
            if constexpr (S == 0) epilogue(integral_constant<int, 3>{});
            if constexpr (S == 1) epilogue(integral_constant<int, 2>{});
            if constexpr (S == 2) { epilogue(integral_constant<int, 1>{}); epilogue(integral_constant<int, 0>{}); }
        }
    };
    for (int g = 0; g < ngroups; ++g) {
        step(integral_constant<int, 0>{}, g);
        step(integral_constant<int, 1>{}, g);
        prev_last = false;
        step(integral_constant<int, 2>{}, g);
        if (++c == nchunks) {
            c = 0;
            prev_last = true;
            x0 += TW;
        }
    }
    wait_vm<0>();
}

}  // namespace

namespace emd {

bool deconv_pipe_covers(const DeconvPipeParams& p) {
    return g_knobs.deconv_direct == 3 && p.H % 8 == 0 && p.W % 32 == 0 && p.Cin % 32 == 0 && p.Cin >= 32 && p.Cin <= 4064 && p.N % 4 == 0 &&
           p.N <= 1024 && (long)p.H * p.W * p.ldx_bytes < (1L << 32);     // (patch sources are 32-bit offsets inside an image)
}

int deconv_pipe_launch(const DeconvPipeParams& p, int B, int out_split, hipStream_t st) {
    DeconvPipeParams q = p;
    q.ablate = g_knobs.sep_ablate;
    const int tiles_w = p.W / 32;
    q.n_ntiles = (p.N + 63) / 64;
    const long wgs1 = (long)tiles_w * (p.H / 8) * B * q.n_ntiles;
    int tpw = 1;
    for (int t : {8, 4, 2})
        if (tiles_w % t == 0 && wgs1 / t >= 1024) { tpw = t; break; }
    if (g_knobs.sep_tpw > 0 && tiles_w % g_knobs.sep_tpw == 0) tpw = g_knobs.sep_tpw;
    q.tpw = tpw;
    const dim3 grid(tiles_w / tpw * q.n_ntiles, p.H / 8, B);
    const int epi = g_knobs.epi_width ? g_knobs.epi_width : 1;
    if (q.ablate) hipLaunchKernelGGL((deconv_pipe_kernel<false, true, 1>), grid, dim3(512), 0, st, q);
    else if (out_split) {
        if (epi == 1) hipLaunchKernelGGL((deconv_pipe_kernel<true, false, 1>), grid, dim3(512), 0, st, q);
        else hipLaunchKernelGGL((deconv_pipe_kernel<true, false, 4>), grid, dim3(512), 0, st, q);
    } else {
        if (epi == 1) hipLaunchKernelGGL((deconv_pipe_kernel<false, false, 1>), grid, dim3(512), 0, st, q);
        else hipLaunchKernelGGL((deconv_pipe_kernel<false, false, 4>), grid, dim3(512), 0, st, q);
    }
    return emd::check_launch("deconv_pipe_kernel");
}

}  // namespace emd
