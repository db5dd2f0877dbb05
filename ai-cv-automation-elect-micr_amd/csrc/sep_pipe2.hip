// Fused separable convolution, software-pipelined form (round 4): depthwise 3x3 (stride 1, rate 1) -> 1x1 on the matrix cores ->
// folded batch norms + relu6 (+ second affine, + residual), one or two outputs, 8 waves on 8 x 32 pixel tiles, one workgroup per CU.
// replaces: the same reference code as sep_pipe.hip (slim.separable_convolution2d + _batch_norm_fn + batch_then_activ =
//           strided_conv_block of machine_learning/denoiser.py:110-136; with two outputs also conv_block_not_sep(kernel_size=1) of the
//           same input, denoiser.py:356-359, :368-371, :380-383), whose arithmetic it repeats bit for bit (same depthwise sums in the
//           same order, same three MFMA passes in the same order along K).
//
// Why a third kernel.  sep_pipe.hip runs a 32-channel chunk as  wait -> barrier -> stage 1 (depthwise: LDS reads + VALU + LDS writes)
// -> barrier -> stage 2 (fragment reads + MFMAs): with one workgroup per CU in lockstep the matrix pipe idles during stage 1 and the
// vector unit during stage 2, and its ablation runs show the phases ADD (profiles/r03_sep_ablation.txt): the two-output launch
// 384 -> 128 | 128 at 256^2 sits at 0.32 of the HBM roof and 0.23 of the MFMA roof at once.  Here every barrier
// interval ("slot") carries BOTH kinds of work for every wave: the MFMAs of K half h of chunk t and, interleaved with them in the same
// instruction stream, the depthwise stage of the NEXT K half -- so whichever of a SIMD's two waves stalls on the matrix pipe, the other
// (and the wave itself, between dependent MFMAs) has vector and LDS work to issue.
//
//   slot s = 2 t + h  (h = 0, 1: channels 16 h .. 16 h + 15 of chunk t)
//     matrix cores : A_h (256 rows x [16 hi | 16 lo] bf16 = 64 B) x B_h (BN rows x 64 B)   -- one 32x32x16 K step, 3 passes
//     vector + LDS : depthwise of step s + 1 from the fp32 patch -> A_{1-h}                 -- 4 pixels x 2 channels per thread
//     DMA          : B_{1-h} of step s + 1 (one slot ahead, L2 hits), and in odd slots the patch of chunk t + 2 (one chunk ahead)
//   one barrier per slot (= two per chunk, as before, but nothing waits in between).
// The A and B tiles are split by K HALF instead of double-buffered: A_0 | A_1 are exactly the old 32 KiB A tile, each half read in one
// slot and rewritten in the next, so the pipelining costs no LDS: 2 x 44 KiB patch stages + 32 KiB A + BN x 128 B weights (154 KiB at
// 256 columns).
// Two outputs: waves 0-3 compute the block (A = depthwise rows), waves 4-7 the 1x1 projection of the block's input (A = the centre
// pixels of the fp32 patch, split to bf16 hi / lo at fragment-load time) -- one of each on every SIMD.  The projection runs ONE SLOT
// AHEAD of the block (its operand needs no depthwise stage), so that a patch stage is read in two consecutive slots only and can be
// refilled a whole chunk ahead; its epilogue therefore also comes one slot earlier.
// Epilogue: from the accumulators (sep_pipe.hip's two forms), at the TOP of the slot after a tile's last MFMA, behind that slot's DMA
// issue: the stores are then the youngest entries of the in-order vmcnt queue and a later wait for DMA pieces does not wait for them.
// Patch swizzle: 16-byte chunk c of patch pixel column px is stored at c ^ ((px >> 1) & 7) (applied to the DMA's source address): the
// depthwise stage's 8-byte reads (a 32-lane group = 2 rows x pixel groups {g, g + 2} x 8 channel pairs), and the projection's 16-byte
// centre reads (16 consecutive pixels of a row) then cover the 64 banks evenly.  A / B rows (64 B): chunk ^ ((row >> 2) & 3).
#include "sep_pipe_common.hpp"

namespace {

using namespace emd;
using namespace emd::sp;

// source of the zero-padding pixels (TF SAME) and of the unused slots: 16 KB, so that "+ chunk offset" stays inside for Cin <= 4064
__device__ __attribute__((aligned(16))) float g_zero_pipe2[4096];

typedef __attribute__((ext_vector_type(2))) float f32x2v;

template <int BN, bool DUAL, bool OSPLIT, int EPI>
__global__ __launch_bounds__(512, 2) void sep_pipe2_kernel(const SepParams p) {
    constexpr int NW = 8, TW = 32, TH = 8, BM = TH * TW;
    constexpr int PW = TW + 2, PH = TH + 2, PWS = PW | 1;          // 34 x 10 pixel patch, slot pitch 35
    constexpr int NPATCH = PH * PWS, NSLOT = NPATCH;               // the nine depthwise-weight slots ride in the pad column (px = 34) of rows 0-8
    constexpr int WK0 = PW, WKS = PWS;
    constexpr int NPIECE = (NSLOT + 7) / 8, PP = (NPIECE + NW - 1) / NW, STAGE = NPIECE * 1024;
    constexpr int A_HALF = BM * 64, B_HALF = BN * 64;
    constexpr int A_OFF = 2 * STAGE, B_OFF = A_OFF + 2 * A_HALF, C_OFF = B_OFF + 2 * B_HALF, SMEM = C_OFF + 16 * BN;   // + the epilogue's [4][BN] affines
    constexpr int NBP = BN / 16;                                   // weight pieces (16 rows x 64 B) per slot
    constexpr int PB = NBP >= NW ? NBP / NW : 1;                   // per wave (4 pieces at 64 columns: waves 4-7 repeat 0-3)
    constexpr int WN = BN / 64, WM = NW / WN, TM = BM / WM / 32, TN = 2;   // a wave owns 32 TM rows x 64 columns
    constexpr int E = 16 / EPI * TM * TN;                          // stores per wave and tile (exact when no lane is masked)
    constexpr bool RPRE = TM == 1 && !DUAL;                        // residual values requested one slot before the epilogue (64 columns: 32 registers; 64 more at 128 columns spill)
    constexpr int R = RPRE ? 16 / EPI * TM * TN : 0;
    static_assert(SMEM <= 160 * 1024 && PP >= 2, "shape");
    static_assert(!(DUAL && OSPLIT), "split32 output: one-output instances only");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave -> (row block, column block): with two outputs waves 0-3 take the block's columns, waves 4-7 the projection's
    int wm, wn;
    if constexpr (DUAL) {
        const int idx = wv & 3;
        wm = idx % WM;
        wn = (wv >> 2) * (WN / 2) + idx / WM;
    } else {
        wm = wv / WN;
        wn = wv % WN;
    }
    const bool out2 = DUAL && wv >= 4;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd) {   // XCD k (workgroup id mod 8) takes the k-th contiguous eighth of the tile list: halo rows meet in one L2
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t = (id & 7) * (total >> 3) + (id >> 3);
        bx = t % gridDim.x;
        by = (t / gridDim.x) % gridDim.y;
        bz = t / (gridDim.x * gridDim.y);
    }
    const int xbase = bx * p.tpw * TW, y0 = by * TH;
    const int Wo = p.W;
    const long img = (long)bz * p.H * p.W;
    const long img_o = img;

    // ---- patch DMA sources.  Lane l of a piece fills 16-byte chunk (l & 7) of slot 8 * piece + (l >> 3).
    const float* psrc[PP];
    unsigned pmove = 0;   // bit j: source j is a pixel of the image (moves with the tile), not padding / weights
    auto set_tile = [&](int xt) {
        int drow = lane >> 3, dk = lane & 7;
        asm volatile("" : "+v"(drow), "+v"(dk));    // opaque: nothing of this (rare: image edges) computation is held in registers across the slots
        pmove = 0;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            const int slot = q * 8 + drow;
            const int py = slot / PWS, px = slot - py * PWS;
            int gy = y0 - 1 + py, gx = xt - 1 + px;
            if (p.reflect) {   // tf.pad(REFLECT, 1): -1 -> 1, H -> H - 2
                gy = gy < 0 ? -gy : (gy >= p.H ? 2 * p.H - 2 - gy : gy);
                gx = gx < 0 ? -gx : (gx >= p.W ? 2 * p.W - 2 - gx : gx);
            }
            const bool real = slot < NPATCH && px < PW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const bool wk = slot < NPATCH && px == PW && py < 9;
            const int kk = dk ^ ((px >> 1) & 7);                                   // patch swizzle (not the weight slots)
            const float* o = g_zero_pipe2 + dk * 4;                                 // padding pixels, unused slots
            const float* o_px = p.x + (img + (long)gy * p.W + gx) * p.ldx + kk * 4;
            const float* o_wk = p.dw + (long)py * p.Cin + dk * 4;                  // the chunk's depthwise weights, tap py in this slot
            o = real ? o_px : o;
            o = wk ? o_wk : o;
            psrc[j] = o;
            pmove |= real ? 1u << j : 0u;
        }
    };
    auto issue_patch = [&](int stage, int coff, int j0 = 0, int j1 = PP) {   // coff: channel offset of the chunk (floats); pieces j0 .. j1-1 of this wave
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[j] + coff), (lptr_t)(smem + stage * STAGE + q * 1024), 16, 0, 0);
        }
    };
    // ---- weight DMA: piece q = 16 rows x 64 B of one K half; lane l fills physical chunk (l & 3) of row 16 q + (l >> 2); with two outputs
    // pieces 0 .. NBP/2-1 are the block's rows, the rest the projection's (which runs one step ahead: other K half, maybe other chunk)
    const uint16_t* bsrc[PB];
    int bq[PB];
    bool bproj[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int q = (wv * PB + j) % NBP;
        const int row = q * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);           // logical chunk: 0, 1 = hi (k 0-7, 8-15), 2, 3 = lo
        const bool second = DUAL && row >= BN / 2;
        const int rr = second ? row - BN / 2 : row;
        const uint16_t* plane = (c & 2) ? (second ? p.W2lo : p.Wlo) : (second ? p.W2hi : p.Whi);
        bsrc[j] = plane + (long)rr * p.Cpad + (c & 1) * 8;
        bq[j] = q;
        bproj[j] = DUAL && q >= NBP / 2;
    }
    // main rows: K half hm of the chunk at channel offset cm; projection rows: half hp at offset cp
    auto issue_B = [&](int hm, int cm, int hp, int cp) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int h = bproj[j] ? hp : hm, c = bproj[j] ? cp : cm;
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[j] + c + h * 16), (lptr_t)(smem + B_OFF + h * B_HALF + bq[j] * 1024), 16, 0, 0);
        }
    };

    // ---- depthwise role: 4 consecutive output pixels of one tile row, 2 channels of the K half.  A 32-lane group = 2 rows x pixel groups
    // {g, g + 2} x 8 channel pairs (conflict-free 8-byte reads under the patch swizzle, see the header)
    const int c2 = lane & 7, pgi = (lane >> 3) & 3, q16 = 2 * wv + (lane >> 5);
    const int xpair = q16 & 3, dy = 2 * (q16 >> 2) + (pgi >> 1), dx = 4 * ((xpair & 1) + 4 * (xpair >> 1) + 2 * (pgi & 1));
    int rdv[2][3];    // byte offset inside a stage of this thread's 8 bytes of patch pixel (dy, dx + 2 m [+ 1]), K half h
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < 3; ++m)
            rdv[h][m] = (dy * PWS + dx) * 128 + (((h * 4 + (c2 >> 1)) ^ (((dx >> 1) + m) & 7)) << 4) + (c2 & 1) * 8;
    const int wk_off = WK0 * 128 + c2 * 8;                       // + h * 64 + tap * WKS * 128
    const int ga_w = (dx >> 2) & 3;
    const int a_wr = A_OFF + (dy * TW + dx) * 64 + (((c2 >> 2) ^ ga_w) << 4) + (c2 & 3) * 4;   // hi dword of pixel j: + 64 j; lo: ^ 32; half: + A_HALF
    // ---- matrix role
    const int fr = lane & 31, fh = lane >> 5, gf = (fr >> 2) & 3;
    const int row0 = wm * TM * 32;                               // first of this wave's GEMM rows; row r = pixel (r / TW, r % TW) of the tile
    // fragment addressing, one form for both roles (the MFMAs then are the same code for every wave):
    //   block      : A rows,   hi at a_rd + h A_HALF + 2048 i, lo at that ^ 32
    //   projection : fp32 centre pixels of the patch, chunks (2 fh, 2 fh + 1) of K half h: stage + (cen ^ 64 h) + 35 * 128 i, and ^ 16
    const int a_rd = A_OFF + (row0 + fr) * 64 + ((fh ^ gf) << 4);
    const int cen = ((row0 / TW + 1) * PWS + fr + 1) * 128 + (((fh * 2) ^ (((fr + 1) >> 1) & 7)) << 4);
    int fav[2][2];                                               // [K half][first / second 16-byte read]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) fav[h][e] = out2 ? (cen ^ (h * 64) ^ (e * 16)) : ((a_rd + h * A_HALF) ^ (e * 32));
    const int fstride = (DUAL && out2) ? PWS * 128 : 2048;       // (a compile-time 2048 in the one-output instances: immediates)
    const int b_rd = B_OFF + (wn * 64 + fr) * 64 + ((fh ^ gf) << 4);              // N tile j: + 2048 j; K half: + B_HALF
    const int b_rd2 = b_rd ^ 32;                                                   // the lo fragment

    // ---- epilogue constants: lane = output channel.  The per-channel affines are fetched IN the epilogue (once per tile, L2 hits): held
    // in registers across the slots they were 8 of the 256 a lane has, and the slots spill without them
    const bool two = p.scale2 != nullptr && !out2;
    const int nlim = out2 ? p.N2 : p.N;
    const bool full = DUAL ? (p.N == BN / 2 && p.N2 == BN / 2) : p.N == BN;   // no lane is masked in the epilogue: store counts are exact
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    const int nchunks = p.Cin / 32;
    const int total = p.tpw * nchunks;
    // patch issue stream (clamped at the last chunk: the surplus groups re-read it into a stage nobody computes on, so that every
    // wave's vmcnt arithmetic stays uniform to the end)
    int istep = 0, ic = 0, ixt = xbase;
    set_tile(xbase);
    auto advance_issue = [&]() {
        if (istep + 1 >= total) return;
        ++istep;
        if (++ic == nchunks) {
            ic = 0;
            const int xn = ixt + TW;
            if (ixt >= 1 && xn + TW + 1 <= p.W) {   // both tiles clear of the left / right image edges: every real pixel moves one tile on
                const long step = (long)TW * p.ldx;
#pragma unroll
                for (int j = 0; j < PP; ++j) psrc[j] += ((pmove >> j) & 1) ? step : 0;
            } else {
                set_tile(xn);
            }
            ixt = xn;
        }
    };

    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (p.stamps) tprev = __builtin_amdgcn_s_memtime();
#define PIPE_STAMP(i) if (p.stamps) { const long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tprev; tprev = t_; }

    // The per-channel affines of the epilogue live in LDS ([scale1 | shift1 | scale2 | shift2][BN], identity / zero beyond the outputs'
    // widths; with two outputs the projection's pair in columns BN/2 ..): read per tile by the epilogue, instead of 8 registers through
    // every slot or global loads the epilogue would have to wait for behind the slot's DMA pieces.
    if (tid < BN) {
        const bool second = DUAL && tid >= BN / 2;
        const int nn = second ? tid - BN / 2 : tid;
        const bool valid = nn < (second ? p.N2 : p.N);
        const bool two_ = p.scale2 != nullptr && !second;
        float* cst = reinterpret_cast<float*>(smem + C_OFF);
        cst[tid] = valid ? (second ? p.scale_b : p.scale1)[nn] : 0.f;
        cst[BN + tid] = valid ? (second ? p.shift_b : p.shift1)[nn] : 0.f;
        cst[2 * BN + tid] = (valid && two_) ? p.scale2[nn] : 1.f;
        cst[3 * BN + tid] = (valid && two_) ? p.shift2[nn] : 0.f;
    }
    // Residual values of a tile (one-output instances up to 128 columns): requested at the top of the slot that holds the tile's last
    // MFMAs, in front of that slot's patch DMA; consumed by the epilogue at the top of the next slot.
    f32x4 rpre[RPRE ? TM : 1][TN][4];
    const bool res_on = p.res != nullptr && !out2;
    auto res_prefetch = [&](int x0) {
        int ldr = p.ldres;
        asm volatile("" : "+s"(ldr));
        int fr = lane & 31, fh = lane >> 5;
        asm volatile("" : "+v"(fr), "+v"(fh));      // opaque: per-lane offsets are computed here, per tile, not held across the slots
        const int li = fr & 3, cq = fr >> 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n4 = wn * 64 + j * 32 + 4 * cq, n1 = wn * 64 + j * 32 + fr;
            const unsigned roff = EPI == 1 ? (unsigned)(4 * fh * ldr + n1) * 4u : (unsigned)((4 * fh + li) * ldr + n4) * 4u;
#pragma unroll
            for (int i = 0; i < (RPRE ? TM : 1); ++i) {
                const float* rbase = p.res + (img_o + (long)(y0 + (row0 + i * 32) / TW) * Wo + x0) * ldr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rpre[i][j][q] = f32x4{-0.f, -0.f, -0.f, -0.f};
                    const float* rb = rbase + ((8 * q / TW) * Wo + 8 * q % TW) * ldr;
                    // plain loads: the compiler owns their wait (a register an inline-asm load is still writing may be COPIED by the allocator before our
                    // own wait -- it was, in the first version, and the copy held stale data)
                    if constexpr (EPI == 1) {      // element k = this lane's channel at pixel 8 q + 4 fh + k
                        if (n1 < nlim) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) rpre[i][j][q][k] = rb[k * ldr + (roff >> 2)];
                        }
                    } else {
                        if (n4 < nlim) rpre[i][j][q] = *reinterpret_cast<const f32x4*>(rb + (roff >> 2));
                    }
                }
            }
        }
    };
    // epilogue from the accumulators (sep_pipe.hip's: C/D layout of the 32x32 MFMA: column = lane & 31 (the channel), row = (e & 3) + 8 (e >> 2)
    // + 4 (lane >> 5) of the M tile = one tile row).  The caller has waited for the prefetched residual values.
    auto epilogue = [&](int x0) {
        int fr = lane & 31, fh = lane >> 5;
        asm volatile("" : "+v"(fr), "+v"(fh));      // opaque: per-lane offsets are computed here, per tile, not held across the slots
        const bool has_res = res_on;
        int actc = out2 ? 1 : p.act;         // the projection of a two-output launch is conv + BN + relu6 (conv_block_not_sep)
        asm volatile("" : "+s"(actc));       // opaque: the activation constants are made here, not held across the slots
        const float hi = actc == 1 ? 6.f : __builtin_inff();
        const float hi2 = actc == 2 ? __builtin_inff() : 6.f;
        const float slope = actc == 4 ? 0.2f : 1.f, lo = (actc == 1 || actc == 2) ? 0.f : -__builtin_inff();
        const bool simple = actc == 1 && !two;   // conv + BN + relu6, nothing else: three VALU operations per value instead of eight
        float* __restrict__ outp = out2 ? p.y2 : p.y;
        int ldo = out2 ? p.ldy2 : p.ldy, ldr = p.ldres;
        asm volatile("" : "+s"(ldo), "+s"(ldr));
        float es1[TN], et1[TN], es2[TN], et2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* cst = reinterpret_cast<const float*>(smem + C_OFF) + wn * 64 + j * 32 + fr;
            es1[j] = cst[0];
            et1[j] = cst[BN];
            es2[j] = cst[2 * BN];
            et2[j] = cst[3 * BN];
        }
        if constexpr (EPI == 1) {
            const bool odd = fr & 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float s1 = es1[j], t1 = et1[j], s2 = es2[j], t2 = et2[j];
                const int n = wn * 64 + j * 32 - (out2 ? BN / 2 : 0) + fr;
                unsigned voff;
                bool live;
                if constexpr (OSPLIT) {
                    voff = (unsigned)(4 * fh * ldo) * 4u + (n >> 5) * 128u + (odd ? 64u + 2u * ((n & 31) - 1) : 2u * (n & 31));
                    live = n < ((nlim + 31) & ~31);     // (scales and shifts are 0 / 1 / 0 past N: the padding is written as zeros)
                } else {
                    voff = (unsigned)(4 * fh * ldo + n) * 4u;
                    live = n < nlim;
                }
                const unsigned roff = (unsigned)(4 * fh * ldr + n) * 4u;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long pixr = img_o + (long)(y0 + (row0 + i * 32) / TW) * Wo + x0;   // uniform: the M tile's first pixel
                    const float* rbase = has_res ? p.res + pixr * ldr : nullptr;
                    float* obase = outp + pixr * ldo;
                    f32x4 rv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] = f32x4{-0.f, -0.f, -0.f, -0.f};   // x + (-0) == x for every x
                    if constexpr (RPRE) {
                        if (has_res) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = rpre[i][j][q];
                        }
                    } else if (has_res) {
                        if (n < nlim) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int k = 0; k < 4; ++k) rv[q][k] = rbase[((8 * q / TW) * Wo + 8 * q % TW + k) * ldr + (roff >> 2)];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r[4];
                        if (simple) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_fmed3f(fmaf(acc[i][j][4 * q + k], s1, t1), 0.f, 6.f) + rv[q][k];
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float u = fmaf(acc[i][j][4 * q + k], s1, t1);
                                u = __builtin_amdgcn_fmed3f(fmaxf(u, slope * u), lo, hi);
                                const float u2 = __builtin_amdgcn_fmed3f(fmaf(u, s2, t2), 0.f, hi2);
                                r[k] = (two ? u2 : u) + rv[q][k];
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            float* ob0 = obase + ((8 * q / TW) * Wo + 8 * q % TW + k) * ldo;   // rows 8q + k.. of the M tile
                            float* ob1 = ob0 + ldo;
                            if constexpr (!OSPLIT) {
                                if (live) {
                                    store_nt_d(ob0, voff, __builtin_bit_cast(unsigned, r[k]));
                                    store_nt_d(ob1, voff, __builtin_bit_cast(unsigned, r[k + 1]));
                                }
                            } else {
                                unsigned h, l;                                 // (pixel k | pixel k + 1) halves of this channel
                                split2(r[k], r[k + 1], h, l);
                                const unsigned got = swap_pair(odd ? h : l);   // even lane: the odd channel's hi pair; odd lane: the even channel's lo pair
                                const unsigned first = odd ? got : h, second = odd ? l : got;
                                if (live) {
                                    store_nt_d(ob0, voff, __builtin_amdgcn_perm(second, first, 0x05040100u));
                                    store_nt_d(ob1, voff, __builtin_amdgcn_perm(second, first, 0x07060302u));
                                }
                            }
                        }
                    }
                }
            }
        } else {
            const int li = fr & 3, cq = fr >> 2;       // after the transpose: lane = pixel (e >> 2) * 8 + 4 fh + li, channels 4 cq .. 4 cq + 3
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float s1 = es1[j], t1 = et1[j], s2 = es2[j], t2 = et2[j];
                const int nb = wn * 64 + j * 32 - (out2 ? BN / 2 : 0);   // first channel of this 32-column group in its output
                const int n4 = nb + 4 * cq;
                const bool valid = n4 < nlim;                             // Cout % 4 == 0: a lane's four channels are all in or all out
                const unsigned roff = (unsigned)((4 * fh + li) * ldr + n4) * 4u;
                unsigned voff;
                if constexpr (OSPLIT) {
                    voff = (unsigned)((4 * fh + li) * ldo) * 4u + (n4 >> 5) * 128u + ((cq & 1) ? 64u : 0u) + ((n4 & 31) >> 3) * 16u;
                } else {
                    voff = (unsigned)((4 * fh + li) * ldo + n4) * 4u;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long pixr = img_o + (long)(y0 + (row0 + i * 32) / TW) * Wo + x0;
                    const float* rbase = has_res ? p.res + pixr * ldr : nullptr;
                    float* obase = outp + pixr * ldo;
                    f32x4 rv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] = f32x4{-0.f, -0.f, -0.f, -0.f};
                    if constexpr (RPRE) {
                        if (has_res) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = rpre[i][j][q];
                        }
                    } else if (has_res) {
                        if (valid) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const f32x4*>(rbase + ((8 * q / TW) * Wo + 8 * q % TW) * ldr + (roff >> 2));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r[4];
                        if (simple) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) r[k] = fminf(fmaxf(fmaf(acc[i][j][4 * q + k], s1, t1), 0.f), 6.f);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float u = fmaf(acc[i][j][4 * q + k], s1, t1);
                                u = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                                const float u2 = fminf(fmaxf(fmaf(u, s2, t2), 0.f), hi2);
                                r[k] = two ? u2 : u;
                            }
                        }
                        quad_transpose(r, li);
                        f32x4 v = f32x4{r[0], r[1], r[2], r[3]} + rv[q];
                        float* ob = obase + ((8 * q / TW) * Wo + 8 * q % TW) * ldo;
                        if constexpr (!OSPLIT) {
                            if (valid) store_nt_s(ob, voff, v);
                        } else {
                            if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};   // the padding channels of a split32 tensor (up to a multiple of 32) are zero
                            unsigned h0, l0, h1, l1;
                            split2(v[0], v[1], h0, l0);
                            split2(v[2], v[3], h1, l1);
                            const bool oddq = cq & 1;
                            const unsigned r0 = xchg4(oddq ? h0 : l0, oddq), r1 = xchg4(oddq ? h1 : l1, oddq);
                            if (n4 < ((nlim + 31) & ~31))
                                store_nt_s(ob, voff, oddq ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1});
                        }
                    }
                }
            }
        }
        zero_acc();
    };

    // ---- one slot's arithmetic.  odd = parity of the slot: the block's MFMAs read K half `odd` of A / B, the projection's (one step ahead)
    // the patch centre and B half 1 - odd; the depthwise stage writes A half 1 - odd.  stg: byte offset of the patch stage of chunk k + 1.
    // Issue order: [MFMAs of a quarter | FMAs of one patch row | the reads the next quarter needs] x 3, then [MFMAs | hi / lo split + A
    // writes]: the MFMAs between a read and its use hide the LDS latency, only ONE row of the 3 x 6 window is live at a time, and
    // sched_barrier pins the segments (inside one the hardware interleaves: an MFMA holds the vector issue for 8 of its 32 cycles).
    struct SlotDma { int hm, cm, hp, cp, patch, pstage, pcoff, rp, x; };   // what a slot requests (see the loop); patch / rp: flags
    auto slot_body = [&](const int odd, const int stg, const bool DIN, const SlotDma& dq) {   // DIN: the slot's DMA pieces go out between the segments
        const int hs = 1 - odd;                          // K half the depthwise stage produces
        const int hm = out2 ? 1 - odd : odd;             // K half of this wave's MFMAs
        // every LDS access is smem + an integer byte offset (a pointer that went through an integer XOR loses its address space: flat loads)
        const int fa_1 = (out2 ? stg : 0) + (hm ? fav[1][0] : fav[0][0]), fa_2 = (out2 ? stg : 0) + (hm ? fav[1][1] : fav[0][1]);
        const int fb = b_rd + hm * B_HALF, fb2 = b_rd2 + hm * B_HALF;
        const int swk = stg + wk_off + hs * 64;
        const int aw = a_wr + hs * A_HALF;
        int srd[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) srd[m] = stg + (hs ? rdv[1][m] : rdv[0][m]);
        f32x2v wk[3], pr[6], o[4];
        auto rd_row = [&](int i) {
#pragma unroll
            for (int d = 0; d < 3; ++d) wk[d] = *reinterpret_cast<const f32x2v*>(smem + swk + (i * 3 + d) * (WKS * 128));
#pragma unroll
            for (int d = 0; d < 6; ++d) pr[d] = *reinterpret_cast<const f32x2v*>(smem + srd[d >> 1] + (i * PWS + d) * 128);
        };
        auto fma_row = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int d = 0; d < 3; ++d) o[j] += wk[d] * pr[j + d];
            // a use in THIS segment: otherwise the optimizer sinks all 36 FMAs (pure, needed only by the split at the end) into the last
            // segment -- the projection's uniform branches cut the slot into basic blocks -- and keeps the three window rows live
            asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
        };
        bf16x8 bh[TN], bl[TN];
        u32x4 f0[TM], f1[TM];        // block: hi / lo fragment of M tile i; projection: eight fp32 centre values, split in cvt_a
        auto ld_b = [&]() {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(smem + fb + j * 2048);
                bl[j] = *reinterpret_cast<const bf16x8*>(smem + fb2 + j * 2048);
            }
        };
        auto ld_a = [&](int i) {
            if constexpr (DUAL) {
                f0[i] = *reinterpret_cast<const u32x4*>(smem + fa_1 + i * fstride);
                f1[i] = *reinterpret_cast<const u32x4*>(smem + fa_2 + i * fstride);
            } else {
                f0[i] = *reinterpret_cast<const u32x4*>(smem + fa_1 + i * 2048);
                f1[i] = *reinterpret_cast<const u32x4*>(smem + fa_2 + i * 2048);
            }
        };
        auto cvt_a = [&](int i) {
            if constexpr (DUAL) {
                if (out2) {   // the projection's A operand = the block's INPUT at the tile's own pixels, split here
                    const f32x4 v0 = __builtin_bit_cast(f32x4, f0[i]), v1 = __builtin_bit_cast(f32x4, f1[i]);
                    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                    split2(v0[0], v0[1], h0, l0);
                    split2(v0[2], v0[3], h1, l1);
                    split2(v1[0], v1[1], h2, l2);
                    split2(v1[2], v1[3], h3, l3);
                    f0[i] = u32x4{h0, h1, h2, h3};
                    f1[i] = u32x4{l0, l1, l2, l3};
                }
            }
        };
        auto mm = [&](int i, int j) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, f0[i]), al = __builtin_bit_cast(bf16x8, f1[i]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
        };
        constexpr int NA = TM * TN;      // atoms (i, j) in row-major order, a quarter of them per segment
        auto seg_of = [](int a) { return a < NA / 4 ? 0 : a < NA * 2 / 4 ? 1 : a < NA * 3 / 4 ? 2 : 3; };   // segment that runs atom a
        auto seg_mm = [&](int q) {
#pragma unroll
            for (int a = NA * q / 4; a < NA * (q + 1) / 4; ++a) {
                if (a % TN == 0) cvt_a(a / TN);
                mm(a / TN, a % TN);
            }
        };
        auto seg_pre = [&](int q) {   // the A fragments whose first atom runs in segment q
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (seg_of(i * TN) == q) ld_a(i);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f32x2v{0.f, 0.f};
        // The slot's patch pieces go out BETWEEN the segments, two per segment (the weight pieces, one or two, at the top of the slot:
        // they are what the slot's end waits for): issued in one block at the top of the slot they cost every wave ~1000 cycles in which
        // nothing else runs (a piece occupies the CU's one address path for >= 16 cycles and the issuing wave until it is accepted).
        constexpr int P3 = (PP + 2) / 3;
        ld_b();
        seg_pre(0);
        rd_row(0);
        __builtin_amdgcn_sched_barrier(0);
        seg_mm(0);
        fma_row();
        rd_row(1);
        seg_pre(1);
        if (DIN && dq.patch) issue_patch(dq.pstage, dq.pcoff, 0, P3);
        __builtin_amdgcn_sched_barrier(0);
        seg_mm(1);
        fma_row();
        rd_row(2);
        seg_pre(2);
        if (DIN && dq.patch) issue_patch(dq.pstage, dq.pcoff, P3, 2 * P3 < PP ? 2 * P3 : PP);
        __builtin_amdgcn_sched_barrier(0);
        seg_mm(2);
        fma_row();
        seg_pre(3);
        if (DIN && dq.patch) issue_patch(dq.pstage, dq.pcoff, 2 * P3 < PP ? 2 * P3 : PP, PP);
        __builtin_amdgcn_sched_barrier(0);
        seg_mm(3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h, l;
            split2(o[j][0], o[j][1], h, l);
            *reinterpret_cast<unsigned*>(smem + aw + j * 64) = h;
            *reinterpret_cast<unsigned*>(smem + ((aw + j * 64) ^ 32)) = l;
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: weights of the projection's step 0, patches of chunks 0 and 1
    issue_B(1, 0, 0, 0);                 // (the block's half is a dummy here: its step 0 is requested at the top of slot -1)
    issue_patch(0, 0);
    advance_issue();
    issue_patch(1, ic * 32);
    wait_vm<PP>();
    __builtin_amdgcn_s_barrier();

    // Slots s = -1 .. 2 total.  Iteration k = odd slot 2k + 1 [block: chunk k half 1; projection + depthwise: chunk k + 1 half 0] and even
    // slot 2k + 2 [block: chunk k + 1 half 0; projection + depthwise: chunk k + 1 half 1]; ct1 = index of chunk k + 1 inside its tile.
    // Slot -1 is the fill (the block's MFMAs run on nothing: their sums are dropped), slot 2 total the drain (it carries the block's last
    // epilogue; everything else in it works on stale data nobody reads) -- one loop body, one epilogue site.
    int k = -1, ct1 = 0, x_epi = xbase;
    const int S = 2 * total;
    for (int s = -1; s <= S; ++s) {
        const int odd = s & 1;
        const int stg = ((k + 1) & 1) * STAGE;
        const int ct2 = ct1 + 1 == nchunks ? 0 : ct1 + 1;
        const bool last0 = k >= 0 && ct1 == 0;        // chunk k is the last of its tile
        const bool rp = RPRE && res_on && last0;
        // ---- what this slot requests, the epilogue of the tile whose last MFMAs ran in the slot before (projection waves: odd slot, the
        // block's: even slot), the slot's arithmetic
        const bool epi_here = last0 && (odd != 0) == out2;
        SlotDma dq;
        dq.hm = odd ? 0 : 1; dq.cm = ct1 * 32;                  // block: step 2k + 2 (odd slot) / 2k + 3 (even slot)
        dq.hp = odd ? 1 : 0; dq.cp = (odd ? ct1 : ct2) * 32;    // projection: step 2k + 3 / 2k + 4
        dq.patch = odd && k >= 0; dq.pstage = k & 1;            // odd slot: chunk k + 2 into the stage chunk k has left
        dq.rp = rp && odd; dq.x = x_epi;
        if (dq.patch) advance_issue();
        dq.pcoff = ic * 32;
        if (epi_here) {
            // a slot with an epilogue issues its DMA pieces here, in one block: the epilogue must precede the slot's MFMAs, and its stores
            // should be the YOUNGEST entries of the in-order queue (they then have until the end of the next slot to drain) -- except with
            // a residual: the compiler's wait for the residual values would also cover the pieces just issued, so it goes first
            const bool epi_first = res_on;
            if (epi_first) epilogue(x_epi);
            issue_B(dq.hm, dq.cm, dq.hp, dq.cp);
            if (dq.patch) issue_patch(dq.pstage, dq.pcoff);
            if (!epi_first) epilogue(x_epi);
        } else {
            // the weights at the top: they have the whole slot to land; the patch pieces between the segments.  (Also tried: the weight pieces
            // inside the body too, behind segment wave % 3 -- deconv1_a + residual1_d 1991 -> 2175 us, every shape slower: less time to land.)
            issue_B(dq.hm, dq.cm, dq.hp, dq.cp);
            if (dq.rp) res_prefetch(dq.x);
        }
        const bool epi_last = epi_here && !res_on;            // its stores are younger than every DMA piece of this slot
        PIPE_STAMP(0)
        slot_body(odd, stg, !epi_here, dq);
        if (s == -1 && !out2) zero_acc();
        PIPE_STAMP(1)
        wait_lgkm0();
        PIPE_STAMP(2)
        if (odd) {
            // need: this slot's weight pieces; younger: residual loads, patch pieces, the projection's stores
            if (k < 0 || (!full && (epi_last || rp))) wait_vm<0>();
            else if (epi_last) wait_vm<PP + E>();
            else if (rp) wait_vm<PP + R>();
            else wait_vm<PP>();
        } else {
            // need: the patch pieces of the odd slot and this slot's weight pieces; younger: only this slot's stores
            if (epi_last && full) wait_vm<E>();
            else wait_vm<0>();
        }
        PIPE_STAMP(3)
        __builtin_amdgcn_s_barrier();
        PIPE_STAMP(4)
        if (!odd) {
            if (last0) x_epi += TW;
            ct1 = ct2;
            ++k;
        }
    }
    wait_vm<0>();   // the surplus DMA groups must have landed before this workgroup's LDS goes to the next one
    if (p.stamps && tid == p.ablate * 64) {   // (dev: the wave whose stamps are reported rides in the otherwise unused ablate field)
        long long* o = p.stamps + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#undef PIPE_STAMP
}

template <int BN, bool DUAL, bool OSPLIT, int EPI>
int launch2(const SepParams& q, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((sep_pipe2_kernel<BN, DUAL, OSPLIT, EPI>), grid, dim3(512), 0, st, q);
    return emd::check_launch("sep_pipe2_kernel");
}

}  // namespace

namespace emd {

// Shapes the software-pipelined kernel has an instance for: stride 1, 8 x 32 tiles (H % 8 == 0, W % 32 == 0), split-bf16, one output of up
// to 256 channels (fp32, or split32 from 128 channels on) or two of up to 128 each (no residual / second affine there).
bool sep_pipe2_covers(const SepParams& p) {
    if (p.stride != 1 || p.gen_a || p.H % 8 != 0 || p.W % 32 != 0 || p.Cin % 32 != 0 || p.Cin < 32 || p.Cin > 4064) return false;
    if (p.N2 > 0) return p.N <= 128 && p.N2 <= 128 && !p.out_split && !p.res && !p.scale2;
    if (p.out_split && p.N <= 64) return false;
    return p.N <= 256;
}

int sep_pipe2_launch(const SepParams& p, int B, hipStream_t st) {
    SepParams q = p;
    const int tiles_w = p.W / 32;
    const long wgs1 = (long)tiles_w * (p.H / 8) * B;
    int tpw = 1;   // several tiles per workgroup (the DMA ring runs on across them) where >= 4 workgroups per CU remain
    for (int t = 8; t >= 2; t >>= 1)
        if (tiles_w % t == 0 && wgs1 / t >= 1024) { tpw = t; break; }
    if (g_knobs.sep_tpw > 0 && tiles_w % g_knobs.sep_tpw == 0) tpw = g_knobs.sep_tpw;
    q.tpw = tpw;
    q.stamps = g_knobs.sep_stamps;
    q.ablate = g_knobs.sep_stamp_wave & 7;
    const dim3 grid(tiles_w / tpw, p.H / 8, B);
    q.xcd = g_knobs.sep_xcd && ((long)grid.x * grid.y * grid.z) % 8 == 0;
    const int epi = g_knobs.epi_width ? g_knobs.epi_width : ((p.res && p.N > 128) ? 4 : 1);
    if (p.N2 > 0) {
        if (p.N > 64 || p.N2 > 64) return launch2<256, true, false, 1>(q, grid, st);
        return launch2<128, true, false, 1>(q, grid, st);
    }
    if (p.N <= 64) return launch2<64, false, false, 1>(q, grid, st);
    if (p.N <= 128) return p.out_split ? launch2<128, false, true, 1>(q, grid, st) : launch2<128, false, false, 1>(q, grid, st);
    if (p.out_split) return epi == 4 ? launch2<256, false, true, 4>(q, grid, st) : launch2<256, false, true, 1>(q, grid, st);
    return epi == 4 ? launch2<256, false, false, 4>(q, grid, st) : launch2<256, false, false, 1>(q, grid, st);
}

}  // namespace emd
