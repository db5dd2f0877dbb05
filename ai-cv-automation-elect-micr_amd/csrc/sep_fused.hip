// Fused separable convolution: depthwise 3x3 (stride 1, rate 1, TF SAME) -> 1x1 on the matrix cores ->
// folded batch norms + relu6 (+ second affine + relu6, + residual), in ONE kernel.
// replaces: slim.separable_convolution2d + _batch_norm_fn + batch_then_activ, i.e. the whole
//           strided_conv_block of machine_learning/denoiser.py:110-136 (for stride 1), and the "+=" after it.
//
// Why: for the layers whose Cout fits one N-tile (<= 128) and that run at 256x256 / 512x512 the unfused
// pair (emd_dw3x3_f32 + emd_conv1x1_f32) is HBM-bound and moves the depthwise result through HBM twice
// (write, then read): in + dw + dw + out bytes.  Fused, the depthwise result only ever exists as the bf16
// hi/lo A-operand planes in LDS: in(+halo) + out bytes.
//
// Block = 256 threads = 4 waves; output tile = 8 x 16 pixels (128 GEMM rows) x all Cout (BN = 64 or 128).
// Per 32-channel chunk of Cin:
//   1. the (8+2) x (16+2) pixel fp32 input patch of the chunk is staged global -> registers (one chunk
//      ahead; unconditional loads, padding pixels read a zero buffer) -> LDS; pixel pitch 40 floats so the
//      depthwise reads below are bank-conflict free; the chunk's 9 x 32 depthwise weights take the same route;
//      GEN instances rebuild the chunk from a one-value-per-pixel tensor instead (emd_sep3x3_fused_gen_f32);
//   2. thread (pixel group of 4 along W, 4 channels) reads 3 x 6 patch vectors (18 ds_read_b128 for 16
//      outputs), accumulates the 9 taps in fp32, splits to bf16 hi/lo and writes the A planes;
//   3. v_mfma_f32_32x32x16_bf16, split-bf16 (3 passes) as in gemm_conv.hip; W tile staged like there.
// Epilogue identical to gemm_conv.hip (fp32 LDS staging, 16-byte stores / residual loads).  A workgroup walks up to 8
// tiles side by side with the staging pipeline running on across them; DESIGN.md 3.3 has the measurements.
#include <cstdlib>

#include "sep_params.hpp"

using namespace emd;

namespace {

// source of the zero-padding pixels: the patch loads are unconditional (a select on the loaded value would make the wave wait for
// the load where it is issued instead of a whole depthwise + MFMA phase later)
__device__ __attribute__((aligned(16))) float g_zero_px[4096];

// Knobs of the 512^2 / 256^2 layers' cache behaviour (dev, emd_debug_knob): nt_mask bit 1 = non-temporal output stores here (default on), sep_xcd = 0
// turns the XCD-contiguous tile order off.
inline int sep_nt() { return (g_knobs.nt_mask >> 1) & 1; }
inline int sep_xcd() { return g_knobs.sep_xcd; }

// BN = columns of the workgroup's GEMM tile: 64 / 128 (one output), or with DUAL the two outputs side by side: 128 = 64 | 64 on an
// 8 x 16 pixel tile, 256 = 128 | 128 on a 4 x 16 pixel tile (TH = 4: the accumulators of both outputs fit the same registers).
// WRES (BN = 64, Cin <= 64): the whole pointwise weight matrix (at most two 32-channel chunks, 10 KB each with the lo plane) is put into
// LDS once per workgroup instead of once per (tile, chunk): with a generated input it was 90 % of the bytes the loader moved.
template <int BN, int PASSES, bool GEN, int TH = 8, bool DUAL = false, bool WRES = false>
__global__ __launch_bounds__(256, 2) void sep_fused_kernel(const SepParams p) {
    constexpr int TW = 16, BM = TH * TW, BK = 32;
    constexpr int PH = TH + 2, PW = TW + 2, NPX = PH * PW;  // 10 x 18 = 180 patch pixels (6 x 18 = 108 for TH = 4)
    constexpr int PLD = 40;                                 // floats per patch pixel (32 + 8 pad)
    constexpr int LDK = BK + 8;                             // bf16 per A/B row (80 B)
    constexpr int WN = BN / 64, WM = 4 / WN;                // every wave owns 64 columns: 4 x 1, 2 x 2 or 1 x 4 waves
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NPL = PASSES == 3 ? 2 : 1;
    constexpr int P_PASSES = (NPX * 8 + 255) / 256;          // 1440 float4 -> 6 passes (864 -> 4)
    constexpr int W_PASSES = BN / 64;                        // BN rows x 4 chunks of 16 B
    constexpr int NPT = BM / 32;                             // pixels per thread along W in the depthwise role (4 or 2)
    static_assert(TM >= 1 && TN == 2 && (!DUAL || (!GEN && PASSES == 3 && WN >= 2)), "tile shape");
    constexpr int LDS_STAGE = BN + 4;
    constexpr int B_CHUNK = NPL * BN * LDK * 2;               // one chunk's W tile (hi [+ lo] plane)
    constexpr int PATCH_BYTES = NPX * PLD * 4, A_BYTES = NPL * BM * LDK * 2, B_BYTES = B_CHUNK * (WRES ? 2 : 1);
    static_assert(!WRES || (BN == 64 && !DUAL), "the resident-W form is the 64-column single-output instance");
    constexpr int STAGE_BYTES = BM * LDS_STAGE * 4;
    constexpr int DW_BYTES = 9 * BK * 4;                     // the chunk's depthwise weights [9][32]
    constexpr int DW_OFF = PATCH_BYTES + A_BYTES + B_BYTES > STAGE_BYTES ? PATCH_BYTES + A_BYTES + B_BYTES : STAGE_BYTES;   // clear of the staging tile
    constexpr int TILE_BYTES = DW_OFF + DW_BYTES;

    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES];
    float* patch = reinterpret_cast<float*>(smem);
    auto As = reinterpret_cast<uint16_t(*)[BM][LDK]>(smem + PATCH_BYTES);
    auto Bs = reinterpret_cast<uint16_t(*)[BN][LDK]>(smem + PATCH_BYTES + A_BYTES);
    float(*stage)[LDS_STAGE] = reinterpret_cast<float(*)[LDS_STAGE]>(smem);
    float* wks = reinterpret_cast<float*>(smem + DW_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    // Workgroups are handed to the 8 XCDs round-robin (id mod 8), so in launch order the tile under this one -- which shares two of
    // its ten patch rows -- runs on another XCD and the halo is fetched once per L2.  p.xcd: XCD k takes the k-th eighth of the tiles.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;   // a multiple of 8 (host check)
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t = (id & 7) * (total >> 3) + (id >> 3);
        bx = t % gridDim.x;
        by = (t / gridDim.x) % gridDim.y;
        bz = t / (gridDim.x * gridDim.y);
    }
    const int xbase = bx * p.tpw * TW, y0 = by * TH;
    int x0 = xbase;       // tile whose chunks are being computed (the epilogue's tile)
    const long img = (long)bz * p.H * p.W;  // pixel index of this image's (0,0)

    // ---- patch loader role: float4 #idx of the patch = (patch pixel idx/8, channel group idx%8)
    const float* psrc[P_PASSES];  // source pixel + channel group of chunk 0, or the zero buffer (padding / unused slots)
    bool pok[P_PASSES];           // generated input only: a real pixel (not padding)
    auto set_tile = [&](int xt) {
#pragma unroll
        for (int q = 0; q < P_PASSES; ++q) {
            const int idx = tid + q * 256;
            const float* o = GEN ? p.x : g_zero_px;
            bool ok = false;
            if (idx < NPX * 8) {
                const int ppx = idx >> 3, py = ppx / PW, px = ppx - py * PW;
                int gy = y0 - 1 + py, gx = xt - 1 + px;
                if (p.reflect) {  // index -1 -> 1, H -> H-2
                    gy = gy < 0 ? -gy : (gy >= p.H ? 2 * p.H - 2 - gy : gy);
                    gx = gx < 0 ? -gx : (gx >= p.W ? 2 * p.W - 2 - gx : gx);
                }
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    o = p.x + (img + (long)gy * p.W + gx) * p.ldx + (GEN ? 0 : (idx & 7) * 4);
                    ok = true;
                }
            }
            psrc[q] = o;
            pok[q] = ok;
        }
    };
    set_tile(xbase);
    int ptile = 0;             // tile the patch offsets belong to
    // ---- W loader role
    const int w_col = (tid & 3) * 8, w_row = tid >> 2;  // + 64 rows per pass
    const uint16_t* __restrict__ whi = p.Whi + (long)w_row * p.Cpad + w_col;
    const uint16_t* __restrict__ wlo = NPL == 2 ? p.Wlo + (long)w_row * p.Cpad + w_col : nullptr;
    // DUAL: the second half of the B tile's rows are the 1x1 projection's weights
    const uint16_t* __restrict__ w2hi = DUAL ? p.W2hi + (long)w_row * p.Cpad + w_col : nullptr;
    const uint16_t* __restrict__ w2lo = DUAL ? p.W2lo + (long)w_row * p.Cpad + w_col : nullptr;
    // ---- depthwise role: NPT consecutive pixels of tile row ty, 4 channels
    const int c4 = tid & 7, pg = tid >> 3;
    const int ty = pg / (TW / NPT), tx0 = (pg % (TW / NPT)) * NPT;

    f32x4 preg[P_PASSES];
    f32x4 gga = {0.f, 0.f, 0.f, 0.f}, ggt = gga;   // generated input: the chunk's a / t vectors
    u32x4 wh[W_PASSES], wl[W_PASSES];
#pragma unroll
    for (int q = 0; q < W_PASSES; ++q) wh[q] = wl[q] = u32x4{0, 0, 0, 0};
    f32x4 wkreg = {0.f, 0.f, 0.f, 0.f};

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = p.Cin / BK;
    if constexpr (WRES) {   // both chunks' W tiles, once (visible after the loop's first barrier)
        for (int c = 0; c < nchunks; ++c) {
            auto Bc = reinterpret_cast<uint16_t(*)[BN][LDK]>(smem + PATCH_BYTES + A_BYTES + c * B_CHUNK);
            *reinterpret_cast<u32x4*>(&Bc[0][w_row][w_col]) = *reinterpret_cast<const u32x4*>(whi + c * BK);
            if (NPL == 2) *reinterpret_cast<u32x4*>(&Bc[NPL - 1][w_row][w_col]) = *reinterpret_cast<const u32x4*>(wlo + c * BK);
        }
    }
    const int total = p.tpw * nchunks;   // (tile, chunk) steps of this workgroup: the staging pipeline runs on across tiles, so
                                         // a tile's epilogue overlaps the loads of the next tile's first chunk
    const int fr = lane & 31, fh = lane >> 5;

    constexpr bool EARLY_RES = BN == 64 && !DUAL;
    constexpr int E_C4 = BN / 4, E_RPP = 256 / E_C4, E_NROWS = BM / E_RPP;
    f32x4 rve[EARLY_RES ? E_NROWS : 1];
#pragma unroll
    for (int k = 0; k < (EARLY_RES ? E_NROWS : 1); ++k) rve[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (p.stamps) tprev = __builtin_amdgcn_s_memtime();
#define SEP_STAMP(i) if (p.stamps) { const long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tprev; tprev = t_; }
    for (int it = -1; it < total; ++it) {
        if (it >= 0) {
            // staged registers (chunk `it`) -> LDS
            if (GEN) {
                const float ghi = p.gen_act == 1 ? 6.f : __builtin_inff();
                const float gsl = p.gen_act == 4 ? 0.2f : 1.f, glo = (p.gen_act == 1 || p.gen_act == 2) ? 0.f : -__builtin_inff();
#pragma unroll
                for (int q = 0; q < P_PASSES; ++q) {
                    const float dv = preg[q][0];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float u = fmaf(dv, gga[c], ggt[c]);
                        preg[q][c] = pok[q] ? fminf(fmaxf(fmaxf(u, glo), gsl * u), ghi) : 0.f;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < P_PASSES; ++q) {
                const int idx = tid + q * 256;
                if (idx < NPX * 8) *reinterpret_cast<f32x4*>(patch + (idx >> 3) * PLD + (idx & 7) * 4) = preg[q];
            }
            if (tid < 72) *reinterpret_cast<f32x4*>(wks + tid * 4) = wkreg;
            if constexpr (!WRES) {
#pragma unroll
                for (int q = 0; q < W_PASSES; ++q) {
                    *reinterpret_cast<u32x4*>(&Bs[0][w_row + 64 * q][w_col]) = wh[q];
                    if (NPL == 2) *reinterpret_cast<u32x4*>(&Bs[NPL - 1][w_row + 64 * q][w_col]) = wl[q];
                }
            }
            __syncthreads();  // (1) patch + W tile of chunk `it` visible
        }
        SEP_STAMP(0)
        // stage chunk it+1 (the last iteration re-loads its own chunk: branch-free).  Issued BEFORE the
        // depthwise work below so the loads have the depthwise + MFMA phases (not just the MFMAs) to land.
        const int nx = it + 1 < total ? it + 1 : it;
        const int ntile = nx / nchunks;
        const int c0n = (nx - ntile * nchunks) * BK;
        if (ntile != ptile) {   // block-uniform: the next step belongs to the next tile, 16 pixels to the right
            const int xt = xbase + ntile * TW;
            // between two tiles that both lie off the image's left and right edges only the x origin changes: every real source
            // pixel moves 16 pixels on, the padding rows above / below the image stay padding
            if (xt - TW > 0 && xt + TW < p.W) {
                const long step = (long)TW * p.ldx;
#pragma unroll
                for (int q = 0; q < P_PASSES; ++q) psrc[q] += pok[q] ? step : 0;
            } else {
                set_tile(xt);
            }
            ptile = ntile;
        }
        if (GEN) {   // the chunk is generated from the one-value-per-pixel tensor when it is written to LDS
            gga = *reinterpret_cast<const f32x4*>(p.gen_a + c0n + (tid & 7) * 4);
            ggt = *reinterpret_cast<const f32x4*>(p.gen_t + c0n + (tid & 7) * 4);
#pragma unroll
            for (int q = 0; q < P_PASSES; ++q) preg[q][0] = *psrc[q];
        } else {
#pragma unroll
            for (int q = 0; q < P_PASSES; ++q) preg[q] = *reinterpret_cast<const f32x4*>(psrc[q] + c0n);
        }
        // the chunk's 9 x 32 depthwise weights travel through LDS too: one 16-byte load for 72 threads instead of nine for every thread
        if (tid < 72) wkreg = *reinterpret_cast<const f32x4*>(p.dw + ((long)(tid >> 3) * p.Cin + c0n) + (tid & 7) * 4);
        if constexpr (!WRES)
#pragma unroll
        for (int q = 0; q < W_PASSES; ++q) {
            constexpr int HALF = W_PASSES / 2;
            if (DUAL && q >= HALF) {
                wh[q] = *reinterpret_cast<const u32x4*>(w2hi + (long)(q - HALF) * 64 * p.Cpad + c0n);
                wl[q] = *reinterpret_cast<const u32x4*>(w2lo + (long)(q - HALF) * 64 * p.Cpad + c0n);
            } else {
                wh[q] = *reinterpret_cast<const u32x4*>(whi + (long)q * 64 * p.Cpad + c0n);
                if (NPL == 2) wl[q] = *reinterpret_cast<const u32x4*>(wlo + (long)q * 64 * p.Cpad + c0n);
            }
        }
        // 64-column instances: a thread's 8 residual vectors (32 registers) of the CURRENT tile are requested here, at the start of the
        // tile's last chunk: they land under its depthwise and MFMA phases (the 128-column instances have no room: they ask four
        // rows at a time in the epilogue, two exposed round trips per tile)
        if (EARLY_RES && p.res && it >= 0 && (it + 1) % nchunks == 0) {
            const int ecol = (tid % E_C4) * 4, eer = tid / E_C4;
            if (ecol < p.N) {
                const float* __restrict__ rt = p.res + (img + (long)y0 * p.W) * p.ldres + ecol;
#pragma unroll
                for (int k = 0; k < E_NROWS; ++k) {
                    const int r = eer + k * E_RPP;
                    rve[k] = *reinterpret_cast<const f32x4*>(rt + ((r >> 4) * p.W + (r & 15) + x0) * p.ldres);
                }
            }
        }
        SEP_STAMP(1)
        if (it >= 0) {
            // depthwise 3x3 from the LDS patch -> bf16 hi/lo A planes
            f32x4 wk[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const f32x4*>(wks + k * BK + c4 * 4);
            f32x4 o[NPT];
#pragma unroll
            for (int j = 0; j < NPT; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 pr[NPT + 2];
#pragma unroll
                for (int d = 0; d < NPT + 2; ++d)
                    pr[d] = *reinterpret_cast<const f32x4*>(patch + ((ty + i) * PW + tx0 + d) * PLD + c4 * 4);
#pragma unroll
                for (int j = 0; j < NPT; ++j)
#pragma unroll
                    for (int d = 0; d < 3; ++d) o[j] += wk[i * 3 + d] * pr[j + d];
            }
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int r = ty * TW + tx0 + j;
                unsigned h0, l0, h1, l1;
                split2(o[j][0], o[j][1], h0, l0);
                split2(o[j][2], o[j][3], h1, l1);
                *reinterpret_cast<u32x2*>(&As[0][r][c4 * 4]) = u32x2{h0, h1};
                if (NPL == 2) *reinterpret_cast<u32x2*>(&As[NPL - 1][r][c4 * 4]) = u32x2{l0, l1};
            }
        }
        SEP_STAMP(2)
        if (it < 0) continue;
        __syncthreads();  // (2) A planes of chunk `it` visible
        SEP_STAMP(3)
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
            if (DUAL && wn >= WN / 2) {
                // the 1x1 projection's A operand = the block's INPUT at the tile's own pixels: the centre of the fp32 patch, split
                // into bf16 hi / lo here (8 channels per lane) instead of going through a second pair of A planes
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm * (BM / WM) + i * 32 + fr;
                    const float* src = patch + (((r >> 4) + 1) * PW + (r & 15) + 1) * PLD + ks * 16 + fh * 8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                    split2(v0[0], v0[1], h0, l0);
                    split2(v0[2], v0[3], h1, l1);
                    split2(v1[0], v1[1], h2, l2);
                    split2(v1[2], v1[3], h3, l3);
                    ah[i] = __builtin_bit_cast(bf16x8, (u32x4{h0, h1, h2, h3}));
                    al[i] = __builtin_bit_cast(bf16x8, (u32x4{l0, l1, l2, l3}));
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm * (BM / WM) + i * 32 + fr;
                    ah[i] = *reinterpret_cast<const bf16x8*>(&As[0][r][ks * 16 + fh * 8]);
                    if (NPL == 2) al[i] = *reinterpret_cast<const bf16x8*>(&As[NPL - 1][r][ks * 16 + fh * 8]);
                }
            }
            auto Bc = WRES ? reinterpret_cast<uint16_t(*)[BN][LDK]>(smem + PATCH_BYTES + A_BYTES + (it % nchunks) * B_CHUNK) : Bs;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / WN) + j * 32 + fr;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bc[0][r][ks * 16 + fh * 8]);
                if (NPL == 2) bl[j] = *reinterpret_cast<const bf16x8*>(&Bc[NPL - 1][r][ks * 16 + fh * 8]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (PASSES == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        SEP_STAMP(4)
        __syncthreads();  // (3) fragment reads done before the next chunk overwrites patch / A / B
        SEP_STAMP(5)
        if ((it + 1) % nchunks != 0) continue;   // more chunks of this tile to come

        // ---- epilogue (see gemm_conv.hip): accumulators -> fp32 LDS tile -> 16-byte stores along the channel axis
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    stage[r][wn * (BN / WN) + j * 32 + fr] = acc[i][j][e];
                }
        __syncthreads();
        constexpr int C4 = BN / 4;
        constexpr int ROWS_PER_PASS = 256 / C4;
        const int ncol = (tid % C4) * 4, er = tid / C4;      // column of the staging tile
        const bool out2 = DUAL && ncol >= BN / 2;             // DUAL: the right half of the tile is the 1x1 projection
        const int n = out2 ? ncol - BN / 2 : ncol;            // channel of the output it belongs to
        if (n < (out2 ? p.N2 : p.N)) {
            const f32x4 s1 = *reinterpret_cast<const f32x4*>((out2 ? p.scale_b : p.scale1) + n);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>((out2 ? p.shift_b : p.shift1) + n);
            f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
            if (p.scale2 && !out2) {
                s2 = *reinterpret_cast<const f32x4*>(p.scale2 + n);
                t2 = *reinterpret_cast<const f32x4*>(p.shift2 + n);
            }
            float* __restrict__ outp = out2 ? p.y2 : p.y;
            const int ldo = out2 ? p.ldy2 : p.ldy;
            // one clamp form for every activation code: v = min(max(max(v, lo), slope*v), hi) -- (lo, slope, hi) = none: (-inf, 1, inf); relu6: (0, 1, 6);
            // relu: (0, 1, inf); leaky relu (graph G): (-inf, 0.2, inf); a clamped negative comes out as +0, as tf.nn.relu6 gives it
            const int actc = out2 ? 1 : p.act;   // the projection of a DUAL launch is conv + BN + relu6 (conv_block_not_sep)
            const float hi = actc == 1 ? 6.f : __builtin_inff();
            const float hi2 = actc == 2 ? __builtin_inff() : 6.f;   // second stage (extra BN): relu6, or relu with act code relu
            const float slope = actc == 4 ? 0.2f : 1.f, lo = (actc == 1 || actc == 2) ? 0.f : -__builtin_inff();
            const bool two = p.scale2 != nullptr && !out2;
            auto finish = [&](f32x4 v) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float u = fmaf(v[c], s1[c], t1[c]);
                    u = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                    if (two) u = fminf(fmaxf(fmaf(u, s2[c], t2[c]), 0.f), hi2);
                    v[c] = u;
                }
                return v;
            };
            constexpr int NROWS = BM / ROWS_PER_PASS;
            // tile row origin as a 64-bit pointer, the pixel relative to it as a 32-bit offset (the host checks 9 rows fit in 2^31 floats)
            // (x0 is folded into the 32-bit part on purpose: tile-invariant offsets would be hoisted out of the tile loop and pinned
            // in registers for the whole kernel)
            const long pix0 = img + (long)y0 * p.W;
            float* __restrict__ ytile = outp + pix0 * ldo + n;
            const float* __restrict__ rtile = p.res ? p.res + pix0 * p.ldres + n : nullptr;
            auto tpix = [&](int r) { return (r >> 4) * p.W + (r & 15) + x0; };
            // output stage: fp32 NHWC, or (single-output instances) the split32 layout through the pair exchange of emd::dw_store
            auto put = [&](int r, f32x4 v) {
                if (DUAL || !p.out_split) {
                    if (p.nt) store_nt16(ytile + tpix(r) * ldo, v);
                    else *reinterpret_cast<f32x4*>(ytile + tpix(r) * ldo) = v;
                    return;
                }
                unsigned h0, l0, h1, l1;
                split2(v[0], v[1], h0, l0);
                split2(v[2], v[3], h1, l1);
                const int q = n >> 2;
                const bool odd = q & 1;
                const unsigned r0 = emd::swap_pair(odd ? h0 : l0), r1 = emd::swap_pair(odd ? h1 : l1);
                unsigned char* g = reinterpret_cast<unsigned char*>(outp) + (pix0 + tpix(r)) * (long)ldo * 4 + (n >> 5) * 128;
                if (!odd) *reinterpret_cast<u32x4*>(g + (q & 7) * 8) = u32x4{h0, h1, r0, r1};
                else *reinterpret_cast<u32x4*>(g + 64 + ((q - 1) & 7) * 8) = u32x4{r0, r1, l0, l1};
            };
            if (EARLY_RES && p.res) {
#pragma unroll
                for (int k = 0; k < NROWS; ++k) {
                    const int r = er + k * ROWS_PER_PASS;
                    put(r, finish(*reinterpret_cast<const f32x4*>(&stage[r][ncol])) + rve[k]);
                }
            } else if (p.res && !DUAL) {
                // residual values are requested four rows at a time, before the first of them is used (the registers of the
                // next chunk's prefetch are live here: no room for all NROWS at once)
#pragma unroll
                for (int k0 = 0; k0 < NROWS; k0 += 4) {
                    f32x4 rv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = er + (k0 + k) * ROWS_PER_PASS;
                        rv[k] = *reinterpret_cast<const f32x4*>(rtile + tpix(r) * p.ldres);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = er + (k0 + k) * ROWS_PER_PASS;
                        put(r, finish(*reinterpret_cast<const f32x4*>(&stage[r][ncol])) + rv[k]);
                    }
                }
            } else {
#pragma unroll 4
                for (int r = er; r < BM; r += ROWS_PER_PASS)
                    put(r, finish(*reinterpret_cast<const f32x4*>(&stage[r][ncol])));
            }
        }
        __syncthreads();  // the staging tile is read out before the next tile's patch overwrites it
        SEP_STAMP(6)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        x0 += TW;
    }
    if (p.stamps && tid == 0) {
        long long* o = p.stamps + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#undef SEP_STAMP
}


// several tiles per workgroup where that still leaves >= 8 workgroups per CU
int tiles_per_workgroup(int tiles_w, long wgs1) {
    const int force = g_knobs.sep_tpw;
    int tpw = 1;
    for (int t = 8; t >= 2; t >>= 1)
        if (tiles_w % t == 0 && wgs1 / t >= 2048) { tpw = t; break; }
    if (force > 0 && tiles_w % force == 0) tpw = force;
    return tpw;
}

// BN = 64, Cin <= 64, split-bf16: the pointwise weights stay in LDS for the workgroup's whole life
int launch_wres(const SepParams& p, int B, hipStream_t st) {
    SepParams q = p;
    const int tiles_w = p.W / 16;
    q.tpw = tiles_per_workgroup(tiles_w, (long)tiles_w * (p.H / 8) * B);
    q.stamps = g_knobs.sep_stamps;
    const dim3 grid(tiles_w / q.tpw, p.H / 8, B);
    q.nt = sep_nt();
    q.xcd = sep_xcd() && ((long)grid.x * grid.y * grid.z) % 8 == 0;
    if (p.gen_a) hipLaunchKernelGGL((sep_fused_kernel<64, 3, true, 8, false, true>), grid, dim3(256), 0, st, q);
    else hipLaunchKernelGGL((sep_fused_kernel<64, 3, false, 8, false, true>), grid, dim3(256), 0, st, q);
    return emd::check_launch("sep_fused_kernel<resident W>");
}

template <int BN>
int launch(const SepParams& p, int B, int passes, hipStream_t st) {
    if constexpr (BN == 64) {
        const int wres = g_knobs.sep_wres;   // dev knob: 0 = per-chunk W loads
        if (wres && p.Cin <= 64 && passes == 3) return launch_wres(p, B, st);
    }
    SepParams q = p;
    const int tiles_w = p.W / 16;
    q.tpw = tiles_per_workgroup(tiles_w, (long)tiles_w * (p.H / 8) * B);
    q.stamps = g_knobs.sep_stamps;
    const dim3 grid(tiles_w / q.tpw, p.H / 8, B);
    q.nt = sep_nt();
    q.xcd = sep_xcd() && ((long)grid.x * grid.y * grid.z) % 8 == 0;
    if (p.gen_a) {
        if (passes == 3)
            hipLaunchKernelGGL((sep_fused_kernel<BN, 3, true>), grid, dim3(256), 0, st, q);
        else
            hipLaunchKernelGGL((sep_fused_kernel<BN, 1, true>), grid, dim3(256), 0, st, q);
    } else if (passes == 3)
        hipLaunchKernelGGL((sep_fused_kernel<BN, 3, false>), grid, dim3(256), 0, st, q);
    else
        hipLaunchKernelGGL((sep_fused_kernel<BN, 1, false>), grid, dim3(256), 0, st, q);
    return emd::check_launch("sep_fused_kernel");
}

// DUAL: 64 | 64 columns on 8 x 16 pixel tiles, or 128 | 128 columns on 4 x 16 pixel tiles
int launch_dual(const SepParams& p, int B, hipStream_t st) {
    SepParams q = p;
    const int tiles_w = p.W / 16;
    const bool wide = p.N > 64 || p.N2 > 64;
    const int th = wide ? 4 : 8;
    q.tpw = tiles_per_workgroup(tiles_w, (long)tiles_w * (p.H / th) * B);
    q.stamps = g_knobs.sep_stamps;
    const dim3 grid(tiles_w / q.tpw, p.H / th, B);
    q.nt = sep_nt();
    q.xcd = sep_xcd() && ((long)grid.x * grid.y * grid.z) % 8 == 0;
    if (wide)
        hipLaunchKernelGGL((sep_fused_kernel<256, 3, false, 4, true>), grid, dim3(256), 0, st, q);
    else
        hipLaunchKernelGGL((sep_fused_kernel<128, 3, false, 8, true>), grid, dim3(256), 0, st, q);
    return emd::check_launch("sep_fused_kernel<dual>");
}

}  // namespace

// 128 < Cout <= 256: one N tile of 256 columns on 4 x 16 pixel tiles (sep_fused_kernel<256, PASSES, false, 4>).  Against depthwise ->
// HBM -> pointwise at [32,128,128,.]: 128 -> 256 256 vs 335 us, 256 -> 256 (+ residual) 537 vs 609 us, 384 -> 256 773 vs 734 us: the
// rule takes it up to Cin = 256.  dev knob sep_wide: 0 = never, 2 = whenever it fits.
inline int sep_wide() { return g_knobs.sep_wide; }

// stride 2 (round 3): only the LDS-DMA pipelined kernel has the form (csrc/sep_pipe.hip, STRIDE = 2) -- even sizes with H % 8 == 0,
// W % 32 == 0 (output tiles of 4 x 16 pixels), Cout <= 256; emd_sep3x3_fused_s2_f32
static bool sep_s2_supported(int H, int W, int Cin, int Cout) {
    // ask the kernel that has the form (its dev knob sep_pipe = 0 switches it off: hosts then take depthwise + pointwise, they do not get an error)
    emd::SepParams q{};
    q.H = H; q.W = W; q.Cin = Cin; q.N = Cout; q.stride = 2;
    return Cout % 4 == 0 && Cout >= 4 && emd::sep_pipe_covers(q, 3);
}

extern "C" int emd_sep3x3_fused_supported(int H, int W, int Cin, int Cout, int stride, int rate) {
    if (stride == 2 && rate == 1) return sep_s2_supported(H, W, Cin, Cout);
    const bool wide = Cout > 128 && Cout <= 256 && (sep_wide() == 2 || (sep_wide() == 1 && Cin <= 256));
    return stride == 1 && rate == 1 && H % 8 == 0 && W % 16 == 0 && Cin % 32 == 0 && Cin >= 32 && Cin <= 4096 && Cout % 4 == 0 &&
           Cout >= 4 && (Cout <= 128 || wide);
}

static int sep_fused_entry(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                           const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                           const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                           int precision, int reflect, emd_stream_t stream, const float* gen_a = nullptr,
                           const float* gen_t = nullptr, int gen_act = 0, int out_split = 0) {
    EMD_REQUIRE(x && dw && whi && scale1 && shift1 && y, EMD_E_INVALID, "emd_sep3x3_fused_f32: null pointer");
    EMD_REQUIRE(precision == 1 || precision == 3, EMD_E_INVALID, "emd_sep3x3_fused_f32: bad precision");
    EMD_REQUIRE(precision == 1 || wlo, EMD_E_INVALID, "emd_sep3x3_fused_f32: the split-bf16 mode needs the lo plane");
    EMD_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), EMD_E_INVALID, "emd_sep3x3_fused_f32: scale2/shift2 pair");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_sep3x3_fused_f32: bad shape");
    EMD_REQUIRE(emd_sep3x3_fused_supported(H, W, Cin, Cout, 1, 1), EMD_E_UNSUPPORTED,
                "emd_sep3x3_fused_f32: needs H%8==0, W%16==0, Cin%32==0, Cout%4==0, Cout<=128 or Cout<=256 with Cin<=256 (use emd_dw3x3_f32 + emd_conv1x1_f32)");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_sep3x3_fused_f32: B > 65535");
    EMD_REQUIRE(9L * W * (ldy > ldres ? ldy : ldres) < (1L << 31), EMD_E_UNSUPPORTED,
                "emd_sep3x3_fused_f32: 9 image rows of the output must span fewer than 2^31 floats");
    EMD_REQUIRE(!out_split || (Cout % 32 == 0 && ldy % 32 == 0 && (reinterpret_cast<uintptr_t>(y) & 127u) == 0), EMD_E_ALIGN,
                "emd_sep3x3_fused_out_f32: a split32 output needs Cout % 32 == 0, ldy % 32 == 0 and y 128-byte aligned");
    EMD_REQUIRE((gen_a ? ldx >= 1 : (ldx % 4 == 0 && ldx >= Cin)) && ldy % 4 == 0 && ldy >= Cout &&
                    (!res || (ldres % 4 == 0 && ldres >= Cout)),
                EMD_E_ALIGN, "emd_sep3x3_fused_f32: pixel strides must be multiples of 4 and >= the channel count");
    EMD_REQUIRE(!gen_a || (gen_t && emd::aligned16(gen_a) && emd::aligned16(gen_t) && gen_act >= 0 && gen_act <= 4 && gen_act != 3),
                EMD_E_INVALID, "emd_sep3x3_fused_gen_f32: gen_a / gen_t must be 16-byte aligned device vectors, gen_act an EMD_ACT_* code");
    EMD_REQUIRE(emd::aligned16(x) && emd::aligned16(dw) && emd::aligned16(whi) && (!wlo || emd::aligned16(wlo)) &&
                    emd::aligned16(y) && (!res || emd::aligned16(res)) && emd::aligned16(scale1) &&
                    emd::aligned16(shift1) && (!scale2 || (emd::aligned16(scale2) && emd::aligned16(shift2))),
                EMD_E_ALIGN, "emd_sep3x3_fused_f32: pointers must be 16-byte aligned");
    if (B == 0) return EMD_OK;
    SepParams p{};
    p.x = x; p.dw = dw; p.Whi = whi; p.Wlo = wlo; p.y = y; p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.H = H; p.W = W; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK; p.N = Cout;
    p.ldx = ldx; p.ldy = ldy; p.ldres = ldres; p.act = act; p.reflect = reflect; p.stride = 1;
    p.gen_a = gen_a; p.gen_t = gen_t; p.gen_act = gen_act; p.out_split = out_split ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (emd::sep_pipe_covers(p, precision)) return emd::sep_pipe_launch(p, B, st);   // W % 32 == 0: the LDS-DMA pipelined kernel
    if (Cout > 128) {   // the wide single-output form (dev): 4 x 16 pixel tiles, split-bf16 only, no generated input
        EMD_REQUIRE(!gen_a, EMD_E_UNSUPPORTED, "emd_sep3x3_fused_gen_f32: Cout > 128 has no generated-input form");
        SepParams q = p;
        const int tiles_w = p.W / 16;
        q.tpw = tiles_per_workgroup(tiles_w, (long)tiles_w * (p.H / 4) * B);
        q.stamps = g_knobs.sep_stamps;
        const dim3 grid(tiles_w / q.tpw, p.H / 4, B);
        q.nt = sep_nt();
        q.xcd = sep_xcd() && ((long)grid.x * grid.y * grid.z) % 8 == 0;
        if (precision == 3) hipLaunchKernelGGL((sep_fused_kernel<256, 3, false, 4, false>), grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((sep_fused_kernel<256, 1, false, 4, false>), grid, dim3(256), 0, st, q);
        return emd::check_launch("sep_fused_kernel<wide>");
    }
    return Cout <= 64 ? launch<64>(p, B, precision, st) : launch<128>(p, B, precision, st);
}

extern "C" int emd_sep3x3_fused_f32(const float* x, int ldx, const float* dw, const uint16_t* whi,
                                    const uint16_t* wlo, const float* scale1, const float* shift1,
                                    const float* scale2, const float* shift2, const float* res, int ldres,
                                    float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                    int precision, emd_stream_t stream) {
    return sep_fused_entry(x, ldx, dw, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, B, H, W, Cin, Cout, act,
                           precision, 0, stream);
}

// The stride-2 separable block (strided_conv_block(stride=2), machine_learning/denoiser.py:258, :273, :288) in one launch: x [B,H,W,Cin]
// (H, W even; TF SAME = no padding before, one pixel after) -> y [B,H/2,W/2,Cout].  Split-bf16.  Same arithmetic as emd_dw3x3_f32(stride 2)
// followed by emd_conv1x1_f32; the depthwise result never exists in memory.
static int sep_s2_entry(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                        const float* shift1, const float* scale2, const float* shift2, const float* res, int ldres, float* y, int ldy,
                        int B, int H, int W, int Cin, int Cout, int act, int reflect, emd_stream_t stream);

extern "C" int emd_sep3x3_fused_s2_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                                       const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                       const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                       emd_stream_t stream) {
    return sep_s2_entry(x, ldx, dw, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, B, H, W, Cin, Cout, act, 0, stream);
}

// The same with the depthwise stage on the tf.pad(REFLECT, 1) image and VALID padding: graph G's strided_conv_block(stride 2,
// pad_size = (1, 1)) (misc_py/gan-infilling-100.py:205-243, the down-sampling layers :345-352): output pixel (i, j) reads input rows
// 2 i - 1 .. 2 i + 1 (row -1 = row 1); on even sizes nothing is read beyond the last row / column.
extern "C" int emd_sep3x3_fused_s2_reflect_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                                               const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                               const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout,
                                               int act, emd_stream_t stream) {
    return sep_s2_entry(x, ldx, dw, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, B, H, W, Cin, Cout, act, 1, stream);
}

static int sep_s2_entry(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                        const float* shift1, const float* scale2, const float* shift2, const float* res, int ldres, float* y, int ldy,
                        int B, int H, int W, int Cin, int Cout, int act, int reflect, emd_stream_t stream) {
    EMD_REQUIRE(x && dw && whi && wlo && scale1 && shift1 && y, EMD_E_INVALID, "emd_sep3x3_fused_s2_f32: null pointer");
    EMD_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), EMD_E_INVALID, "emd_sep3x3_fused_s2_f32: scale2/shift2 pair");
    EMD_REQUIRE(B >= 0 && H >= 2 && W >= 2, EMD_E_INVALID, "emd_sep3x3_fused_s2_f32: bad shape");
    EMD_REQUIRE(sep_s2_supported(H, W, Cin, Cout), EMD_E_UNSUPPORTED,
                "emd_sep3x3_fused_s2_f32: needs H%8==0, W%32==0, Cin%32==0, Cout%4==0, Cout<=256 (use emd_dw3x3_f32 + emd_conv1x1_f32)");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_sep3x3_fused_s2_f32: B > 65535");
    EMD_REQUIRE(ldx % 4 == 0 && ldx >= Cin && ldy % 4 == 0 && ldy >= Cout && (!res || (ldres % 4 == 0 && ldres >= Cout)), EMD_E_ALIGN,
                "emd_sep3x3_fused_s2_f32: pixel strides must be multiples of 4 and >= the channel count");
    EMD_REQUIRE(9L * W * (ldy > ldres ? ldy : ldres) < (1L << 31), EMD_E_UNSUPPORTED,
                "emd_sep3x3_fused_s2_f32: 9 image rows of the output must span fewer than 2^31 floats");
    EMD_REQUIRE(emd::aligned16(x) && emd::aligned16(dw) && emd::aligned16(whi) && emd::aligned16(wlo) && emd::aligned16(y) &&
                    (!res || emd::aligned16(res)) && emd::aligned16(scale1) && emd::aligned16(shift1) &&
                    (!scale2 || (emd::aligned16(scale2) && emd::aligned16(shift2))),
                EMD_E_ALIGN, "emd_sep3x3_fused_s2_f32: pointers must be 16-byte aligned");
    if (B == 0) return EMD_OK;
    SepParams p{};
    p.x = x; p.dw = dw; p.Whi = whi; p.Wlo = wlo; p.y = y; p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.H = H; p.W = W; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK; p.N = Cout;
    p.ldx = ldx; p.ldy = ldy; p.ldres = ldres; p.act = act; p.stride = 2; p.reflect = reflect ? 1 : 0;
    EMD_REQUIRE(emd::sep_pipe_covers(p, 3), EMD_E_UNSUPPORTED, "emd_sep3x3_fused_s2_f32: the pipelined kernel is switched off (dev knob sep_pipe)");
    return emd::sep_pipe_launch(p, B, static_cast<hipStream_t>(stream));
}

// emd_sep3x3_fused_f32 writing y as a split32 tensor (Cout % 32 == 0; pitch ldy 4-byte units, % 32): the producer of a split32
// convolution's input (graph D: deconv1_b -> deconv1to0) then writes no fp32 activation and needs no converter pass.
extern "C" int emd_sep3x3_fused_out_f32(const float* x, int ldx, const float* dw, const uint16_t* whi,
                                        const uint16_t* wlo, const float* scale1, const float* shift1,
                                        const float* scale2, const float* shift2, const float* res, int ldres,
                                        void* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                        emd_stream_t stream) {
    return sep_fused_entry(x, ldx, dw, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, static_cast<float*>(y), ldy, B, H, W, Cin,
                           Cout, act, 3, 0, stream, nullptr, nullptr, 0, 1);
}

// The same with the depthwise stage reading the tf.pad(REFLECT, 1) border instead of zeros: graph G's
// strided_conv_block(stride 1, pad_size=(1,1)) (misc_py/gan-infilling-100.py:205-243); act is usually EMD_ACT_LEAKY.
extern "C" int emd_sep3x3_fused_reflect_f32(const float* x, int ldx, const float* dw, const uint16_t* whi,
                                            const uint16_t* wlo, const float* scale1, const float* shift1,
                                            const float* scale2, const float* shift2, const float* res, int ldres,
                                            float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                            int precision, emd_stream_t stream) {
    return sep_fused_entry(x, ldx, dw, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, B, H, W, Cin, Cout, act,
                           precision, 1, stream);
}

// The fused separable conv on a GENERATED input: the layer's Cin-channel input is act(d[pixel] * gen_a[c] + gen_t[c]), d a
// one-value-per-pixel tensor (pitch ldd floats) -- what emd_cin1_f32 / emd_cin1_k7_reflect_f32 would write out for the layer
// that follows the one fed by the 1-channel image (cnn0 -> cnn0_last, machine_learning/denoiser.py:252-255; the generator's first
// two layers, misc_py/gan-infilling-100.py:343-349).  The Cin-channel tensor never exists in memory.
extern "C" int emd_sep3x3_fused_gen_f32(const float* d, int ldd, const float* gen_a, const float* gen_t, int gen_act,
                                        const float* dw, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                                        const float* shift1, const float* scale2, const float* shift2, const float* res,
                                        int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                        int precision, int reflect, emd_stream_t stream) {
    EMD_REQUIRE(gen_a && gen_t, EMD_E_INVALID, "emd_sep3x3_fused_gen_f32: null gen_a / gen_t");
    return sep_fused_entry(d, ldd, dw, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, B, H, W, Cin, Cout, act,
                           precision, reflect, stream, gen_a, gen_t, gen_act);
}

// dev hook (not in the header): per-workgroup phase cycle sums for tools/sep_bench.py
extern "C" void emd_debug_sep_stamps(void* buf) { g_knobs.sep_stamps = static_cast<long long*>(buf); }

// The decoder pair "separable conv + 1x1 residual projection of the same input" in ONE launch (machine_learning/denoiser.py:356-359,
// :368-371, :380-383: deconv*_a = strided_conv_block(concat) and residual*_d = conv_block_not_sep(concat, kernel_size=1)):
//   y  = relu6(BN2(BN1(pointwise(depthwise3x3(x)))))        the fused separable conv of emd_sep3x3_fused_f32 (scale1 / shift1)
//   y2 = relu6(BN(x * W2 + bias))                           a 1x1 conv of x itself (scale_b / shift_b: bias and BN folded)
// Both read the 384- or 128-channel input, the largest tensors of the decoder; fused, it comes from HBM once.  The input patch of
// a tile is staged in LDS for the depthwise stage anyway; its centre pixels are the projection's A operand.
extern "C" int emd_sep3x3_dual_supported(int H, int W, int Cin, int Cout, int Cout2) {
    const bool wide = Cout > 64 || Cout2 > 64;
    return H % (wide ? 4 : 8) == 0 && W % 16 == 0 && Cin % 32 == 0 && Cin >= 32 && Cin <= 4096 && Cout % 4 == 0 && Cout2 % 4 == 0 &&
           Cout >= 4 && Cout2 >= 4 && Cout <= 128 && Cout2 <= 128;
}

// 1 when the one-launch form is the faster route for the shape (a pure function of the shape; graph D's decoder pairs): always for two
// outputs of up to 64 channels; for wider ones only where the LDS-DMA pipelined kernel takes the launch (W % 32 == 0, H % 8 == 0:
// 384 -> 128 | 128 at 256^2 2.07 ms against 1.32 + 0.99 for the pair; the 4 x 16-tile form of sep_fused.hip is slower than the pair).
extern "C" int emd_sep3x3_dual_preferred(int H, int W, int Cin, int Cout, int Cout2) {
    if (!emd_sep3x3_dual_supported(H, W, Cin, Cout, Cout2)) return 0;
    if (Cout <= 64 && Cout2 <= 64) return 1;
    return H % 8 == 0 && W % 32 == 0;
}

extern "C" int emd_sep3x3_dual_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                                   const float* scale1, const float* shift1, float* y, int ldy, const uint16_t* w2hi,
                                   const uint16_t* w2lo, const float* scale_b, const float* shift_b, float* y2, int ldy2, int B, int H,
                                   int W, int Cin, int Cout, int Cout2, int act, emd_stream_t stream) {
    EMD_REQUIRE(x && dw && whi && wlo && scale1 && shift1 && y && w2hi && w2lo && scale_b && shift_b && y2, EMD_E_INVALID,
                "emd_sep3x3_dual_f32: null pointer");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_sep3x3_dual_f32: bad shape");
    EMD_REQUIRE(emd_sep3x3_dual_supported(H, W, Cin, Cout, Cout2), EMD_E_UNSUPPORTED,
                "emd_sep3x3_dual_f32: needs H%8==0 (H%4==0 above 64 output channels), W%16==0, Cin%32==0, Cout%4==0, Cout<=128 (both outputs)");
    EMD_REQUIRE(B <= 65535, EMD_E_UNSUPPORTED, "emd_sep3x3_dual_f32: B > 65535");
    EMD_REQUIRE(9L * W * (ldy > ldy2 ? ldy : ldy2) < (1L << 31), EMD_E_UNSUPPORTED,
                "emd_sep3x3_dual_f32: 9 image rows of an output must span fewer than 2^31 floats");
    EMD_REQUIRE(ldx % 4 == 0 && ldx >= Cin && ldy % 4 == 0 && ldy >= Cout && ldy2 % 4 == 0 && ldy2 >= Cout2, EMD_E_ALIGN,
                "emd_sep3x3_dual_f32: pixel strides must be multiples of 4 and >= the channel count");
    EMD_REQUIRE(emd::aligned16(x) && emd::aligned16(dw) && emd::aligned16(whi) && emd::aligned16(wlo) && emd::aligned16(w2hi) &&
                    emd::aligned16(w2lo) && emd::aligned16(y) && emd::aligned16(y2) && emd::aligned16(scale1) && emd::aligned16(shift1) &&
                    emd::aligned16(scale_b) && emd::aligned16(shift_b),
                EMD_E_ALIGN, "emd_sep3x3_dual_f32: pointers must be 16-byte aligned");
    EMD_REQUIRE(y != y2, EMD_E_INVALID, "emd_sep3x3_dual_f32: the two outputs alias");
    if (B == 0) return EMD_OK;
    SepParams p{};
    p.x = x; p.dw = dw; p.Whi = whi; p.Wlo = wlo; p.y = y; p.res = nullptr;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = nullptr; p.shift2 = nullptr;
    p.H = H; p.W = W; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK; p.N = Cout;
    p.ldx = ldx; p.ldy = ldy; p.ldres = 0; p.act = act; p.reflect = 0; p.stride = 1;
    p.W2hi = w2hi; p.W2lo = w2lo; p.y2 = y2; p.scale_b = scale_b; p.shift_b = shift_b; p.N2 = Cout2; p.ldy2 = ldy2;
    if (emd::sep_pipe_covers(p, 3)) return emd::sep_pipe_launch(p, B, static_cast<hipStream_t>(stream));
    return launch_dual(p, B, static_cast<hipStream_t>(stream));
}
