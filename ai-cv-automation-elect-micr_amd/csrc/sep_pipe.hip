// Fused separable convolution, pipelined form: depthwise 3x3 (stride 1, rate 1) -> 1x1 on the matrix cores -> folded batch norms +
// relu6 (+ second affine, + residual), with the input patch brought into LDS by LDS-DMA into a two-stage ring.
// replaces (like sep_fused.hip, whose arithmetic it repeats bit for bit): slim.separable_convolution2d + _batch_norm_fn +
//           batch_then_activ = strided_conv_block of machine_learning/denoiser.py:110-136 (stride 1) and the "+=" after it; with
//           two outputs also the decoder's 1x1 residual projection of the same input (denoiser.py:356-359, :368-371, :380-383).
//
// Why a second kernel: sep_fused.hip stages a chunk's patch global -> registers -> LDS one chunk ahead; its in-kernel stamps show
// 40 % of a workgroup's life waiting for / issuing those loads, and the prefetch registers keep it at 222-256 VGPRs.  Here nothing
// is staged through registers: the loads of chunk t+1 (and t+2) are in flight while chunk t is computed, 46 KB per stage.
//
// Workgroup = 512 threads = 8 waves, ONE per CU (all of its 160 KB of LDS): output tile 8 x 32 pixels (256 GEMM rows) x all Cout
// (BN = 64 / 128 / 256 columns; with two outputs 64 | 64 or 128 | 128).  Per 32-channel chunk of Cin ("step"):
//   DMA    the (8+2) x (32+2) pixel fp32 patch of the chunk + the chunk's 9 x 32 depthwise weights: 359 slots of 128 B, 45 pieces of
//          1 KiB (global_load_lds_dwordx4: one wave instruction = 8 slots); padding pixels come from a zero buffer; the pointwise
//          weight tile (BN rows x [32 hi | 32 lo] bf16) the same way, rows XOR-swizzled on the SOURCE side as in gemm_split.hip;
//   stage 1  thread (4 pixels along W, 4 channels) reads 3 x 6 patch vectors (18 ds_read_b128), 9 taps in fp32, splits to bf16
//          hi / lo and writes the A rows (swizzled like the weight rows).  The slot pitch of a patch row is 35 (odd) and the four
//          pixel groups of a 32-lane half wave form a 2 x 2 block (2 rows x 2 groups): every 16-lane group of a ds_read_b128
//          then covers the 64 banks exactly once with the plain, unswizzled patch;
//   stage 2  v_mfma_f32_32x32x16_bf16, split-bf16 (Alo*Whi + Ahi*Wlo + Ahi*Whi), every wave 64 columns x (256 / WM) rows.
// Two barriers per step.  Epilogue straight from the accumulators (lane = output channel, a wave store = two 128-byte runs): no
// staging tile -- the LDS that would hold it is the landing zone of the next tile's first two steps.
//
// Schedules (template parameter MODE):
//   0  all waves stage 1 then stage 2; the patch of step t+2 is requested right after barrier B of step t;
//   1  patch of step t+1 requested after barrier A of step t (the two-output instances: the projection reads the patch in
//      stage 2, so its stage cannot be refilled earlier);
// (Measured and removed, round 3: a ping-pong schedule -- waves 0-3 on tile rows 0-3, waves 4-7 on rows 4-7 half a step apart, one
// wave of each SIMD in stage 1 while its partner is in stage 2.  With two stages the patch can only be requested one step ahead
// there, and a lone wave's stage 1 takes 2100 cycles against 1400 for two waves side by side: 1.71 ms against 1.48 on the
// 512^2 x 128 -> 64 layer, slower on every shape.  profiles/r03_experiments.txt.)
// vmcnt bookkeeping is exact for full tiles (every wave issues the same number of DMA pieces per step and, per tile, a fixed number
// of stores), conservative otherwise.
#include "sep_pipe_common.hpp"

namespace {

using namespace emd;
using namespace emd::sp;

// source of the zero-padding pixels (TF SAME) and of the unused slots: 16 KB, so that "+ chunk offset" stays inside for Cin <= 4064
__device__ __attribute__((aligned(16))) float g_zero_pipe[4096];

// NW = 8: 512 threads, 8 x 32 pixel tiles, one workgroup per CU; NW = 4: 256 threads, 8 x 16 pixel tiles, two independent workgroups per
// CU (80 KB of LDS each: one weight tile, up to 64 output channels) whose phases overlap each other without any schedule.
// STRIDE = 2 (round 3; strided_conv_block with stride 2, machine_learning/denoiser.py:258, :273, :288: TF SAME on even sizes = no padding
// before, one pixel after): output tile 4 x 16 pixels (64 GEMM rows) from a 9 x 33 pixel patch, a thread one output pixel x 4 channels
// in stage 1 (9 patch + 9 weight reads), every wave one 32 x (BN / 4) accumulator block -- the depthwise result of the strided
// blocks no longer goes through HBM (cnn0_strided: 0.54 GB written and read back per batch).
// GEN (round 4): the layer's input is GENERATED -- act(d[pixel] * gen_a[c] + gen_t[c]), d a one-value-per-pixel tensor (the 1 -> 64
// channel separable conv in front of it, whose pointwise half is rank 1; sep_fused.hip's generated-input form, same arithmetic) -- so
// there is no patch to fetch: where the other instances request the DMA pieces of a chunk, every lane computes the 16 bytes its DMA
// lane would have received and writes them to the same place.  d of the NEXT tile is loaded one tile ahead; the depthwise weights of
// all chunks (Cin <= 64) sit in their own 2.25 KiB of LDS for the workgroup's life.
template <int BN, bool DUAL, int MODE, bool OSPLIT, int NW = 8, int STRIDE = 1, int EPI = 1, bool GEN = false>     // EPI: dwords a lane stores at a time (epilogue)
__global__ __launch_bounds__(NW * 64, 2) void sep_pipe_kernel(const SepParams p) {
    constexpr int TW = STRIDE == 2 ? 16 : 4 * NW, TH = STRIDE == 2 ? 4 : 8, BM = TH * TW;
    constexpr int PW = STRIDE * TW + 3 - STRIDE, PH = STRIDE * TH + 3 - STRIDE, PWS = PW | 1;   // patch pixels per row; slot pitch odd (see above): 34 -> 35, 18 -> 19, 33
    // The chunk's depthwise weights (nine taps x 32 channels = nine 128-byte slots) ride in the PAD column of the patch rows where there
    // is one (stride 1: 34 -> 35, 18 -> 19; tap t in row t), else in nine slots behind the patch: one KiB less per stage, which is what
    // lets the 4-wave form with 128 columns fit two workgroups per CU (2 x 24 + 16 + 16 = 80 KiB).
    constexpr bool WPAD = PWS > PW && PH >= 9;
    constexpr int NPATCH = PH * PWS, NSLOT = WPAD ? NPATCH : NPATCH + 9;
    constexpr int WK0 = WPAD ? PW : NPATCH, WKS = WPAD ? PWS : 1;    // slot of tap t: WK0 + t * WKS
    constexpr int NPIECE = (NSLOT + 7) / 8;                   // DMA pieces of 8 slots (8 x 32 tiles: 44)
    constexpr int PP = (NPIECE + NW - 1) / NW;                // 6 per wave (the surplus ones repeat the wave's previous piece)
    constexpr int STAGE = NPIECE * 1024;
    constexpr int A_BYTES = BM * 128, B_ONE = BN * 128;
    constexpr bool BDBL = BN <= 128 && NW == 8;               // two weight tiles in LDS
    constexpr int PB = BN / 8 / NW;                           // weight pieces per wave and step
    constexpr bool LEAD2 = MODE == 0;
    constexpr int WN = STRIDE == 2 ? 4 : BN / 64, WM = NW / WN, TM = BM / WM / 32, TN = BN / WN / 32;   // a wave owns 32 TM rows x 32 TN columns
    constexpr int A_OFF = 2 * STAGE, B_OFF = A_OFF + A_BYTES;
    constexpr int GW_OFF = B_OFF + B_ONE * (BDBL ? 2 : 1);    // GEN: depthwise weights [chunk][tap][32] fp32, two chunks at most
    constexpr int SMEM = GW_OFF + (GEN ? 2 * 9 * 128 : 0);
    constexpr bool SWZ = DUAL;                                // patch chunks XORed with (pixel >> 1) & 3: the projection's centre reads
    constexpr int E = 16 / EPI * TM * TN;                     // stores per wave and tile (exact when no lane is masked: full tiles)
    static_assert(!(DUAL && MODE != 1), "the two-output instances read the patch in stage 2");
    static_assert(!(DUAL && OSPLIT), "split32 output: one-output instances only");
    static_assert(PP >= 2 && PB >= 1 && SMEM <= (NW == 8 ? 160 : 80) * 1024, "shape");
    static_assert(NW == 8 || NW == 4, "4 or 8 waves");
    static_assert(!GEN || (MODE == 1 && !DUAL && !OSPLIT && STRIDE == 1 && !BDBL), "generated input: one fp32 output, stride 1, schedule 1, one weight tile");
    static_assert(STRIDE == 1 || (STRIDE == 2 && NW == 8 && !DUAL && BN >= 128 && TM == 1), "stride 2: 8 waves, one output, 128 or 256 columns");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd) {   // XCD k (workgroup id mod 8) takes the k-th contiguous eighth of the tile list: halo rows meet in one L2
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t = (id & 7) * (total >> 3) + (id >> 3);
        bx = t % gridDim.x;
        by = (t / gridDim.x) % gridDim.y;
        bz = t / (gridDim.x * gridDim.y);
    }
    const int xbase = bx * p.tpw * TW, y0 = by * TH;        // OUTPUT coordinates of the workgroup's first tile
    const int Wo = p.W / STRIDE;                             // output row pitch in pixels (p.H, p.W: the input's)
    const long img = (long)bz * p.H * p.W;                   // pixel index of this image's (0, 0) in the input ...
    const long img_o = (long)bz * (p.H / STRIDE) * Wo;       // ... and in the output / residual

    // ---- DMA sources.  Lane l of a piece fills 16-byte chunk (l & 7) of slot 8 * piece + (l >> 3).
    const int drow = lane >> 3, dk = lane & 7;
    const float* psrc[PP];
    unsigned pmove = 0;   // bit j: source j is a pixel of the image (moves with the tile), not padding / weights
    auto set_tile = [&](int xt) {   // branch-free on purpose (selects): it is inlined at three places
        pmove = 0;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            const int slot = q * 8 + drow;
            const int py = slot / PWS, px = slot - py * PWS;
            // patch origin: one pixel up / left of the tile for stride 1 and for the REFLECT-padded stride 2 (tf.pad(1) then VALID), at the
            // tile for TF SAME stride 2 on even sizes (which pads after only)
            const int org = (STRIDE == 1 || p.reflect) ? 1 : 0;
            int gy = STRIDE * y0 - org + py, gx = STRIDE * xt - org + px;
            if (p.reflect) {   // tf.pad(REFLECT, 1): -1 -> 1, H -> H - 2
                gy = gy < 0 ? -gy : (gy >= p.H ? 2 * p.H - 2 - gy : gy);
                gx = gx < 0 ? -gx : (gx >= p.W ? 2 * p.W - 2 - gx : gx);
            }
            const bool real = slot < NPATCH && px < PW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const bool wk = WPAD ? (slot < NPATCH && px == PW && py < 9) : (slot >= NPATCH && slot < NSLOT);
            const int tap = WPAD ? py : slot - NPATCH;
            const int kk = SWZ ? (dk ^ ((px >> 1) & 3)) : dk;
            const float* o = g_zero_pipe + dk * 4;                                  // padding pixels, unused slots
            const float* o_px = p.x + (img + (long)gy * p.W + gx) * p.ldx + (GEN ? 0 : kk * 4);   // GEN: the pixel's one value
            const float* o_wk = p.dw + (long)tap * p.Cin + dk * 4;                 // the chunk's depthwise weights, one tap per slot
            o = real ? o_px : o;
            o = (wk && !GEN) ? o_wk : o;                                            // (GEN: the weights are not part of the patch)
            psrc[j] = o;
            pmove |= real ? 1u << j : 0u;
        }
    };
    // dev ablations (knob sep_ablate; results are then wrong on purpose): 1 no stage 1, 2 no stage 2, 4 no epilogue arithmetic / stores,
    // 8 no patch DMA after the prologue, 16 no weight DMA after the prologue, 32 no residual loads
    const int abl = p.ablate;
    bool primed = false;
    auto issue_patch = [&](int stage, int coff) {   // coff: channel offset of the chunk (floats)
        if ((abl & 8) && primed) return;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[j] + coff), (lptr_t)(smem + stage * STAGE + q * 1024), 16, 0, 0);
        }
    };
    const uint16_t* bsrc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = (wv * PB + j) * 8 + drow;
        const int c = dk ^ ((row >> 1) & 7);                 // logical chunk: 0-3 hi, 4-7 lo
        const bool second = DUAL && row >= BN / 2;           // right half of the tile: the 1x1 projection's weights
        const int rr = second ? row - BN / 2 : row;
        const uint16_t* plane = (c & 4) ? (second ? p.W2lo : p.Wlo) : (second ? p.W2hi : p.Whi);
        bsrc[j] = plane + (long)rr * p.Cpad + (c & 3) * 8;
    }
    auto issue_B = [&](int buf, int coff) {
        if ((abl & 16) && primed) return;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[j] + coff), (lptr_t)(smem + B_OFF + buf * B_ONE + (wv * PB + j) * 1024), 16, 0, 0);
    };

    // ---- generated input: this lane's a / t of both chunks, d of the tile being generated (dcur, real-pixel mask greal) and of the one
    // after it (dnx; psrc / pmove describe THAT tile), and the patch chunk written where the DMA would have put it
    f32x4 gga[2], ggt[2];
    float dcur[PP], dnx[PP];
    unsigned greal = 0;
    float ghi = 0.f, gsl = 0.f, glo = 0.f;
    if constexpr (GEN) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = (c * 32 < p.Cin ? c * 32 : 0) + dk * 4;
            gga[c] = *reinterpret_cast<const f32x4*>(p.gen_a + ch);
            ggt[c] = *reinterpret_cast<const f32x4*>(p.gen_t + ch);
        }
        ghi = p.gen_act == 1 ? 6.f : __builtin_inff();
        gsl = p.gen_act == 4 ? 0.2f : 1.f;
        glo = (p.gen_act == 1 || p.gen_act == 2) ? 0.f : -__builtin_inff();
        for (int i = tid; i < (p.Cin / 32) * 72; i += NW * 64) {   // depthwise weights -> LDS [chunk][tap][32] (visible after the first barrier)
            const int ch = i / 72, rem = i - ch * 72;
            *reinterpret_cast<f32x4*>(smem + GW_OFF + i * 16) =
                *reinterpret_cast<const f32x4*>(p.dw + (long)(rem >> 3) * p.Cin + ch * 32 + (rem & 7) * 4);
        }
    }
    auto load_d = [&](float (&dst)[PP]) {
#pragma unroll
        for (int j = 0; j < PP; ++j) dst[j] = *psrc[j];
    };
    auto gen_patch = [&](int stage, int chunk) {
        const f32x4 ga = chunk ? gga[1] : gga[0], gt = chunk ? ggt[1] : ggt[0];
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int q = wv + NW * j;
            if (q >= NPIECE) continue;          // (wave-uniform: the surplus pieces of the DMA form have nothing to repeat here)
            const bool real = (greal >> j) & 1;
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float u = fmaf(dcur[j], ga[c], gt[c]);
                v[c] = real ? fminf(fmaxf(fmaxf(u, glo), gsl * u), ghi) : 0.f;
            }
            *reinterpret_cast<f32x4*>(smem + stage * STAGE + q * 1024 + lane * 16) = v;
        }
    };

    // ---- stage 1 role: 4 consecutive output pixels of one tile row, 4 channels.  A wave covers two 2 x 8 pixel blocks; inside a
    // block the four pixel groups of a half wave are (row, x group) = (0,0) (0,1) (1,1) (1,0)
    const int c4 = tid & 7, pgw = (lane >> 3);
    const int blk = 2 * wv + (pgw >> 2), pgl = pgw & 3;
    const int rb = pgl >> 1, xb = (pgl & 1) ^ rb;
    constexpr int XO = TW / 8;                                 // 8-pixel blocks per tile row
    // stride 2: a thread = ONE output pixel (tid >> 3 of the tile's 64) x 4 channels, patch pixel (2 oy + i, 2 ox + d)
    const int dy = STRIDE == 2 ? (tid >> 3) / TW : 2 * (blk / XO) + rb, dx = STRIDE == 2 ? (tid >> 3) % TW : 8 * (blk % XO) + 4 * xb;
    const int rd_base = (STRIDE * dy * PWS + STRIDE * dx) * 128;
    int rd_k[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) rd_k[m] = (SWZ ? (c4 ^ (((dx >> 1) + m) & 3)) : c4) * 16;
    int a_wr[4];   // byte offset of this thread's hi words of output pixel j in the A rows
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = dy * TW + dx + j;
        a_wr[j] = A_OFF + r * 128 + (((c4 >> 1) ^ ((r >> 1) & 7)) << 4) + (c4 & 1) * 8;
    }
    // ---- stage 2 role
    const int fr = lane & 31, fh = lane >> 5, sw = (fr >> 1) & 7;
    const int row0 = wm * TM * 32;                       // first of this wave's GEMM rows; row r = pixel (r / TW, r % TW) of the tile
    const int a_off = A_OFF + (row0 + fr) * 128;
    const int b_off = B_OFF + (wn * (TN * 32) + fr) * 128;
    const bool out2 = DUAL && wn >= WN / 2;

    // ---- epilogue constants: lane = output channel
    float es1[TN], et1[TN], es2[TN], et2[TN];
    const bool two = p.scale2 != nullptr && !out2;
    const int nlim = out2 ? p.N2 : p.N;
    const bool full = DUAL ? (p.N == BN / 2 && p.N2 == BN / 2) : p.N == BN;   // no lane is masked in the epilogue: store counts are exact
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * (TN * 32) + j * 32 + fr;
        const int nn = out2 ? n - BN / 2 : n;
        const bool valid = nn < nlim;
        es1[j] = valid ? (out2 ? p.scale_b : p.scale1)[nn] : 0.f;
        et1[j] = valid ? (out2 ? p.shift_b : p.shift1)[nn] : 0.f;
        es2[j] = (valid && two) ? p.scale2[nn] : 1.f;
        et2[j] = (valid && two) ? p.shift2[nn] : 0.f;
        // first use here: the wait for these loads stays in front of the loop (inside it, it would also wait for every older DMA)
        asm volatile("" ::"v"(es1[j]), "v"(et1[j]), "v"(es2[j]), "v"(et2[j]));
    }
    const int actc = out2 ? 1 : p.act;   // the projection of a two-output launch is conv + BN + relu6 (conv_block_not_sep)
    const float hi = actc == 1 ? 6.f : __builtin_inff();
    const float hi2 = actc == 2 ? __builtin_inff() : 6.f;
    const float slope = actc == 4 ? 0.2f : 1.f, lo = (actc == 1 || actc == 2) ? 0.f : -__builtin_inff();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = p.Cin / 32;
    const int total = p.tpw * nchunks;
    // issue stream: the step the next DMA group belongs to (clamped at the last step: the surplus groups re-read it into a stage
    // nobody computes on, so that every wave's vmcnt arithmetic stays uniform to the end)
    int istep = 0, ic = 0, ixt = xbase;
    set_tile(xbase);
    auto advance_issue = [&]() {
        if (istep + 1 >= total) return;
        ++istep;
        if (++ic == nchunks) {
            ic = 0;
            const int xn = ixt + TW;
            if constexpr (GEN) {   // the tile whose d values were requested a tile ago becomes current; request the one after it
#pragma unroll
                for (int j = 0; j < PP; ++j) dcur[j] = dnx[j];
                greal = pmove;
                ixt = xn;
                if (istep + nchunks < total) {
                    set_tile(xn + TW);
                    load_d(dnx);
                }
                return;
            }
            if (((STRIDE == 2 && !p.reflect) || ixt >= 1) && STRIDE * (xn + TW) + 1 <= p.W) {   // both tiles clear of the left / right image edges: every real
                const long step = (long)STRIDE * TW * p.ldx;                     // pixel moves one tile on
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

    auto stage1 = [&](const unsigned char* stg, int chunk) {   // depthwise 3x3 from the patch -> bf16 hi / lo A rows
        if (abl & 1) return;
        constexpr int WKT = GEN ? 128 : WKS * 128;                 // bytes from one tap's weights to the next
        if constexpr (STRIDE == 2) {
            const unsigned char* wkp = stg + WK0 * 128 + c4 * 16;
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    o += *reinterpret_cast<const f32x4*>(wkp + (i * 3 + d) * (WKS * 128)) *
                         *reinterpret_cast<const f32x4*>(stg + rd_base + (i * PWS + d) * 128 + c4 * 16);
            unsigned h0, l0, h1, l1;
            split2(o[0], o[1], h0, l0);
            split2(o[2], o[3], h1, l1);
            *reinterpret_cast<u32x2*>(smem + a_wr[0]) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(smem + (a_wr[0] ^ 64)) = u32x2{l0, l1};
            return;
        }
            const unsigned char* wkp = GEN ? smem + GW_OFF + chunk * (9 * 128) + c4 * 16 : stg + WK0 * 128 + c4 * 16;
            f32x4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 wk[3], pr[6];
#pragma unroll
                for (int d = 0; d < 3; ++d) wk[d] = *reinterpret_cast<const f32x4*>(wkp + (i * 3 + d) * WKT);
#pragma unroll
                for (int d = 0; d < 6; ++d)
                    pr[d] = *reinterpret_cast<const f32x4*>(stg + rd_base + (i * PWS + d) * 128 + rd_k[d >> 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int d = 0; d < 3; ++d) o[j] += wk[d] * pr[j + d];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned h0, l0, h1, l1;
                split2(o[j][0], o[j][1], h0, l0);
                split2(o[j][2], o[j][3], h1, l1);
                *reinterpret_cast<u32x2*>(smem + a_wr[j]) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(smem + (a_wr[j] ^ 64)) = u32x2{l0, l1};
            }
    };
    // Residual values of the tile, requested after stage 1 of the tile's last step (TM <= 2: 16 TM TN registers, free once stage 1's
    // are dead): they land under stage 2 instead of stalling the epilogue behind every older DMA piece (vmcnt is in order).
    constexpr bool RPRE = TM <= 2 && !DUAL;
    constexpr int R = RPRE ? 16 / EPI * TM * TN : 0;         // residual loads per wave and tile
    f32x4 rpre[RPRE ? TM : 1][TN][4];
    const bool res_on = p.res != nullptr && !out2 && !(abl & 32);
    auto res_prefetch = [&](int x0) {
        int ldr = p.ldres;
        asm volatile("" : "+s"(ldr));
        const int li = fr & 3, cq = fr >> 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n4 = wn * (TN * 32) + j * 32 + 4 * cq, n1 = wn * (TN * 32) + j * 32 + fr;
            const unsigned roff = EPI == 1 ? (unsigned)(4 * fh * ldr + n1) * 4u : (unsigned)((4 * fh + li) * ldr + n4) * 4u;
#pragma unroll
            for (int i = 0; i < (RPRE ? TM : 1); ++i) {
                const float* rbase = p.res + (img_o + (long)(y0 + (row0 + i * 32) / TW) * Wo + x0) * ldr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rpre[i][j][q] = f32x4{-0.f, -0.f, -0.f, -0.f};
                    const float* rb = rbase + ((8 * q / TW) * Wo + 8 * q % TW) * ldr;
                    if constexpr (EPI == 1) {      // element k = this lane's channel at pixel 8 q + 4 fh + k
                        if (n1 < nlim) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) rpre[i][j][q][k] = load_s1(rb + k * ldr, roff);
                        }
                    } else {
                        if (n4 < nlim) rpre[i][j][q] = load_s(rb, roff);
                    }
                }
            }
        }
    };
    auto stage2 = [&](const unsigned char* stg, int bbuf) {
        if (abl & 2) return;
        if constexpr (TM == 1 && !DUAL) {   // (TM = 2: the 64 fragment registers beside the 64 residual ones spill)
            // all fragment reads of the step first, then its MFMAs (otherwise every read sits in front of its first use and its LDS
            // latency is exposed 4-6 times per step)
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ((ks * 2 + fh) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[ks][i] = *reinterpret_cast<const bf16x8*>(smem + a_off + i * 4096 + ch);
                    al[ks][i] = *reinterpret_cast<const bf16x8*>(smem + a_off + i * 4096 + (ch ^ 64));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[ks][j] = *reinterpret_cast<const bf16x8*>(smem + b_off + bbuf * B_ONE + j * 4096 + ch);
                    bl[ks][j] = *reinterpret_cast<const bf16x8*>(smem + b_off + bbuf * B_ONE + j * 4096 + (ch ^ 64));
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);   // DS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);     // MFMAs
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 2 + fh) ^ sw) << 4;
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
            if (DUAL && out2) {
                // the projection's A operand = the block's INPUT at the tile's own pixels: the centre of the fp32 patch, split here
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = row0 + i * 32 + fr, cx = m % TW + 1;
                    const int slot = (m / TW + 1) * PWS + cx;
                    const int s = SWZ ? ((cx >> 1) & 3) : 0;
                    const int c0 = ks * 4 + fh * 2;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + slot * 128 + ((c0 ^ s) << 4));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + slot * 128 + (((c0 + 1) ^ s) << 4));
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
                    ah[i] = *reinterpret_cast<const bf16x8*>(smem + a_off + i * 4096 + ch);
                    al[i] = *reinterpret_cast<const bf16x8*>(smem + a_off + i * 4096 + (ch ^ 64));
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(smem + b_off + bbuf * B_ONE + j * 4096 + ch);
                bl[j] = *reinterpret_cast<const bf16x8*>(smem + b_off + bbuf * B_ONE + j * 4096 + (ch ^ 64));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    // epilogue from the accumulators.  C/D layout of the 32x32 MFMA: column = lane & 31 (the channel), row = (e & 3) + 8 (e >> 2) +
    // 4 (lane >> 5) of the M tile = 32 consecutive pixels of the tile in row-major order (one tile row at TW = 32, two at 16).
    auto epilogue = [&](int x0) {
        if (abl & 4) return;
        const bool has_res = res_on;
        const bool simple = actc == 1 && !two;   // conv + BN + relu6, nothing else: three VALU operations per value instead of eight
        float* __restrict__ outp = out2 ? p.y2 : p.y;
        int ldo = out2 ? p.ldy2 : p.ldy, ldr = p.ldres;
        asm volatile("" : "+s"(ldo), "+s"(ldr));   // opaque: the per-pixel bases below are recomputed per tile, not hoisted out of the
                                                  // step loop into (spilled) SGPRs
        if constexpr (EPI == 1) {
            // a lane keeps its channel (nb + fr) and stores the sixteen pixels of an accumulator one dword each: the 32 lanes of a half
            // wave write a pixel's 128 contiguous bytes.  split32: two pixels are split together, the (even, odd) channel pair trades
            // halves, the even lane stores hi (c, c + 1), the odd lane lo (c - 1, c).
            const bool odd = fr & 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float s1 = es1[j], t1 = et1[j], s2 = es2[j], t2 = et2[j];
                const int n = wn * (TN * 32) + j * 32 - (out2 ? BN / 2 : 0) + fr;
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
                            if (i == 0 && j == 0) {   // requested by res_prefetch; younger than them: the DMA groups issued after barrier B
                                if (!full) wait_vm<0>();
                                else if constexpr (LEAD2) wait_vm<(BDBL ? PB : 0) + PP>();
                                else wait_vm<0>();
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = rpre[i][j][q];
                        }
                    } else if (has_res) {
                        if (n < nlim) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int k = 0; k < 4; ++k) rv[q][k] = load_s1(rbase + ((8 * q / TW) * Wo + 8 * q % TW + k) * ldr, roff);
                        }
                        wait_vm<0>();
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
                const int nb = wn * (TN * 32) + j * 32 - (out2 ? BN / 2 : 0);   // first channel of this 32-column group in its output
                const int n4 = nb + 4 * cq;
                const bool valid = n4 < nlim;                             // Cout % 4 == 0: a lane's four channels are all in or all out
                const unsigned roff = (unsigned)((4 * fh + li) * ldr + n4) * 4u;
                unsigned voff;
                if constexpr (OSPLIT) {
                    // split32 output: the lanes of an (even, odd) pair of channel quads swap halves -- the even one stores both quads' hi
                    // words (16 bytes of the channel group's 128-byte line), the odd one both quads' lo words (64 bytes further on)
                    voff = (unsigned)((4 * fh + li) * ldo) * 4u + (n4 >> 5) * 128u + ((cq & 1) ? 64u : 0u) + ((n4 & 31) >> 3) * 16u;
                } else {
                    voff = (unsigned)((4 * fh + li) * ldo + n4) * 4u;
                }
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
                            if (i == 0 && j == 0) {   // requested by res_prefetch; younger than them: the DMA groups issued after barrier B
                                if (!full) wait_vm<0>();
                                else if constexpr (LEAD2) wait_vm<(BDBL ? PB : 0) + PP>();
                                else wait_vm<0>();
                            }
    #pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = rpre[i][j][q];
                        }
                    } else if (has_res) {
                        if (valid) {
    #pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = load_s(rbase + ((8 * q / TW) * Wo + 8 * q % TW) * ldr, roff);
                        }
                        wait_vm<0>();
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
                        float* ob = obase + ((8 * q / TW) * Wo + 8 * q % TW) * ldo;   // rows 8q.. of the M tile: pixel (8q / TW, 8q % TW)
                        if constexpr (!OSPLIT) {
                            if (valid) store_nt_s(ob, voff, v);
                        } else {
                            if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};   // the padding channels of a split32 tensor (up to a multiple of 32) are zero
                            unsigned h0, l0, h1, l1;
                            split2(v[0], v[1], h0, l0);
                            split2(v[2], v[3], h1, l1);
                            const bool oddq = cq & 1;
                            const unsigned r0 = xchg4(oddq ? h0 : l0, oddq), r1 = xchg4(oddq ? h1 : l1, oddq);
                            if (n4 < ((nlim + 31) & ~31))   // nothing lies beyond the padding (both lanes of a pair agree: the bound is a multiple of 8)
                                store_nt_s(ob, voff, oddq ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1});
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };

    if constexpr (LEAD2) {
        issue_patch(0, 0);
        if constexpr (BDBL) issue_B(0, 0);
        advance_issue();
        issue_patch(1, ic * 32);
    } else if constexpr (GEN) {
        load_d(dcur);
        greal = pmove;
        if (nchunks < total) {     // a second tile: its d values one tile ahead
            set_tile(xbase + TW);
            load_d(dnx);
        }
        gen_patch(0, 0);
    } else {
        if constexpr (BDBL) issue_B(0, 0);
        issue_patch(0, 0);
    }
    primed = true;

    int ct = 0, x0 = xbase;   // chunk of the step being computed, x origin of its tile
    bool epi1 = false;        // an epilogue ran at the end of the previous step
    for (int t = 0; t < total; ++t) {
        const int st = t & 1;
        const unsigned char* stg = smem + st * STAGE;
        const int bbuf = BDBL ? st : 0;
        // ---- A: the patch of step t has landed (this wave's pieces; the barrier adds everybody else's); stage 2 of step t-1 is over
        if constexpr (LEAD2) {
            if (!BDBL && t == 0) wait_vm<PP>();
            else if (epi1 && full) wait_vm<PB + PP + E>();
            else wait_vm<PB + PP>();
        } else {
            if (epi1 && full) wait_vm<E>();
            else wait_vm<0>();
            if constexpr (GEN) wait_lgkm0();    // this wave's part of the generated patch of step t is written
        }
        __builtin_amdgcn_s_barrier();
        PIPE_STAMP(0)
        if constexpr (!LEAD2) {
            if constexpr (BDBL) {
                const int cn = ct + 1 < nchunks ? ct + 1 : (t + 1 < total ? 0 : ct);
                issue_B(st ^ 1, cn * 32);
            } else {
                issue_B(0, ct * 32);
            }
            if constexpr (GEN) {
                // (a new tile: the d values of the tile after it are requested here, beside this step's weight pieces; barrier B waits
                // for everything -- vmcnt 0 -- and by then both have had stage 1 to land)
                const int before = istep;
                advance_issue();
                if (istep != before) gen_patch(st ^ 1, ic);
            } else {
                advance_issue();
                issue_patch(st ^ 1, ic * 32);
            }
        } else if constexpr (!BDBL) {
            issue_B(0, ct * 32);
        }
        stage1(stg, ct);
        const bool rp = RPRE && res_on && ct + 1 == nchunks;   // last step of a tile with a residual: request its values now
        if (rp) res_prefetch(x0);
        PIPE_STAMP(1)
        // ---- B: A rows and the weight tile of step t visible; everybody is done with the patch of step t (unless DUAL)
        wait_lgkm0();
        if constexpr (LEAD2) {
            if constexpr (!BDBL) wait_vm<0>();
            else if (rp && full) { if (epi1) wait_vm<PP + R + E>(); else wait_vm<PP + R>(); }
            else if (epi1 && full && !rp) wait_vm<PP + E>();
            else wait_vm<PP>();
        } else if constexpr (GEN) {
            wait_vm<0>();
        } else if constexpr (!BDBL) {
            if (rp && full) wait_vm<PP + R>();
            else wait_vm<PP>();
        }
        __builtin_amdgcn_s_barrier();
        PIPE_STAMP(2)
        if constexpr (LEAD2) {
            if constexpr (BDBL) {
                const int cn = ct + 1 < nchunks ? ct + 1 : (t + 1 < total ? 0 : ct);
                issue_B(st ^ 1, cn * 32);
            }
            advance_issue();
            issue_patch(st, ic * 32);
        }
        stage2(stg, bbuf);
        PIPE_STAMP(3)
        epi1 = false;
        if (++ct < nchunks) continue;
        ct = 0;
        epi1 = true;
        epilogue(x0);
        x0 += TW;
        PIPE_STAMP(4)
    }
    wait_vm<0>();   // the surplus DMA groups must have landed before this workgroup's LDS goes to the next one
    if (p.stamps && tid == 0) {
        long long* o = p.stamps + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#undef PIPE_STAMP
}

template <int BN, bool DUAL, bool OSPLIT, int EPI>
int launch_mode(const SepParams& q, dim3 grid, int mode, int nw, hipStream_t st) {
    if constexpr (DUAL) {
        if constexpr (BN == 128) {
            if (nw == 4) {
                hipLaunchKernelGGL((sep_pipe_kernel<BN, true, 1, false, 4, 1, EPI>), grid, dim3(256), 0, st, q);
                return emd::check_launch("sep_pipe_kernel<4 waves, two outputs>");
            }
        }
        hipLaunchKernelGGL((sep_pipe_kernel<BN, true, 1, false, 8, 1, EPI>), grid, dim3(512), 0, st, q);
    } else {
        if constexpr (BN <= 128) {
            if (nw == 4) {
                if (mode == 1) hipLaunchKernelGGL((sep_pipe_kernel<BN, false, 1, OSPLIT, 4, 1, EPI>), grid, dim3(256), 0, st, q);
                else hipLaunchKernelGGL((sep_pipe_kernel<BN, false, 0, OSPLIT, 4, 1, EPI>), grid, dim3(256), 0, st, q);
                return emd::check_launch("sep_pipe_kernel<4 waves>");
            }
        }
        if (mode == 1) {
            hipLaunchKernelGGL((sep_pipe_kernel<BN, false, 1, OSPLIT, 8, 1, EPI>), grid, dim3(512), 0, st, q);
        } else {
            hipLaunchKernelGGL((sep_pipe_kernel<BN, false, 0, OSPLIT, 8, 1, EPI>), grid, dim3(512), 0, st, q);
        }
    }
    return emd::check_launch("sep_pipe_kernel");
}

template <int BN, int EPI>
int launch_s2(const SepParams& q, dim3 grid, int mode, hipStream_t st) {
    if (mode == 1) hipLaunchKernelGGL((sep_pipe_kernel<BN, false, 1, false, 8, 2, EPI>), grid, dim3(512), 0, st, q);
    else hipLaunchKernelGGL((sep_pipe_kernel<BN, false, 0, false, 8, 2, EPI>), grid, dim3(512), 0, st, q);
    return emd::check_launch("sep_pipe_kernel<stride 2>");
}

template <int BN, bool DUAL, int EPI>
int launch_bn(const SepParams& q, dim3 grid, int mode, int nw, hipStream_t st) {
    if constexpr (!DUAL && BN >= 128) {
        if (q.out_split) return launch_mode<BN, false, true, EPI>(q, grid, mode, nw, st);
    }
    return launch_mode<BN, DUAL, false, EPI>(q, grid, mode, nw, st);
}

template <int EPI>
int launch_epi(const SepParams& p, const SepParams& q, dim3 grid, int mode, int nw, hipStream_t st) {
    if (p.stride == 2) return p.N <= 128 ? launch_s2<128, EPI>(q, grid, mode, st) : launch_s2<256, EPI>(q, grid, mode, st);
    if (p.N2 > 0) {
        const bool wide = p.N > 64 || p.N2 > 64;
        if (wide) return launch_bn<256, true, EPI>(q, grid, 1, 8, st);
        return launch_bn<128, true, EPI>(q, grid, 1, nw, st);
    }
    if (p.N <= 64) return launch_bn<64, false, EPI>(q, grid, mode, nw, st);
    if (p.N <= 128) return launch_bn<128, false, EPI>(q, grid, mode, nw, st);
    return launch_bn<256, false, EPI>(q, grid, mode, nw, st);
}

}  // namespace

namespace emd {

// 4-wave form (8 x 16 tiles, 80 KiB of LDS or less: two workgroups per CU, whose phases -- DMA wait, depthwise stage, MFMAs, epilogue --
// interleave): instances up to 128 output columns (one output, fp32 or split32) and 64 | 64 (two outputs).  Rule: wherever there is an
// instance.  Measured on graph D's shapes (tools/sep_epi_bench.py with SEB_KNOB=sep_nw, [32, ., ., .], two boxes): the two-output launch
// 512^2 x 128 -> 64 | 64 2078 -> 1892 us, 2084 -> 1921; 64 -> 64 961 -> 859, 943 -> 873; 128 -> 64 1382 -> 1354; 64 -> 64 with a
// residual 1161 / 1288 -> 1212 / 1210; 256^2 x 128 -> 128 480 -> 467, 486 -> 469; with residual and split32 output 669 -> 647 / 653;
// 384 -> 128 1155 -> 1144; the whole D step 23.36 -> 23.24 ms in one process (tools/d_knob_ab.py).  Dev knob sep_nw = 8 / 4 forces.
static bool use_nw4(const SepParams& p) {
    if (g_knobs.sep_nw == 8 || p.stride == 2) return false;
    return p.N2 > 0 ? (p.N <= 64 && p.N2 <= 64) : (p.N <= 128 && !(p.out_split && p.N <= 64));    // instances: launch_mode
}

bool sep_pipe_covers(const SepParams& p, int precision) {
    if (!g_knobs.sep_pipe || precision != 3) return false;
    if (p.gen_a)   // generated input (round 4, opt-in: dev knob sep_gen_pipe): the 4-wave 64-column instance only -- cnn0_last of graphs D / X and its likes
        return g_knobs.sep_gen_pipe && p.stride == 1 && p.H % 8 == 0 && p.W % 16 == 0 && (p.Cin == 32 || p.Cin == 64) && p.N <= 64 && p.N2 == 0 &&
               !p.out_split && !p.res && g_knobs.sep_nw != 8;
    if (p.stride == 2)   // output tiles of 4 x 16 pixels: H % 8 == 0, W % 32 == 0 (input sizes); one fp32 output of up to 256 channels
        return p.H % 8 == 0 && p.W % 32 == 0 && p.Cin % 32 == 0 && p.Cin >= 32 && p.Cin <= 4064 && p.N2 == 0 && !p.out_split && p.N <= 256;
    if (p.H % 8 != 0 || p.W % (use_nw4(p) ? 16 : 32) != 0 || p.Cin % 32 != 0 || p.Cin < 32 || p.Cin > 4064) return false;
    if (p.N2 > 0) return p.N <= 128 && p.N2 <= 128 && !p.out_split && !p.res && !p.scale2;
    if (p.out_split && p.N <= 64) return false;
    return p.N <= 256;
}

// Which launches go to the software-pipelined kernel (sep_pipe2.hip; same bits).  Dev knob sep_pipe2: 0 (default) none, 1 the two-output
// launches with more than 64 columns per output (deconv1_a + residual1_d, 384 -> 128 | 128: the one shape whose slots are matrix-core
// bound), 2 everything it has an instance for (the parity tests).  Measured (profiles/r04_experiments.txt 3): standalone at parity on
// that shape (2034 / 2070 us against 2059 / 2004 on two boxes), 5-50 % SLOWER on every one-output shape -- where this file's 4-wave
// two-workgroups-per-CU form wins: both kernels are bound by instruction issue and by phases no other wave fills, and the pipelined one
// issues ~1.9 x the instructions per chunk -- and graph D 22.10 ms with 0, 22.26 with 1, 23.07 with 2 in one process.  A forced 4-wave
// form (dev knob sep_nw = 4) always means this file's kernel.
static bool use_pipe2(const SepParams& p) {
    if (p.gen_a) return false;
    if (!g_knobs.sep_pipe2 || g_knobs.sep_ablate || g_knobs.sep_nw == 4 || !sep_pipe2_covers(p)) return false;
    if (g_knobs.sep_pipe2 == 2) return true;
    return p.N2 > 0 && (p.N > 64 || p.N2 > 64);
}

int sep_pipe_launch(const SepParams& p, int B, hipStream_t st) {
    if (use_pipe2(p)) return sep_pipe2_launch(p, B, st);
    SepParams q = p;
    const bool s2 = p.stride == 2;
    const int nw = (p.gen_a || (!s2 && use_nw4(p))) ? 4 : 8, tw = s2 ? 16 : 4 * nw, th = s2 ? 4 : 8;
    const int Ho = p.H / (s2 ? 2 : 1), Wo = p.W / (s2 ? 2 : 1);
    const int tiles_w = Wo / tw;
    const long wgs1 = (long)tiles_w * (Ho / th) * B;
    int tpw = 1;   // several tiles per workgroup (the DMA ring runs on across them) where >= 4 workgroups per CU remain
    for (int t = 8; t >= 2; t >>= 1)
        if (tiles_w % t == 0 && wgs1 / t >= 1024 * (8 / nw)) { tpw = t; break; }
    if (g_knobs.sep_tpw > 0 && tiles_w % g_knobs.sep_tpw == 0) tpw = g_knobs.sep_tpw;
    q.tpw = tpw;
    q.stamps = g_knobs.sep_stamps;
    q.ablate = g_knobs.sep_ablate;
    const dim3 grid(tiles_w / tpw, Ho / th, B);
    q.xcd = g_knobs.sep_xcd && ((long)grid.x * grid.y * grid.z) % 8 == 0;
    // schedule (see the kernel): rule = the patch two steps ahead; one step ahead for two outputs, with a residual (its loads then
    // queue behind one DMA group instead of two) and in the 4-wave form; the dev knob sep_mode (0 / 1) overrides
    const int mode = g_knobs.sep_mode >= 0 ? g_knobs.sep_mode : ((nw == 4 || p.res) ? 1 : 0);
    // epilogue (see the kernel): per-channel dword stores, 2.7 % over graph D's twelve shapes (tools/sep_epi_bench.py; the two-output
    // launches 5 %) -- except with a residual on more than 128 columns (cnn2_last: 445 against 430 us for the transposed 16-byte form)
    if (p.gen_a) {
        if (g_knobs.epi_width == 4) hipLaunchKernelGGL((sep_pipe_kernel<64, false, 1, false, 4, 1, 4, true>), grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((sep_pipe_kernel<64, false, 1, false, 4, 1, 1, true>), grid, dim3(256), 0, st, q);
        return emd::check_launch("sep_pipe_kernel<generated input>");
    }
    const int epi = g_knobs.epi_width ? g_knobs.epi_width : ((p.res && p.N > 128) ? 4 : 1);
    return epi == 4 ? launch_epi<4>(p, q, grid, mode, nw, st) : launch_epi<1>(p, q, grid, mode, nw, st);
}

}  // namespace emd
