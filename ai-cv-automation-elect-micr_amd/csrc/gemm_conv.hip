// Implicit-GEMM convolution on the matrix cores: pointwise 1x1, strided 1x1 and the four output
// phases of the 3x3 stride-2 transposed convolution, all with a fused per-channel epilogue.
//
// replaces (TensorFlow ops called by machine_learning/denoiser.py):
//   the pointwise half of slim.separable_convolution2d (:113-131) + its normalizer BN (:123) +
//     batch_then_activ (:134)                                  -> emd_conv1x1_f32
//   slim.conv2d(kernel_size=1[, stride=2]) + bias + BN + relu6 (:91-97, :159-164, :208-214,
//     :220-227)                                                -> emd_conv1x1_f32
//   slim.conv2d_transpose(k=3, stride=2, 'same') + bias + BN + relu6 (:141-148)
//                                                              -> emd_deconv3x3s2_f32
//   the residual "+=" that follows them (:264, :279, :294, :309, :322, :246, :360, :372, :384)
//     and tf.concat (:203, :353, :365: outputs are written straight into channel slices).
//
// C[M,N] = epilogue( A[M,K] * W[K,N] ):  M = B*Hg*Wg output positions, K = taps*Cin, N = Cout.
//   A is never materialised: row m = (b,i,j) reads, for tap t, source pixel (b, i*sa+dy_t, j*sa+dx_t)
//   of the NHWC fp32 activation tensor (zero outside the image), channels contiguous.
//   W is pre-packed once on the host (emd_pack_weights_bf16) as two bf16 planes (hi, lo) in
//   [Npad][taps*Cpad] order, i.e. K-contiguous per output channel, zero padded.
//
// Numerics ("split-bf16", PASSES=3): fp32 activations are split in-kernel into bf16 hi + bf16 lo
// (a = hi + lo + O(2^-17 a)) and  acc += Ahi*Whi + Ahi*Wlo + Alo*Whi  in fp32 on
// v_mfma_f32_32x32x16_bf16: ~2^-16 relative per product, which is what keeps a ~60-layer network
// inside the 1e-3 relative-L2 parity bar (plain bf16 inputs, PASSES=1, is ~2^-9 per layer and is
// offered as the fast mode).
//
// Block = 256 threads = 4 waves; block tile 128 x BN (BN 128 or 64), K step 64; each wave owns a
// 64x64 (or 32x64) sub-tile as 32x32 MFMA tiles.  A: global fp32 -> registers (issued one K step
// ahead) -> split -> LDS bf16 planes; W: global bf16 -> registers -> LDS.  LDS rows are 144 B
// (128 + 16 pad): the ds_read_b128 fragment reads and the ds_write_b64/b128 staging writes (one full
// row per lane group) are bank-conflict free.  The epilogue stages the fp32 tile through LDS so that
// stores and residual loads are 16 B per lane, a whole pixel's channel run per 32 lanes.
// Block index -> tile mapping is XCD-aware: the N-tiles of one M-tile (which re-read the same
// activation rows) get consecutive indices inside one XCD's share of the grid.
#include "mfma_common.hpp"
#include "bn_chain_dev.hpp"

namespace {

using namespace emd;

__device__ __attribute__((aligned(16))) float g_zero16[4];   // what padding rows and channel tails load

struct GemmParams {
    const float* A;       // source activations (NHWC), pixel stride lda
    const uint16_t* Whi;  // [Npad][taps*Cpad]
    const uint16_t* Wlo;
    float* C;             // destination, pixel stride ldc
    const float* res;     // optional residual (same pixel indexing as C), pixel stride ldres
    const float* scale1;  // [N] epilogue: y = acc*scale1 + shift1
    const float* shift1;
    const float* scale2;  // optional second affine + relu6 (ASPP extra BN)
    const float* shift2;
    long M;               // B*Hg*Wg
    int N, Cin, Cpad, ntaps;
    int lda, ldc, ldres;
    int act;              // EMD_ACT_*: 0 none, 1 relu6, 2 relu -- after the first affine (and after the second)
    // row map: m -> (b,i,j) on Hg x Wg;  source (b, i*sa+dy, j*sa+dx) in Ha x Wa;  dest (b, i*sc+py, j*sc+px) in Hc x Wc
    int flat;             // 1: source pixel = dest pixel = m (plain pointwise)
    int Hg, Wg, Ha, Wa, Hc, Wc, sa, sc, py, px;
    unsigned long long dyp, dxp;  // per-tap source offsets, 7 bits each, biased by 64 (no dynamic kernarg indexing)
    int n_mtiles, n_ntiles;
    double* stats_part;   // optional [n_mtiles][2][N]: per-channel sum / sum of squares of the values each M tile STORES (training: the
                          // batch norm behind the conv takes batch statistics -- no second pass over y)
    int stats_tpi, stats_nph, stats_ph;   // the transposed conv's four phase launches share one table: M tile mt of phase ph is slab
                                          // ((mt / tpi) * nph + ph) * tpi + mt % tpi  (tpi = tiles per image, or all tiles; nph = 0: slab = mt)
};

// Block tile 128 x BN (BN = 128: 2x2 waves of 64x64; BN = 64: 4x1 waves of 32x64), K step 64.
template <int BN, int PASSES, bool FLAT>
__global__ __launch_bounds__(256, 2) void gemm_conv_kernel(const GemmParams p) {
    constexpr int BM = 128, BK = 64;
    constexpr int LDK = BK + 8;                      // bf16 elements per LDS row: 144 B, conflict-free b128 reads
    constexpr int WM = BN == 128 ? 2 : 4, WN = 4 / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NPL = PASSES == 3 ? 2 : 1;         // bf16 planes kept in LDS
    constexpr int A_PASSES = BM / 16;                // 16 float4 per 64-float row -> 16 rows per pass
    constexpr int W_PASSES = BN / 32;                // 8 x 16-B chunks per 64-bf16 row -> 32 rows per pass
    constexpr int LDS_STAGE = BN + 4;                // fp32 epilogue staging row (floats)
    constexpr int A_BYTES = NPL * BM * LDK * 2, B_BYTES = NPL * BN * LDK * 2;
    constexpr int STAGE_BYTES = BM * LDS_STAGE * 4;
    constexpr int TILE_BYTES = A_BYTES + B_BYTES > STAGE_BYTES ? A_BYTES + B_BYTES : STAGE_BYTES;

    // ONE LDS object, carved by hand (tiles | row maps); the epilogue staging overlays the tiles
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES + 2 * BM * 8];
    auto As = reinterpret_cast<uint16_t(*)[BM][LDK]>(smem);
    auto Bs = reinterpret_cast<uint16_t(*)[BN][LDK]>(smem + A_BYTES);
    float(*stage)[LDS_STAGE] = reinterpret_cast<float(*)[LDS_STAGE]>(smem);
    long long* rowA = reinterpret_cast<long long*>(smem + TILE_BYTES);  // element offset of the source pixel (-1: zero row)
    long long* rowP = rowA + BM;                                        // destination pixel index (-1: beyond M)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;

    // XCD-aware tile mapping (bijective for any grid size): the N-tiles of one M-tile are neighbours in one XCD
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    auto map_rows = [&](int tap, bool with_dest) {
        if (tid < BM) {
            const long m = m0 + tid;
            long long src = -1, dst = -1;
            if (m < p.M) {
                if (FLAT) {
                    src = dst = m;
                } else {
                    const int j = (int)(m % p.Wg);
                    const long t = m / p.Wg;
                    const int i = (int)(t % p.Hg);
                    const long b = t / p.Hg;
                    const int dy = (int)((p.dyp >> (7 * tap)) & 127) - 64;
                    const int dx = (int)((p.dxp >> (7 * tap)) & 127) - 64;
                    const int iy = i * p.sa + dy, ix = j * p.sa + dx;
                    if (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) src = (b * p.Ha + iy) * (long)p.Wa + ix;
                    dst = (b * p.Hc + (i * p.sc + p.py)) * (long)p.Wc + (j * p.sc + p.px);
                }
            }
            rowA[tid] = src < 0 ? -1 : src * p.lda;
            if (with_dest) rowP[tid] = dst;
        }
    };
    const bool rows_all_valid = m0 + BM <= p.M;   // block-uniform

    // global -> register staging, one K step ahead.  The loop starts at it = -1 (stage tile 0 only) so that
    // the load code and the LDS-store code each exist once, straight-line and fully unrolled.
    f32x4 areg[A_PASSES];
    u32x4 whreg[W_PASSES], wlreg[W_PASSES];
    const int a_col = (tid & 15) * 4;     // first of this thread's 4 consecutive channels in the K step
    const int a_row = tid >> 4;           // + 16 per pass
    const int w_col = (tid & 7) * 8;
    const int w_row = tid >> 3;           // + 32 per pass
    const int Ktot = p.ntaps * p.Cpad;
    const uint16_t* __restrict__ whi = p.Whi + (long)(n0 + w_row) * Ktot + w_col;
    const uint16_t* __restrict__ wlo = NPL == 2 ? p.Wlo + (long)(n0 + w_row) * Ktot + w_col : nullptr;
    const long w_pass_stride = 32L * Ktot;
    const float* __restrict__ Ab = p.A + a_col;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ksteps = p.Cpad / BK;
    const int total = p.ntaps * ksteps;
    map_rows(0, true);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    // this thread's 8 source-row offsets live in registers for a whole tap: re-reading them from LDS before
    // every load put 8 serial LDS round trips + 8 divergent branches in front of each K step's loads
    long long aoff[A_PASSES];
#pragma unroll
    for (int q = 0; q < A_PASSES; ++q) aoff[q] = rowA[a_row + q * 16];

    for (int it = -1; it < total; ++it) {
        if (it >= 0) {
            // registers (tile `it`) -> LDS, splitting the activations into bf16 hi/lo on the way
#pragma unroll
            for (int q = 0; q < A_PASSES; ++q) {
                const int r = a_row + q * 16;
                unsigned h0, l0, h1, l1;
                split2(areg[q][0], areg[q][1], h0, l0);
                split2(areg[q][2], areg[q][3], h1, l1);
                *reinterpret_cast<u32x2*>(&As[0][r][a_col]) = u32x2{h0, h1};
                if (NPL == 2) *reinterpret_cast<u32x2*>(&As[NPL - 1][r][a_col]) = u32x2{l0, l1};
            }
#pragma unroll
            for (int q = 0; q < W_PASSES; ++q) {
                const int r = w_row + q * 32;
                *reinterpret_cast<u32x4*>(&Bs[0][r][w_col]) = whreg[q];
                if (NPL == 2) *reinterpret_cast<u32x4*>(&Bs[NPL - 1][r][w_col]) = wlreg[q];
            }
            __syncthreads();  // tile `it` visible
        }
        {
            // stage tile it+1 (the last iteration re-loads its own tile: keeps this path branch-free)
            const int nx = it + 1 < total ? it + 1 : it;
            const int tap = nx / ksteps, c0 = (nx - tap * ksteps) * BK;
            if (c0 == 0 && !FLAT && nx != it && nx > 0) {  // tap change: new source rows (block-uniform)
                map_rows(tap, false);
                __syncthreads();
#pragma unroll
                for (int q = 0; q < A_PASSES; ++q) aoff[q] = rowA[a_row + q * 16];
            }
            const bool kok = c0 + a_col < p.Cin;  // Cin % 4 == 0: a float4 is all inside or all outside
            {   // (round 4: the plain 1x1 forms had a fast path -- all rows real, plain loads -- beside a masked one that selected on the LOADED
                // value; the compiler joined the two with copies of the loaded registers, i.e. uses: s_waitcnt vmcnt(0) right behind the
                // loads in the steady state, the next K step's latency exposed in front of this step's MFMAs.  One path, select on the address.)
                // forms with taps / strides: a row without a source pixel for this tap (padding) or a channel group beyond Cin
                // reads 16 zero bytes.  The select is on the ADDRESS -- a select on the loaded value makes the wave wait for the
                // load right here instead of a whole MFMA phase later (measured: the 3x3 / transposed-conv phases of graph D
                // 6 - 12 % faster).
#pragma unroll
                for (int q = 0; q < A_PASSES; ++q) {
                    const float* src = (aoff[q] >= 0 && kok) ? Ab + aoff[q] + c0 : g_zero16;
                    areg[q] = *reinterpret_cast<const f32x4*>(src);
                }
            }
            const long kk = (long)tap * p.Cpad + c0;
#pragma unroll
            for (int q = 0; q < W_PASSES; ++q) {
                whreg[q] = *reinterpret_cast<const u32x4*>(whi + kk + q * w_pass_stride);
                if (NPL == 2) wlreg[q] = *reinterpret_cast<const u32x4*>(wlo + kk + q * w_pass_stride);
            }
        }
        if (it < 0) continue;
        // 16-wide K sub-steps that hold real channels in this tile (the last tile of Cin = 728 has 24 of 64:
        // the all-padding sub-steps are skipped, block-uniformly)
        const int kvalid = p.Cin - (it % ksteps) * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks * 16 >= kvalid) break;
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (BM / WM) + i * 32 + fr;
                ah[i] = *reinterpret_cast<const bf16x8*>(&As[0][r][ks * 16 + fh * 8]);
                if (NPL == 2) al[i] = *reinterpret_cast<const bf16x8*>(&As[NPL - 1][r][ks * 16 + fh * 8]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / WN) + j * 32 + fr;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bs[0][r][ks * 16 + fh * 8]);
                if (NPL == 2) bl[j] = *reinterpret_cast<const bf16x8*>(&Bs[NPL - 1][r][ks * 16 + fh * 8]);
            }

#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (PASSES == 3) {  // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }

        }
        __syncthreads();  // all fragment reads of tile `it` done before it is overwritten
    }

    // ---- epilogue.  The accumulators (C/D layout of mfma_32x32: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5))
    // go through an fp32 staging tile in LDS so that global traffic is 16 B per lane along the channel axis:
    // one 512-B (BN=128) run per output pixel for the stores and for the residual loads.
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
    constexpr int C4 = BN / 4;            // float4 chunks per staged row
    constexpr int ROWS_PER_PASS = 256 / C4;
    constexpr int NROWS = BM / ROWS_PER_PASS;
    const int ec = (tid % C4) * 4, er = tid / C4;
    const int n = n0 + ec;
    double ssum[4] = {0.0, 0.0, 0.0, 0.0}, ssq[4] = {0.0, 0.0, 0.0, 0.0};
    static_assert(ROWS_PER_PASS * BN * 2 * 8 <= TILE_BYTES && 2 * BN <= 256, "the statistics reduction reuses the tile memory");
    if (n < p.N) {                        // N % 4 == 0: a chunk is all inside or all outside
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.scale1 + n);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.shift1 + n);
        f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        if (p.scale2) {
            s2 = *reinterpret_cast<const f32x4*>(p.scale2 + n);
            t2 = *reinterpret_cast<const f32x4*>(p.shift2 + n);
        }
        float* __restrict__ outp = p.C;
        // one clamp form for every activation code: v = min(max(max(v, lo), slope*v), hi) -- (lo, slope, hi) = none: (-inf, 1, inf); relu6: (0, 1, 6);
        // relu: (0, 1, inf); leaky relu (graph G): (-inf, 0.2, inf); a clamped negative comes out as +0, as tf.nn.relu6 gives it
        const float hi = p.act == 1 ? 6.f : __builtin_inff();
        const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;   // second stage (extra BN): relu6, or relu with act code relu
        const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();
        const bool two = p.scale2 != nullptr;
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
        if (p.res) {
            // all residual values of this thread's rows are requested before the first one is used
            f32x4 rv[NROWS];
#pragma unroll
            for (int k = 0; k < NROWS; ++k) {
                const long long pix = rowP[er + k * ROWS_PER_PASS];
                rv[k] = pix >= 0 ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < NROWS; ++k) {
                const int r = er + k * ROWS_PER_PASS;
                const long long pix = rowP[r];
                if (pix >= 0) *reinterpret_cast<f32x4*>(outp + pix * p.ldc + n) = finish(*reinterpret_cast<const f32x4*>(&stage[r][ec])) + rv[k];
            }
        } else if (p.stats_part) {
            // batch statistics of the output gathered while the tile is in hand (as gemm_split.hip's: per-thread double sums over its
            // rows here, row groups -> 1 below, one partial per (M tile, channel) for the fixed-order final reduction bn_stats_final)
#pragma unroll 4
            for (int r = er; r < BM; r += ROWS_PER_PASS) {
                const long long pix = rowP[r];
                if (pix < 0) continue;
                const f32x4 v = finish(*reinterpret_cast<const f32x4*>(&stage[r][ec]));
                *reinterpret_cast<f32x4*>(outp + pix * p.ldc + n) = v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double d = (double)v[c];
                    ssum[c] += d;
                    ssq[c] += d * d;
                }
            }
        } else {
#pragma unroll 4
            for (int r = er; r < BM; r += ROWS_PER_PASS) {
                const long long pix = rowP[r];
                if (pix < 0) continue;
                *reinterpret_cast<f32x4*>(outp + pix * p.ldc + n) = finish(*reinterpret_cast<const f32x4*>(&stage[r][ec]));
            }
        }
    }
    if (p.stats_part) {   // block-uniform
        __syncthreads();  // the staging tile has been read out
        double(*red)[BN][2] = reinterpret_cast<double(*)[BN][2]>(smem);   // [row groups][BN channels][sum, sum of squares]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[er][ec + c][0] = ssum[c];
            red[er][ec + c][1] = ssq[c];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            if (n0 + col < p.N) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < ROWS_PER_PASS; ++k) t += red[k][col][which];
                const long slab = p.stats_nph ? ((long)(mt / p.stats_tpi) * p.stats_nph + p.stats_ph) * p.stats_tpi + mt % p.stats_tpi : mt;
                p.stats_part[(slab * 2 + which) * p.N + n0 + col] = t;
            }
        }
    }
}

template <int BN>
int launch(const GemmParams& p0, int passes, hipStream_t st) {
    GemmParams p = p0;
    p.n_mtiles = (int)((p.M + 127) / 128);
    p.n_ntiles = (p.N + BN - 1) / BN;
    const long nblk = (long)p.n_mtiles * p.n_ntiles;
    if (nblk <= 0 || nblk > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "gemm_conv: grid too large");
    if (passes == 3)
        if (p.flat) hipLaunchKernelGGL((gemm_conv_kernel<BN, 3, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_conv_kernel<BN, 3, false>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else
        if (p.flat) hipLaunchKernelGGL((gemm_conv_kernel<BN, 1, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_conv_kernel<BN, 1, false>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    return emd::check_launch("gemm_conv_kernel");
}

int dispatch(const GemmParams& p, int passes, hipStream_t st) {
    if (p.N <= 64) return launch<64>(p, passes, st);
    // a grid that cannot even give every CU one 128x128 tile (a single 32x32 training tower: 8 x 6 tiles) runs as
    // 128x64 tiles: twice the workgroups, each latency-bound pass over K shorter
    const long tiles128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (tiles128 < 192) return launch<64>(p, passes, st);
    return launch<128>(p, passes, st);
}

inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

void set_taps(GemmParams& p, int n, const int* dy, const int* dx) {
    p.ntaps = n;
    p.dyp = p.dxp = 0;
    for (int t = 0; t < n; ++t) {
        p.dyp |= (unsigned long long)((dy ? dy[t] : 0) + 64) << (7 * t);
        p.dxp |= (unsigned long long)((dx ? dx[t] : 0) + 64) << (7 * t);
    }
}

int common_checks(const char* who, const float* x, const void* whi, const void* wlo, const float* scale1,
                  const float* shift1, const float* scale2, const float* shift2, const float* res, float* y,
                  int Cin, int N, int ldx, int ldy, int ldres, int passes) {
    (void)who;
    EMD_REQUIRE(x && whi && scale1 && shift1 && y, EMD_E_INVALID, "conv: null pointer");
    EMD_REQUIRE(passes == 1 || passes == 3, EMD_E_INVALID, "conv: precision must be EMD_PREC_BF16 or EMD_PREC_BF16X3");
    EMD_REQUIRE(passes == 1 || wlo, EMD_E_INVALID, "conv: the split-bf16 mode needs the lo weight plane");
    EMD_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), EMD_E_INVALID, "conv: scale2/shift2 must come together");
    EMD_REQUIRE(Cin >= 4 && N >= 1, EMD_E_INVALID, "conv: bad channel counts");
    EMD_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0 && ldx >= Cin, EMD_E_ALIGN, "conv: Cin and ldx must be multiples of 4, ldx >= Cin");
    EMD_REQUIRE(ldy >= N && (!res || ldres >= N), EMD_E_INVALID, "conv: ldy/ldres smaller than Cout");
    EMD_REQUIRE(N % 4 == 0 && ldy % 4 == 0 && emd::aligned16(y) && (!res || (ldres % 4 == 0 && emd::aligned16(res))),
                EMD_E_ALIGN, "conv: Cout, ldy, ldres must be multiples of 4 and y, res 16-byte aligned");
    EMD_REQUIRE(emd::aligned16(scale1) && emd::aligned16(shift1) && (!scale2 || (emd::aligned16(scale2) && emd::aligned16(shift2))),
                EMD_E_ALIGN, "conv: scale/shift vectors must be 16-byte aligned");
    EMD_REQUIRE(emd::aligned16(x) && emd::aligned16(whi) && (!wlo || emd::aligned16(wlo)), EMD_E_ALIGN,
                "conv: x and the weight planes must be 16-byte aligned");
    return EMD_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- host packing
extern "C" size_t emd_packed_weight_elems(int taps, int Cin, int Cout) {
    if (taps < 1 || Cin < 1 || Cout < 1) return 0;
    const size_t cpad = (size_t)(Cin + kBK - 1) / kBK * kBK;
    const size_t npad = (size_t)(Cout + kNPadTo - 1) / kNPadTo * kNPadTo;
    return npad * taps * cpad;
}

extern "C" int emd_pack_weights_bf16(const float* w_host, int taps, int Cin, int Cout, int cout_major,
                                     uint16_t* hi_host, uint16_t* lo_host) {
    EMD_REQUIRE(w_host && hi_host && lo_host, EMD_E_INVALID, "emd_pack_weights_bf16: null pointer");
    EMD_REQUIRE(taps >= 1 && taps <= 9 && Cin >= 1 && Cout >= 1, EMD_E_INVALID, "emd_pack_weights_bf16: bad shape");
    const size_t cpad = (size_t)(Cin + kBK - 1) / kBK * kBK;
    const size_t npad = (size_t)(Cout + kNPadTo - 1) / kNPadTo * kNPadTo;
    const size_t ktot = (size_t)taps * cpad;
    for (size_t i = 0; i < npad * ktot; ++i) hi_host[i] = lo_host[i] = 0;
    for (int t = 0; t < taps; ++t)
        for (int c = 0; c < Cin; ++c)
            for (int n = 0; n < Cout; ++n) {
                // TF layouts: conv [taps][Cin][Cout]; conv2d_transpose [taps][Cout][Cin]
                const float w = cout_major ? w_host[((size_t)t * Cout + n) * Cin + c]
                                           : w_host[((size_t)t * Cin + c) * Cout + n];
                const uint16_t h = f32_to_bf16_rne(w);
                const uint16_t l = f32_to_bf16_rne(w - bf16_to_f32(h));
                const size_t o = (size_t)n * ktot + (size_t)t * cpad + c;
                hi_host[o] = h;
                lo_host[o] = l;
            }
    return EMD_OK;
}

// ---------------------------------------------------------------------------------------------- 1x1 convolution
extern "C" int emd_conv1x1_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo,
                               const float* scale1, const float* shift1, const float* scale2,
                               const float* shift2, const float* res, int ldres, float* y, int ldy, int B,
                               int H, int W, int Cin, int Cout, int stride, int act, int precision,
                               emd_stream_t stream) {
    int rc = common_checks("emd_conv1x1_f32", x, whi, wlo, scale1, shift1, scale2, shift2, res, y, Cin, Cout, ldx,
                           ldy, ldres, precision);
    if (rc != EMD_OK) return rc;
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv1x1_f32: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_conv1x1_f32: stride must be 1 or 2");
    if (B == 0) return EMD_OK;
    GemmParams p{};
    p.A = x; p.Whi = whi; p.Wlo = wlo; p.C = y; p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.N = Cout; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK; p.ntaps = 1;
    p.lda = ldx; p.ldc = ldy; p.ldres = ldres; p.act = act;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;  // TF SAME, k=1: pad 0, samples x[0::s]
    p.M = (long)B * Ho * Wo;
    p.flat = stride == 1;
    p.Hg = Ho; p.Wg = Wo; p.Ha = H; p.Wa = W; p.Hc = Ho; p.Wc = Wo; p.sa = stride; p.sc = 1; p.py = p.px = 0;
    set_taps(p, 1, nullptr, nullptr);
    return dispatch(p, precision, static_cast<hipStream_t>(stream));
}

// The convolutions of a TRAINING forward pass (misc_py/denoiser-multi-gpu.py:200-540 with phase = True: every conv is followed by a batch
// norm on batch statistics): y = conv(x) with no affine and no activation, plus the per-channel mean and biased variance of y gathered in
// the GEMM's epilogue (one partial per 128-row tile, reduced in a fixed order by bn_stats_final): what emd_bn_stats_f32 /
// emd_bn_stats_images_f32 would return for y, without their pass over it.  images = 0: statistics over all B * Ho * Wo pixels (mean / var
// [Cout]); images = 1: per image ([B][Cout]; needs Ho * Wo % 128 == 0 so that no tile straddles two images: EMD_E_UNSUPPORTED
// otherwise -- call the conv and the statistics separately).  workspace: emd_conv_stats_workspace_bytes(B * Ho * Wo, Cout) bytes.
extern "C" size_t emd_conv_stats_workspace_bytes(long M, int Cout) {
    if (M <= 0 || Cout <= 0) return 0;
    return (size_t)((M + 127) / 128) * 2 * Cout * sizeof(double);
}

static int conv_stats_run(GemmParams& p, int B, long npix_img, int images, float* mean, float* var, void* workspace, int precision,
                          emd_stream_t stream, const emd_bn_train_fold_t* fold = nullptr) {
    emd::BnFoldArgs fa;
    if (fold) {
        int rcf = emd::bn_fold_args(fold, &fa);
        if (rcf != EMD_OK) return rcf;
    }
    const emd::BnFoldArgs* fp = fold ? &fa : nullptr;
    EMD_REQUIRE(mean && var && workspace && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EMD_E_INVALID,
                "emd_conv*_stats_f32: mean, var and an 8-byte aligned workspace are required");
    EMD_REQUIRE(!images || npix_img % 128 == 0, EMD_E_UNSUPPORTED,
                "emd_conv*_stats_f32: per-image statistics need Ho * Wo % 128 == 0 (a 128-row tile must not straddle two images)");
    p.stats_part = static_cast<double*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (always 128-row tiles: the number of partials per image must not depend on the column tile dispatch picks)
    int rc = dispatch(p, precision, st);
    if (rc != EMD_OK) return rc;
    if (images)
        return emd::launch_bn_stats_final(p.stats_part, (int)(npix_img / 128), p.N, npix_img, mean, var, st, nullptr, nullptr, 0.f, nullptr,
                                          nullptr, B, fp);
    return emd::launch_bn_stats_final(p.stats_part, (int)((p.M + 127) / 128), p.N, p.M, mean, var, st, nullptr, nullptr, 0.f, nullptr, nullptr, 1, fp);
}

static int conv1x1_stats_impl(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones,
                                     const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride,
                                     int precision, int images, float* mean, float* var, void* workspace, emd_stream_t stream,
                                     const emd_bn_train_fold_t* fold) {
    int rc = common_checks("emd_conv1x1_stats_f32", x, whi, wlo, ones, zeros, nullptr, nullptr, nullptr, y, Cin, Cout, ldx, ldy, 0,
                           precision);
    if (rc != EMD_OK) return rc;
    EMD_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv1x1_stats_f32: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_conv1x1_stats_f32: stride must be 1 or 2");
    GemmParams p{};
    p.A = x; p.Whi = whi; p.Wlo = wlo; p.C = y; p.res = nullptr;
    p.scale1 = ones; p.shift1 = zeros; p.scale2 = p.shift2 = nullptr;
    p.N = Cout; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK; p.ntaps = 1;
    p.lda = ldx; p.ldc = ldy; p.ldres = 0; p.act = 0;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    p.M = (long)B * Ho * Wo;
    p.flat = stride == 1;
    p.Hg = Ho; p.Wg = Wo; p.Ha = H; p.Wa = W; p.Hc = Ho; p.Wc = Wo; p.sa = stride; p.sc = 1; p.py = p.px = 0;
    set_taps(p, 1, nullptr, nullptr);
    return conv_stats_run(p, B, (long)Ho * Wo, images, mean, var, workspace, precision, stream, fold);
}

extern "C" int emd_conv1x1_stats_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones,
                                     const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride,
                                     int precision, int images, float* mean, float* var, void* workspace, emd_stream_t stream) {
    return conv1x1_stats_impl(x, ldx, whi, wlo, ones, zeros, y, ldy, B, H, W, Cin, Cout, stride, precision, images, mean, var, workspace, stream,
                              nullptr);
}

// ... and the training-mode fold of the norm behind it in the statistics' final kernel (emd_bn_train_fold[_images]_f32's step; one launch less)
extern "C" int emd_conv1x1_stats_fold_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones,
                                          const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride,
                                          int precision, int images, float* mean, float* var, void* workspace,
                                          const emd_bn_train_fold_t* fold, emd_stream_t stream) {
    EMD_REQUIRE(fold, EMD_E_INVALID, "emd_conv1x1_stats_fold_f32: null fold block");
    return conv1x1_stats_impl(x, ldx, whi, wlo, ones, zeros, y, ldy, B, H, W, Cin, Cout, stride, precision, images, mean, var, workspace, stream,
                              fold);
}

static int conv3x3_stats_impl(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones,
                                     const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int rate,
                                     int precision, int images, float* mean, float* var, void* workspace, emd_stream_t stream,
                                     const emd_bn_train_fold_t* fold) {
    int rc = common_checks("emd_conv3x3_stats_f32", x, whi, wlo, ones, zeros, nullptr, nullptr, nullptr, y, Cin, Cout, ldx, ldy, 0,
                           precision);
    if (rc != EMD_OK) return rc;
    EMD_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_stats_f32: bad shape");
    EMD_REQUIRE(rate >= 1 && rate <= 31, EMD_E_UNSUPPORTED, "emd_conv3x3_stats_f32: rate must be 1..31 (stride 1)");
    GemmParams p{};
    p.A = x; p.Whi = whi; p.Wlo = wlo; p.C = y; p.res = nullptr;
    p.scale1 = ones; p.shift1 = zeros; p.scale2 = p.shift2 = nullptr;
    p.N = Cout; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK;
    p.lda = ldx; p.ldc = ldy; p.ldres = 0; p.act = 0;
    p.M = (long)B * H * W;
    p.flat = 0;
    p.Hg = H; p.Wg = W; p.Ha = H; p.Wa = W; p.Hc = H; p.Wc = W; p.sa = 1; p.sc = 1; p.py = p.px = 0;
    int dy[9], dx[9];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {   // TF SAME at stride 1: (rate, rate) on both sides
            dy[ky * 3 + kx] = (ky - 1) * rate;
            dx[ky * 3 + kx] = (kx - 1) * rate;
        }
    set_taps(p, 9, dy, dx);
    return conv_stats_run(p, B, (long)H * W, images, mean, var, workspace, precision, stream, fold);
}

extern "C" int emd_conv3x3_stats_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones,
                                     const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int rate,
                                     int precision, int images, float* mean, float* var, void* workspace, emd_stream_t stream) {
    return conv3x3_stats_impl(x, ldx, whi, wlo, ones, zeros, y, ldy, B, H, W, Cin, Cout, rate, precision, images, mean, var, workspace, stream,
                              nullptr);
}

extern "C" int emd_conv3x3_stats_fold_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* ones,
                                          const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int rate,
                                          int precision, int images, float* mean, float* var, void* workspace,
                                          const emd_bn_train_fold_t* fold, emd_stream_t stream) {
    EMD_REQUIRE(fold, EMD_E_INVALID, "emd_conv3x3_stats_fold_f32: null fold block");
    return conv3x3_stats_impl(x, ldx, whi, wlo, ones, zeros, y, ldy, B, H, W, Cin, Cout, rate, precision, images, mean, var, workspace, stream,
                              fold);
}

// The transposed 3x3 stride-2 conv of a training forward pass (emd_deconv3x3s2_f32 with no affine, no activation) + the batch statistics of
// its output from the four phase GEMMs' epilogues: mean / var [Cout] over all B * 2H * 2W output pixels, or images != 0: [B][Cout] per
// image (needs H * W % 128 == 0).  workspace: emd_conv_stats_workspace_bytes(4 * B * H * W, Cout) bytes.
static int deconv_stats_impl(const float* x, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4], const float* ones,
                                         const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int precision,
                                         int images, float* mean, float* var, void* workspace, emd_stream_t stream,
                                         const emd_bn_train_fold_t* fold) {
    EMD_REQUIRE(whi, EMD_E_INVALID, "emd_deconv3x3s2_stats_f32: null weight table");
    emd::BnFoldArgs fa;
    if (fold) {
        int rcf = emd::bn_fold_args(fold, &fa);
        if (rcf != EMD_OK) return rcf;
    }
    const emd::BnFoldArgs* fp = fold ? &fa : nullptr;
    for (int ph = 0; ph < 4; ++ph) {
        int rc = common_checks("emd_deconv3x3s2_stats_f32", x, whi[ph], wlo ? wlo[ph] : nullptr, ones, zeros, nullptr, nullptr, nullptr, y,
                               Cin, Cout, ldx, ldy, 0, precision);
        if (rc != EMD_OK) return rc;
    }
    EMD_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_deconv3x3s2_stats_f32: bad shape");
    EMD_REQUIRE(mean && var && workspace && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EMD_E_INVALID,
                "emd_deconv3x3s2_stats_f32: mean, var and an 8-byte aligned workspace are required");
    const long npix_in = (long)H * W, M = (long)B * npix_in;
    EMD_REQUIRE(!images || npix_in % 128 == 0, EMD_E_UNSUPPORTED,
                "emd_deconv3x3s2_stats_f32: per-image statistics need H * W % 128 == 0 (a 128-row tile must not straddle two images)");
    const int n_mt = (int)((M + 127) / 128);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int ph = 0; ph < 4; ++ph) {
        GemmParams p{};
        int ky[4], kx[4];
        p.ntaps = emd_deconv_phase_taps(ph, ky, kx);
        p.A = x; p.Whi = whi[ph]; p.Wlo = wlo ? wlo[ph] : nullptr; p.C = y; p.res = nullptr;
        p.scale1 = ones; p.shift1 = zeros; p.scale2 = p.shift2 = nullptr;
        p.N = Cout; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK;
        p.lda = ldx; p.ldc = ldy; p.ldres = 0; p.act = 0;
        p.M = M;
        p.flat = 0;
        p.Hg = H; p.Wg = W; p.Ha = H; p.Wa = W; p.Hc = 2 * H; p.Wc = 2 * W; p.sa = 1; p.sc = 2;
        p.py = ph >> 1; p.px = ph & 1;
        int dy[4], dx[4];
        for (int t = 0; t < p.ntaps; ++t) {
            dy[t] = ky[t] == 2 ? -1 : 0;
            dx[t] = kx[t] == 2 ? -1 : 0;
        }
        set_taps(p, p.ntaps, dy, dx);
        p.stats_part = static_cast<double*>(workspace);
        p.stats_nph = 4; p.stats_ph = ph; p.stats_tpi = images ? (int)(npix_in / 128) : n_mt;
        int rc = dispatch(p, precision, st);
        if (rc != EMD_OK) return rc;
    }
    if (images)
        return emd::launch_bn_stats_final(static_cast<const double*>(workspace), 4 * (int)(npix_in / 128), Cout, 4 * npix_in, mean, var, st, nullptr,
                                          nullptr, 0.f, nullptr, nullptr, B, fp);
    return emd::launch_bn_stats_final(static_cast<const double*>(workspace), 4 * n_mt, Cout, 4 * M, mean, var, st, nullptr, nullptr, 0.f, nullptr,
                                      nullptr, 1, fp);
}

extern "C" int emd_deconv3x3s2_stats_f32(const float* x, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4], const float* ones,
                                         const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int precision,
                                         int images, float* mean, float* var, void* workspace, emd_stream_t stream) {
    return deconv_stats_impl(x, ldx, whi, wlo, ones, zeros, y, ldy, B, H, W, Cin, Cout, precision, images, mean, var, workspace, stream, nullptr);
}

extern "C" int emd_deconv3x3s2_stats_fold_f32(const float* x, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4],
                                              const float* ones, const float* zeros, float* y, int ldy, int B, int H, int W, int Cin, int Cout,
                                              int precision, int images, float* mean, float* var, void* workspace,
                                              const emd_bn_train_fold_t* fold, emd_stream_t stream) {
    EMD_REQUIRE(fold, EMD_E_INVALID, "emd_deconv3x3s2_stats_fold_f32: null fold block");
    return deconv_stats_impl(x, ldx, whi, wlo, ones, zeros, y, ldy, B, H, W, Cin, Cout, precision, images, mean, var, workspace, stream, fold);
}

// Data gradient of the stride-2 1x1 convolution: dx[b, 2i, 2j, :] (+)= dy[b, i, j, :] * W^T, the other pixels of dx
// are not touched (the caller zero-fills dx, or passes res == dx to add into a gradient that already exists).
// whi/wlo: W packed transposed (emd_pack_weights_dev with Cin/Cout swapped and cout_major = 1).
extern "C" int emd_conv1x1_s2_bwd_data_f32(const float* dy, int ldd, const uint16_t* whi, const uint16_t* wlo,
                                           const float* scale1, const float* shift1, const float* res, int ldres,
                                           float* dx, int ldx, int B, int H, int W, int Cout, int Cin, int precision,
                                           emd_stream_t stream) {
    int rc = common_checks("emd_conv1x1_s2_bwd_data_f32", dy, whi, wlo, scale1, shift1, nullptr, nullptr, res, dx, Cout, Cin,
                           ldd, ldx, ldres, precision);
    if (rc != EMD_OK) return rc;
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv1x1_s2_bwd_data_f32: bad shape");
    if (B == 0) return EMD_OK;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    GemmParams p{};
    p.A = dy; p.Whi = whi; p.Wlo = wlo; p.C = dx; p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = p.shift2 = nullptr;
    p.N = Cin; p.Cin = Cout; p.Cpad = (Cout + kBK - 1) / kBK * kBK; p.ntaps = 1;
    p.lda = ldd; p.ldc = ldx; p.ldres = ldres; p.act = 0;
    p.M = (long)B * Ho * Wo;
    p.flat = 0;
    p.Hg = Ho; p.Wg = Wo; p.Ha = Ho; p.Wa = Wo; p.Hc = H; p.Wc = W; p.sa = 1; p.sc = 2; p.py = p.px = 0;
    set_taps(p, 1, nullptr, nullptr);
    return dispatch(p, precision, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------------- 3x3 stride-2 transposed convolution
// y[2i+k] += x[i]*w[k] cropped to [0,2N) (gradient of the SAME stride-2 conv, denoiser.py:141-148):
//   even output index 2i  : taps k=0 (from x[i]) and k=2 (from x[i-1]);  odd 2i+1 : tap k=1 (from x[i]).
// The four (row-parity, column-parity) phases are four GEMMs over the INPUT grid with 4/2/2/1 taps,
// each with its own packed weight block (see emd_deconv_phase_taps).
extern "C" int emd_deconv_phase_taps(int phase, int* ky, int* kx) {
    // phase = 2*py + px; returns the number of taps and their 3x3 kernel coordinates, in the order
    // the packed weights of that phase must be laid out
    if (phase < 0 || phase > 3 || !ky || !kx) return 0;
    const int py = phase >> 1, px = phase & 1;
    int kys[2], kxs[2], ny, nx;
    if (py) { kys[0] = 1; ny = 1; } else { kys[0] = 0; kys[1] = 2; ny = 2; }
    if (px) { kxs[0] = 1; nx = 1; } else { kxs[0] = 0; kxs[1] = 2; nx = 2; }
    int n = 0;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) { ky[n] = kys[a]; kx[n] = kxs[b]; ++n; }
    return n;
}

extern "C" int emd_deconv3x3s2_f32(const float* x, int ldx, const uint16_t* const whi[4],
                                   const uint16_t* const wlo[4], const float* scale1, const float* shift1,
                                   float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                   int precision, emd_stream_t stream) {
    EMD_REQUIRE(whi, EMD_E_INVALID, "emd_deconv3x3s2_f32: null weight table");
    for (int ph = 0; ph < 4; ++ph) {
        int rc = common_checks("emd_deconv3x3s2_f32", x, whi[ph], wlo ? wlo[ph] : nullptr, scale1, shift1, nullptr,
                               nullptr, nullptr, y, Cin, Cout, ldx, ldy, 0, precision);
        if (rc != EMD_OK) return rc;
    }
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_deconv3x3s2_f32: bad shape");
    if (B == 0) return EMD_OK;
    for (int ph = 0; ph < 4; ++ph) {
        GemmParams p{};
        int ky[4], kx[4];
        p.ntaps = emd_deconv_phase_taps(ph, ky, kx);
        p.A = x; p.Whi = whi[ph]; p.Wlo = wlo ? wlo[ph] : nullptr; p.C = y; p.res = nullptr;
        p.scale1 = scale1; p.shift1 = shift1; p.scale2 = p.shift2 = nullptr;
        p.N = Cout; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK;
        p.lda = ldx; p.ldc = ldy; p.ldres = 0; p.act = act;
        p.M = (long)B * H * W;
        p.flat = 0;
        p.Hg = H; p.Wg = W; p.Ha = H; p.Wa = W; p.Hc = 2 * H; p.Wc = 2 * W; p.sa = 1; p.sc = 2;
        p.py = ph >> 1; p.px = ph & 1;
        int dy[4], dx[4];
        for (int t = 0; t < p.ntaps; ++t) {  // kernel index 2 reads the previous input sample
            dy[t] = ky[t] == 2 ? -1 : 0;
            dx[t] = kx[t] == 2 ? -1 : 0;
        }
        set_taps(p, p.ntaps, dy, dx);
        int rc = dispatch(p, precision, static_cast<hipStream_t>(stream));
        if (rc != EMD_OK) return rc;
    }
    return EMD_OK;
}

// ---------------------------------------------------------------------------------------------- dense 3x3 convolution
// tf.layers.conv2d / slim.conv2d with kernel_size 3, TF SAME padding, stride 1 or 2, dilation `rate`
// (stride 1 only) as a 9-tap implicit GEMM: tap (ky,kx) reads source pixel (i*s + ky*rate - pad_top,
// j*s + kx*rate - pad_left), zero outside.  Weights: emd_pack_weights_bf16(taps = 9, [ky][kx][Cin][Cout]).
extern "C" int emd_conv3x3_f32(const float* x, int ldx, const uint16_t* whi, const uint16_t* wlo,
                               const float* scale1, const float* shift1, const float* scale2,
                               const float* shift2, const float* res, int ldres, float* y, int ldy, int B,
                               int H, int W, int Cin, int Cout, int stride, int rate, int act, int precision,
                               emd_stream_t stream) {
    int rc = common_checks("emd_conv3x3_f32", x, whi, wlo, scale1, shift1, scale2, shift2, res, y, Cin, Cout, ldx,
                           ldy, ldres, precision);
    if (rc != EMD_OK) return rc;
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_f32: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_conv3x3_f32: stride must be 1 or 2");
    EMD_REQUIRE(rate >= 1 && rate <= 31 && (rate == 1 || stride == 1), EMD_E_UNSUPPORTED,
                "emd_conv3x3_f32: rate must be 1..31, and 1 when stride is 2");
    if (B == 0) return EMD_OK;
    GemmParams p{};
    p.A = x; p.Whi = whi; p.Wlo = wlo; p.C = y; p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.N = Cout; p.Cin = Cin; p.Cpad = (Cin + kBK - 1) / kBK * kBK;
    p.lda = ldx; p.ldc = ldy; p.ldres = ldres; p.act = act;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int eff = 2 * rate + 1;
    int pth = (Ho - 1) * stride + eff - H, ptw = (Wo - 1) * stride + eff - W;  // TF SAME: total padding
    if (pth < 0) pth = 0;
    if (ptw < 0) ptw = 0;
    const int pt = pth / 2, pl = ptw / 2;
    p.M = (long)B * Ho * Wo;
    p.flat = 0;
    p.Hg = Ho; p.Wg = Wo; p.Ha = H; p.Wa = W; p.Hc = Ho; p.Wc = Wo; p.sa = stride; p.sc = 1; p.py = p.px = 0;
    int dy[9], dx[9];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            dy[ky * 3 + kx] = ky * rate - pt;
            dx[ky * 3 + kx] = kx * rate - pl;
        }
    set_taps(p, 9, dy, dx);
    return dispatch(p, precision, static_cast<hipStream_t>(stream));
}
