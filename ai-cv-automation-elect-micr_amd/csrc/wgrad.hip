// Backward ("training") kernels of graph D' (misc_py/denoiser-multi-gpu.py:752-782 tf.gradients of the tower
// loss; :1011-1077 the Nesterov train op).  Data gradients of the convolutions reuse the forward implicit
// GEMM (gemm_conv.hip) with transposed packed weights; this file holds what has no forward twin:
//   emd_conv_wgrad_f32        dW[t][k][n] = sum_m A[src_t(m)][k] * dY[m][n]   (any 1x1 / 3x3 / transposed conv)
//   emd_pack_weights_dev      fp32 weights on the device -> bf16 hi/lo planes, either orientation, any tap subset
// (bn_train.hip: training-mode batch norm forward fold / backward; bwd_misc.hip: depthwise, resize, pooling,
// 1-channel conv backward, the loss and the optimizer step.)
// Numerics: split-bf16 MFMA (~2^-16 relative) or fp32 VALU per block, fp32 accumulation, float atomics across blocks.
#include "mfma_common.hpp"

using namespace emd;

namespace {

// ------------------------------------------------------------------------------------------------
// Weight gradient.  Block = 256 threads, output tile 64 (k) x 64 (n), thread = 4x4 micro-tile; the block walks
// its slice of M in chunks of 32 positions staged in LDS (A chunk [32][64], dY chunk [32][64]); per position a
// thread does 2 ds_read_b128 and 16 FMAs.  Slices of M are combined with float atomics into dW (zeroed by the
// caller).  Row map as in gemm_conv.hip: m -> (b,i,j); A is read at (i*sa+dy_t, j*sa+dx_t), dY at m.
struct WgradParams {
    const float* A;   // [.., lda]  activations (or, for the transposed conv, the upstream gradient image)
    const float* dY;  // [M, ldd]   gradient w.r.t. the conv output (or, for the transposed conv, its input x)
    float* dW;        // [taps][K][N]
    long M;
    int K, N, lda, ldd, ntaps, msplit;
    int flat, Hg, Wg, Ha, Wa, sa;
    unsigned long long dyp, dxp;
};

__device__ __attribute__((aligned(16))) float g_zero_w[4];   // what the pixels past a slice's end and the padded taps load

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int TK = 64, TN = 64, MC = 32;
    __shared__ __attribute__((aligned(16))) float As[MC][TK + 4];
    __shared__ __attribute__((aligned(16))) float Ds[MC][TN + 4];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * TK, n0 = blockIdx.y * TN;
    const int tap = blockIdx.z / p.msplit, split = blockIdx.z % p.msplit;
    const long mper = ((p.M + p.msplit - 1) / p.msplit + MC - 1) / MC * MC;
    const long mbeg = (long)split * mper, mend = mbeg + mper < p.M ? mbeg + mper : p.M;
    const int dyo = (int)((p.dyp >> (7 * tap)) & 127) - 64, dxo = (int)((p.dxp >> (7 * tap)) & 127) - 64;
    const int tk = (tid >> 4) * 4, tn = (tid & 15) * 4;  // micro-tile origin
    const int lr = tid >> 4, lc = (tid & 15) * 4;        // loader: rows lr, lr+16; 4 consecutive channels
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (long mc = mbeg; mc < mend; mc += MC) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = lr + h * 16;
            const long m = mc + r;
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f), dv = av;
            if (m < mend) {
                long src = m;
                if (!p.flat) {
                    const int j = (int)(m % p.Wg);
                    const long t = m / p.Wg;
                    const int iy = (int)(t % p.Hg) * p.sa + dyo, ix = j * p.sa + dxo;
                    src = (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) ? ((t / p.Hg) * p.Ha + iy) * (long)p.Wa + ix : -1;
                }
                if (src >= 0 && k0 + lc < p.K) av = *reinterpret_cast<const float4*>(p.A + src * p.lda + k0 + lc);
                if (n0 + lc < p.N) dv = *reinterpret_cast<const float4*>(p.dY + m * p.ldd + n0 + lc);
            }
            *reinterpret_cast<float4*>(&As[r][lc]) = av;
            *reinterpret_cast<float4*>(&Ds[r][lc]) = dv;
        }
        __syncthreads();
#pragma unroll 8
        for (int mm = 0; mm < MC; ++mm) {
            const float4 a = *reinterpret_cast<const float4*>(&As[mm][tk]);
            const float4 d = *reinterpret_cast<const float4*>(&Ds[mm][tn]);
            const float av[4] = {a.x, a.y, a.z, a.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], dv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* out = p.dW + (long)tap * p.K * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + tk + i;
        if (k >= p.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn + j;
            if (n < p.N) atomicAdd(out + (long)k * p.N + n, acc[i][j]);
        }
    }
}

// K = 4 (the network's first pointwise convs: a 1-channel image padded to 4 channels, cnn0 and residual0 of graph D'): the 64 x 64 tile
// above is 1/16 full there and walks M in 32-pixel chunks between barriers -- 214 us for a pair of 512^2 images whose dY a stream
// kernel reads in 30 (round 4 profile: 2.1 ms of the step's kernel time, at the very end of every tower's reverse pass).  Here a thread
// owns four output columns and every 16th pixel of its slab: one 16-byte load of A, one of dY, sixteen FMAs per pixel, four pixels' loads
// in flight; 16 pixel lanes -> LDS -> one float atomic per (k, n) and workgroup.  One tap, any stride (row map as above).
__global__ __launch_bounds__(256) void conv_wgrad_k4_kernel(const WgradParams p, long pix_per_slab) {
    const int nq = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int n = blockIdx.x * 64 + nq * 4;
    const bool live = n < p.N;
    const int nc = live ? n : 0;
    const long p0 = (long)blockIdx.y * pix_per_slab, p1 = p0 + pix_per_slab < p.M ? p0 + pix_per_slab : p.M;
    const int dyo = (int)(p.dyp & 127) - 64, dxo = (int)(p.dxp & 127) - 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    auto src_of = [&](long m) -> long {
        if (p.flat) return m;
        const int j = (int)(m % p.Wg);
        const long t = m / p.Wg;
        const int iy = (int)(t % p.Hg) * p.sa + dyo, ix = j * p.sa + dxo;
        return (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) ? ((t / p.Hg) * p.Ha + iy) * (long)p.Wa + ix : -1;
    };
    auto fma16 = [&](const f32x4 a, const f32x4 d) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], d[j], acc[i][j]);
    };
    long m = p0 + lane;
    for (; m + 48 < p1; m += 64) {
        f32x4 a[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long mm = m + 16 * u, src = src_of(mm);
            a[u] = *reinterpret_cast<const f32x4*>(src >= 0 ? p.A + src * p.lda : g_zero_w);
            d[u] = *reinterpret_cast<const f32x4*>(p.dY + mm * p.ldd + nc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) fma16(a[u], d[u]);
    }
    for (; m < p1; m += 16) {
        const long src = src_of(m);
        fma16(*reinterpret_cast<const f32x4*>(src >= 0 ? p.A + src * p.lda : g_zero_w), *reinterpret_cast<const f32x4*>(p.dY + m * p.ldd + nc));
    }
    __shared__ float red[16][4][64 + 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[lane][i][nq * 4 + j] = acc[i][j];
    __syncthreads();
    const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int nn = blockIdx.x * 64 + c;
    if (nn < p.N && k < p.K) {
        float sum = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) sum += red[l][k][c];
        atomicAdd(p.dW + (long)k * p.N + nn, sum);
    }
}

// ------------------------------------------------------------------------------------------------
// The same weight gradient on the matrix cores (the default path; the VALU kernel above remains for K or N < 32,
// where a 128x128 tile would be nearly empty).  dW[k][n] = sum_m A[m][k] * dY[m][n]: the CONTRACTION runs over
// pixels, which is the strided direction of both NHWC operands, while an MFMA fragment wants 8 consecutive
// contraction elements per lane.  So every 64-pixel chunk of A (128 channels) and dY (128 channels) is split into
// bf16 hi/lo and TRANSPOSED on its way into LDS (planes [channel][pixel], one ds_write_b64 = 4 pixels of one
// channel), after which the k-loop is the forward GEMM's: ds_read_b128 fragments, three
// mfma_f32_32x32x16_bf16 per fragment pair (lo*hi + hi*lo + hi*hi), fp32 accumulators.  Block = 4 waves, each a
// 64x64 sub-tile; the next chunk's 16 global loads per thread are in flight during the MFMA phase.  The M range is
// split over blockIdx.z and combined with float atomics, as above.
template <int UA, int UD>  // tile = 64*UA channels of A x 64*UD channels of dY
__global__ __launch_bounds__(256, 2) void conv_wgrad_mfma_kernel(const WgradParams p) {
    constexpr int TK = 64 * UA, TN = 64 * UD, MC = 64;
    constexpr int LDM = MC + 8;  // bf16 elements per LDS row (144 B): conflict-free ds_read_b128 fragments
    __shared__ __attribute__((aligned(16))) uint16_t Ah[TK][LDM], Al[TK][LDM], Dh[TN][LDM], Dl[TN][LDM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wk = wv >> 1, wn = wv & 1;   // 2 x 2 waves, each (32*UA) x (32*UD)
    const int fr = lane & 31, fh = lane >> 5;
    // XCD-aware block order: workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  All (k,n)
    // tiles of one (tap, M-slice) read the same A and dY rows, so they are given ids that land on ONE XCD and are
    // dispatched together: the slice is fetched into that L2 once instead of once per tile.
    const int ntile = gridDim.x * gridDim.y, nblk = ntile * gridDim.z;
    int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile = bid % ntile, zz = bid / ntile;
    const int k0 = (tile % gridDim.x) * TK, n0 = (tile / gridDim.x) * TN;
    const int tap = zz / p.msplit, split = zz % p.msplit;
    const int M = (int)p.M;  // the host guarantees M < 2^31: pixel indices are 32-bit, byte offsets 64-bit
    const int mper = ((M + p.msplit - 1) / p.msplit + MC - 1) / MC * MC;
    const int mbeg = split * mper, mend = mbeg + mper < M ? mbeg + mper : M;
    const int dyo = (int)((p.dyp >> (7 * tap)) & 127) - 64, dxo = (int)((p.dxp >> (7 * tap)) & 127) - 64;
    // loader: 4 consecutive pixels (pixel quad pq) x channel quads cq (and cq+16 when the tile is 128 wide).
    // Channel quads beyond K / N are read from a clamped (in-bounds) address and left as they come: they only feed
    // rows / columns of the tile that are never written out.  A chunk whose 64 pixels are all real and ungathered
    // takes the straight-line path (back-to-back loads); chunk tails and tap-shifted / strided reads are masked.
    const int pq = tid & 15, cq = tid >> 4;
    int ca[UA], cd[UD];
#pragma unroll
    for (int u = 0; u < UA; ++u) ca[u] = (k0 + 4 * cq + 64 * u < p.K) ? k0 + 4 * cq + 64 * u : 0;
#pragma unroll
    for (int u = 0; u < UD; ++u) cd[u] = (n0 + 4 * cq + 64 * u < p.N) ? n0 + 4 * cq + 64 * u : 0;

    f32x16 acc[UA][UD];
#pragma unroll
    for (int i = 0; i < UA; ++i)
#pragma unroll
        for (int j = 0; j < UD; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[UA][4], rd[UD][4];  // staging registers: [channel quad][pixel]
    // ONE straight-line path (round 4): every load is issued unconditionally from an address that is either the element's or a 16-byte
    // zero buffer's.  The first version had a fast path (full, ungathered chunk) and a masked path (zero, then load under `if`): the
    // compiler joined them with register copies, and a copy of a loaded register is a use -- s_waitcnt vmcnt right behind the loads, the
    // whole latency of the NEXT chunk's loads exposed in front of this chunk's MFMAs (in-kernel stamps: 38 % of the kernel; same
    // disease and cure as gemm_conv.hip's loader).
    auto load_chunk = [&](int mc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mc + 4 * pq + j;
            const bool in = m < mend;
            long src = m;
            if (!p.flat) {   // kernel-uniform; integer arithmetic only
                const unsigned um = (unsigned)(in ? m : mbeg), jx = um % (unsigned)p.Wg, t = um / (unsigned)p.Wg;
                const int iy = (int)(t % (unsigned)p.Hg) * p.sa + dyo, ix = (int)jx * p.sa + dxo;
                src = (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) ? ((long)(t / (unsigned)p.Hg) * p.Ha + iy) * (long)p.Wa + ix : -1;
            }
            const bool oka = in && src >= 0;
            const float* ap = p.A + (oka ? src : 0) * p.lda;
            const float* dp = p.dY + (long)(in ? m : 0) * p.ldd;
#pragma unroll
            for (int u = 0; u < UA; ++u) ra[u][j] = *reinterpret_cast<const f32x4*>(oka ? ap + ca[u] : g_zero_w);
#pragma unroll
            for (int u = 0; u < UD; ++u) rd[u][j] = *reinterpret_cast<const f32x4*>(in ? dp + cd[u] : g_zero_w);
        }
    };
    // 4 pixels of one channel -> 4 bf16 hi + 4 bf16 lo, one ds_write_b64 each
    auto store_unit = [&](uint16_t(*H)[LDM], uint16_t(*L)[LDM], int row, const f32x4* r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned h01, l01, h23, l23;
            split2(r[0][c], r[1][c], h01, l01);
            split2(r[2][c], r[3][c], h23, l23);
            const u32x2 hv = {h01, h23}, lv = {l01, l23};
            *reinterpret_cast<u32x2*>(&H[row + c][4 * pq]) = hv;
            *reinterpret_cast<u32x2*>(&L[row + c][4 * pq]) = lv;
        }
    };

    if (mbeg < mend) load_chunk(mbeg);
    for (int mc = mbeg; mc < mend; mc += MC) {
        __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
        for (int u = 0; u < UA; ++u) store_unit(Ah, Al, 4 * cq + 64 * u, ra[u]);
#pragma unroll
        for (int u = 0; u < UD; ++u) store_unit(Dh, Dl, 4 * cq + 64 * u, rd[u]);
        __syncthreads();
        if (mc + MC < mend) load_chunk(mc + MC);  // in flight during the MFMA phase
#pragma unroll
        for (int ks = 0; ks < MC / 16; ++ks) {
            bf16x8 ah[UA], al[UA], bh[UD], bl[UD];
#pragma unroll
            for (int i = 0; i < UA; ++i) {
                const int r = wk * 32 * UA + i * 32 + fr;
                ah[i] = *reinterpret_cast<const bf16x8*>(&Ah[r][ks * 16 + fh * 8]);
                al[i] = *reinterpret_cast<const bf16x8*>(&Al[r][ks * 16 + fh * 8]);
            }
#pragma unroll
            for (int j = 0; j < UD; ++j) {
                const int r = wn * 32 * UD + j * 32 + fr;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Dh[r][ks * 16 + fh * 8]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&Dl[r][ks * 16 + fh * 8]);
            }
#pragma unroll
            for (int i = 0; i < UA; ++i)
#pragma unroll
                for (int j = 0; j < UD; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    // C/D layout of mfma_32x32: col (N index) = lane & 31, row (K index) = (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* out = p.dW + (long)tap * p.K * p.N;
#pragma unroll
    for (int i = 0; i < UA; ++i)
#pragma unroll
        for (int j = 0; j < UD; ++j) {
            const int n = n0 + wn * 32 * UD + j * 32 + fr;
            if (n >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wk * 32 * UA + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (k < p.K) atomicAdd(out + (long)k * p.N + n, acc[i][j][e]);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// fp32 weights on the DEVICE, [src_taps][Cin][Cout] (cout_major = 0) or [src_taps][Cout][Cin] (cout_major = 1)
// -> packed bf16 hi/lo planes [Npad][ntaps][Cpad] (the layout of emd_pack_weights_bf16).  Packed tap t comes from
// source tap (sel >> 4t) & 15, so one kernel serves the forward pack, the flipped pack of the data gradient
// and the four tap subsets of the transposed conv.  One thread per packed element (padding written as zero).
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int ntaps, unsigned long long sel,
                                                           int Cin, int Cout, int cout_major, int cpad, long total,
                                                           uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cpad);
    const long r = idx / cpad;
    const int t = (int)(r % ntaps);
    const int n = (int)(r / ntaps);
    float v = 0.f;
    if (c < Cin && n < Cout) {
        const long st = (long)((sel >> (4 * t)) & 15);
        v = cout_major ? w[(st * Cout + n) * Cin + c] : w[(st * Cin + c) * Cout + n];
    }
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    hi[idx] = __builtin_bit_cast(uint16_t, h);
    lo[idx] = __builtin_bit_cast(uint16_t, l);
}

// every pack of a model in one launch: block -> job by binary search over the jobs' first blocks, then pack_weights_kernel's body
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const emd_pack_job_t* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    const long blk = blockIdx.x;
    while (lo < hi) {   // the last job whose first_block <= blk (block-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blk) lo = mid;
        else hi = mid - 1;
    }
    const emd_pack_job_t j = jobs[lo];
    const long lb = blk - j.first_block;
    auto split_store = [&](long o, const float (&v)[4]) {   // four consecutive packed elements: one 8-byte store per plane
        unsigned h01, l01, h23, l23;
        split2(v[0], v[1], h01, l01);
        split2(v[2], v[3], h23, l23);
        *reinterpret_cast<u32x2*>(j.hi + o) = u32x2{h01, h23};
        *reinterpret_cast<u32x2*>(j.lo + o) = u32x2{l01, l23};
    };
    if (!j.cout_major) {
        // [taps][Cin][Cout] -> [Npad][taps][Cpad] is a transpose: one thread per packed element read with a stride of Cout floats -- a
        // 32-byte sector per 4-byte value, 2 GB fetched for the model's 160 MB of weights (round 4: profiles/r04_t_traffic_by_kernel.txt).
        // A block moves a 64 (c) x 16 (n) tile of one tap through LDS: 64-byte runs read (a thread = four consecutive n of one c), 128-byte
        // runs written (a thread = four consecutive c of one n, 8 bytes per plane).
        __shared__ float tile[64][17];
        const int ntc = (j.cpad + 63) >> 6;
        const int tc = (int)(lb % ntc);
        const long rest = lb / ntc;
        const int t = (int)(rest % j.ntaps);
        const int tn = (int)(rest / j.ntaps);
        const int npad = (int)(j.total / ((long)j.ntaps * j.cpad));
        const long st = (long)((j.tap_sel >> (4 * t)) & 15);
        {
            const int c = tc * 64 + (threadIdx.x >> 2), n = tn * 16 + (threadIdx.x & 3) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (c < j.cin) {
                const float* src = j.w + (st * j.cin + c) * j.cout + n;
                if (n + 3 < j.cout && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {   // (a variable's slice of the flat parameter vector need not be 16-byte aligned)
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (n + k < j.cout) v[k] = src[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[threadIdx.x >> 2][(threadIdx.x & 3) * 4 + k] = v[k];
        }
        __syncthreads();
        const int n = tn * 16 + (threadIdx.x >> 4), c = tc * 64 + (threadIdx.x & 15) * 4;
        if (n < npad && c < j.cpad) {   // cpad % 64 == 0: four packed elements are all inside
            const float v[4] = {tile[(threadIdx.x & 15) * 4 + 0][threadIdx.x >> 4], tile[(threadIdx.x & 15) * 4 + 1][threadIdx.x >> 4],
                                tile[(threadIdx.x & 15) * 4 + 2][threadIdx.x >> 4], tile[(threadIdx.x & 15) * 4 + 3][threadIdx.x >> 4]};
            split_store(((long)n * j.ntaps + t) * j.cpad + c, v);
        }
        return;
    }
    // cout_major: source rows are contiguous along c -- a thread packs four consecutive c (cpad % 64 == 0)
    const long idx = (lb * 256 + threadIdx.x) * 4;
    if (idx >= j.total) return;
    const int c = (int)(idx % j.cpad);
    const long r = idx / j.cpad;
    const int t = (int)(r % j.ntaps);
    const int n = (int)(r / j.ntaps);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < j.cout) {
        const long st = (long)((j.tap_sel >> (4 * t)) & 15);
        const float* src = j.w + (st * j.cout + n) * j.cin + c;
        if (c + 3 < j.cin && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const float4 q = *reinterpret_cast<const float4*>(src);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c + k < j.cin) v[k] = src[k];
        }
    }
    split_store(idx, v);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int emd_conv_wgrad_f32(const float* a, int lda, const float* dy, int ldd, float* dw, int B, int Hg, int Wg,
                                  int Ha, int Wa, int K, int N, int ntaps, const int* tap_dy, const int* tap_dx, int sa,
                                  emd_stream_t stream) {
    EMD_REQUIRE(a && dy && dw, EMD_E_INVALID, "emd_conv_wgrad_f32: null pointer");
    EMD_REQUIRE(B >= 0 && Hg >= 1 && Wg >= 1 && Ha >= 1 && Wa >= 1 && K >= 4 && N >= 4, EMD_E_INVALID, "emd_conv_wgrad_f32: bad shape");
    EMD_REQUIRE(ntaps >= 1 && ntaps <= 9 && (ntaps == 1 || (tap_dy && tap_dx)) && sa >= 1, EMD_E_INVALID, "emd_conv_wgrad_f32: bad taps");
    EMD_REQUIRE(K % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldd % 4 == 0 && lda >= K && ldd >= N && emd::aligned16(a) &&
                    emd::aligned16(dy), EMD_E_ALIGN, "emd_conv_wgrad_f32: K, N, lda, ldd multiples of 4; 16-byte aligned pointers");
    if (B == 0) return EMD_OK;
    WgradParams p{};
    p.A = a; p.dY = dy; p.dW = dw;
    p.M = (long)B * Hg * Wg; p.K = K; p.N = N; p.lda = lda; p.ldd = ldd; p.ntaps = ntaps;
    p.Hg = Hg; p.Wg = Wg; p.Ha = Ha; p.Wa = Wa; p.sa = sa;
    p.flat = ntaps == 1 && sa == 1 && Ha == Hg && Wa == Wg && (!tap_dy || (tap_dy[0] == 0 && tap_dx[0] == 0));
    for (int t = 0; t < ntaps; ++t) {
        const int dyv = tap_dy ? tap_dy[t] : 0, dxv = tap_dx ? tap_dx[t] : 0;
        EMD_REQUIRE(dyv >= -64 && dyv < 64 && dxv >= -64 && dxv < 64, EMD_E_UNSUPPORTED, "emd_conv_wgrad_f32: tap offset out of range");
        p.dyp |= (unsigned long long)(dyv + 64) << (7 * t);
        p.dxp |= (unsigned long long)(dxv + 64) << (7 * t);
    }
    const bool mfma = K >= 32 && N >= 32;
    EMD_REQUIRE(p.M < (1L << 31), EMD_E_UNSUPPORTED, "emd_conv_wgrad_f32: more than 2^31 pixels");
    const bool small_m = p.M <= 8192;   // (the dev knobs below act on the 1/16-resolution layers of a tower only)
    const bool t64 = emd::g_knobs.wgrad_tile == 64 && small_m;
    const int tk = mfma ? (K > 64 && !t64 ? 128 : 64) : 64, tn = mfma ? (N > 64 && !t64 ? 128 : 64) : 64;
    const int kt = (K + tk - 1) / tk, nt = (N + tn - 1) / tn;
    const long tiles = (long)kt * nt * ntaps;
    const long maxsplit = (p.M + (mfma ? 511 : 2047)) / (mfma ? 512 : 2048);
    long want;
    if (mfma) {
        // MI355X: 256 CUs x 2 resident workgroups.  A grid a little over one round (576 workgroups on 512 slots) takes two rounds
        // of time: aim at r whole rounds, r = 4 / 2 while a workgroup still gets 16 chunks of 64 pixels, else one round.
        const long slots = 512;
        want = slots / tiles;
        for (int r = 4; r >= 2; r >>= 1)
            if ((p.M / 64) / (slots * r / tiles > 0 ? slots * r / tiles : 1) >= 16) { want = slots * r / tiles; break; }
    } else {
        want = 2048 / tiles;  // enough workgroups to fill the chip a few times
    }
    if (want < 1) want = 1;
    if (emd::g_knobs.wgrad_msplit > 0 && small_m && want > emd::g_knobs.wgrad_msplit) want = emd::g_knobs.wgrad_msplit;
    p.msplit = (int)(want < maxsplit ? want : maxsplit);
    if (p.msplit < 1) p.msplit = 1;
    if (mfma) {   // slices are whole 64-pixel chunks: drop the ones that rounding left empty
        const long mper = ((p.M + p.msplit - 1) / p.msplit + 63) / 64 * 64;
        p.msplit = (int)((p.M + mper - 1) / mper);
    }
    EMD_REQUIRE((long)ntaps * p.msplit <= 65535, EMD_E_UNSUPPORTED, "emd_conv_wgrad_f32: grid too large");
    if (mfma) {
        const dim3 grid(kt, nt, ntaps * p.msplit);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (tk == 128 && tn == 128) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<2, 2>), grid, dim3(256), 0, st, p);
        else if (tk == 128) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<2, 1>), grid, dim3(256), 0, st, p);
        else if (tn == 128) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<1, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_mfma_kernel<1, 1>), grid, dim3(256), 0, st, p);
        return emd::check_launch("conv_wgrad_mfma_kernel");
    }
    if (K == 4 && ntaps == 1 && lda % 4 == 0 && N % 4 == 0) {
        long nslab = (p.M + 2047) / 2048;
        if (nslab > 512) nslab = 512;
        const long pps = ((p.M + nslab - 1) / nslab + 15) / 16 * 16;
        hipLaunchKernelGGL(conv_wgrad_k4_kernel, dim3((N + 63) / 64, (unsigned)((p.M + pps - 1) / pps)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           p, pps);
        return emd::check_launch("conv_wgrad_k4_kernel");
    }
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(kt, nt, ntaps * p.msplit), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return emd::check_launch("conv_wgrad_kernel");
}

extern "C" int emd_pack_weights_dev(const float* w, int src_taps, int ntaps, const int* tap_sel, int Cin, int Cout,
                                    int cout_major, uint16_t* hi, uint16_t* lo, emd_stream_t stream) {
    EMD_REQUIRE(w && hi && lo, EMD_E_INVALID, "emd_pack_weights_dev: null pointer");
    EMD_REQUIRE(src_taps >= 1 && src_taps <= 9 && ntaps >= 1 && ntaps <= 9 && Cin >= 1 && Cout >= 1, EMD_E_INVALID,
                "emd_pack_weights_dev: bad shape");
    EMD_REQUIRE(tap_sel || ntaps == src_taps, EMD_E_INVALID, "emd_pack_weights_dev: a tap subset needs tap_sel");
    unsigned long long sel = 0;
    for (int t = 0; t < ntaps; ++t) {
        const int s = tap_sel ? tap_sel[t] : t;
        EMD_REQUIRE(s >= 0 && s < src_taps, EMD_E_INVALID, "emd_pack_weights_dev: tap_sel out of range");
        sel |= (unsigned long long)s << (4 * t);
    }
    const int cpad = (Cin + kBK - 1) / kBK * kBK;
    const long npad = (Cout + kNPadTo - 1) / kNPadTo * kNPadTo;
    const long total = npad * ntaps * cpad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, ntaps, sel, Cin, Cout, cout_major, cpad, total, hi, lo);
    return emd::check_launch("pack_weights_kernel");
}

extern "C" int emd_pack_job_fill(emd_pack_job_t* job, const float* w, int src_taps, int ntaps, const int* tap_sel, int Cin, int Cout,
                                 int cout_major, uint16_t* hi, uint16_t* lo) {
    EMD_REQUIRE(job && w && hi && lo, EMD_E_INVALID, "emd_pack_job_fill: null pointer");
    EMD_REQUIRE(src_taps >= 1 && src_taps <= 9 && ntaps >= 1 && ntaps <= 9 && Cin >= 1 && Cout >= 1, EMD_E_INVALID,
                "emd_pack_job_fill: bad shape");
    EMD_REQUIRE(tap_sel || ntaps == src_taps, EMD_E_INVALID, "emd_pack_job_fill: a tap subset needs tap_sel");
    unsigned long long sel = 0;
    for (int t = 0; t < ntaps; ++t) {
        const int s = tap_sel ? tap_sel[t] : t;
        EMD_REQUIRE(s >= 0 && s < src_taps, EMD_E_INVALID, "emd_pack_job_fill: tap_sel out of range");
        sel |= (unsigned long long)s << (4 * t);
    }
    const int cpad = (Cin + kBK - 1) / kBK * kBK;
    const long npad = (Cout + kNPadTo - 1) / kNPadTo * kNPadTo;
    job->w = w; job->hi = hi; job->lo = lo; job->tap_sel = sel;
    job->total = npad * ntaps * cpad;
    job->first_block = 0;
    // cout_major: a thread per four packed elements; else 16 (n) x 64 (c) tiles per tap (see pack_weights_batch_kernel)
    job->n_blocks = cout_major ? (job->total / 4 + 255) / 256 : (long)ntaps * ((npad + 15) / 16) * ((cpad + 63) / 64);
    job->ntaps = ntaps; job->cin = Cin; job->cout = Cout; job->cout_major = cout_major ? 1 : 0; job->cpad = cpad; job->pad_ = 0;
    return EMD_OK;
}

extern "C" int emd_pack_weights_batch_dev(const emd_pack_job_t* jobs_dev, int n_jobs, long n_blocks, emd_stream_t stream) {
    EMD_REQUIRE(jobs_dev && n_jobs >= 1, EMD_E_INVALID, "emd_pack_weights_batch_dev: null table / no jobs");
    EMD_REQUIRE(n_blocks >= 1 && n_blocks <= 0x7fffffffL, EMD_E_UNSUPPORTED, "emd_pack_weights_batch_dev: bad block count");
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), jobs_dev, n_jobs);
    return emd::check_launch("pack_weights_batch_kernel");
}
