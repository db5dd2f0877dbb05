// sep_gemm.hip -- the separable convs of the 728-channel flow as ONE kernel: the depthwise 3x3 stage is a K-chunked PRODUCER inside
// the pointwise GEMM (machine_learning/denoiser.py:110-136 for the 36 blocks of :312-325, cnn3 / cnn3_last of :297-302).
//
// Why: as two kernels (emd_dw3x3_split32_f32 -> emd_conv1x1_split32_f32) every block launches twice, the depthwise result goes
// through memory, and the GEMM's 256 x 128 tiles re-read the activation tile once per N tile (6x) while the matrix cores idle
// during 24 % of a tile's life (prologue + epilogue).  sep_fused.hip covers Cout <= 128 only: its depthwise result must serve
// every N tile, and at 728 channels a tile's A operand (128 pixels x 728 channels, hi + lo) does not fit in LDS.  It does not
// have to: K is walked in 32-channel steps, and a step's A slice is 16 KB.
//
// Workgroup = 256 threads (one wave per SIMD, up to 512 registers per lane) = 128 pixels (4 rows x 32 columns of one image) x 384
// output channels (half of N padded to 768): 192 accumulator registers per lane, 4 x 3 MFMA tiles of 32 x 32 per wave.  The
// two N halves of a pixel tile are two workgroups (the depthwise stage is computed twice: 144 FMAs per lane and K step against
// 72 MFMAs of 32 cycles).  Per K step of 32 channels:
//   DMA   (global_load_lds_dwordx4, issued one step ahead): the 6 x 34 pixel fp32 input patch of the step's 32 channels (26 KB;
//         pixels outside the image and channels beyond Cin come from a zero buffer = TF SAME padding), the step's 9 x 32
//         depthwise weights, and the step's 384 x 32 weight slice in bf16 hi / lo (48 KB) from L2;
//   dw    lane = 4 consecutive pixels x 4 channels: 18 ds_read_b128 of the patch, 144 FMAs, split to bf16 hi / lo, 8 ds_write_b64
//         into the A slice (128 rows x 128 B, the XOR-swizzled row layout of gemm_split.hip);
//   MFMA  2 sub-steps of 16 channels x (8 A + 6 B fragment reads, 36 v_mfma_f32_32x32x16_bf16): acc += Alo Whi + Ahi Wlo + Ahi Whi.
// Two barriers per step.  LDS: patch 26 KB + A 16 KB + W 2 x 48 KB + depthwise weights = 141 KB.  Every LDS image is laid out so
// that its 16-byte reads are bank-conflict free (patch: 8-pixel blocks with pixels 4..7 pairwise swapped so that the two pixel
// groups a 16-lane read group touches sit in different bank halves; W: 64-byte rows with chunk ^ ((row >> 2) & 3); A: as
// gemm_split.hip).  Epilogue: per-channel affine (both folded batch norms) + activation + residual straight from the MFMA C/D
// layout, 32 lanes x 4 B = one 128-byte run of one pixel per half wave.
// The GEMM sums the same products in the same order along K as emd_conv1x1_split32_f32; the depthwise stage adds its 9 taps in
// sep_fused.hip's order (row by row, left to right), not in the rolling kernel's, so results agree with the two-kernel route to
// float32 rounding of the depthwise sums (1e-7), not bit for bit.  The kernel is selected by (H, W, Cin, Cout) only, never by the
// batch size: image b of a batch still equals the image run alone, bit for bit.
#include "mfma_common.hpp"

using namespace emd;

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(128))) unsigned char g_sg_zero[4096];

struct SepGemmParams {
    const float* x;       // [B,H,W,Cin] fp32, pixel pitch ldx floats
    const float* dw;      // [9][Cin]
    const uint16_t* Whi;  // [Npad][Ktot] bf16 planes (emd_pack_weights_bf16, 1 tap)
    const uint16_t* Wlo;
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    const float* res;
    float* y;
    int H, W, Cin, N, Npad, Ktot;
    int ldx, ldy, ldres, act;
    int tiles_x, tiles_y;   // tiles per image row / column
    long long* stamps;      // dev hook (emd_debug_sepgemm_stamps): per-workgroup phase cycle sums, NULL otherwise
};

constexpr int TM = 128, NH = 384, TROWS = 4, TCOLS = 32;
constexpr int PCOLS = TCOLS + 2, PROWS = TROWS + 2, PPIX = PROWS * PCOLS;   // 6 x 34 = 204 patch pixels
constexpr int P_INSTR = (PPIX + 7) / 8;                                     // 26 DMA instructions of 8 pixels x 128 B
constexpr int P_BYTES = P_INSTR * 1024;
constexpr int A_BYTES = TM * 128;
constexpr int B16_BYTES = NH * 64;                                          // one 16-channel sub-step of the weight slice
constexpr int WK_BYTES = 2048;                                              // 9 x 32 floats, padded to two DMA instructions
constexpr int OFF_A = P_BYTES, OFF_B = OFF_A + A_BYTES, OFF_WK = OFF_B + 4 * B16_BYTES, SMEM = OFF_WK + WK_BYTES;
static_assert(SMEM <= 160 * 1024, "LDS budget");

// patch pixel L (row-major over 6 x 34) -> byte offset of its 128-byte line: 8-pixel blocks, pixels 4..7 of a block pairwise swapped
__device__ __forceinline__ int patch_off(int L) {
    const int j = L & 7;
    return (L >> 3) * 1024 + ((j ^ ((j >> 2) & 1)) << 7);
}

__global__ __launch_bounds__(256, 1) void sep_gemm_kernel(const SepGemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nh = blockIdx.x & 1;                 // N half
    int tile = blockIdx.x >> 1;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty_ = tile % p.tiles_y;
    const int img = tile / p.tiles_y;
    const int x0 = tx * TCOLS, y0 = ty_ * TROWS;
    const long img_pix = (long)img * p.H * p.W;
    const int nk = (p.Cin + 31) / 32;

    // ---- DMA sources.  Patch: instruction t = wv * 7 + q covers patch pixels [8 t, 8 t + 8); lane l fills LDS slot l >> 3,
    // chunk l & 7, which holds pixel 8 t + sigma(l >> 3) (sigma swaps 4<->5, 6<->7: see patch_off)
    const unsigned char* psrc[7];
    bool pin[7];
    const int pchunk = lane & 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int t = wv * 7 + q;
        const int s = lane >> 3;
        const int L = t * 8 + (s ^ ((s >> 2) & 1));
        bool ok = t < P_INSTR && L < PPIX;
        const int pr = L / PCOLS, pc = L - pr * PCOLS;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        ok = ok && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        pin[q] = ok;
        psrc[q] = reinterpret_cast<const unsigned char*>(p.x + (img_pix + (long)(ok ? gy : 0) * p.W + (ok ? gx : 0)) * p.ldx) + pchunk * 16;
    }
    // W slice: instruction u = wv * 6 + q covers rows [16 u, 16 u + 16) of this N half; lane l fills row 16 u + (l >> 2),
    // physical chunk l & 3 = logical chunk (l & 3) ^ ((row >> 2) & 3); logical chunks 0,1: hi plane k 0-7 / 8-15, 2,3: lo plane
    const unsigned char* wsrc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int row = (wv * 6 + q) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        const int n = nh * NH + row;
        const uint16_t* plane = (c & 2) ? p.Wlo : p.Whi;
        wsrc[q] = n < p.Npad ? reinterpret_cast<const unsigned char*>(plane + (long)n * p.Ktot + (c & 1) * 8) : nullptr;
    }
    // depthwise weights of a step: 9 taps x 32 channels = 72 chunks of 16 B: wave 0, lanes 0..63 and wave 1, lanes 0..7
    const int wk_idx = wv * 64 + lane;             // chunk index when wv < 2
    const bool wk_on = wk_idx < 72;
    const int wk_tap = wk_idx >> 3, wk_c = wk_idx & 7;

    auto issue = [&](int k, int bslot) {
        const int cbase = k * 32;
        const bool chan_ok = cbase + pchunk * 4 < p.Cin;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            if (wv * 7 + q < P_INSTR) {   // wave-uniform
                const unsigned char* s = (pin[q] && chan_ok) ? psrc[q] + (long)cbase * 4 : g_sg_zero + pchunk * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + (wv * 7 + q) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const unsigned char* s = wsrc[q] ? wsrc[q] + ((long)cbase + ks * 16) * 2 : g_sg_zero + (lane & 3) * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + OFF_B + (bslot * 2 + ks) * B16_BYTES + (wv * 6 + q) * 1024), 16, 0, 0);
            }
        if (wv < 2) {   // wave-uniform
            const bool ok = wk_on && cbase + wk_c * 4 < p.Cin;
            const unsigned char* s = ok ? reinterpret_cast<const unsigned char*>(p.dw + (long)wk_tap * p.Cin + cbase + wk_c * 4) : g_sg_zero;
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + OFF_WK + wv * 1024), 16, 0, 0);
        }
    };

    // ---- depthwise role: lane = 4 consecutive pixels of tile row dty, 4 channels (chunk c4 of the step)
    const int c4 = tid & 7, pg = tid >> 3;
    const int dty = pg >> 3, dtx0 = (pg & 7) * 4;
    int poff[3][6];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int d = 0; d < 6; ++d) poff[i][d] = patch_off((dty + i) * PCOLS + dtx0 + d) + c4 * 16;
    // A slice destination of the lane's 4 pixels: row r, hi chunk = (c4 >> 2) * 4 + ((c4 >> 1) & 1), lo chunk = hi + 2, 8 bytes at (c4 & 1) * 8
    const int a_lc = (c4 >> 2) * 4 + ((c4 >> 1) & 1);

    // ---- MFMA role: wave wv owns output columns [96 wv, 96 wv + 96) of the N half, all 128 rows
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    if (p.stamps) tprev = __builtin_amdgcn_s_memtime();
#define SG_STAMP(i) if (p.stamps) { const long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tprev; tprev = t_; }
    issue(0, 0);
    for (int k = 0; k < nk; ++k) {
        const int bslot = k & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SG_STAMP(0)                     // own DMA landed
        __builtin_amdgcn_s_barrier();   // (a) patch, depthwise weights and W slice of step k have landed; every wave is done with step k-1
        SG_STAMP(1)

        // depthwise 3x3 of the step's 32 channels -> bf16 hi / lo A slice
        {
            const float* wks = reinterpret_cast<const float*>(smem + OFF_WK);
            f32x4 wk[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const f32x4*>(wks + t * 32 + c4 * 4);
            f32x4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 pr[6];
#pragma unroll
                for (int d = 0; d < 6; ++d) pr[d] = *reinterpret_cast<const f32x4*>(smem + poff[i][d]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int d = 0; d < 3; ++d) o[j] += wk[i * 3 + d] * pr[j + d];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = dty * TCOLS + dtx0 + j;
                unsigned h0, l0, h1, l1;
                split2(o[j][0], o[j][1], h0, l0);
                split2(o[j][2], o[j][3], h1, l1);
                const int sw = (r >> 1) & 7;
                unsigned char* row = smem + OFF_A + r * 128 + (c4 & 1) * 8;
                *reinterpret_cast<u32x2*>(row + ((a_lc ^ sw) << 4)) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(row + (((a_lc + 2) ^ sw) << 4)) = u32x2{l0, l1};
            }
        }
        SG_STAMP(2)                     // depthwise stage
        __builtin_amdgcn_s_barrier();   // (b) A slice visible; patch and depthwise weights are free
        SG_STAMP(3)
        // fragments of BOTH 16-channel sub-steps are requested up front (one wave per SIMD: nothing else hides an LDS read's
        // latency; hipcc otherwise waits for every fragment right where it is first used, a dozen times per step), the next step's
        // DMA is issued between them
        bf16x8 ah[2][4], al[2][4], bh[2][3], bl[2][3];
        auto load_frags = [&](int ks) {
            const unsigned char* bb = smem + OFF_B + (bslot * 2 + ks) * B16_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 32 + fr;
                const int sw = (r >> 1) & 7;
                const unsigned char* row = smem + OFF_A + r * 128;
                ah[ks][i] = *reinterpret_cast<const bf16x8*>(row + (((ks * 4 + fh) ^ sw) << 4));
                al[ks][i] = *reinterpret_cast<const bf16x8*>(row + (((ks * 4 + 2 + fh) ^ sw) << 4));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = wv * 96 + j * 32 + fr;
                const int sw = (n >> 2) & 3;
                const unsigned char* row = bb + n * 64;
                bh[ks][j] = *reinterpret_cast<const bf16x8*>(row + ((fh ^ sw) << 4));
                bl[ks][j] = *reinterpret_cast<const bf16x8*>(row + (((2 + fh) ^ sw) << 4));
            }
        };
        load_frags(0);
        issue(k + 1 < nk ? k + 1 : k, bslot ^ 1);   // past the end: the last step again, into the slot nobody reads (no branch here)
        load_frags(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {   // small terms first (as gemm_conv.hip / gemm_split.hip)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                }
        // keep the issue order: 28 fragment reads (with the DMA pieces between the two sets), then the 72 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 20, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 72, 0);
        if (p.stamps) {   // the MFMA results are needed for the stamp to mean "MFMAs done": touch one accumulator
            asm volatile("" ::"v"(acc[3][2][15]));
        }
        SG_STAMP(4)
    }

    if (p.stamps && tid == 0) {
        long long* o = p.stamps + (long)blockIdx.x * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = ph[i];
        o[6] = __builtin_amdgcn_s_memtime();
    }
#undef SG_STAMP
    // ---- epilogue straight from the C/D layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();
    const bool two = p.scale2 != nullptr;
    const long pix0 = img_pix + (long)y0 * p.W + x0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = nh * NH + wv * 96 + j * 32 + fr;
        if (n >= p.N) continue;
        const float s1 = p.scale1[n], t1 = p.shift1[n];
        const float s2 = two ? p.scale2[n] : 1.f, t2 = two ? p.shift2[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float rv[16];
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    rv[e] = p.res[(pix0 + (long)(r >> 5) * p.W + (r & 31)) * p.ldres + n];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                float u = fmaf(acc[i][j][e], s1, t1);
                u = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                if (two) u = fminf(fmaxf(fmaf(u, s2, t2), 0.f), hi2);
                if (p.res) u += rv[e];
                p.y[(pix0 + (long)(r >> 5) * p.W + (r & 31)) * p.ldy + n] = u;
            }
        }
    }
}

long long* g_sg_stamps = nullptr;

}  // namespace

// dev hook (include/emdenoise_dev.h): device buffer (8 x int64 per workgroup) for per-phase s_memtime sums, NULL = off
extern "C" void emd_debug_sepgemm_stamps(void* buf) { g_sg_stamps = static_cast<long long*>(buf); }

extern "C" int emd_sep3x3_gemm_supported(int H, int W, int Cin, int Cout) {
    // matrix-core bound separable convs: stride 1, rate 1, whole 4 x 32 pixel tiles, the K loop long enough to amortise a tile
    return H % TROWS == 0 && W % TCOLS == 0 && Cin % 4 == 0 && Cin >= 256 && Cin <= 4096 && Cout % 4 == 0 && Cout > NH && Cout <= 2 * NH;
}

extern "C" int emd_sep3x3_gemm_f32(const float* x, int ldx, const float* dw, const uint16_t* whi, const uint16_t* wlo,
                                   const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                   const float* res, int ldres, float* y, int ldy, int B, int H, int W, int Cin, int Cout, int act,
                                   emd_stream_t stream) {
    EMD_REQUIRE(x && dw && whi && wlo && scale1 && shift1 && y, EMD_E_INVALID, "emd_sep3x3_gemm_f32: null pointer");
    EMD_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), EMD_E_INVALID, "emd_sep3x3_gemm_f32: scale2/shift2 pair");
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_sep3x3_gemm_f32: bad shape");
    EMD_REQUIRE(emd_sep3x3_gemm_supported(H, W, Cin, Cout), EMD_E_UNSUPPORTED,
                "emd_sep3x3_gemm_f32: needs H%4==0, W%32==0, 256 <= Cin <= 4096, 384 < Cout <= 768, both multiples of 4 (use emd_dw3x3_split32_f32 + emd_conv1x1_split32_f32)");
    EMD_REQUIRE(ldx % 4 == 0 && ldx >= Cin && ldy >= Cout && (!res || ldres >= Cout), EMD_E_ALIGN,
                "emd_sep3x3_gemm_f32: ldx a multiple of 4 and >= Cin; ldy, ldres >= Cout");
    EMD_REQUIRE(emd::aligned16(x) && emd::aligned16(dw) && emd::aligned16(whi) && emd::aligned16(wlo), EMD_E_ALIGN,
                "emd_sep3x3_gemm_f32: x, dw and the weight planes must be 16-byte aligned");
    EMD_REQUIRE(y != x && (!res || res != y), EMD_E_INVALID, "emd_sep3x3_gemm_f32: the output aliases an input");
    if (B == 0) return EMD_OK;
    SepGemmParams p{};
    p.x = x; p.dw = dw; p.Whi = whi; p.Wlo = wlo; p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.res = res; p.y = y; p.H = H; p.W = W; p.Cin = Cin; p.N = Cout;
    p.Npad = (Cout + kNPadTo - 1) / kNPadTo * kNPadTo;
    p.Ktot = (Cin + kBK - 1) / kBK * kBK;
    p.ldx = ldx; p.ldy = ldy; p.ldres = ldres; p.act = act;
    p.tiles_x = W / TCOLS; p.tiles_y = H / TROWS; p.stamps = g_sg_stamps;
    const long tiles = (long)B * p.tiles_x * p.tiles_y;
    EMD_REQUIRE(tiles * 2 <= 0x7fffffffL, EMD_E_UNSUPPORTED, "emd_sep3x3_gemm_f32: grid too large");
    hipLaunchKernelGGL(sep_gemm_kernel, dim3((unsigned)(tiles * 2)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return emd::check_launch("sep_gemm_kernel");
}
