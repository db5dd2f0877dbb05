// Pointwise 1x1 convolution on the matrix cores from PRE-SPLIT activations ("split32" tensors), both operands
// brought into LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR staging, no in-kernel split, no ds_write.
//
// replaces (TensorFlow ops called by machine_learning/denoiser.py): the pointwise half of
//   slim.separable_convolution2d (:113-131) + its normalizer BN (:123) + batch_then_activ (:134) + the residual
//   "+=" that follows (:246, :322) -- for the layers whose depthwise half runs as its own launch (the 728-channel
//   middle flow, the strided and dilated blocks): emd_dw3x3_split32_f32 writes the depthwise result already split,
//   emd_conv1x1_split32_f32 consumes it.  Same arithmetic as emd_conv1x1_f32 (gemm_conv.hip), bit for bit:
//   a = hi + lo (round-to-nearest bf16 twice), acc += Alo*Whi + Ahi*Wlo + Ahi*Whi in fp32 on
//   v_mfma_f32_32x32x16_bf16, K ascending.
//
// split32 layout of an activation tensor [npix][C]: pixel pitch ld 4-byte units (ld % 32 == 0, ld >= ceil32(C));
//   inside a pixel, channel group g = c/32 occupies bytes [128g, 128g+128): 32 bf16 "hi" then 32 bf16 "lo".
//   A tensor therefore has exactly the size and pitch of its fp32 NHWC twin, one K step (32 channels) of one pixel is
//   one 128-byte line, and channels C..ceil32(C) are zero.
//
// Block = 512 threads = 8 waves (4 along M x 2 along N, 64x64 each as 2x2 MFMA tiles of 32x32); block tile 256 x 128,
// K step 32.  LDS: two stages of (256 + 128) rows x 128 B (hi | lo of one K step), filled by LDS-DMA one K step ahead
// (one barrier per K step); rows are XOR-swizzled by 16-byte chunk, chunk' = chunk ^ ((row >> 1) & 7), applied on the
// DMA's SOURCE address (the DMA writes LDS linearly: wave base + lane*16) and on the ds_read_b128 fragment address,
// which makes every 16-lane group of a fragment read cover all 64 banks once.  Against the 128x128 register-staged
// kernel this moves 25 % fewer L2 bytes per flop and removes ~250 VALU + ~24 ds_write per wave and K step.
#include "mfma_common.hpp"

namespace {

using namespace emd;

struct SplitGemmParams {
    const unsigned char* A;   // split32 activations
    const uint16_t* Whi;      // [Npad][Ktot] (emd_pack_weights_bf16, taps = 1: Ktot = Cin padded to 64)
    const uint16_t* Wlo;
    float* C;
    const float* res;
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    long M;
    long lda_bytes;           // pixel pitch of A in bytes
    int N, Cin, Ktot;
    int ldc, ldres, act;
    int n_mtiles, n_ntiles;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SBM = 256, SBN = 128, SBK = 32;
constexpr int A_STAGE = SBM * 128, W_STAGE = SBN * 128, STAGE = A_STAGE + W_STAGE;   // 48 KB
constexpr int EPI_LD = SBN + 4;                                                       // fp32 staging row (floats)
constexpr int EPI_BYTES = SBM * EPI_LD * 4;                                           // 132 KB
constexpr int SMEM_BYTES = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;

__global__ __launch_bounds__(512, 2) void gemm_split_kernel(const SplitGemmParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;           // 0..7
    const int wm = wv >> 1, wn = wv & 1;

    // XCD-aware tile mapping (bijective for any grid size): the N-tiles of one M-tile are neighbours in one XCD
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * SBM;
    const int n0 = nt * SBN;

    // ---- LDS-DMA source addresses.  One wave instruction = 64 lanes x 16 B = 8 rows x 128 B, written linearly.
    // Lane l fills physical chunk (l & 7) of row (l >> 3); that chunk holds logical chunk (l & 7) ^ ((row >> 1) & 7).
    // A: wave wv fills rows [32 wv, 32 wv + 32) in 4 pieces; W: rows [16 wv, 16 wv + 16) in 2 pieces.
    const int drow = lane >> 3, dchunk = lane & 7;
    const unsigned char* asrc[4];
    const unsigned char* wsrc[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;                       // M tail: re-read the last pixel, never stored
        asrc[q] = p.A + m * p.lda_bytes + c * 16;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wv * 16 + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        const uint16_t* plane = (c & 4) ? p.Wlo : p.Whi;  // logical chunks 0-3: hi, 4-7: lo
        wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * p.Ktot + (c & 3) * 8);
    }

    auto issue = [&](int stage, int kt) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kt * 128), (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + (long)kt * 64),
                                             (lptr_t)(sb + A_STAGE + (wv * 16 + q * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing: lane (fr = lane & 31, fh = lane >> 5) reads row fr of a 32-row tile, logical chunk
    // plane*4 + ks*2 + fh; tile bases are multiples of 16 rows, so the swizzle term depends on fr only
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int a_off = (wm * 64 + fr) * 128;              // + i*32*128
    const int w_off = A_STAGE + (wn * 64 + fr) * 128;    // + j*32*128

    const int nk = (p.Cin + SBK - 1) / SBK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed (every wave waits for its own DMA pieces, then the barrier), and every wave has
        // finished reading the other stage (its MFMAs of step kt-1 consumed those reads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
        const unsigned char* sb = smem + (kt & 1) * STAGE;
        const int kvalid = p.Cin - kt * SBK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks * 16 >= kvalid) break;                // all-padding half step (block-uniform)
            const int ch = ((ks * 2 + fh) ^ sw) << 4;    // hi plane chunk; lo = same with bit 2 flipped
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + ch);
                al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + (ch ^ 64));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + ch);
                bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + (ch ^ 64));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {           // small terms first (as gemm_conv.hip)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    __syncthreads();  // all fragment reads done before the staging tile overlays the stages

    // ---- epilogue (as gemm_conv.hip): accumulators -> fp32 LDS tile -> per-channel affine(s), activation, residual,
    // 16-byte loads and stores along the channel axis.  C/D layout of mfma_32x32: col = lane&31,
    // row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    float(*stage)[EPI_LD] = reinterpret_cast<float(*)[EPI_LD]>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                stage[r][wn * 64 + j * 32 + fr] = acc[i][j][e];
            }
    __syncthreads();
    constexpr int C4 = SBN / 4;             // 32 float4 chunks per staged row
    constexpr int ROWS_PER_PASS = 512 / C4; // 16
    const int ec = (tid % C4) * 4, er = tid / C4;
    const int n = n0 + ec;
    if (n < p.N) {                          // N % 4 == 0: a chunk is all inside or all outside
        const float4 s1 = *reinterpret_cast<const float4*>(p.scale1 + n);
        const float4 t1 = *reinterpret_cast<const float4*>(p.shift1 + n);
        float4 s2 = make_float4(1.f, 1.f, 1.f, 1.f), t2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale2) {
            s2 = *reinterpret_cast<const float4*>(p.scale2 + n);
            t2 = *reinterpret_cast<const float4*>(p.shift2 + n);
        }
        const float* __restrict__ resp = p.res;
        float* __restrict__ outp = p.C;
        const float hi = p.act == 2 ? __builtin_inff() : 6.f;
#pragma unroll 4
        for (int r = er; r < SBM; r += ROWS_PER_PASS) {
            const long pix = m0 + r;
            if (pix >= p.M) break;
            float4 v = *reinterpret_cast<const float4*>(&stage[r][ec]);
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (resp) rv = *reinterpret_cast<const float4*>(resp + pix * p.ldres + n);
            v.x = fmaf(v.x, s1.x, t1.x); v.y = fmaf(v.y, s1.y, t1.y); v.z = fmaf(v.z, s1.z, t1.z); v.w = fmaf(v.w, s1.w, t1.w);
            if (p.act == 4) {  // tf.nn.leaky_relu, alpha 0.2 (graph G)
                v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y;
                v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
            } else if (p.act) {  // hi = 6 (relu6) or +inf (relu)
                v.x = fminf(fmaxf(v.x, 0.f), hi); v.y = fminf(fmaxf(v.y, 0.f), hi);
                v.z = fminf(fmaxf(v.z, 0.f), hi); v.w = fminf(fmaxf(v.w, 0.f), hi);
            }
            if (p.scale2) {
                v.x = fminf(fmaxf(fmaf(v.x, s2.x, t2.x), 0.f), hi); v.y = fminf(fmaxf(fmaf(v.y, s2.y, t2.y), 0.f), hi);
                v.z = fminf(fmaxf(fmaf(v.z, s2.z, t2.z), 0.f), hi); v.w = fminf(fmaxf(fmaf(v.w, s2.w, t2.w), 0.f), hi);
            }
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            *reinterpret_cast<float4*>(outp + pix * p.ldc + n) = v;
        }
    }
}

// fp32 [npix][C] (pitch ldx floats) -> split32 (pitch ldy 4-byte units): one thread per (pixel, 4 channels),
// channels C..ceil32(C) written as zero
__global__ __launch_bounds__(256) void to_split32_kernel(const float* __restrict__ x, int ldx, unsigned char* __restrict__ y,
                                                         int ldy, long npix, int C4, int C4p) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= npix * C4p) return;
    const int c4 = (int)(tid % C4p);
    const long pix = tid / C4p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) v = *reinterpret_cast<const f32x4*>(x + pix * ldx + c4 * 4);
    unsigned h0, l0, h1, l1;
    split2(v[0], v[1], h0, l0);
    split2(v[2], v[3], h1, l1);
    unsigned char* o = y + pix * (long)ldy * 4 + (c4 >> 3) * 128 + (c4 & 7) * 8;
    *reinterpret_cast<u32x2*>(o) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(o + 64) = u32x2{l0, l1};
}

}  // namespace

extern "C" int emd_split32_ld(int C) { return C < 1 ? 0 : (C + 31) / 32 * 32; }

extern "C" int emd_to_split32_f32(const float* x, int ldx, void* y, int ldy, long npix, int C, emd_stream_t stream) {
    EMD_REQUIRE(x && y, EMD_E_INVALID, "emd_to_split32_f32: null pointer");
    EMD_REQUIRE(npix >= 0 && C >= 4, EMD_E_INVALID, "emd_to_split32_f32: bad shape");
    EMD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldx >= C && ldy % 32 == 0 && ldy >= emd_split32_ld(C), EMD_E_ALIGN,
                "emd_to_split32_f32: C, ldx multiples of 4; ldy a multiple of 32, >= ceil32(C)");
    EMD_REQUIRE(emd::aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 127u) == 0, EMD_E_ALIGN,
                "emd_to_split32_f32: x 16-byte, y 128-byte aligned");
    if (npix == 0) return EMD_OK;
    const int C4 = C / 4, C4p = emd_split32_ld(C) / 4;
    const long nb = (npix * C4p + 255) / 256;
    if (nb > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "emd_to_split32_f32: grid too large");
    hipLaunchKernelGGL(to_split32_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                       static_cast<unsigned char*>(y), ldy, npix, C4, C4p);
    return emd::check_launch("to_split32_kernel");
}

extern "C" int emd_conv1x1_split32_supported(long M, int Cin, int Cout) {
    // worth it where the GEMM is matrix-core bound and the grid fills the chip with 256 x 128 tiles
    if (Cin < 128 || Cout < 128 || Cout % 4) return 0;
    const long tiles = ((M + SBM - 1) / SBM) * ((Cout + SBN - 1) / SBN);
    return tiles >= 256 ? 1 : 0;
}

extern "C" int emd_conv1x1_split32_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo,
                                       const float* scale1, const float* shift1, const float* scale2,
                                       const float* shift2, const float* res, int ldres, float* y, int ldy, long M,
                                       int Cin, int Cout, int act, emd_stream_t stream) {
    EMD_REQUIRE(xs && whi && wlo && scale1 && shift1 && y, EMD_E_INVALID, "emd_conv1x1_split32_f32: null pointer");
    EMD_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), EMD_E_INVALID, "emd_conv1x1_split32_f32: scale2/shift2 must come together");
    EMD_REQUIRE(M >= 0 && Cin >= 1 && Cout >= 4, EMD_E_INVALID, "emd_conv1x1_split32_f32: bad shape");
    EMD_REQUIRE(ldx % 32 == 0 && ldx >= emd_split32_ld(Cin) && (reinterpret_cast<uintptr_t>(xs) & 127u) == 0, EMD_E_ALIGN,
                "emd_conv1x1_split32_f32: xs 128-byte aligned, ldx a multiple of 32, >= ceil32(Cin)");
    EMD_REQUIRE(ldy >= Cout && (!res || ldres >= Cout), EMD_E_INVALID, "emd_conv1x1_split32_f32: ldy/ldres smaller than Cout");
    EMD_REQUIRE(Cout % 4 == 0 && ldy % 4 == 0 && emd::aligned16(y) && (!res || (ldres % 4 == 0 && emd::aligned16(res))),
                EMD_E_ALIGN, "emd_conv1x1_split32_f32: Cout, ldy, ldres multiples of 4; y, res 16-byte aligned");
    EMD_REQUIRE(emd::aligned16(scale1) && emd::aligned16(shift1) && (!scale2 || (emd::aligned16(scale2) && emd::aligned16(shift2))) &&
                    emd::aligned16(whi) && emd::aligned16(wlo),
                EMD_E_ALIGN, "emd_conv1x1_split32_f32: weight planes and scale/shift vectors must be 16-byte aligned");
    if (M == 0) return EMD_OK;
    SplitGemmParams p{};
    p.A = static_cast<const unsigned char*>(xs); p.Whi = whi; p.Wlo = wlo; p.C = y; p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.M = M; p.lda_bytes = (long)ldx * 4; p.N = Cout; p.Cin = Cin; p.Ktot = (Cin + kBK - 1) / kBK * kBK;
    p.ldc = ldy; p.ldres = ldres; p.act = act;
    p.n_mtiles = (int)((M + SBM - 1) / SBM);
    p.n_ntiles = (Cout + SBN - 1) / SBN;
    const long nblk = (long)p.n_mtiles * p.n_ntiles;
    if (nblk > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "emd_conv1x1_split32_f32: grid too large");
    hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)nblk), dim3(512), 0, static_cast<hipStream_t>(stream), p);
    return emd::check_launch("gemm_split_kernel");
}
