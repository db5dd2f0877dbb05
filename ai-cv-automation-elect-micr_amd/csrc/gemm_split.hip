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
#include <cstdlib>

#include "mfma_common.hpp"
#include "conv3_params.hpp"

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
    double* stats_part;       // optional [n_mtiles][2][N]: per-channel sum / sum of squares of the STORED values of each M tile
    unsigned wlo_delta;       // persistent kernel: byte distance Wlo - Whi (one allocation)
    long long* stamps;        // dev builds only: 5 s_memtime stamps per workgroup (NULL otherwise)
    int out_split;            // pointwise kernel: C is a split32 tensor (pitch ldc 4-byte units), for a following split32 GEMM
    int nt;                   // non-temporal output stores: the output is not re-read by this launch, L2 is kept for the operands
};

// Non-temporal output stores are the default (graph D: 26.0 -> 25.5 ms, PMC fetch of the transposed convs 5.97 -> 3.83 GB per launch:
// the outputs no longer push the re-read input rows out of L2).  The dev knob nt_mask masks them: bit 0 = the implicit-GEMM convolutions here,
// bit 2 = the pointwise GEMM (bit 1: sep_fused.hip).
inline int split_nt(int bit) { return (emd::g_knobs.nt_mask >> bit) & 1; }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SBN = 128, SBK = 32;

// BM = 256: 8 waves, one workgroup per CU; BM = 128: 4 waves, two workgroups per CU.  NS = LDS stages (DMA runs NS-1 K steps ahead).
// WREG (with PIPE): the W tile travels global -> registers -> ds_write_b128 instead of by LDS-DMA, one K step earlier than the A tile's
// DMA.  Why: a K step moves 32 KB of A + 16 KB of W into LDS in 1850 cycles = 26 B/clk/CU, and LDS-DMA from L2 tops out at about 30
// B/clk/CU (MI355X_MICROARCH.md, gather into LDS: 66-73 GB/s per CU) -- the K loop runs at the DMA rate, not at the MFMA rate
// (1536 cycles).  With the W third of the bytes on the vector-load path the DMA carries 32 KB per step.  Same LDS image, same bits.
// DIRECT: the MFMA operands trade places (D^T = W x A^T: the same products in the same K order), so a lane ends up with four
// CONSECUTIVE channels of one pixel in each accumulator quad (pixel = lane & 31, channels 8g + 4(lane >> 5) .. +3) and the epilogue
// leaves straight from the registers in 16-byte stores: no staging tile, no barrier, and a wave that has issued its 16 stores is done --
// the workgroup's LDS and registers go to the next one while the stores drain.
template <int BM, int NS, bool PIPE = false, bool WREG = false, bool DIRECT = false>
__global__ __launch_bounds__(BM * 2, 2) void gemm_split_kernel(const SplitGemmParams p) {
    constexpr int NW = BM / 32;                                                       // waves
    constexpr int NT = NW * 64;
    constexpr int A_STAGE = BM * 128, W_STAGE = SBN * 128, STAGE = A_STAGE + W_STAGE;
    constexpr int EPI_LD = SBN + 4;                                                   // fp32 staging row (floats)
    constexpr int EPI_BYTES = BM * EPI_LD * 4;
    constexpr int SMEM_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
    constexpr int WQ = SBN / NW / 8;                                                  // W pieces (8 rows) per wave
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;

    // XCD-aware tile mapping (bijective for any grid size): the N-tiles of one M-tile are neighbours in one XCD
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * SBN;

    // ---- LDS-DMA source addresses.  One wave instruction = 64 lanes x 16 B = 8 rows x 128 B, written linearly.
    // Lane l fills physical chunk (l & 7) of row (l >> 3); that chunk holds logical chunk (l & 7) ^ ((row >> 1) & 7).
    // A: wave wv fills rows [32 wv, 32 wv + 32) in 4 pieces; W: rows [16 wv, 16 wv + 16) in 2 pieces.
    const int drow = lane >> 3, dchunk = lane & 7;
    const unsigned char* asrc[4];
    const unsigned char* wsrc[WQ];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;                       // M tail: re-read the last pixel, never stored
        asrc[q] = p.A + m * p.lda_bytes + c * 16;
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int row = wv * (WQ * 8) + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        const uint16_t* plane = (c & 4) ? p.Wlo : p.Whi;  // logical chunks 0-3: hi, 4-7: lo
        wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * p.Ktot + (c & 3) * 8);
    }

    auto issue = [&](int stage, int kt) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kt * 128), (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
        if constexpr (!WREG) {
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + (long)kt * 64),
                                                 (lptr_t)(sb + A_STAGE + (wv * (WQ * 8) + q * 8) * 128), 16, 0, 0);
        }
    };
    // WREG: this lane's WQ pieces of a W tile, global -> registers / registers -> the LDS image the DMA would have written
    u32x4 wreg[WQ];
    auto w_load = [&](int kt) {
#pragma unroll
        for (int q = 0; q < WQ; ++q) wreg[q] = *reinterpret_cast<const u32x4*>(wsrc[q] + (long)kt * 64);
    };
    auto w_store = [&](int stage) {
        unsigned char* sb = smem + stage * STAGE + A_STAGE;
#pragma unroll
        for (int q = 0; q < WQ; ++q) *reinterpret_cast<u32x4*>(sb + (wv * (WQ * 8) + q * 8) * 128 + lane * 16) = wreg[q];
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
    long long t0 = 0, t1 = 0, r0 = 0;
    if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    struct Frags { bf16x8 ah[2], al[2], bh[2], bl[2]; };
    auto load_frags = [&](Frags& f, const unsigned char* sb, int ks) {
        const int ch = ((ks * 2 + fh) ^ sw) << 4;    // hi plane chunk; lo = same with bit 2 flipped
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + ch);
            f.al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + (ch ^ 64));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + ch);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + (ch ^ 64));
        }
    };
    auto mfma12 = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {           // small terms first (as gemm_conv.hip)
                if constexpr (DIRECT) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
                } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
                }
            }
    };
    if constexpr (PIPE) {
        // Three stages; the barrier at the top of step kt certifies tile kt+1 (DMA issued one step earlier), so the
        // fragments of a tile's first half step are read during the previous tile's last MFMAs: no ds_read latency
        // is exposed at the barrier.  Fragment sets f0 / f1 alternate between the two half steps.
        static_assert(NS == 3, "the pipelined loop needs three stages");
        issue(0, 0);
        issue(1, 1 < nk ? 1 : nk - 1);
        if constexpr (WREG) {   // W tiles 0 and 1 into their stages, tile 2 into the registers
            w_load(0);
            w_store(0);
            w_load(1 < nk ? 1 : nk - 1);
            w_store(1);
            w_load(2 < nk ? 2 : nk - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p.stamps) t1 = __builtin_amdgcn_s_memtime();
        Frags f0, f1;
        load_frags(f0, smem, 0);
        int s0 = 0, s1 = 1, s2 = 2;   // stage of tile kt / kt+1 / the one being refilled (held tile kt-1)
        for (int kt = 0; kt < nk; ++kt) {
            if (kt > 0) {
                if constexpr (WREG) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // + the previous step's W stores
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // WREG: the registers hold W tile kt+2 (loaded a step ago, waited for above): into the stage the A DMA below fills.
            // The stores go BEFORE the DMA pieces: hipcc puts vmcnt(0) in front of a ds_write that follows LDS-DMA.
            if constexpr (WREG) w_store(s2);
            {
                const int nx = kt + 2;
                issue(s2, nx < nk ? nx : nk - 1);   // beyond the end: re-read the last tile into a stage nobody computes on
            }
            if constexpr (WREG) {
                const int nx = kt + 3;
                w_load(nx < nk ? nx : nk - 1);
            }
            load_frags(f1, smem + s0 * STAGE, 1);
            mfma12(f0);
            load_frags(f0, smem + s1 * STAGE, 0);
            mfma12(f1);
            // issue order of the step (hipcc otherwise sinks every fragment read next to its first use and waits for it
            // there).  The DMA pieces write LDS, so the compiler keeps every ds_read of the step behind them: first half step
            // = 12 MFMAs on f0 with the 6 DMA pieces, then the 8 reads of f1, between them; second half step = 12 MFMAs
            // on f1 with the 8 reads of the next tile's f0 between them
            if constexpr (WREG) {   // the 2 W stores, 4 DMA pieces, the 2 W loads, then the reads of f1
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);       // DS write (W tile kt+2)
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM (4 LDS-DMA pieces, then W tile kt+3 into registers)
                }
            } else {
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM (LDS-DMA piece)
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            const int t = s0; s0 = s1; s1 = s2; s2 = t;
        }
    } else {
    // prologue: NS-1 tiles in flight.  One DMA "group" = the 4 + WQ pieces a wave issues per tile; a tile that does not
    // exist is still issued (re-reading the last one into a stage nobody reads) so that the counted waits stay uniform
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s < nk ? s : nk - 1);
    int st_c = 0, st_l = NS - 1;   // stage being computed / stage being loaded
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed: every wave waits until only its NS-2 youngest groups are outstanding, then the barrier;
        // and every wave has finished reading the stage that is about to be refilled (its MFMAs of step kt-1 consumed them)
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (4 + WQ)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (kt == 0 && p.stamps) t1 = __builtin_amdgcn_s_memtime();
        {
            const int nx = kt + NS - 1;
            issue(st_l, nx < nk ? nx : nk - 1);
        }
        const unsigned char* sb = smem + st_c * STAGE;
        const int kvalid = p.Cin - kt * SBK;
        Frags f;
        load_frags(f, sb, 0);
        mfma12(f);
        if (kvalid > 16) {                               // else: all-padding half step (block-uniform)
            load_frags(f, sb, 1);
            mfma12(f);
        }
        st_c = st_c + 1 == NS ? 0 : st_c + 1;
        st_l = st_l + 1 == NS ? 0 : st_l + 1;
    }
    }
    long long t2 = 0, t3 = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMA groups of the last steps
    if constexpr (DIRECT) {
        if (p.stamps) t2 = t3 = __builtin_amdgcn_s_memtime();
        const float hi = p.act == 1 ? 6.f : __builtin_inff();
        const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;
        const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();
        const bool two = p.scale2 != nullptr;
        const int Np = p.out_split ? (p.N + 31) / 32 * 32 : p.N;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + j * 32 + g * 8 + fh * 4;
                if (n >= Np) continue;
                const bool real = n < p.N;
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, t1v = s1, s2 = {1.f, 1.f, 1.f, 1.f}, t2v = s1;
                if (real) {
                    s1 = *reinterpret_cast<const f32x4*>(p.scale1 + n);
                    t1v = *reinterpret_cast<const f32x4*>(p.shift1 + n);
                    if (two) {
                        s2 = *reinterpret_cast<const f32x4*>(p.scale2 + n);
                        t2v = *reinterpret_cast<const f32x4*>(p.shift2 + n);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long pix = m0 + wm * 64 + i * 32 + fr;
                    if (pix >= p.M) continue;
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float u = fmaf(acc[i][j][4 * g + c], s1[c], t1v[c]);
                        u = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                        if (two) u = fminf(fmaxf(fmaf(u, s2[c], t2v[c]), 0.f), hi2);
                        v[c] = u;
                    }
                    if (p.res && real) v += *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n);
                    if (!p.out_split) {
                        if (p.nt) store_nt16(p.C + pix * p.ldc + n, v);
                        else *reinterpret_cast<f32x4*>(p.C + pix * p.ldc + n) = v;
                    } else {   // four channels = 8 bytes in the hi half of the 128-byte line, 8 in the lo half
                        if (!real) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        unsigned h0, l0, h1, l1;
                        split2(v[0], v[1], h0, l0);
                        split2(v[2], v[3], h1, l1);
                        unsigned char* gl = reinterpret_cast<unsigned char*>(p.C) + pix * (long)p.ldc * 4 + (n >> 5) * 128 + (n & 31) * 2;
                        *reinterpret_cast<u32x2*>(gl) = u32x2{h0, h1};
                        *reinterpret_cast<u32x2*>(gl + 64) = u32x2{l0, l1};
                    }
                }
            }
        if (p.stamps && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            long long* o = p.stamps + (long)blockIdx.x * 8;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = __builtin_amdgcn_s_memtime();
            o[5] = 0; o[6] = r0; o[7] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    __syncthreads();  // all fragment reads (and DMA writes) done before the staging tile overlays the stages
    if (p.stamps) t2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue (as gemm_conv.hip): accumulators -> fp32 LDS tile -> per-channel affine(s), activation, residual,
    // 16-byte loads and stores along the channel axis.  C/D layout of mfma_32x32: col = lane&31,
    // row = (e&3) + 8*(e>>2) + 4*(lane>>5).  The residual values of this thread's rows are requested BEFORE the
    // accumulators go through LDS: their latency hides behind the staging pass and its barrier (a residual epilogue took
    // 20-27 k cycles per tile with the loads inside the store loop, 7 k without a residual).
    float(*stage)[EPI_LD] = reinterpret_cast<float(*)[EPI_LD]>(smem);
    constexpr int C4 = SBN / 4;             // 32 float4 chunks per staged row
    constexpr int ROWS_PER_PASS = NT / C4;
    constexpr int NROWS = BM / ROWS_PER_PASS;   // 16 rows per thread
    const int ec = (tid % C4) * 4, er = tid / C4;
    const int n = n0 + ec;
    const bool real = n < p.N;              // N % 4 == 0: a chunk is all inside or all outside
    // split32 output: the padding channels up to a multiple of 32 are written too (zeros), and both lanes of a channel-quad pair
    // take the same branches (ceil32(N) is a multiple of 8: a pair is all inside or all outside)
    const bool ncol = p.out_split ? n < (p.N + 31) / 32 * 32 : real;
    f32x4 rv[NROWS];
    if (p.res) {
#pragma unroll
        for (int k = 0; k < NROWS; ++k) {
            const long pix = m0 + er + k * ROWS_PER_PASS;
            rv[k] = (real && pix < p.M) ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
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
    if (p.stamps) t3 = __builtin_amdgcn_s_memtime();
    double ssum[4] = {0.0, 0.0, 0.0, 0.0}, ssq[4] = {0.0, 0.0, 0.0, 0.0};
    if (ncol) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, t1 = s1;
        f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        if (real) {
            s1 = *reinterpret_cast<const f32x4*>(p.scale1 + n);
            t1 = *reinterpret_cast<const f32x4*>(p.shift1 + n);
            if (p.scale2) {
                s2 = *reinterpret_cast<const f32x4*>(p.scale2 + n);
                t2 = *reinterpret_cast<const f32x4*>(p.shift2 + n);
            }
        }
        float* __restrict__ outp = p.C;
        // the output stage: fp32 NHWC, or the split32 layout (16-byte stores through the pair exchange of emd::dw_store)
        auto put = [&](long pix, f32x4 v) {
            if (!p.out_split) {
                if (p.nt) store_nt16(outp + pix * p.ldc + n, v);
                else *reinterpret_cast<f32x4*>(outp + pix * p.ldc + n) = v;
                return;
            }
            if (!real) v = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned h0, l0, h1, l1;
            split2(v[0], v[1], h0, l0);
            split2(v[2], v[3], h1, l1);
            const int q = n >> 2;
            const bool odd = q & 1;
            const unsigned r0 = emd::swap_pair(odd ? h0 : l0), r1 = emd::swap_pair(odd ? h1 : l1);
            unsigned char* g = reinterpret_cast<unsigned char*>(outp) + pix * (long)p.ldc * 4 + (n >> 5) * 128;
            if (!odd) *reinterpret_cast<u32x4*>(g + (q & 7) * 8) = u32x4{h0, h1, r0, r1};
            else *reinterpret_cast<u32x4*>(g + 64 + ((q - 1) & 7) * 8) = u32x4{r0, r1, l0, l1};
        };
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
#pragma unroll
            for (int k = 0; k < NROWS; ++k) {
                const int r = er + k * ROWS_PER_PASS;
                const long pix = m0 + r;
                if (pix < p.M) put(pix, finish(*reinterpret_cast<const f32x4*>(&stage[r][ec])) + rv[k]);
            }
        } else if (p.stats_part) {
            // batch statistics of the output (misc_py/modified_Xception.py:302-323: the norm that follows runs on batch
            // statistics) gathered while the tile is in hand: per-thread double sums over its 16 rows here, 16 -> 1 below,
            // one partial per (M tile, channel) for the fixed-order final reduction (bn_stats_final): no second pass over y
#pragma unroll 4
            for (int r = er; r < BM; r += ROWS_PER_PASS) {
                const long pix = m0 + r;
                if (pix >= p.M) break;
                const f32x4 v = finish(*reinterpret_cast<const f32x4*>(&stage[r][ec]));
                put(pix, v);
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
                const long pix = m0 + r;
                if (pix >= p.M) break;
                put(pix, finish(*reinterpret_cast<const f32x4*>(&stage[r][ec])));
            }
        }
    }
    if (p.stats_part) {   // block-uniform
        __syncthreads();  // the staging tile has been read out
        double(*red)[SBN][2] = reinterpret_cast<double(*)[SBN][2]>(smem);   // [row groups][128 channels][sum, sum of squares]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[er][ec + c][0] = ssum[c];
            red[er][ec + c][1] = ssq[c];
        }
        __syncthreads();
        if (tid < 2 * SBN) {
            const int which = tid / SBN, col = tid % SBN;
            if (n0 + col < p.N) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < ROWS_PER_PASS; ++k) t += red[k][col][which];
                p.stats_part[((long)mt * 2 + which) * p.N + n0 + col] = t;
            }
        }
    }
    if (p.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* o = p.stamps + (long)blockIdx.x * 8;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = __builtin_amdgcn_s_memtime();
        o[5] = ((long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) << 32) |   // HW_REG_XCC_ID
               (unsigned)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);              // HW_REG_HW_ID
        o[6] = r0; o[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// The pipelined kernel on v_mfma_f32_16x16x32_bf16 (dev variant 5): a lane's 16-byte fragment read covers a whole 32-wide K
// step of one row (row = lane & 15, chunk = lane >> 4), so a K step is 16 fragment reads + 48 MFMAs of 16 cycles instead of
// 2 x (8 + 12 x 32 cycles); MI355X_MICROARCH.md reports the 16x16x32 shape holding a ~1.15x higher clock under load.
// Two whole fragment sets alternate between K steps.  The hardware sums a K step in a different order than 32x32x16 does,
// so results are not bit-identical to emd_conv1x1_f32 (same error class, checked against the oracle).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
// BM = 256: 8 waves (4 x 2 of 64 x 64), the chip-filling form; BM = 128: 4 waves (2 x 2), one workgroup per CU, for the small batches
// whose 256-row tiles would leave most CUs idle (M = 4096: 96 workgroups of 256 rows, 192 of 128).  Same products in the same order:
// the two forms are bit-identical, so a result does not depend on the batch size that selected one of them.
// BN = 64 (with BM = 128): 128 x 64 tiles, two workgroups per CU -- the batch-of-4 shapes (M = 4096) then run as 384 workgroups instead of 192.
// LEAD2: the DMA of tile kt + 3 is issued in step kt into the stage of tile kt itself -- whose fragments are in registers since step
// kt - 1 -- so that TWO tiles are in flight on the same three stages and a tile has two steps to land instead of one (the wait is
// vmcnt(pieces of one tile), not vmcnt(0)).  A K step of a small-M launch is shorter than an L2 round trip: M = 4096 x 728 x 728 ran
// 23 steps of 0.8 us.
template <int BM, int BN = SBN, bool LEAD2 = false>
__global__ __launch_bounds__(BM * 2, 2) void gemm_split16_kernel(const SplitGemmParams p) {
    constexpr int NS = 3, NT = BM * 2, WQ = BN / (BM / 32) / 8, TJ = BN / 32;   // TJ 16-column MFMA tiles per wave
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128, STAGE = A_STAGE + W_STAGE;
    constexpr int EPI_LD = BN + 4;
    constexpr int EPI_BYTES = BM * EPI_LD * 4;
    constexpr int SMEM_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int drow = lane >> 3, dchunk = lane & 7;
    const unsigned char* asrc[4];
    const unsigned char* wsrc[WQ];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc[q] = p.A + m * p.lda_bytes + c * 16;
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int row = wv * (WQ * 8) + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        const uint16_t* plane = (c & 4) ? p.Wlo : p.Whi;
        wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * p.Ktot + (c & 3) * 8);
    }
    auto issue = [&](int stage, int kt) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kt * 128), (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + (long)kt * 64),
                                             (lptr_t)(sb + A_STAGE + (wv * (WQ * 8) + q * 8) * 128), 16, 0, 0);
    };

    f32x4v acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: lane reads row r16 = lane & 15 of a 16-row block, logical chunk plane*4 + (lane >> 4)
    const int r16 = lane & 15, q4 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int ch_hi = ((q4 ^ sw) & 7) << 4, ch_lo = (((4 + q4) ^ sw) & 7) << 4;
    const int a_off = (wm * 64 + r16) * 128;              // + i*16*128
    const int w_off = A_STAGE + (wn * (BN / 2) + r16) * 128;    // + j*16*128
    struct Frags { bf16x8 ah[4], al[4], bh[TJ], bl[TJ]; };
    auto load_frags = [&](Frags& f, const unsigned char* sb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 2048 + ch_hi);
            f.al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 2048 + ch_lo);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 2048 + ch_hi);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 2048 + ch_lo);
        }
    };
    auto mfma48 = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
            }
    };

    const int nk = (p.Cin + SBK - 1) / SBK;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, r0 = 0;
    if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int P = 4 + WQ;      // DMA pieces per wave and tile
    issue(0, 0);
    issue(1, 1 < nk ? 1 : nk - 1);
    if constexpr (LEAD2) {
        issue(2, 2 < nk ? 2 : nk - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (p.stamps) t1 = __builtin_amdgcn_s_memtime();
    Frags f0, f1;
    load_frags(f0, smem);
    int s0 = 0, s1 = 1, s2 = 2;
    auto step = [&](Frags& cur, Frags& nxt, int kt) {
        if constexpr (LEAD2) {
            // tile kt + 1 has landed (tile kt + 2 may still fly); this wave's reads of stage s0 (tile kt: issued a step ago) are over --
            // with everybody's, behind the barrier, the stage can take tile kt + 3
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(s0, kt + 3 < nk ? kt + 3 : nk - 1);
        } else {
            if (kt > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            issue(s2, kt + 2 < nk ? kt + 2 : nk - 1);
        }
        load_frags(nxt, smem + s1 * STAGE);      // tile kt+1: certified by this step's barrier
        mfma48(cur);
        if constexpr (TJ == 4) {
#pragma unroll
            for (int g = 0; g < 4 + WQ; ++g) {   // the step's DMA pieces, two MFMAs apart
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (48 - 2 * (4 + WQ) - 32 > 0) __builtin_amdgcn_sched_group_barrier(0x008, 48 - 2 * (4 + WQ) - 32, 0);
        } else {   // 24 MFMAs, 4 + WQ DMA pieces, 12 fragment reads
#pragma unroll
            for (int g = 0; g < 4 + WQ; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 24 - (4 + WQ) - 12, 0);
        }
        const int t = s0; s0 = s1; s1 = s2; s2 = t;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(f0, f1, kt);
        if (kt + 1 < nk) step(f1, f0, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.stamps) t2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue: C/D layout of mfma_16x16: col = lane & 15, rows 4*(lane >> 4) + e; from the staging tile on it is the 32x32x16
    // kernel's (residual, statistics, split32 output)
    float(*stage)[EPI_LD] = reinterpret_cast<float(*)[EPI_LD]>(smem);
    constexpr int C4 = BN / 4;              // float4 chunks per staged row
    constexpr int ROWS_PER_PASS = NT / C4;
    constexpr int NROWS = BM / ROWS_PER_PASS;   // 16 rows per thread
    const int ec = (tid % C4) * 4, er = tid / C4;
    const int n = n0 + ec;
    const bool real = n < p.N;              // N % 4 == 0: a chunk is all inside or all outside
    // split32 output: the padding channels up to a multiple of 32 are written too (zeros), and both lanes of a channel-quad pair
    // take the same branches (ceil32(N) is a multiple of 8: a pair is all inside or all outside)
    const bool ncol = p.out_split ? n < (p.N + 31) / 32 * 32 : real;
    f32x4 rv[NROWS];
    if (p.res) {
#pragma unroll
        for (int k = 0; k < NROWS; ++k) {
            const long pix = m0 + er + k * ROWS_PER_PASS;
            rv[k] = (real && pix < p.M) ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) stage[wm * 64 + i * 16 + 4 * q4 + e][wn * (BN / 2) + j * 16 + r16] = acc[i][j][e];
    __syncthreads();
    if (p.stamps) t3 = __builtin_amdgcn_s_memtime();
    double ssum[4] = {0.0, 0.0, 0.0, 0.0}, ssq[4] = {0.0, 0.0, 0.0, 0.0};
    if (ncol) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, t1 = s1;
        f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        if (real) {
            s1 = *reinterpret_cast<const f32x4*>(p.scale1 + n);
            t1 = *reinterpret_cast<const f32x4*>(p.shift1 + n);
            if (p.scale2) {
                s2 = *reinterpret_cast<const f32x4*>(p.scale2 + n);
                t2 = *reinterpret_cast<const f32x4*>(p.shift2 + n);
            }
        }
        float* __restrict__ outp = p.C;
        // the output stage: fp32 NHWC, or the split32 layout (16-byte stores through the pair exchange of emd::dw_store)
        auto put = [&](long pix, f32x4 v) {
            if (!p.out_split) {
                if (p.nt) store_nt16(outp + pix * p.ldc + n, v);
                else *reinterpret_cast<f32x4*>(outp + pix * p.ldc + n) = v;
                return;
            }
            if (!real) v = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned h0, l0, h1, l1;
            split2(v[0], v[1], h0, l0);
            split2(v[2], v[3], h1, l1);
            const int q = n >> 2;
            const bool odd = q & 1;
            const unsigned r0 = emd::swap_pair(odd ? h0 : l0), r1 = emd::swap_pair(odd ? h1 : l1);
            unsigned char* g = reinterpret_cast<unsigned char*>(outp) + pix * (long)p.ldc * 4 + (n >> 5) * 128;
            if (!odd) *reinterpret_cast<u32x4*>(g + (q & 7) * 8) = u32x4{h0, h1, r0, r1};
            else *reinterpret_cast<u32x4*>(g + 64 + ((q - 1) & 7) * 8) = u32x4{r0, r1, l0, l1};
        };
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
#pragma unroll
            for (int k = 0; k < NROWS; ++k) {
                const int r = er + k * ROWS_PER_PASS;
                const long pix = m0 + r;
                if (pix < p.M) put(pix, finish(*reinterpret_cast<const f32x4*>(&stage[r][ec])) + rv[k]);
            }
        } else if (p.stats_part) {
            // batch statistics of the output (misc_py/modified_Xception.py:302-323: the norm that follows runs on batch
            // statistics) gathered while the tile is in hand: per-thread double sums over its 16 rows here, 16 -> 1 below,
            // one partial per (M tile, channel) for the fixed-order final reduction (bn_stats_final): no second pass over y
#pragma unroll 4
            for (int r = er; r < BM; r += ROWS_PER_PASS) {
                const long pix = m0 + r;
                if (pix >= p.M) break;
                const f32x4 v = finish(*reinterpret_cast<const f32x4*>(&stage[r][ec]));
                put(pix, v);
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
                const long pix = m0 + r;
                if (pix >= p.M) break;
                put(pix, finish(*reinterpret_cast<const f32x4*>(&stage[r][ec])));
            }
        }
    }
    if (p.stats_part) {   // block-uniform
        __syncthreads();  // the staging tile has been read out
        double(*red)[BN][2] = reinterpret_cast<double(*)[BN][2]>(smem);   // [row groups][128 channels][sum, sum of squares]
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
                p.stats_part[((long)mt * 2 + which) * p.N + n0 + col] = t;
            }
        }
    }
    if (p.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* o = p.stamps + (long)blockIdx.x * 8;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = __builtin_amdgcn_s_memtime();
        o[5] = ((long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) << 32) |
               (unsigned)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
        o[6] = r0; o[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// 256 x 192 tiles of the same GEMM (round 3).  The K loop above is held at 0.83 of its MFMA issue time by the LDS pipe: a 64 x 64 wave
// tile reads (4 + 4) x 2 fragments of 1 KB per 48 MFMAs and the workgroup's DMA writes another 48 KB per K step -- 176 KB per 1536
// clocks, 0.9 of the 128 B/clk a CU's LDS delivers.  Here a wave owns 64 x 96 (NWM = 4: 8 waves) or 128 x 96 (NWM = 2: 4 waves):
// (4 + 6) x 2 fragments per 72 MFMAs, and the tile's DMA (256 + 192 rows) is shared by 1.5x the MFMAs: 216 KB per 2304 clocks = 0.73
// (0.57 with 4 waves).  N = 728 is four column tiles (768 columns, as the 128-wide tiles pad it), M = 32768 gives 512 workgroups =
// two full rounds of one per CU.  LDS: the A ring has three stages (the activations stream from HBM: two K steps of lead), the W ring
// two (the weights sit in L2) = 144 KB.  Same products in the same order per output element as gemm_split16_kernel: identical bits,
// so that the choice of tile by M (the host rule) changes nothing in a result.  fp32 output, optional residual and second affine.
// MEASURED (profiles/r03_experiments.txt 11): 414 vs 433 us on 131072 x 728 x 728 and 53.7 vs 60.6 us on 16384 x 728 x 728, but 113.7 vs
// 101.2 us on the headline 32768 x 728 x 728 (two lockstep rounds of one workgroup per CU: every CU streams its 196 KB of output at the
// same moment, nothing computes meanwhile) and graph D 24.9-25.1 vs 24.3-24.4 ms -- so it is OFF by default (dev knob split_wide).
template <int NWM>
__global__ __launch_bounds__(NWM * 128, 1) void gemm_split16_wide_kernel(const SplitGemmParams p) {
    constexpr int BM = 256, BN = 192, NW = NWM * 2, NT = NW * 64;
    constexpr int WROWS = BM / NWM, TI = WROWS / 16, TJ = 6;       // wave tile: TI x TJ MFMA tiles of 16 x 16
    constexpr int PA = 32 / NW, PW = 24 / NW;                       // 1 KiB DMA pieces per wave and K step (A: 32, W: 24 per tile)
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128, NSA = 3, NSW = 2;
    constexpr int W_OFF = NSA * A_STAGE, SMEM_BYTES = W_OFF + NSW * W_STAGE;
    constexpr int EPI_LD = BN + 4, HALF = 128;
    static_assert(HALF * EPI_LD * 4 <= SMEM_BYTES, "staging half tile");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;   // the column tiles of a row tile are neighbours in one XCD: its A rows come from L2
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int drow = lane >> 3, dchunk = lane & 7;
    const unsigned char* asrc[PA];
    const unsigned char* wsrc[PW];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int row = (wv * PA + q) * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        asrc[q] = p.A + m * p.lda_bytes + c * 16;
    }
#pragma unroll
    for (int q = 0; q < PW; ++q) {
        const int row = (wv * PW + q) * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        const uint16_t* plane = (c & 4) ? p.Wlo : p.Whi;
        wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * p.Ktot + (c & 3) * 8);
    }
    auto issue_a = [&](int stage, int kt) {
#pragma unroll
        for (int q = 0; q < PA; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kt * 128), (lptr_t)(smem + stage * A_STAGE + (wv * PA + q) * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int stage, int kt) {
#pragma unroll
        for (int q = 0; q < PW; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + (long)kt * 64), (lptr_t)(smem + W_OFF + stage * W_STAGE + (wv * PW + q) * 1024), 16, 0, 0);
    };

    f32x4v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q4 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int ch_hi = ((q4 ^ sw) & 7) << 4, ch_lo = (((4 + q4) ^ sw) & 7) << 4;
    const int a_off = (wm * WROWS + r16) * 128;              // + i * 2048
    const int w_off = W_OFF + (wn * (BN / 2) + r16) * 128;   // + j * 2048

    const int nk = (p.Cin + SBK - 1) / SBK;
    // Pipeline.  The W fragments of a step (12 x 16 bytes per lane) are read one step ahead into a second register set; the A
    // fragments are read just in time, one row tile ahead of their MFMAs.  So at barrier kt the wave needs A(kt) and W(kt+1) in LDS:
    // issue order A0 W0 W1 A1 | step kt: W(kt+2), A(kt+2); the only group younger than W(kt+1) at the top of step kt is A(kt+1):
    // vmcnt(PA).  A(kt+2) goes into the stage tile kt-1 was read from (free since everybody passed barrier kt), W(kt+2) into the one
    // the current W fragments were read from during step kt-1.  A has two K steps to land (HBM), W one (L2).
    issue_a(0, 0);
    issue_w(0, 0);
    issue_w(1, 1 < nk ? 1 : nk - 1);
    issue_a(1, 1 < nk ? 1 : nk - 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA) : "memory");
    __builtin_amdgcn_s_barrier();
    struct BFrags { bf16x8 h[TJ], l[TJ]; };
    BFrags b0, b1;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        b0.h[j] = *reinterpret_cast<const bf16x8*>(smem + w_off + j * 2048 + ch_hi);
        b0.l[j] = *reinterpret_cast<const bf16x8*>(smem + w_off + j * 2048 + ch_lo);
    }
    int sa = 0, sa2 = 2;
    auto step = [&](const BFrags& cur, BFrags& nxt, int kt) {
        // (lgkmcnt(0): this wave's reads of the stages about to be refilled have returned, not merely been issued)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PA) : "memory");
        __builtin_amdgcn_s_barrier();
        issue_w(kt & 1, kt + 2 < nk ? kt + 2 : nk - 1);
        issue_a(sa2, kt + 2 < nk ? kt + 2 : nk - 1);
        const unsigned char* ab = smem + sa * A_STAGE + a_off;
        const unsigned char* wb = smem + ((kt + 1) & 1) * W_STAGE + w_off;
        bf16x8 ah[2], al[2];
        ah[0] = *reinterpret_cast<const bf16x8*>(ab + ch_hi);
        al[0] = *reinterpret_cast<const bf16x8*>(ab + ch_lo);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (i + 1 < TI) {
                ah[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(ab + (i + 1) * 2048 + ch_hi);
                al[(i + 1) & 1] = *reinterpret_cast<const bf16x8*>(ab + (i + 1) * 2048 + ch_lo);
            }
            // the next step's W fragments, spread over the row tiles
#pragma unroll
            for (int j = i * TJ / TI; j < (i + 1) * TJ / TI; ++j) {
                nxt.h[j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + ch_hi);
                nxt.l[j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + ch_lo);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], cur.h[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], cur.l[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], cur.h[j], acc[i][j], 0, 0, 0);
            }
        }
        // issue order pinned: the DMA pieces and the LDS reads sit between MFMAs, not in front of them
        constexpr int NDS = 2 * (TI - 1) + 2 * TJ, NMF = 3 * TI * TJ;
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);            // A fragments of row tile 0
#pragma unroll
        for (int g = 0; g < PA + PW; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
#pragma unroll
        for (int g = 0; g < NDS; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NMF - 2 * (PA + PW) - 2 * NDS > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - 2 * (PA + PW) - 2 * NDS, 0);
        sa = sa + 1 == NSA ? 0 : sa + 1;
        sa2 = sa2 + 1 == NSA ? 0 : sa2 + 1;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(b0, b1, kt);
        if (kt + 1 < nk) step(b1, b0, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: the tile leaves in two halves of 128 rows through a staging tile (full 768-byte row runs per 48 lanes)
    float(*stage)[EPI_LD] = reinterpret_cast<float(*)[EPI_LD]>(smem);
    constexpr int C4 = BN / 4;
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();
    const bool two = p.scale2 != nullptr;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (wm * WROWS / HALF == h) {   // wave-uniform
            const int rbase = wm * WROWS - h * HALF;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) stage[rbase + i * 16 + 4 * q4 + e][wn * (BN / 2) + j * 16 + r16] = acc[i][j][e];
        }
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < HALF * C4; idx += NT) {
            const int r = idx / C4, c = (idx - r * C4) * 4;
            const long pix = m0 + h * HALF + r;
            const int n = n0 + c;
            if (pix >= p.M) break;
            if (n >= p.N) continue;
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.scale1 + n), t1 = *reinterpret_cast<const f32x4*>(p.shift1 + n);
            f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
            if (two) {
                s2 = *reinterpret_cast<const f32x4*>(p.scale2 + n);
                t2 = *reinterpret_cast<const f32x4*>(p.shift2 + n);
            }
            f32x4 v = *reinterpret_cast<const f32x4*>(&stage[r][c]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float u = fmaf(v[k], s1[k], t1[k]);
                u = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                if (two) u = fminf(fmaxf(fmaf(u, s2[k], t2[k]), 0.f), hi2);
                v[k] = u;
            }
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + n);
            if (p.nt) store_nt16(p.C + pix * p.ldc + n, v);
            else *reinterpret_cast<f32x4*>(p.C + pix * p.ldc + n) = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolutions from split32 activations: dense 3x3 (stride 1/2, dilation), the four output phases of the
// 3x3 stride-2 transposed convolution, strided 1x1 -- the row map and tap list of gemm_conv.hip on the pipelined LDS-DMA
// structure above.  K runs over (tap, 32-channel step); the DMA source of an A row is the tap's source pixel, or a line of
// zeros for TF-SAME padding and rows beyond M (an 8 KB zero buffer, so that "+ K step" needs no per-row select).  The
// row -> destination pixel table lives in LDS behind the staging tile.  Output: fp32 NHWC, or split32 (out_split) when the
// consumer is another of these GEMMs -- a chain of convolutions then never materialises an fp32 activation.
struct SplitConvParams {
    SplitGemmParams g;
    int ntaps, Cpad, nkc;        // W tap stride (elements), 32-channel steps per tap
    int flat;                    // 1: source pixel = dest pixel = m
    int Hg, Wg, Ha, Wa, Hc, Wc, sa, sc, py, px;
    unsigned long long dyp, dxp; // per-tap source offsets, 7 bits each, biased by 64
    int out_split;
    // FOUR instances (the 3x3 stride-2 transposed conv as ONE launch): a workgroup runs the four output phases of its 256 input
    // pixels back to back, so the input rows come from HBM once (the later phases' DMA re-reads them from L2) instead of once per
    // phase launch.  Per phase: weight planes, tap count and tap offsets; the output phase (py, px) = (ph >> 1, ph & 1).
    const uint16_t* Whi4[4];
    const uint16_t* Wlo4[4];
    int ntaps4[4];
    unsigned long long dyp4[4], dxp4[4];
};

__device__ __attribute__((aligned(128))) unsigned char g_zero_buf[8192];

// BN = 128: 2 x 2 MFMA tiles per wave, 3 stages.  BN = 64 (the 64-channel 512^2 layers): 2 x 1 tiles per wave, a K step is half
// as long, so 4 stages (the whole 160 KB) keep the DMA three K steps ahead and one of its groups may stay in flight across the barrier.
template <int BN, bool FOUR = false>
__global__ __launch_bounds__(512, 2) void gemm_split_conv_kernel(const SplitConvParams cp) {
    const SplitGemmParams& p = cp.g;
    constexpr int BM = 256, NS = BN == 128 ? 3 : 4, WQ = BN / 64, TN = BN / 64;
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128, STAGE = A_STAGE + W_STAGE;
    constexpr int EPI_LD = BN + 4;
    constexpr int EPI_BYTES = BM * EPI_LD * 4;
    constexpr int SMEM_BYTES = NS * STAGE > EPI_BYTES + BM * 8 ? NS * STAGE : EPI_BYTES + BM * 8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    // this lane's four DMA rows: grid position (b, i, j) of row m, kept as (pixel index of (b,0,0) in the source, i, j)
    const int drow = lane >> 3, dchunk = lane & 7;
    int pi[4], pj[4], pb[4], pc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        pc[q] = (dchunk ^ ((row >> 1) & 7)) * 16;
        const long m = m0 + row;
        if (m < p.M) {
            if (cp.flat) {
                pi[q] = 0; pj[q] = 0; pb[q] = (int)m;
            } else {
                const int j = (int)(m % cp.Wg);
                const long t = m / cp.Wg;
                pi[q] = (int)(t % cp.Hg); pj[q] = j; pb[q] = (int)(t / cp.Hg) * cp.Ha * cp.Wa;
            }
        } else {
            pi[q] = -(1 << 20); pj[q] = 0; pb[q] = 0;    // beyond M: every tap reads zeros
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int a_off = (wm * 64 + fr) * 128;
    const int w_off = A_STAGE + (wn * (BN / 2) + fr) * 128;
#pragma unroll 1
    for (int ph = 0; ph < (FOUR ? 4 : 1); ++ph) {
    const uint16_t* __restrict__ Whi = FOUR ? cp.Whi4[ph] : p.Whi;
    const uint16_t* __restrict__ Wlo = FOUR ? cp.Wlo4[ph] : p.Wlo;
    const int ntaps = FOUR ? cp.ntaps4[ph] : cp.ntaps;
    const unsigned long long dyp = FOUR ? cp.dyp4[ph] : cp.dyp, dxp = FOUR ? cp.dxp4[ph] : cp.dxp;
    const int py = FOUR ? (ph >> 1) : cp.py, px = FOUR ? (ph & 1) : cp.px;
    const int Ktot = FOUR ? ntaps * cp.Cpad : p.Ktot;
    const unsigned char* asrc[4];
    auto set_tap = [&](int tap) {
        const int dy = (int)((dyp >> (7 * tap)) & 127) - 64, dx = (int)((dxp >> (7 * tap)) & 127) - 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long pix;
            bool ok;
            if (cp.flat) {
                pix = pb[q];
                ok = pi[q] >= 0;
            } else {
                const int iy = pi[q] * cp.sa + dy, ix = pj[q] * cp.sa + dx;
                ok = iy >= 0 && iy < cp.Ha && ix >= 0 && ix < cp.Wa;
                pix = (long)pb[q] + (long)iy * cp.Wa + ix;
            }
            asrc[q] = (ok ? p.A + pix * p.lda_bytes : g_zero_buf) + pc[q];
        }
    };
    const unsigned char* wsrc[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int row = wv * (WQ * 8) + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        const uint16_t* plane = (c & 4) ? Wlo : Whi;
        wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * Ktot + (c & 3) * 8);
    }
    auto issue = [&](int stage, int tap, int kc) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kc * 128), (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
        const long wk = ((long)tap * cp.Cpad + (long)kc * 32) * 2;
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + wk), (lptr_t)(sb + A_STAGE + (wv * (WQ * 8) + q * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frags { bf16x8 ah[2], al[2], bh[TN], bl[TN]; };
    auto load_frags = [&](Frags& f, const unsigned char* sb, int ks) {
        const int ch = ((ks * 2 + fh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + ch);
            f.al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + (ch ^ 64));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + ch);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + (ch ^ 64));
        }
    };
    auto mfma12 = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
            }
    };

    // (dtap, dkc): the K step the next DMA fetches; it stops advancing at the last one (the two surplus issues at the end
    // re-read it into a stage nobody computes on)
    const int total = ntaps * cp.nkc;
    int dtap = 0, dkc = 0, dstep = 0;
    auto advance = [&]() {
        if (dstep + 1 < total) {
            ++dstep;
            if (++dkc == cp.nkc) {
                dkc = 0;
                ++dtap;
                set_tap(dtap);
            }
        }
    };
    set_tap(0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        issue(s, dtap, dkc);
        advance();
    }
    // K steps 0 and 1 landed; NS-3 younger DMA groups (4 + WQ pieces each) may stay in flight across every barrier
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * (4 + WQ)) : "memory");
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    load_frags(f0, smem, 0);
    int s0 = 0, s1 = 1, s2 = NS - 1;   // stage of K step st / st+1 / the one being refilled (held step st-1)
    for (int st = 0; st < total; ++st) {
        if (st > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * (4 + WQ)) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        issue(s2, dtap, dkc);
        load_frags(f1, smem + s0 * STAGE, 1);
        mfma12(f0);
        load_frags(f0, smem + s1 * STAGE, 0);
        mfma12(f1);
        if constexpr (TN == 2) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {   // 6 MFMAs per half step: 5 DMA pieces, then the 6 reads of f1; second half: the 6 reads of the next f0
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        s2 = s0;                                   // the stage just computed on is refilled next
        s0 = s1;
        s1 = s1 + 1 == NS ? 0 : s1 + 1;
        __builtin_amdgcn_sched_barrier(0);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: accumulators -> fp32 LDS tile; row -> destination pixel table behind it
    float(*stage)[EPI_LD] = reinterpret_cast<float(*)[EPI_LD]>(smem);
    long long* rowP = reinterpret_cast<long long*>(smem + EPI_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                stage[r][wn * (BN / 2) + j * 32 + fr] = acc[i][j][e];
            }
    if (tid < BM) {
        const long m = m0 + tid;
        long long dst = -1;
        if (m < p.M) {
            if (cp.flat) {
                dst = m;
            } else {
                const int j = (int)(m % cp.Wg);
                const long t = m / cp.Wg;
                const int i = (int)(t % cp.Hg);
                const long b = t / cp.Hg;
                dst = (b * cp.Hc + (i * cp.sc + py)) * (long)cp.Wc + (j * cp.sc + px);
            }
        }
        rowP[tid] = dst;
    }
    __syncthreads();
    constexpr int C4 = BN / 4;
    constexpr int ROWS_PER_PASS = 512 / C4;
    const int ec = (tid % C4) * 4, er = tid / C4;
    const int n = n0 + ec;
    const int Np = cp.out_split ? (p.N + 31) / 32 * 32 : p.N;   // split32 output: the padding channels are written (zeros)
    if (n < Np) {
        const bool real = n < p.N;                               // N % 4 == 0: a chunk is all inside or all outside
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = s1, s2 = make_float4(1.f, 1.f, 1.f, 1.f), t2 = s1;
        if (real) {
            s1 = *reinterpret_cast<const float4*>(p.scale1 + n);
            t1 = *reinterpret_cast<const float4*>(p.shift1 + n);
            if (p.scale2) {
                s2 = *reinterpret_cast<const float4*>(p.scale2 + n);
                t2 = *reinterpret_cast<const float4*>(p.shift2 + n);
            }
        }
        const float* __restrict__ resp = real ? p.res : nullptr;
        const float hi = p.act == 1 ? 6.f : __builtin_inff();
        const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;   // second stage (extra BN): relu6, or relu with act code relu
        const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();   // v = min(max(max(v, lo), slope*v), hi): every act code
#pragma unroll 4
        for (int r = er; r < BM; r += ROWS_PER_PASS) {
            const long long pix = rowP[r];
            if (pix < 0) continue;
            float4 v = *reinterpret_cast<const float4*>(&stage[r][ec]);
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (resp) rv = *reinterpret_cast<const float4*>(resp + pix * p.ldres + n);
            v.x = fmaf(v.x, s1.x, t1.x); v.y = fmaf(v.y, s1.y, t1.y); v.z = fmaf(v.z, s1.z, t1.z); v.w = fmaf(v.w, s1.w, t1.w);
            v.x = fminf(fmaxf(fmaxf(v.x, lo), slope * v.x), hi); v.y = fminf(fmaxf(fmaxf(v.y, lo), slope * v.y), hi);
            v.z = fminf(fmaxf(fmaxf(v.z, lo), slope * v.z), hi); v.w = fminf(fmaxf(fmaxf(v.w, lo), slope * v.w), hi);
            if (p.scale2) {
                v.x = fminf(fmaxf(fmaf(v.x, s2.x, t2.x), 0.f), hi2); v.y = fminf(fmaxf(fmaf(v.y, s2.y, t2.y), 0.f), hi2);
                v.z = fminf(fmaxf(fmaf(v.z, s2.z, t2.z), 0.f), hi2); v.w = fminf(fmaxf(fmaf(v.w, s2.w, t2.w), 0.f), hi2);
            }
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            if (!real) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cp.out_split) {
                // 16-byte stores through an exchange between the two lanes of a channel-quad pair (see emd::dw_store): both
                // lanes of a pair share the row and the n < Np test (Np is a multiple of 32)
                unsigned h0, l0, h1, l1;
                split2(v.x, v.y, h0, l0);
                split2(v.z, v.w, h1, l1);
                const int q = n >> 2;
                const bool odd = q & 1;
                const unsigned r0 = emd::swap_pair(odd ? h0 : l0), r1 = emd::swap_pair(odd ? h1 : l1);
                unsigned char* g = reinterpret_cast<unsigned char*>(p.C) + pix * (long)p.ldc * 4 + (n >> 5) * 128;
                u32x4* dst = reinterpret_cast<u32x4*>(!odd ? g + (q & 7) * 8 : g + 64 + ((q - 1) & 7) * 8);
                const u32x4 val = !odd ? u32x4{h0, h1, r0, r1} : u32x4{r0, r1, l0, l1};
                if (p.nt) store_nt16(dst, val);
                else *dst = val;
            } else if (p.nt) {
                store_nt16(p.C + pix * p.ldc + n, f32x4{v.x, v.y, v.z, v.w});
            } else {
                *reinterpret_cast<float4*>(p.C + pix * p.ldc + n) = v;
            }
        }
    }
    if (FOUR) __syncthreads();   // the staging tile and the row table are read out before the next phase's DMA lands on them
    }   // phase loop
}

// Persistent form of the same GEMM: one workgroup per CU walks over its tiles (vb = blockIdx.x, + gridDim.x, ...) and the
// three-stage DMA / fragment pipeline simply runs on across tile boundaries: during a tile's last two K steps the first
// two K steps of the NEXT tile are already being fetched, and they land while this tile's epilogue runs -- only the first
// tile of a workgroup pays a prologue.  The epilogue does not go through LDS (the stages hold the next tile): the
// accumulators get the per-channel affine(s) + activation and leave straight from the MFMA C/D layout, 32 lanes x 4 B = one
// 128-byte run of one pixel per half wave, in 8 blocks of 8 dwords per lane; the residual values of block s+1 are requested
// while block s is stored.  Needs Cin >= 64 (two K steps), M % 256 == 0, and the lo weight plane within 2 GB behind the hi plane.
__global__ __launch_bounds__(512, 2) void gemm_split_persist_kernel(const SplitGemmParams p) {
    constexpr int BM = 256, NS = 3, WQ = 2;
    constexpr int A_STAGE = BM * 128, W_STAGE = SBN * 128, STAGE = A_STAGE + W_STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int drow = lane >> 3, dchunk = lane & 7;
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int a_off = (wm * 64 + fr) * 128;
    const int w_off = A_STAGE + (wn * 64 + fr) * 128;
    const int nblk = p.n_mtiles * p.n_ntiles;
    const int nk = (p.Cin + SBK - 1) / SBK;

    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, r0 = 0;
    if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    f32x16 acc[2][2];
    struct Frags { bf16x8 ah[2], al[2], bh[2], bl[2]; };
    auto load_frags = [&](Frags& f, const unsigned char* sb, int ks) {
        const int ch = ((ks * 2 + fh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + ch);
            f.al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + (ch ^ 64));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + ch);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + (ch ^ 64));
        }
    };
    auto mfma12 = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
            }
    };

    // lane l of a DMA piece fills physical chunk (l & 7) of row (l >> 3) with logical chunk (l & 7) ^ ((row >> 1) & 7);
    // rows of pieces q and q+2 (A) differ by 16 (same swizzle), those of q and q+1 by 8 (chunk ^ 4).  DMA sources =
    // uniform tile base (scalar registers) + per-lane 32-bit offset that does not depend on the tile.
    unsigned aoff[2], woff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        const int c = dchunk ^ ((row >> 1) & 7);
        aoff[q] = (unsigned)row * (unsigned)p.lda_bytes + c * 16;
        const int wrow = wv * 16 + q * 8 + drow;
        const int wc = dchunk ^ ((wrow >> 1) & 7);
        woff[q] = (unsigned)wrow * (unsigned)p.Ktot * 2u + (wc & 3) * 16 + ((wc & 4) ? p.wlo_delta : 0u);
    }
    auto issue = [&](int stage, const unsigned char* abase, const unsigned char* wbase) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(abase + (long)(q >> 1) * 16 * p.lda_bytes + aoff[q & 1]),
                                             (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wbase + woff[q]), (lptr_t)(sb + A_STAGE + (wv * (WQ * 8) + q * 8) * 128), 16, 0, 0);
    };
    auto tile_origin = [&](int vb, long& m0, int& n0) {   // XCD-aware tile mapping (bijective for any grid size)
        const int q = nblk >> 3, r = nblk & 7, xcd = vb & 7, loc = vb >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        m0 = (long)(bid / p.n_ntiles) * BM;
        n0 = (bid % p.n_ntiles) * SBN;
    };

    int vb = blockIdx.x;           // the launcher guarantees gridDim.x <= nblk
    long m0;
    int n0;
    tile_origin(vb, m0, n0);
    const unsigned char* abase = p.A + m0 * p.lda_bytes;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.Whi + (long)n0 * p.Ktot);
    issue(0, abase, wbase);
    issue(1, abase + 128, wbase + 64);     // nk >= 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.stamps) t1 = __builtin_amdgcn_s_memtime();
    Frags f0, f1;
    load_frags(f0, smem, 0);
    int s0 = 0, s1 = 1, s2 = 2;    // stage of K step kt / kt+1 / the one being refilled
    bool first_tile = true;
    // the DMA of the coming step, prepared one step ahead so that the scalar arithmetic sits between MFMAs
    const unsigned char* dma_a = abase + 2 * 128;
    const unsigned char* dma_w = wbase + 2 * 64;

    const unsigned lrow = wm * 64 + 4 * fh, lcol = wn * 64 + fr;
    const unsigned loff_c = lrow * (unsigned)p.ldc + lcol, loff_r = lrow * (unsigned)p.ldres + lcol;
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();

    while (true) {
        // the tile after this one (its first two K steps are fetched during this tile's last two)
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < nblk;
        long m0n = m0;
        int n0n = n0;
        if (has_next) tile_origin(vbn, m0n, n0n);
        const unsigned char* abase_n = p.A + m0n * p.lda_bytes;
        const unsigned char* wbase_n = reinterpret_cast<const unsigned char*>(p.Whi + (long)n0n * p.Ktot);

#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            issue(s2, dma_a, dma_w);
            load_frags(f1, smem + s0 * STAGE, 1);
            mfma12(f0);
            load_frags(f0, smem + s1 * STAGE, 0);
            mfma12(f1);
            // the step after this one fetches K step kt+3: beyond this tile's end that is the next tile's first K steps (for the
            // last tile: its own again, into a stage nobody computes on)
            dma_a += 128;
            dma_w += 64;
            if (kt + 3 == nk) {
                dma_a = abase_n;
                dma_w = wbase_n;
            }
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            const int t = s0; s0 = s1; s1 = s2; s2 = t;
            // the step ENDS with the wait + barrier that certify the next one: K step kt+2 has landed for every wave (the
            // previous tile's stores are older than this step's DMA, so they are covered too), and every wave is done
            // reading the stage the next step refills
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (p.stamps && first_tile) t3 = __builtin_amdgcn_s_memtime();
        first_tile = false;

        // ---- epilogue of this tile, straight from the accumulators.  Block s = (i, j, h) = (s>>2, (s>>1)&1, s&1): elements
        // e = 8h .. 8h+7 of MFMA tile (i, j): rows (e&3) + 8*(e>>2) + 4*(lane>>5), column lane & 31.  Address = uniform
        // base (scalar unit) + one per-lane 32-bit offset.  One clamp form for every activation code (see gemm_split_kernel).
        {
            float s1v[2], t1v[2], s2v[2], t2v[2];
            bool nv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + fr;
                nv[j] = n < p.N;
                s1v[j] = nv[j] ? p.scale1[n] : 0.f;
                t1v[j] = nv[j] ? p.shift1[n] : 0.f;
                s2v[j] = (p.scale2 && nv[j]) ? p.scale2[n] : 1.f;
                t2v[j] = (p.scale2 && nv[j]) ? p.shift2[n] : 0.f;
            }
            float rres[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) rres[c] = 0.f;
            auto load_res = [&](int i, int j, int h) {
                const float* ub = p.res + (m0 + i * 32 + h * 16) * p.ldres + n0 + j * 32;
                if (nv[j]) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) rres[c] = (ub + ((c & 3) + 8 * (c >> 2)) * p.ldres)[loff_r];
                }
            };
            if (p.res) load_res(0, 0, 0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int i = s >> 2, j = (s >> 1) & 1, h = s & 1;
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float u = fmaf(acc[i][j][8 * h + c], s1v[j], t1v[j]);
                    u = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                    if (p.scale2) u = fminf(fmaxf(fmaf(u, s2v[j], t2v[j]), 0.f), hi2);
                    v[c] = u + rres[c];
                }
                if (p.res && s < 7) load_res((s + 1) >> 2, ((s + 1) >> 1) & 1, (s + 1) & 1);
                float* ub = p.C + (m0 + i * 32 + h * 16) * p.ldc + n0 + j * 32;
                if (nv[j]) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) (ub + ((c & 3) + 8 * (c >> 2)) * p.ldc)[loff_c] = v[c];
                }
            }
        }
        if (!has_next) break;
        vb = vbn; m0 = m0n; n0 = n0n; abase = abase_n; wbase = wbase_n;
    }
    if (p.stamps) t2 = __builtin_amdgcn_s_memtime();
    if (p.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* o8 = p.stamps + (long)blockIdx.x * 8;
        o8[0] = t0; o8[1] = t1; o8[2] = t2; o8[3] = t3; o8[4] = __builtin_amdgcn_s_memtime();
        o8[5] = ((long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) << 32) |
                (unsigned)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
        o8[6] = r0; o8[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// fp32 [npix][C] (pitch ldx floats) -> split32 (pitch ldy 4-byte units): one thread per (pixel, 4 channels),
// channels C..ceil32(C) written as zero
__global__ __launch_bounds__(256) void to_split32_kernel(const float* __restrict__ x, int ldx, unsigned char* __restrict__ y,
                                                         int ldy, long npix, int C4, int C4p) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= npix * C4p) return;
    int c4;
    const long pix = emd::divmod(tid, C4p, c4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4) v = *reinterpret_cast<const f32x4*>(x + pix * ldx + c4 * 4);
    unsigned h0, l0, h1, l1;
    split2(v[0], v[1], h0, l0);
    split2(v[2], v[3], h1, l1);
    // 16-byte stores through an exchange between the two lanes of a quad pair (see emd::dw_store); C4p is even, so a pair is
    // never split by the bounds check above
    const bool odd = c4 & 1;
    const unsigned r0 = emd::swap_pair(odd ? h0 : l0), r1 = emd::swap_pair(odd ? h1 : l1);
    unsigned char* g = y + pix * (long)ldy * 4 + (c4 >> 3) * 128;
    if (!odd) *reinterpret_cast<u32x4*>(g + (c4 & 7) * 8) = u32x4{h0, h1, r0, r1};
    else *reinterpret_cast<u32x4*>(g + 64 + ((c4 - 1) & 7) * 8) = u32x4{r0, r1, l0, l1};
}

}  // namespace

static int g_variant_override = -1;
static long long* g_stamps = nullptr;

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
    // Where the pair (depthwise with split32 output, this GEMM) beats (depthwise, emd_conv1x1_f32) on MI355X
    // (tools/gemm_split_bench.py): matrix-core bound shapes whose grid fills the chip with 256 x 128 tiles.  On the
    // HBM-bound layers (N = 128 at 256^2, K <= 128) the register-staged kernel's 128 x 128 tiles at two workgroups per CU are as fast or faster.
    // Round 3: at every M -- below 192 tiles of 256 rows the 128-row form of the same kernel takes over (bit-identical), so that
    // "image b of a batch == the image alone" holds although the 16x16x32 MFMAs sum a K step in another order than the register-staged
    // kernel's 32x32x16.
    (void)M;
    if (Cout < 128 || Cout % 4 || Cin % 4) return 0;
    return (Cin >= 512 || (Cin >= 256 && Cout >= 256)) ? 1 : 0;
}

static int conv1x1_split32_impl(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo,
                                const float* scale1, const float* shift1, const float* scale2,
                                const float* shift2, const float* res, int ldres, float* y, int ldy, long M,
                                int Cin, int Cout, int act, emd_stream_t stream, double* stats_part, int out_split = 0) {
    EMD_REQUIRE(xs && whi && wlo && scale1 && shift1 && y, EMD_E_INVALID, "emd_conv1x1_split32_f32: null pointer");
    EMD_REQUIRE(!out_split || (!stats_part && ldy % 32 == 0 && ldy >= emd_split32_ld(Cout) && (reinterpret_cast<uintptr_t>(y) & 127u) == 0),
                EMD_E_ALIGN, "emd_conv1x1_split32_out_f32: a split32 output needs y 128-byte aligned, ldy a multiple of 32, >= ceil32(Cout)");
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
    p.ldc = ldy; p.ldres = ldres; p.act = act; p.stats_part = stats_part; p.out_split = out_split ? 1 : 0;
    p.nt = split_nt(2);
    // kernel variant: 3 = 256-row tiles, 3 stages, pipelined K loop, 32x32x16 MFMAs; 5 = the same on 16x16x32 MFMAs;
    // dev knobs for A/B runs: emd_debug_knob("split_variant") / emd_debug_split_variant = 0 (256 rows, 2 stages), 1 (256, 3, plain loop),
    // 2 (128 rows, 2 stages, two workgroups per CU), 4 (persistent, epilogue stores inside the next tile's K loop),
    // 6 (variant 3 with the W tile through registers instead of LDS-DMA: WREG), 7 (variant 3 with the epilogue straight from
    // the registers: DIRECT)
    int v = emd::g_knobs.split_variant;
    if (g_variant_override >= 0) v = g_variant_override;
    if ((stats_part || out_split) && v >= 0 && v != 3 && v != 5 && v != 6 && !(out_split && v == 7)) v = 3;   // the statistics epilogue and
                                                // the split32 output live in the two default kernels (and the WREG / DIRECT dev forms)
    // default (round 3): variant 5, the 16x16x32-MFMA kernel -- same cycles per K step, but the chip holds 1.86 instead of 1.73 GHz
    // under it (97.9 vs 103.7 us on 32768 x 728 x 728; MI355X_MICROARCH.md, DVFS give-back 7).  It sums a K step in another order
    // than the 32x32x16 kernels (2e-7 relative: the contract is the oracle at 1e-3, not identity with a sibling kernel); batch
    // independence is kept by using it at EVERY M of the layers it serves (128-row tiles below 192 tiles of 256 rows).
    if (v < 0) v = 5;
    p.stamps = g_stamps;
    const bool small = v == 5 && !stats_part && ((M + 255) / 256) * ((Cout + SBN - 1) / SBN) < 192;   // (statistics partials: one per 256 rows)
    // ... and 64-column tiles (two workgroups per CU) where even the 128-row tiles leave CUs idle: a batch of 4 at 512^2 is M = 4096,
    // 192 tiles of 128 x 128 on 256 CUs, 384 of 128 x 64.  All forms give the same bits (same products, same K order).
    const bool narrow = small && !out_split && ((M + 127) / 128) * ((Cout + SBN - 1) / SBN) < 256 && emd::g_knobs.split_narrow;
    // ... and 256 x 192 tiles (gemm_split16_wide_kernel) where they fill the chip and stay inside the padded weight planes (N = 728, 384, 192, ...)
    const int n192 = (Cout + 191) / 192;
    const int wide = (v == 5 && !stats_part && !out_split && n192 * 192 <= (Cout + SBN - 1) / SBN * SBN && ((M + 255) / 256) * n192 >= 256)
                         ? emd::g_knobs.split_wide : 0;
    const int bm = (v == 2 || small) ? 128 : 256, bn = wide ? 192 : (narrow ? 64 : SBN);
    p.n_mtiles = (int)((M + bm - 1) / bm);
    p.n_ntiles = (Cout + bn - 1) / bn;
    const long nblk = (long)p.n_mtiles * p.n_ntiles;
    if (nblk > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "emd_conv1x1_split32_f32: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (v == 1) hipLaunchKernelGGL((gemm_split_kernel<256, 3>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    else if (v == 0) hipLaunchKernelGGL((gemm_split_kernel<256, 2>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    else if (v == 4 && Cin >= 64 && M % 256 == 0 && wlo > whi &&
             (reinterpret_cast<uintptr_t>(wlo) - reinterpret_cast<uintptr_t>(whi)) < 0x7fffffffu && (long)256 * p.lda_bytes < 0x7fffffffL) {
        p.wlo_delta = (unsigned)(reinterpret_cast<uintptr_t>(wlo) - reinterpret_cast<uintptr_t>(whi));
        int ncu = 256;
        static const int cus = [] { int d = 0, n = 0; if (hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) return n; return 256; }();
        ncu = cus;
        const unsigned grid = (unsigned)(nblk < ncu ? nblk : ncu);
        hipLaunchKernelGGL(gemm_split_persist_kernel, dim3(grid), dim3(512), 0, st, p);
    }
    else if (v == 2) hipLaunchKernelGGL((gemm_split_kernel<128, 2>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (wide == 2) hipLaunchKernelGGL((gemm_split16_wide_kernel<2>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (wide) hipLaunchKernelGGL((gemm_split16_wide_kernel<4>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    else if (v == 5 && emd::g_knobs.split_lead == 2) {
        if (narrow) hipLaunchKernelGGL((gemm_split16_kernel<128, 64, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
        else if (small) hipLaunchKernelGGL((gemm_split16_kernel<128, SBN, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_split16_kernel<256, SBN, true>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    }
    else if (v == 5 && narrow) hipLaunchKernelGGL((gemm_split16_kernel<128, 64>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (v == 5 && small) hipLaunchKernelGGL(gemm_split16_kernel<128>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    else if (v == 5) hipLaunchKernelGGL(gemm_split16_kernel<256>, dim3((unsigned)nblk), dim3(512), 0, st, p);
    else if (v == 6) hipLaunchKernelGGL((gemm_split_kernel<256, 3, true, true>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    else if (v == 7) hipLaunchKernelGGL((gemm_split_kernel<256, 3, true, false, true>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_split_kernel<256, 3, true>), dim3((unsigned)nblk), dim3(512), 0, st, p);
    return emd::check_launch("gemm_split_kernel");
}

// ---------------------------------------------------------------------------------------------- the transposed conv, one launch
namespace {

// 4 x 4 transpose inside a lane quad (two DPP exchange rounds): in, lane i holds column i of a block (r[k] = a[k][i]); out, row i.
// Turns four rows x one channel of the 32x32 MFMA C/D layout into one row x four consecutive channels: 16-byte stores without
// a staging tile (sep_pipe.hip has the same helper).
__device__ __forceinline__ void quad_transpose4(float (&r)[4], int li) {
    const bool b0 = li & 1, b1 = li & 2;
    float s0 = b0 ? r[0] : r[1], s1 = b0 ? r[2] : r[3];
    s0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
    s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
    r[0] = b0 ? s0 : r[0]; r[1] = b0 ? r[1] : s0;
    r[2] = b0 ? s1 : r[2]; r[3] = b0 ? r[3] : s1;
    float t0 = b1 ? r[0] : r[2], t1 = b1 ? r[1] : r[3];
    t0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t0), 0x4E, 0xF, 0xF, true));
    t1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1), 0x4E, 0xF, 0xF, true));
    r[0] = b1 ? t0 : r[0]; r[2] = b1 ? r[2] : t0;
    r[1] = b1 ? t1 : r[1]; r[3] = b1 ? r[3] : t1;
}

// slim.conv2d_transpose(k = 3, s = 2) (machine_learning/denoiser.py:138-150) as ONE launch, round 3 form: the four output phases of a
// workgroup's 256 input pixels back to back as in gemm_split_conv_kernel<BN, true> (same K loops, same products in the same order:
// bit-identical), but the epilogue leaves straight from the accumulators (quad transpose, 16-byte non-temporal stores) and touches no
// LDS -- so the DMA of the NEXT phase's first K steps is issued before the stores of this one, and the stores (the layer writes four
// times what it reads: 4.3 GB for deconv1to0) drain under the next K loop instead of between two of them.  fp32 output, no residual.
template <int BN>
__global__ __launch_bounds__(512, 2) void deconv4_split_kernel(const SplitConvParams cp) {
    const SplitGemmParams& p = cp.g;
    constexpr int BM = 256, NS = BN == 128 ? 3 : 4, WQ = BN / 64, TN = BN / 64;
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128, STAGE = A_STAGE + W_STAGE;
    constexpr int E = 8 * TN;   // stores per wave and phase
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int drow = lane >> 3, dchunk = lane & 7;
    int pi[4], pj[4], pb[4], pc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        pc[q] = (dchunk ^ ((row >> 1) & 7)) * 16;
        const long m = m0 + row;
        if (m < p.M) {
            const int j = (int)(m % cp.Wg);
            const long t = m / cp.Wg;
            pi[q] = (int)(t % cp.Hg); pj[q] = j; pb[q] = (int)(t / cp.Hg) * cp.Ha * cp.Wa;
        } else {
            pi[q] = -(1 << 20); pj[q] = 0; pb[q] = 0;    // beyond M: every tap reads zeros
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int a_off = (wm * 64 + fr) * 128;
    const int w_off = A_STAGE + (wn * (BN / 2) + fr) * 128;

    // ---- epilogue roles.  Before the transpose a lane holds channel n0 + wn BN/2 + 32 j + fr of rows (e & 3) + 8 (e >> 2) + 4 fh;
    // after it, row 8 q + 4 fh + li and channels 4 cq .. 4 cq + 3 of the 32-column group
    const int li = fr & 3, cq = fr >> 2;
    long dst0[2][4];     // output pixel of phase (0, 0) for this lane's rows, -1 beyond M
    bool full = true;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long m = m0 + wm * 64 + i * 32 + 8 * q + 4 * fh + li;
            long d = -1;
            if (m < p.M) {
                const int j = (int)(m % cp.Wg);
                const long t = m / cp.Wg;
                d = ((t / cp.Hg) * cp.Hc + (t % cp.Hg) * 2) * (long)cp.Wc + 2 * j;
            }
            dst0[i][q] = d;
            full = full && d >= 0;
        }
    float es1[TN], et1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + fr;
        es1[j] = n < p.N ? p.scale1[n] : 0.f;
        et1[j] = n < p.N ? p.shift1[n] : 0.f;
        asm volatile("" ::"v"(es1[j]), "v"(et1[j]));   // waited for here, not behind the DMA groups in the loop
        full = full && (n0 + wn * (BN / 2) + j * 32 + 31 < p.N);
    }
    full = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(!full) == 0);   // per wave: no masked store, the store count is exact
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();

    // ---- per-phase state
    const uint16_t* __restrict__ Whi = cp.Whi4[0];
    const uint16_t* __restrict__ Wlo = cp.Wlo4[0];
    int ntaps = cp.ntaps4[0];
    unsigned long long dyp = cp.dyp4[0], dxp = cp.dxp4[0];
    const unsigned char* asrc[4];
    const unsigned char* wsrc[WQ];
    auto set_tap = [&](int tap) {
        const int dy = (int)((dyp >> (7 * tap)) & 127) - 64, dx = (int)((dxp >> (7 * tap)) & 127) - 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int iy = pi[q] * cp.sa + dy, ix = pj[q] * cp.sa + dx;
            const bool ok = iy >= 0 && iy < cp.Ha && ix >= 0 && ix < cp.Wa;
            const long pix = (long)pb[q] + (long)iy * cp.Wa + ix;
            asrc[q] = (ok ? p.A + pix * p.lda_bytes : g_zero_buf) + pc[q];
        }
    };
    auto set_phase = [&](int ph) {
        Whi = cp.Whi4[ph]; Wlo = cp.Wlo4[ph]; ntaps = cp.ntaps4[ph]; dyp = cp.dyp4[ph]; dxp = cp.dxp4[ph];
        const int Ktot = ntaps * cp.Cpad;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int row = wv * (WQ * 8) + q * 8 + drow;
            const int c = dchunk ^ ((row >> 1) & 7);
            const uint16_t* plane = (c & 4) ? Wlo : Whi;
            wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * Ktot + (c & 3) * 8);
        }
    };
    auto issue = [&](int stage, int tap, int kc) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kc * 128), (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
        const long wk = ((long)tap * cp.Cpad + (long)kc * 32) * 2;
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + wk), (lptr_t)(sb + A_STAGE + (wv * (WQ * 8) + q * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frags { bf16x8 ah[2], al[2], bh[TN], bl[TN]; };
    auto load_frags = [&](Frags& f, const unsigned char* sb, int ks) {
        const int ch = ((ks * 2 + fh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + ch);
            f.al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + (ch ^ 64));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + ch);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + (ch ^ 64));
        }
    };
    auto mfma12 = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
            }
    };

    int total = 0, dtap = 0, dkc = 0, dstep = 0;
    auto advance = [&]() {
        if (dstep + 1 < total) {
            ++dstep;
            if (++dkc == cp.nkc) {
                dkc = 0;
                ++dtap;
                set_tap(dtap);
            }
        }
    };
    auto prologue = [&](int ph) {   // the first NS - 1 K steps of phase ph into stages 0 .. NS - 2
        set_phase(ph);
        total = ntaps * cp.nkc;
        dtap = dkc = dstep = 0;
        set_tap(0);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) {
            issue(s, dtap, dkc);
            advance();
        }
    };
    prologue(0);
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
        // K steps 0 and 1 landed; NS-3 younger DMA groups (4 + WQ pieces each) may stay in flight across every barrier -- and, from the
        // second phase on, the previous phase's E stores, which were issued after this phase's first groups
        if (ph > 0 && full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * (4 + WQ) + E) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * (4 + WQ)) : "memory");
        __builtin_amdgcn_s_barrier();
        Frags f0, f1;
        load_frags(f0, smem, 0);
        int s0 = 0, s1 = 1, s2 = NS - 1;
        for (int st = 0; st < total; ++st) {
            if (st > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * (4 + WQ)) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            issue(s2, dtap, dkc);
            load_frags(f1, smem + s0 * STAGE, 1);
            mfma12(f0);
            load_frags(f0, smem + s1 * STAGE, 0);
            mfma12(f1);
            if constexpr (TN == 2) {
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            s2 = s0;
            s0 = s1;
            s1 = s1 + 1 == NS ? 0 : s1 + 1;
            __builtin_amdgcn_sched_barrier(0);
            advance();
        }
        // every wave's fragment reads are done and this wave's surplus DMA groups have landed: the stages are free for the next phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int py = ph >> 1, px = ph & 1;
        if (ph < 3) prologue(ph + 1);
        // ---- epilogue of phase ph, straight from the accumulators
        const long poff = (long)py * cp.Wc + px;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n4 = n0 + wn * (BN / 2) + j * 32 + 4 * cq;
            const bool ncol = n4 < p.N;
            const float s1 = es1[j], t1 = et1[j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float u = fmaf(acc[i][j][4 * q + k], s1, t1);
                        r[k] = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                    }
                    quad_transpose4(r, li);
                    if (ncol && dst0[i][q] >= 0) store_nt16(p.C + (dst0[i][q] + poff) * p.ldc + n4, f32x4{r[0], r[1], r[2], r[3]});
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The same transposed conv on 128-row tiles, 4 waves, two LDS stages (64 KB at BN = 128): TWO workgroups per CU.  vmcnt retires in
// order, so a workgroup's output stores (four times the bytes it reads) hold up its own next K loop whatever the issue order -- here
// the partner workgroup's K loop runs meanwhile.  Same wave tile (64 x 64), same products in the same order along K: bit-identical
// to deconv4_split_kernel.  One K step of DMA in flight (two stages), fragments read after the step's barrier.
template <int BN>
__global__ __launch_bounds__(256, 2) void deconv4_half_kernel(const SplitConvParams cp) {
    const SplitGemmParams& p = cp.g;
    constexpr int BM = 128, NS = 2, WQ = BN / 32, TN = BN / 64;
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128, STAGE = A_STAGE + W_STAGE;
    constexpr int E = 8 * TN;   // stores per wave and phase
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int nblk = p.n_mtiles * p.n_ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.n_ntiles, nt = bid % p.n_ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int drow = lane >> 3, dchunk = lane & 7;
    int pi[4], pj[4], pb[4], pc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wv * 32 + q * 8 + drow;
        pc[q] = (dchunk ^ ((row >> 1) & 7)) * 16;
        const long m = m0 + row;
        if (m < p.M) {
            const int j = (int)(m % cp.Wg);
            const long t = m / cp.Wg;
            pi[q] = (int)(t % cp.Hg); pj[q] = j; pb[q] = (int)(t / cp.Hg) * cp.Ha * cp.Wa;
        } else {
            pi[q] = -(1 << 20); pj[q] = 0; pb[q] = 0;    // beyond M: every tap reads zeros
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int a_off = (wm * 64 + fr) * 128;
    const int w_off = A_STAGE + (wn * (BN / 2) + fr) * 128;

    // ---- epilogue roles.  Before the transpose a lane holds channel n0 + wn BN/2 + 32 j + fr of rows (e & 3) + 8 (e >> 2) + 4 fh;
    // after it, row 8 q + 4 fh + li and channels 4 cq .. 4 cq + 3 of the 32-column group
    const int li = fr & 3, cq = fr >> 2;
    long dst0[2][4];     // output pixel of phase (0, 0) for this lane's rows, -1 beyond M
    bool full = true;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long m = m0 + wm * 64 + i * 32 + 8 * q + 4 * fh + li;
            long d = -1;
            if (m < p.M) {
                const int j = (int)(m % cp.Wg);
                const long t = m / cp.Wg;
                d = ((t / cp.Hg) * cp.Hc + (t % cp.Hg) * 2) * (long)cp.Wc + 2 * j;
            }
            dst0[i][q] = d;
            full = full && d >= 0;
        }
    float es1[TN], et1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + fr;
        es1[j] = n < p.N ? p.scale1[n] : 0.f;
        et1[j] = n < p.N ? p.shift1[n] : 0.f;
        asm volatile("" ::"v"(es1[j]), "v"(et1[j]));   // waited for here, not behind the DMA groups in the loop
        full = full && (n0 + wn * (BN / 2) + j * 32 + 31 < p.N);
    }
    full = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(!full) == 0);   // per wave: no masked store, the store count is exact
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();

    // ---- per-phase state
    const uint16_t* __restrict__ Whi = cp.Whi4[0];
    const uint16_t* __restrict__ Wlo = cp.Wlo4[0];
    int ntaps = cp.ntaps4[0];
    unsigned long long dyp = cp.dyp4[0], dxp = cp.dxp4[0];
    const unsigned char* asrc[4];
    const unsigned char* wsrc[WQ];
    auto set_tap = [&](int tap) {
        const int dy = (int)((dyp >> (7 * tap)) & 127) - 64, dx = (int)((dxp >> (7 * tap)) & 127) - 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int iy = pi[q] * cp.sa + dy, ix = pj[q] * cp.sa + dx;
            const bool ok = iy >= 0 && iy < cp.Ha && ix >= 0 && ix < cp.Wa;
            const long pix = (long)pb[q] + (long)iy * cp.Wa + ix;
            asrc[q] = (ok ? p.A + pix * p.lda_bytes : g_zero_buf) + pc[q];
        }
    };
    auto set_phase = [&](int ph) {
        Whi = cp.Whi4[ph]; Wlo = cp.Wlo4[ph]; ntaps = cp.ntaps4[ph]; dyp = cp.dyp4[ph]; dxp = cp.dxp4[ph];
        const int Ktot = ntaps * cp.Cpad;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int row = wv * (WQ * 8) + q * 8 + drow;
            const int c = dchunk ^ ((row >> 1) & 7);
            const uint16_t* plane = (c & 4) ? Wlo : Whi;
            wsrc[q] = reinterpret_cast<const unsigned char*>(plane + (long)(n0 + row) * Ktot + (c & 3) * 8);
        }
    };
    auto issue = [&](int stage, int tap, int kc) {
        unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (long)kc * 128), (lptr_t)(sb + (wv * 32 + q * 8) * 128), 16, 0, 0);
        const long wk = ((long)tap * cp.Cpad + (long)kc * 32) * 2;
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + wk), (lptr_t)(sb + A_STAGE + (wv * (WQ * 8) + q * 8) * 128), 16, 0, 0);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frags { bf16x8 ah[2], al[2], bh[TN], bl[TN]; };
    auto load_frags = [&](Frags& f, const unsigned char* sb, int ks) {
        const int ch = ((ks * 2 + fh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + ch);
            f.al[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 4096 + (ch ^ 64));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + ch);
            f.bl[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 4096 + (ch ^ 64));
        }
    };
    auto mfma12 = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
            }
    };

    int total = 0, dtap = 0, dkc = 0, dstep = 0;
    auto advance = [&]() {
        if (dstep + 1 < total) {
            ++dstep;
            if (++dkc == cp.nkc) {
                dkc = 0;
                ++dtap;
                set_tap(dtap);
            }
        }
    };
    auto prologue = [&](int ph) {   // the first K step of phase ph into stage 0
        set_phase(ph);
        total = ntaps * cp.nkc;
        dtap = dkc = dstep = 0;
        set_tap(0);
        issue(0, 0, 0);
        advance();
    };
    prologue(0);
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
        for (int st = 0; st < total; ++st) {
            // K step st has landed.  Older than its group: nothing but, at st == 0 of a later phase, nothing either (the previous
            // phase's stores were issued AFTER this phase's first group) -- so the E stores may stay in flight across the first barrier
            if (st == 0 && ph > 0 && full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // ... for every wave; and every wave is done with the other stage (step st - 1)
            issue((st + 1) & 1, dtap, dkc);     // step st + 1 (beyond the last step: a re-read of it, so that the counts stay uniform)
            advance();
            Frags f0, f1;
            const unsigned char* sb = smem + (st & 1) * STAGE;
            load_frags(f0, sb, 0);
            load_frags(f1, sb, 1);
            mfma12(f0);
            mfma12(f1);
        }
        // every wave's fragment reads are done and this wave's surplus DMA groups have landed: the stages are free for the next phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int py = ph >> 1, px = ph & 1;
        if (ph < 3) prologue(ph + 1);
        // ---- epilogue of phase ph, straight from the accumulators
        const long poff = (long)py * cp.Wc + px;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n4 = n0 + wn * (BN / 2) + j * 32 + 4 * cq;
            const bool ncol = n4 < p.N;
            const float s1 = es1[j], t1 = et1[j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float u = fmaf(acc[i][j][4 * q + k], s1, t1);
                        r[k] = fminf(fmaxf(fmaxf(u, lo), slope * u), hi);
                    }
                    quad_transpose4(r, li);
                    if (ncol && dst0[i][q] >= 0) store_nt16(p.C + (dst0[i][q] + poff) * p.ldc + n4, f32x4{r[0], r[1], r[2], r[3]});
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// ---------------------------------------------------------------------------------------------- convolutions on split32 input
namespace {

void set_taps(SplitConvParams& c, int n, const int* dy, const int* dx) {
    c.ntaps = n;
    c.dyp = c.dxp = 0;
    for (int t = 0; t < n; ++t) {
        c.dyp |= (unsigned long long)((dy ? dy[t] : 0) + 64) << (7 * t);
        c.dxp |= (unsigned long long)((dx ? dx[t] : 0) + 64) << (7 * t);
    }
}

int conv_checks(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1, const float* shift1,
                const float* scale2, const float* shift2, const float* res, int ldres, void* y, int ldy, int Cin, int Cout,
                int out_split) {
    EMD_REQUIRE(xs && whi && wlo && scale1 && shift1 && y, EMD_E_INVALID, "split32 conv: null pointer");
    EMD_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), EMD_E_INVALID, "split32 conv: scale2/shift2 must come together");
    EMD_REQUIRE(Cin >= 1 && Cin <= 2048 && Cout >= 4 && Cout % 4 == 0, EMD_E_INVALID, "split32 conv: 1 <= Cin <= 2048, Cout a multiple of 4");
    EMD_REQUIRE(ldx % 32 == 0 && ldx >= emd_split32_ld(Cin) && (reinterpret_cast<uintptr_t>(xs) & 127u) == 0, EMD_E_ALIGN,
                "split32 conv: xs 128-byte aligned, ldx a multiple of 32, >= ceil32(Cin)");
    if (out_split)
        EMD_REQUIRE(ldy % 32 == 0 && ldy >= emd_split32_ld(Cout) && (reinterpret_cast<uintptr_t>(y) & 127u) == 0, EMD_E_ALIGN,
                    "split32 conv: split32 output needs y 128-byte aligned, ldy a multiple of 32, >= ceil32(Cout)");
    else
        EMD_REQUIRE(ldy % 4 == 0 && ldy >= Cout && emd::aligned16(y), EMD_E_ALIGN, "split32 conv: ldy a multiple of 4, >= Cout; y 16-byte aligned");
    EMD_REQUIRE(!res || (ldres % 4 == 0 && ldres >= Cout && emd::aligned16(res)), EMD_E_ALIGN, "split32 conv: res alignment");
    EMD_REQUIRE(emd::aligned16(scale1) && emd::aligned16(shift1) && (!scale2 || (emd::aligned16(scale2) && emd::aligned16(shift2))) &&
                    emd::aligned16(whi) && emd::aligned16(wlo),
                EMD_E_ALIGN, "split32 conv: weight planes and scale/shift vectors must be 16-byte aligned");
    return EMD_OK;
}

int launch_conv(SplitConvParams& c, hipStream_t st, bool four = false) {
    SplitGemmParams& p = c.g;
    c.Cpad = (p.Cin + kBK - 1) / kBK * kBK;
    c.nkc = (p.Cin + SBK - 1) / SBK;
    p.Ktot = c.ntaps * c.Cpad;
    const int bn = p.N <= 64 ? 64 : 128;
    p.n_mtiles = (int)((p.M + 255) / 256);
    p.n_ntiles = (p.N + bn - 1) / bn;
    p.stamps = nullptr;
    p.nt = split_nt(0);
    const long nblk = (long)p.n_mtiles * p.n_ntiles;
    if (nblk <= 0 || nblk > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "split32 conv: grid too large");
    if (p.M > 0x7fffffffL || (!c.flat && (long)(p.M / ((long)c.Hg * c.Wg)) * c.Ha * c.Wa > 0x7fffffffL))
        return emd::fail(EMD_E_UNSUPPORTED, "split32 conv: more than 2^31 pixels");   // the kernel keeps pixel indices in 32 bits
    if (four && !c.out_split && !p.res && !p.scale2 && emd::g_knobs.deconv_direct == 2) {   // 128-row tiles, two workgroups per CU
        p.n_mtiles = (int)((p.M + 127) / 128);
        const long nb2 = (long)p.n_mtiles * p.n_ntiles;
        if (nb2 > 0x7fffffffL) return emd::fail(EMD_E_UNSUPPORTED, "split32 conv: grid too large");
        if (bn == 64) hipLaunchKernelGGL((deconv4_half_kernel<64>), dim3((unsigned)nb2), dim3(256), 0, st, c);
        else hipLaunchKernelGGL((deconv4_half_kernel<128>), dim3((unsigned)nb2), dim3(256), 0, st, c);
        return emd::check_launch("deconv4_half_kernel");
    }
    if (four && !c.out_split && !p.res && !p.scale2 && emd::g_knobs.deconv_direct) {   // round 3: epilogue from the registers, next phase's DMA first
        if (bn == 64) hipLaunchKernelGGL((deconv4_split_kernel<64>), dim3((unsigned)nblk), dim3(512), 0, st, c);
        else hipLaunchKernelGGL((deconv4_split_kernel<128>), dim3((unsigned)nblk), dim3(512), 0, st, c);
        return emd::check_launch("deconv4_split_kernel");
    }
    if (four) {
        if (bn == 64) hipLaunchKernelGGL((gemm_split_conv_kernel<64, true>), dim3((unsigned)nblk), dim3(512), 0, st, c);
        else hipLaunchKernelGGL((gemm_split_conv_kernel<128, true>), dim3((unsigned)nblk), dim3(512), 0, st, c);
    } else if (bn == 64) hipLaunchKernelGGL((gemm_split_conv_kernel<64>), dim3((unsigned)nblk), dim3(512), 0, st, c);
    else hipLaunchKernelGGL((gemm_split_conv_kernel<128>), dim3((unsigned)nblk), dim3(512), 0, st, c);
    return emd::check_launch("gemm_split_conv_kernel");
}

}  // namespace

extern "C" int emd_conv3x3_split32_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo, const float* scale1,
                                       const float* shift1, const float* scale2, const float* shift2, const float* res,
                                       int ldres, void* y, int ldy, int B, int H, int W, int Cin, int Cout, int stride,
                                       int rate, int act, int out_split, emd_stream_t stream) {
    int rc = conv_checks(xs, ldx, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, Cin, Cout, out_split);
    if (rc != EMD_OK) return rc;
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_conv3x3_split32_f32: bad shape");
    EMD_REQUIRE(stride == 1 || stride == 2, EMD_E_UNSUPPORTED, "emd_conv3x3_split32_f32: stride must be 1 or 2");
    EMD_REQUIRE(rate >= 1 && rate <= 31 && (rate == 1 || stride == 1), EMD_E_UNSUPPORTED,
                "emd_conv3x3_split32_f32: rate must be 1..31, and 1 when stride is 2");
    if (B == 0) return EMD_OK;
    if (stride == 1 && rate == 1 && !res && H >= 8 && B <= 65535 && 9L * W * ldy < (1L << 31)) {   // round 3: the patch-resident kernel (conv3_pipe.hip)
        emd::Conv3Params q{};
        q.x = static_cast<const unsigned char*>(xs); q.ldx_bytes = (long)ldx * 4; q.Whi = whi; q.Wlo = wlo; q.y = static_cast<float*>(y);
        q.scale1 = scale1; q.shift1 = shift1; q.scale2 = scale2; q.shift2 = shift2;
        q.H = H; q.W = W; q.Cin = (Cin + 31) / 32 * 32; q.Cpad = (Cin + kBK - 1) / kBK * kBK; q.Ktot = 9 * q.Cpad; q.N = Cout; q.ldy = ldy; q.act = act;
        if (emd::conv3_pipe_covers(q)) return emd::conv3_pipe_launch(q, B, out_split, static_cast<hipStream_t>(stream));
    }
    SplitConvParams c{};
    SplitGemmParams& p = c.g;
    p.A = static_cast<const unsigned char*>(xs); p.Whi = whi; p.Wlo = wlo; p.C = static_cast<float*>(y); p.res = res;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2;
    p.lda_bytes = (long)ldx * 4; p.N = Cout; p.Cin = Cin; p.ldc = ldy; p.ldres = ldres; p.act = act;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int eff = 2 * rate + 1;
    int pth = (Ho - 1) * stride + eff - H, ptw = (Wo - 1) * stride + eff - W;  // TF SAME: total padding
    if (pth < 0) pth = 0;
    if (ptw < 0) ptw = 0;
    const int pt = pth / 2, pl = ptw / 2;
    p.M = (long)B * Ho * Wo;
    c.flat = 0; c.out_split = out_split ? 1 : 0;
    c.Hg = Ho; c.Wg = Wo; c.Ha = H; c.Wa = W; c.Hc = Ho; c.Wc = Wo; c.sa = stride; c.sc = 1; c.py = c.px = 0;
    int dy[9], dx[9];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            dy[ky * 3 + kx] = ky * rate - pt;
            dx[ky * 3 + kx] = kx * rate - pl;
        }
    set_taps(c, 9, dy, dx);
    return launch_conv(c, static_cast<hipStream_t>(stream));
}

extern "C" int emd_deconv3x3s2_split32_f32(const void* xs, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4],
                                           const float* scale1, const float* shift1, void* y, int ldy, int B, int H, int W,
                                           int Cin, int Cout, int act, int out_split, emd_stream_t stream) {
    EMD_REQUIRE(whi && wlo, EMD_E_INVALID, "emd_deconv3x3s2_split32_f32: null weight table");
    for (int ph = 0; ph < 4; ++ph) {
        int rc = conv_checks(xs, ldx, whi[ph], wlo[ph], scale1, shift1, nullptr, nullptr, nullptr, 0, y, ldy, Cin, Cout, out_split);
        if (rc != EMD_OK) return rc;
    }
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_deconv3x3s2_split32_f32: bad shape");
    if (B == 0) return EMD_OK;
    for (int ph = 0; ph < 4; ++ph) {
        SplitConvParams c{};
        SplitGemmParams& p = c.g;
        int ky[4], kx[4];
        const int nt = emd_deconv_phase_taps(ph, ky, kx);
        p.A = static_cast<const unsigned char*>(xs); p.Whi = whi[ph]; p.Wlo = wlo[ph]; p.C = static_cast<float*>(y); p.res = nullptr;
        p.scale1 = scale1; p.shift1 = shift1; p.scale2 = p.shift2 = nullptr;
        p.lda_bytes = (long)ldx * 4; p.N = Cout; p.Cin = Cin; p.ldc = ldy; p.ldres = 0; p.act = act;
        p.M = (long)B * H * W;
        c.flat = 0; c.out_split = out_split ? 1 : 0;
        c.Hg = H; c.Wg = W; c.Ha = H; c.Wa = W; c.Hc = 2 * H; c.Wc = 2 * W; c.sa = 1; c.sc = 2;
        c.py = ph >> 1; c.px = ph & 1;
        int dy[4], dx[4];
        for (int t = 0; t < nt; ++t) {  // kernel index 2 reads the previous input sample
            dy[t] = ky[t] == 2 ? -1 : 0;
            dx[t] = kx[t] == 2 ? -1 : 0;
        }
        set_taps(c, nt, dy, dx);
        int rc = launch_conv(c, static_cast<hipStream_t>(stream));
        if (rc != EMD_OK) return rc;
    }
    return EMD_OK;
}

// The same transposed convolution as ONE launch (gemm_split_conv_kernel<BN, true>): each workgroup computes the four output
// phases of its 256 input pixels back to back, so the input is fetched from HBM once instead of once per phase launch (the 9 taps'
// DMA re-reads hit L2).  Same products in the same order as the four-launch form: bit-identical results.
// Where graph hosts should take the one-launch form (emd_deconv3x3s2_fused_split32_f32) rather than the register-staged four-phase GEMM:
// always where the patch-resident kernel covers the layer (H % 8 == 0, W % 32 == 0, Cin % 32 == 0: it sums in another order than the GEMM
// forms, so the choice must not depend on the batch size -- image b of a batch == the image alone, bit for bit), otherwise from 192
// row tiles on (the GEMM forms agree with each other bit for bit, there the choice is speed only).
// ONE predicate for "this layer runs on the patch-resident kernel" (deconv_pipe.hip), used by emd_deconv3x3s2_fused_preferred and by the
// entry point alike, and a function of the LAYER only (H, W, Cin, Cout) -- never of the batch size or of the pitches: the kernel sums in
// another order than the GEMM forms, so a route that flipped with B (or with the buffer a tensor happens to live in) would break "image b
// of a batch == the image alone, bit for bit".  What the kernel additionally needs of a call (32-bit in-image offsets: H * W * ldx_bytes <
// 2^32, 36 * W * ldy < 2^31) is checked by the entry point and REPORTED (EMD_E_UNSUPPORTED) instead of silently falling back to a kernel
// with other bits; batches beyond the grid's 65535 images are cut into launches of the same kernel.
static bool deconv_patch_route(int H, int W, int Cin, int Cout) {
    emd::DeconvPipeParams q{};
    q.H = H; q.W = W; q.Cin = (Cin + 31) / 32 * 32; q.N = Cout;   // (ldx_bytes = 0: the offset bound is the entry point's to check)
    return Cin % 32 == 0 && H >= 8 && emd::deconv_pipe_covers(q);
}

extern "C" int emd_deconv3x3s2_fused_preferred(int B, int H, int W, int Cin, int Cout) {
    if (deconv_patch_route(H, W, Cin, Cout)) return 1;
    return (long)B * H * W >= 256L * 192 ? 1 : 0;
}

extern "C" int emd_deconv3x3s2_fused_split32_f32(const void* xs, int ldx, const uint16_t* const whi[4], const uint16_t* const wlo[4],
                                                 const float* scale1, const float* shift1, void* y, int ldy, int B, int H, int W,
                                                 int Cin, int Cout, int act, int out_split, emd_stream_t stream) {
    EMD_REQUIRE(whi && wlo, EMD_E_INVALID, "emd_deconv3x3s2_fused_split32_f32: null weight table");
    for (int ph = 0; ph < 4; ++ph) {
        int rc = conv_checks(xs, ldx, whi[ph], wlo[ph], scale1, shift1, nullptr, nullptr, nullptr, 0, y, ldy, Cin, Cout, out_split);
        if (rc != EMD_OK) return rc;
    }
    EMD_REQUIRE(B >= 0 && H >= 1 && W >= 1, EMD_E_INVALID, "emd_deconv3x3s2_fused_split32_f32: bad shape");
    if (B == 0) return EMD_OK;
    if (Cin % 32 == 0 && deconv_patch_route(H, W, Cin, Cout)) {   // the patch-resident kernel (deconv_pipe.hip), dev knob deconv_direct = 3
        emd::DeconvPipeParams q{};
        q.x = static_cast<const unsigned char*>(xs); q.ldx_bytes = (long)ldx * 4; q.y = static_cast<float*>(y);
        for (int ph = 0; ph < 4; ++ph) { q.Whi[ph] = whi[ph]; q.Wlo[ph] = wlo[ph]; }
        q.scale1 = scale1; q.shift1 = shift1;
        q.H = H; q.W = W; q.Cin = (Cin + 31) / 32 * 32; q.Cpad = (Cin + kBK - 1) / kBK * kBK; q.N = Cout; q.ldy = ldy; q.act = act;
        EMD_REQUIRE(emd::deconv_pipe_covers(q) && 36L * W * ldy < (1L << 31), EMD_E_UNSUPPORTED,
                    "emd_deconv3x3s2_fused_split32_f32: this layer runs on the patch-resident kernel, whose in-image offsets are 32-bit "
                    "(H * W * ldx * 4 < 2^32, 36 * W * ldy < 2^31): pitch too large (a fallback would change the summation order)");
        const long in_img = (long)H * W * q.ldx_bytes, out_img = 4L * H * W * ldy * (long)sizeof(float);
        for (int b0 = 0; b0 < B; b0 += 65535) {   // grid.z <= 65535: the same kernel on slices of the batch
            const int nb = B - b0 < 65535 ? B - b0 : 65535;
            q.x = static_cast<const unsigned char*>(xs) + (long)b0 * in_img;
            q.y = reinterpret_cast<float*>(static_cast<unsigned char*>(y) + (long)b0 * out_img);
            int rc = emd::deconv_pipe_launch(q, nb, out_split, static_cast<hipStream_t>(stream));
            if (rc != EMD_OK) return rc;
        }
        return EMD_OK;
    }
    SplitConvParams c{};
    SplitGemmParams& p = c.g;
    p.A = static_cast<const unsigned char*>(xs); p.Whi = whi[0]; p.Wlo = wlo[0]; p.C = static_cast<float*>(y); p.res = nullptr;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = p.shift2 = nullptr;
    p.lda_bytes = (long)ldx * 4; p.N = Cout; p.Cin = Cin; p.ldc = ldy; p.ldres = 0; p.act = act;
    p.M = (long)B * H * W;
    c.flat = 0; c.out_split = out_split ? 1 : 0;
    c.Hg = H; c.Wg = W; c.Ha = H; c.Wa = W; c.Hc = 2 * H; c.Wc = 2 * W; c.sa = 1; c.sc = 2; c.py = c.px = 0;
    for (int ph = 0; ph < 4; ++ph) {
        int ky[4], kx[4], dy[4], dx[4];
        const int nt = emd_deconv_phase_taps(ph, ky, kx);
        for (int t = 0; t < nt; ++t) {  // kernel index 2 reads the previous input sample
            dy[t] = ky[t] == 2 ? -1 : 0;
            dx[t] = kx[t] == 2 ? -1 : 0;
        }
        set_taps(c, nt, dy, dx);
        c.Whi4[ph] = whi[ph]; c.Wlo4[ph] = wlo[ph]; c.ntaps4[ph] = nt; c.dyp4[ph] = c.dyp; c.dxp4[ph] = c.dxp;
    }
    c.ntaps = 4;   // launch_conv derives Ktot from it; the kernel uses the per-phase counts
    return launch_conv(c, static_cast<hipStream_t>(stream), true);
}

extern "C" int emd_conv1x1_split32_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo,
                                       const float* scale1, const float* shift1, const float* scale2,
                                       const float* shift2, const float* res, int ldres, float* y, int ldy, long M,
                                       int Cin, int Cout, int act, emd_stream_t stream) {
    return conv1x1_split32_impl(xs, ldx, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, y, ldy, M, Cin, Cout, act, stream, nullptr);
}

// The same GEMM that also delivers the per-channel batch mean and biased variance of its OUTPUT y (what emd_bn_stats_f32 would
// compute in a second pass over y): the epilogue leaves one double partial per (256-row tile, channel), a fixed-order final
// reduction follows (deterministic).  workspace: emd_conv1x1_split32_stats_workspace_bytes(M, Cout) bytes, 8-byte aligned.
// The same GEMM writing y as a split32 tensor (pitch ldy 4-byte units, % 32; channels Cout..ceil32(Cout) zero): the producer
// of a split32 convolution's input (graph D: deconv2_b -> deconv2to1) then writes no fp32 activation and needs no converter pass.
extern "C" int emd_conv1x1_split32_out_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo,
                                           const float* scale1, const float* shift1, const float* scale2,
                                           const float* shift2, const float* res, int ldres, void* y, int ldy, long M,
                                           int Cin, int Cout, int act, emd_stream_t stream) {
    return conv1x1_split32_impl(xs, ldx, whi, wlo, scale1, shift1, scale2, shift2, res, ldres, static_cast<float*>(y), ldy, M, Cin, Cout,
                                act, stream, nullptr, 1);
}

extern "C" size_t emd_conv1x1_split32_stats_workspace_bytes(long M, int Cout) {
    if (M < 1 || Cout < 1) return 0;
    return (size_t)((M + 255) / 256) * 2 * (size_t)Cout * sizeof(double);
}

extern "C" int emd_conv1x1_split32_stats_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo,
                                             const float* scale1, const float* shift1, float* y, int ldy, long M, int Cin,
                                             int Cout, int act, float* mean, float* var, void* workspace, emd_stream_t stream) {
    EMD_REQUIRE(mean && var && workspace && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EMD_E_INVALID,
                "emd_conv1x1_split32_stats_f32: mean, var and an 8-byte aligned workspace are required");
    EMD_REQUIRE(M >= 1, EMD_E_INVALID, "emd_conv1x1_split32_stats_f32: M >= 1");
    int rc = conv1x1_split32_impl(xs, ldx, whi, wlo, scale1, shift1, nullptr, nullptr, nullptr, 0, y, ldy, M, Cin, Cout, act, stream,
                                  static_cast<double*>(workspace));
    if (rc != EMD_OK) return rc;
    return emd::launch_bn_stats_final(static_cast<const double*>(workspace), (int)((M + 255) / 256), Cout, M, mean, var,
                                      static_cast<hipStream_t>(stream));
}

// ... and folds the batch norm that uses those statistics in the same final-reduction launch: scale = gamma / sqrt(var + eps)
// (gamma NULL = 1), shift = beta - mean * scale (beta NULL = 0) -- emd_bn_fold_f32's arithmetic on the same float mean / var.
extern "C" int emd_conv1x1_split32_stats_fold_f32(const void* xs, int ldx, const uint16_t* whi, const uint16_t* wlo,
                                                  const float* scale1, const float* shift1, float* y, int ldy, long M, int Cin,
                                                  int Cout, int act, float* mean, float* var, void* workspace, const float* gamma,
                                                  const float* beta, float eps, float* scale, float* shift, emd_stream_t stream) {
    EMD_REQUIRE(mean && var && scale && shift && workspace && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, EMD_E_INVALID,
                "emd_conv1x1_split32_stats_fold_f32: mean, var, scale, shift and an 8-byte aligned workspace are required");
    EMD_REQUIRE(M >= 1, EMD_E_INVALID, "emd_conv1x1_split32_stats_fold_f32: M >= 1");
    int rc = conv1x1_split32_impl(xs, ldx, whi, wlo, scale1, shift1, nullptr, nullptr, nullptr, 0, y, ldy, M, Cin, Cout, act, stream,
                                  static_cast<double*>(workspace));
    if (rc != EMD_OK) return rc;
    return emd::launch_bn_stats_final(static_cast<const double*>(workspace), (int)((M + 255) / 256), Cout, M, mean, var,
                                      static_cast<hipStream_t>(stream), gamma, beta, eps, scale, shift);
}

// dev hooks (not in the header): kernel variant and stamp buffer for tools/gemm_split_bench.py
extern "C" void emd_debug_split_variant(int v) { g_variant_override = v; }
extern "C" void emd_debug_split_stamps(void* buf) { g_stamps = static_cast<long long*>(buf); }
