// Dense 3x3 convolution (stride 1, rate 1, TF SAME) from a split32 input with the input PATCH resident in LDS across the nine taps.
// replaces: tf.layers.conv2d(k = 3) + bias -> relu -> batch norm -> relu = conv_block of misc_py/modified_Xception.py:215-229 for the
//           decoder's narrow layers (:538-621: 128 -> 64 and 64 -> 64 at full resolution), reached through emd_conv3x3_split32_f32.
//
// Why: gemm_split_conv_kernel<64> brings a tap's 256 A rows into LDS for every (tap, 32-channel step) -- nine times the input per tile,
// 40 KB of DMA writes per K step for 64 output columns -- and a 64 x 32 wave tile reads 1 KB of fragments per MFMA: 1.4 KB of LDS
// traffic per MFMA against the 1 KB/MFMA the LDS pipe can feed at the matrix cores' issue rate; the two 512^2 layers of graph X ran
// at 0.27 of 2.5 PFLOP/s issued.  Here the (8+2) x (32+2) pixel patch of a 32-channel chunk (already bf16 hi | lo lines of 128 B: the
// split32 layout) is brought in ONCE by LDS-DMA and the nine taps read their A fragments from it at shifted slots; only the weights
// stream per tap: 0.8 KB per MFMA.
//
// Workgroup = 512 threads = 8 waves, one per CU; output tile 8 x 32 pixels x 64 columns; wave = 32 pixels (one tile row) x 64 columns.
// Wider layers run as column tiles of 64 (neighbours in the launch order share their patch through L2): measured on graph X's decoder
// (tools/conv3_bench.py, [32, ., ., .]) 128 -> 64 @512^2 4616 -> 3471 us, 64 -> 64 2911 -> 1876, 192 -> 128 @256^2 2698 -> 2461,
// 128 -> 128 1837 -> 1656, 256 -> 192 @128^2 1575 -> 1206, 192 -> 192 1212 -> 905; 256 columns: 621 -> 646 (not dispatched).
// LDS: patch ring 2 x 44 KiB (352 slots of 128 B; slot = one pixel x 32 channels, 16-byte pieces XOR-swizzled by (slot >> 1) & 7 on the
// DMA's source side so that the 32 consecutive slots of a fragment read cover the banks evenly) + weight ring 2 x 24 KiB (3 taps x 64
// rows x [32 hi | 32 lo]) = 136 KiB.  A step = (chunk, tap row): 3 taps x 2 x 6 MFMAs per wave, one barrier.
// Summation order per output element: chunk-major, taps inside -- NOT the order of gemm_split_conv_kernel (tap-major): same error
// class (split-bf16, fp32 accumulate), different last bits; the dispatch depends on the layer's shape only.
// Epilogue from the accumulators (quad transpose, 16-byte non-temporal stores), fp32 or split32 output, two-stage affine.
#include <type_traits>

#include "conv3_params.hpp"

namespace {

using namespace emd;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(128))) unsigned char g_zero_c3[16384];   // padding pixels: "+ chunk offset" stays inside for Cin <= 4064

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
// Fragment reads by hand: the compiler neither sees them nor waits for them (it would wait with lgkmcnt(0), i.e. also for the NEXT
// unit's reads issued behind them); the wait in front of a unit's MFMAs lets the younger reads stay in flight and ties the registers.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(const unsigned char* p) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((lptr_t)p), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void store_nt_d(const void* sbase, unsigned voff, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 3" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store_nt_s(const void* sbase, unsigned voff, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 3" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
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

template <bool OSPLIT, int EPI>      // EPI: dwords a lane stores at a time (1; 4 behind dev knob epi_width)
__global__ __launch_bounds__(512, 1) void conv3_pipe_kernel(const Conv3Params p) {
    constexpr int BN = 64, NW = 8, TW = 32, TH = 8;
    constexpr int PW = TW + 2, PH = TH + 2, PWS = 35, NPATCH = PH * PWS;   // 350 slots, 2 spare
    constexpr int NPIECE = (NPATCH + 7) / 8, PP = (NPIECE + NW - 1) / NW;  // 44 pieces of 1 KiB, 6 per wave (waves 4-7 repeat one)
    constexpr int STAGE = NPIECE * 1024;                                   // 45056
    constexpr int B_ONE = 3 * BN * 128, PB = 3 * BN / 8 / NW;              // 24576 B per (chunk, tap row): 24 pieces, 3 per wave
    constexpr int B_OFF = 2 * STAGE;
    constexpr int TN = 2, E = 16 * TN / EPI;                               // 32-column MFMA tiles / stores per wave and tile
    __shared__ __attribute__((aligned(1024))) unsigned char smem[B_OFF + 2 * B_ONE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD k takes the k-th contiguous eighth of the tile list: halo rows meet in one L2
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t = (id & 7) * (total >> 3) + (id >> 3);
        if ((total & 7) == 0) {
            bx = t % gridDim.x;
            by = (t / gridDim.x) % gridDim.y;
            bz = t / (gridDim.x * gridDim.y);
        }
    }
    // the column tiles (64 output channels each) of one pixel tile are neighbours in the launch order: the patch they share comes from L2
    const int n0 = (bx % p.n_ntiles) * BN;
    bx /= p.n_ntiles;
    const int xbase = bx * p.tpw * TW, y0 = by * TH;
    const long img = (long)bz * p.H * p.W;

    // ---- DMA sources.  Lane l of piece q fills 16-byte piece (l & 7) of slot 8 q + (l >> 3); it holds the slot's LOGICAL piece
    // (l & 7) ^ ((slot >> 1) & 7) (0-3: hi words of channels 0-31 of the chunk, 4-7: lo words)
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
            const bool real = slot < NPATCH && px < PW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const int kk = dk ^ ((slot >> 1) & 7);
            const unsigned char* o = g_zero_c3 + kk * 16;
            const unsigned char* o_px = p.x + (img + (long)gy * p.W + gx) * p.ldx_bytes + kk * 16;
            psrc[j] = real ? o_px : o;
            pmove |= real ? 1u << j : 0u;
        }
    };
    auto issue_patch = [&](int stage, int c) {   // chunk c: 128 bytes further on in every pixel's line
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            int q = wv + NW * j;
            if (q >= NPIECE) q -= NW;
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[j] + c * 128), (lptr_t)(smem + stage * STAGE + q * 1024), 16, 0, 0);
        }
    };
    // weight rows of a step: row = tap-in-row * 64 + output channel; 16-byte pieces XOR-swizzled by (row >> 1) & 7
    const uint16_t* bsrc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = (wv * PB + j) * 8 + drow;     // 0 .. 191
        const int kx = row >> 6, n = row & 63;
        const int c = dk ^ ((row >> 1) & 7);
        const uint16_t* plane = (c & 4) ? p.Wlo : p.Whi;
        bsrc[j] = plane + (long)(n0 + n) * p.Ktot + kx * p.Cpad + (c & 3) * 8;
    }
    auto issue_B = [&](int buf, int c, int ky) {
        const int off = ky * 3 * p.Cpad + c * 32;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[j] + off), (lptr_t)(smem + B_OFF + buf * B_ONE + (wv * PB + j) * 1024), 16, 0, 0);
    };

    // ---- fragment addressing.  A: lane fr = pixel fr of the wave's tile row (wave wv = tile row wv), 8 consecutive k per lane
    // (logical piece ks * 2 + fh for hi, + 4 for lo).  Patch slot of tap (ky, kx): (wv + ky) * PWS + fr + kx.
    const int fr = lane & 31, fh = lane >> 5;
    // (slot >> 1) & 7 is not additive in the tap offsets (a patch row is 35 slots), so every tap gets its own byte offset: 9 registers
    int a_off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int slot = (wv + t / 3) * PWS + fr + t % 3;
        a_off[t] = slot * 128 + ((fh ^ ((slot >> 1) & 7)) << 4);
    }
    const int sw = (fr >> 1) & 7;
    // the four (K half, hi / lo) variants of the lane's weight-fragment address (bits 5 and 6); + buffer; kx * 8192 + j * 4096 ride in the
    // read's offset field
    const unsigned char* b_var[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) b_var[v] = smem + ((B_OFF + fr * 128 + ((fh ^ sw) << 4)) ^ (v << 5));

    // ---- epilogue constants: before the transpose a lane holds channel j * 32 + fr
    float es1[TN], et1[TN], es2[TN], et2[TN];
    const bool two = p.scale2 != nullptr;
    const bool full = n0 + BN <= p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + fr;
        const bool valid = n < p.N;
        es1[j] = valid ? p.scale1[n] : 0.f;
        et1[j] = valid ? p.shift1[n] : 0.f;
        es2[j] = (valid && two) ? p.scale2[n] : 1.f;
        et2[j] = (valid && two) ? p.shift2[n] : 0.f;
        asm volatile("" ::"v"(es1[j]), "v"(et1[j]), "v"(es2[j]), "v"(et2[j]));
    }
    const float hi = p.act == 1 ? 6.f : __builtin_inff();
    const float hi2 = p.act == 2 ? __builtin_inff() : 6.f;
    const float slope = p.act == 4 ? 0.2f : 1.f, lo = (p.act == 1 || p.act == 2) ? 0.f : -__builtin_inff();

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int nchunks = p.Cin / 32;              // (the host pads: split32 tensors are multiples of 32 channels wide, padding zero)
    const int nsteps = 3 * nchunks;              // per tile
    const int total = p.tpw * nsteps;
    // issue stream of the patches: chunk ic of the tile at ixt (clamped at the very last chunk: surplus groups re-read it)
    int ic = 0, ixt = xbase, ichunk = 0;
    const int tchunks = p.tpw * nchunks;
    set_tile(xbase);
    auto advance_patch = [&]() {
        if (ichunk + 1 >= tchunks) return;
        ++ichunk;
        if (++ic == nchunks) {
            ic = 0;
            const int xn = ixt + TW;
            if (ixt >= 1 && xn + TW + 1 <= p.W) {
                const long step = (long)TW * p.ldx_bytes;
#pragma unroll
                for (int j = 0; j < PP; ++j) psrc[j] += ((pmove >> j) & 1) ? step : 0;
            } else {
                set_tile(xn);
            }
            ixt = xn;
        }
    };

    // prologue: patch(0), B(0); the loop's step u issues B(u+1) and, on a chunk's first tap row, patch(chunk + 1) AFTER it
    issue_patch(0, 0);
    issue_B(0, 0, 0);

    int x0 = xbase;
    int c = 0, ky = 0;     // the step being computed
    bool epi = false;      // an epilogue ran at the end of the previous step (its E stores are the youngest entries of the queue)
    for (int u = 0; u < total; ++u) {
        // B(u) -- and with it everything older: the patch of this chunk -- has landed for this wave; the groups issued after B(u)
        // may stay in flight: the patch of the next chunk when the previous step was a chunk's first (ky == 1 now), the E stores of
        // an epilogue at the end of the previous step
        if (ky == 1) wait_vm<PP>();
        else if (epi && full) wait_vm<E>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        {   // next step's weights into the other buffer (everybody is done with step u - 1's)
            int cn = c, kn = ky + 1;
            if (kn == 3) { kn = 0; cn = c + 1 == nchunks ? 0 : c + 1; }
            if (u + 1 >= total) { cn = c; kn = ky; }
            issue_B((u + 1) & 1, cn, kn);
        }
        if (ky == 0) {      // the next chunk's patch into the other stage (its last reader passed this barrier); beyond the last chunk
            const bool more = ichunk + 1 < tchunks;   // of the workgroup: a re-read of the current one, into the OTHER stage all the same
            advance_patch();
            issue_patch(more ? (ichunk & 1) : ((ichunk + 1) & 1), ic);
        }
        const unsigned char* stg = smem + ((u / 3) & 1) * STAGE;
        {   // the step's 36 MFMAs as six units (kx, K half) of six; unit i + 1's twelve fragment registers are read BEFORE unit i's MFMAs
            // are issued (two register sets): a read's latency passes under matrix work instead of in front of every MFMA
            const unsigned char* bv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) bv[v] = b_var[v] + (u & 1) * B_ONE;
            int ao3[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) ao3[kx] = ky == 0 ? a_off[kx] : (ky == 1 ? a_off[3 + kx] : a_off[6 + kx]);
            struct Frag { bf16x8 ah, al, bh[TN], bl[TN]; };
            static_assert(TN == 2, "two column tiles per wave");
            auto load_unit = [&](auto U_) {
                constexpr int kx = decltype(U_)::value / 2, ks = decltype(U_)::value % 2;
                Frag f;
                f.ah = lds_read16<0>(stg + (ao3[kx] ^ (ks << 5)));
                f.al = lds_read16<0>(stg + (ao3[kx] ^ (ks << 5) ^ 64));
                f.bh[0] = lds_read16<kx * (BN * 128)>(bv[ks]);
                f.bl[0] = lds_read16<kx * (BN * 128)>(bv[ks + 2]);
                f.bh[1] = lds_read16<kx * (BN * 128) + 4096>(bv[ks]);
                f.bl[1] = lds_read16<kx * (BN * 128) + 4096>(bv[ks + 2]);
                return f;
            };
            auto mma_unit = [&](Frag& f, auto YOUNGER_) {
                asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.ah), "+v"(f.al), "+v"(f.bh[0]), "+v"(f.bl[0]), "+v"(f.bh[1]), "+v"(f.bl[1]) : "n"(decltype(YOUNGER_)::value));
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bl[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[j], acc[j], 0, 0, 0);
            };
            using std::integral_constant;
            typedef integral_constant<int, 6> Six;
            Frag f0 = load_unit(integral_constant<int, 0>{}), f1;
            f1 = load_unit(integral_constant<int, 1>{});
            mma_unit(f0, Six{});
            __builtin_amdgcn_sched_barrier(0);
            f0 = load_unit(integral_constant<int, 2>{});
            mma_unit(f1, Six{});
            __builtin_amdgcn_sched_barrier(0);
            f1 = load_unit(integral_constant<int, 3>{});
            mma_unit(f0, Six{});
            __builtin_amdgcn_sched_barrier(0);
            f0 = load_unit(integral_constant<int, 4>{});
            mma_unit(f1, Six{});
            __builtin_amdgcn_sched_barrier(0);
            f1 = load_unit(integral_constant<int, 5>{});
            mma_unit(f0, Six{});
            __builtin_amdgcn_sched_barrier(0);
            mma_unit(f1, integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        epi = false;
        if (++ky == 3) {
            ky = 0;
            if (++c == nchunks) {
                c = 0;
                epi = true;
                // ---- epilogue from the accumulators: C/D layout col = lane & 31 (channel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) = pixel
                // of the tile row.  A lane stores its channel's sixteen pixels as they are, one dword each: the 32 lanes of a half wave write
                // the 128 contiguous bytes of a pixel's channel group (see deconv_pipe.hip; EPI = 4 behind dev knob epi_width is the older
                // form -- a 4 x 4 transpose inside lane quads, then 16 bytes per lane).  split32 output: two pixels are split together,
                // the (even, odd) channel pair trades halves, the even lane stores hi (c, c + 1), the odd lane lo (c - 1, c).
                int ldo = p.ldy;
                asm volatile("" : "+s"(ldo));
                const long pixr = img + (long)(y0 + wv) * p.W + x0;
                float* obase = p.y + pixr * ldo;
                auto body = [&](auto LEAKY_, auto TWO_) {
                    constexpr bool LEAKY = decltype(LEAKY_)::value, TWO = decltype(TWO_)::value;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float s1 = es1[j], t1 = et1[j], s2 = es2[j], t2 = et2[j];
                        auto act = [&](float a) {
                            float v = fmaf(a, s1, t1);
                            if constexpr (LEAKY) v = fmaxf(v, slope * v);
                            v = __builtin_amdgcn_fmed3f(v, lo, hi);
                            if constexpr (TWO) v = __builtin_amdgcn_fmed3f(fmaf(v, s2, t2), 0.f, hi2);
                            return v;
                        };
                        if constexpr (EPI == 1) {
                            const int n = n0 + j * 32 + fr;
                            const bool odd = fr & 1;
                            unsigned voff;
                            bool live;
                            if constexpr (OSPLIT) {
                                voff = (unsigned)(4 * fh * ldo) * 4u + (n >> 5) * 128u + (odd ? 64u + 2u * ((n & 31) - 1) : 2u * (n & 31));
                                live = n < ((p.N + 31) & ~31);     // (scale = shift = 0 past N: the padding of the last channel group is zeros)
                            } else {
                                voff = (unsigned)(4 * fh * ldo + n) * 4u;
                                live = n < p.N;
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
#pragma unroll
                                for (int k = 0; k < 4; k += 2) {
                                    const float r0 = act(acc[j][4 * q + k]), r1 = act(acc[j][4 * q + k + 1]);
                                    float* ob0 = obase + (8 * q + k) * ldo;
                                    float* ob1 = ob0 + ldo;
                                    if constexpr (!OSPLIT) {
                                        if (live) {
                                            store_nt_d(ob0, voff, __builtin_bit_cast(unsigned, r0));
                                            store_nt_d(ob1, voff, __builtin_bit_cast(unsigned, r1));
                                        }
                                    } else {
                                        unsigned h, l;                                 // (pixel k | pixel k + 1) halves of this channel
                                        split2(r0, r1, h, l);
                                        const unsigned got = swap_pair(odd ? h : l);   // even lane: the odd channel's hi pair; odd lane: the even channel's lo pair
                                        const unsigned first = odd ? got : h, second = odd ? l : got;
                                        if (live) {
                                            store_nt_d(ob0, voff, __builtin_amdgcn_perm(second, first, 0x05040100u));
                                            store_nt_d(ob1, voff, __builtin_amdgcn_perm(second, first, 0x07060302u));
                                        }
                                    }
                                }
                            }
                        } else {
                            const int li = fr & 3, cq = fr >> 2;
                            const int n4 = n0 + j * 32 + 4 * cq;
                            const bool valid = n4 < p.N;
                            unsigned voff;
                            if constexpr (OSPLIT) voff = (unsigned)((4 * fh + li) * ldo) * 4u + (n4 >> 5) * 128u + ((cq & 1) ? 64u : 0u) + ((n4 & 31) >> 3) * 16u;
                            else voff = (unsigned)((4 * fh + li) * ldo + n4) * 4u;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float r[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) r[k] = act(acc[j][4 * q + k]);
                                quad_transpose(r, li);
                                f32x4 v = f32x4{r[0], r[1], r[2], r[3]};
                                float* ob = obase + (8 * q) * ldo;
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
                    }
                };
                if (p.act == 4) { if (two) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
                else { if (two) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
                x0 += TW;
            }
        }
    }
    wait_vm<0>();
}

}  // namespace

namespace emd {

bool conv3_pipe_covers(const Conv3Params& p) {
    // (column tiles of 64: the packed weight planes are padded to a multiple of 128 rows, so a last partial tile stays inside them)
    return g_knobs.conv3_pipe && p.H % 8 == 0 && p.W % 32 == 0 && p.Cin % 32 == 0 && p.Cin >= 32 && p.Cin <= 4064 && p.N % 4 == 0 &&
           p.N <= (g_knobs.conv3_pipe >= 2 ? 1024 : 256);   // measured (tools/conv3_bench.py): faster up to 192 columns, 3-4 % at 256
}

int conv3_pipe_launch(const Conv3Params& p, int B, int out_split, hipStream_t st) {
    Conv3Params q = p;
    const int tiles_w = p.W / 32;
    q.n_ntiles = (p.N + 63) / 64;
    const long wgs1 = (long)tiles_w * (p.H / 8) * B * q.n_ntiles;
    int tpw = 1;
    for (int t : {8, 4, 2})
        if (tiles_w % t == 0 && wgs1 / t >= 1024) { tpw = t; break; }
    if (g_knobs.sep_tpw > 0 && tiles_w % g_knobs.sep_tpw == 0) tpw = g_knobs.sep_tpw;   // dev knob (shared with the separable kernels)
    q.tpw = tpw;
    const dim3 grid(tiles_w / tpw * q.n_ntiles, p.H / 8, B);
    // epilogue: per-channel dword stores; four or more column tiles (>= 256 columns) do better with the transposed 16-byte form
    if (g_knobs.epi_width ? g_knobs.epi_width == 4 : q.n_ntiles >= 4) {
        if (out_split) hipLaunchKernelGGL((conv3_pipe_kernel<true, 4>), grid, dim3(512), 0, st, q);
        else hipLaunchKernelGGL((conv3_pipe_kernel<false, 4>), grid, dim3(512), 0, st, q);
    } else {
        if (out_split) hipLaunchKernelGGL((conv3_pipe_kernel<true, 1>), grid, dim3(512), 0, st, q);
        else hipLaunchKernelGGL((conv3_pipe_kernel<false, 1>), grid, dim3(512), 0, st, q);
    }
    return emd::check_launch("conv3_pipe_kernel");
}

}  // namespace emd
