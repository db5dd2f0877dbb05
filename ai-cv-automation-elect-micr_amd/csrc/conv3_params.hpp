// Parameter block of the patch-resident dense 3x3 convolution (conv3_pipe.hip), shared with its dispatcher (gemm_split.hip).
#pragma once

#include "mfma_common.hpp"

namespace emd {

struct Conv3Params {
    const unsigned char* x;   // split32 input [B,H,W, ceil32(Cin)]: per pixel and 32-channel group one 128-byte line [32 hi | 32 lo]
    long ldx_bytes;           // pixel pitch
    const uint16_t* Whi;      // packed weights [Npad][9][Cpad], K-contiguous per (output channel, tap)
    const uint16_t* Wlo;
    float* y;                 // fp32 [B,H,W,N] (pitch ldy floats) or, with the OSPLIT instance, a split32 tensor (pitch ldy 4-byte units)
    const float* scale1;
    const float* shift1;
    const float* scale2;      // second affine + relu (conv_block's batch norm), or null
    const float* shift2;
    int H, W, Cin, Cpad, Ktot, N, ldy, act;
    int tpw;                  // tiles per workgroup, side by side along W
    int n_ntiles;             // column tiles of 64 output channels
};

// the patch-resident transposed conv (deconv_pipe.hip)
struct DeconvPipeParams {
    const unsigned char* x;   // split32 input [B,H,W, ceil32(Cin)]
    long ldx_bytes;
    const uint16_t* Whi[4];   // per output phase 2 py + px: packed weights [Npad][taps of the phase: 4, 2, 2, 1][Cpad]
    const uint16_t* Wlo[4];
    float* y;                 // [B,2H,2W,N] fp32 (pitch ldy floats) or split32 (pitch ldy 4-byte units)
    const float* scale1;
    const float* shift1;
    int H, W, Cin, Cpad, N, ldy, act;
    int tpw, n_ntiles;
    int ablate;               // dev (knob sep_ablate; results wrong on purpose): 2 no MFMAs, 4 no stores, 8 no patch DMA, 16 no weight DMA
};

bool deconv_pipe_covers(const DeconvPipeParams& p);
int deconv_pipe_launch(const DeconvPipeParams& p, int B, int out_split, hipStream_t st);

bool conv3_pipe_covers(const Conv3Params& p);
int conv3_pipe_launch(const Conv3Params& p, int B, int out_split, hipStream_t st);

}  // namespace emd
