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

bool conv3_pipe_covers(const Conv3Params& p);
int conv3_pipe_launch(const Conv3Params& p, int B, int out_split, hipStream_t st);

}  // namespace emd
