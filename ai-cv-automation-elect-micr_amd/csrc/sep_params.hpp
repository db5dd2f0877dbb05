// Parameter block shared by the fused separable-convolution kernels (sep_fused.hip: register-staged loader, any W % 16 == 0;
// sep_pipe.hip: LDS-DMA ring, W % 32 == 0) and the dev knobs that choose between them.
#pragma once

#include "mfma_common.hpp"

namespace emd {

struct SepParams {
    const float* x;       // [B,H,W,Cin] pixel stride ldx
    const float* dw;      // [9][Cin]
    const uint16_t* Whi;  // [Npad][Cpad]
    const uint16_t* Wlo;
    float* y;             // [B,H,W,N] pixel stride ldy
    const float* res;
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    int H, W, Cin, Cpad, N;
    int ldx, ldy, ldres, act;
    int stride;           // 1, or 2 (sep_pipe.hip only: TF SAME on even sizes; H, W are the INPUT sizes, the output is H/2 x W/2)
    int reflect;          // 1: the patch border is tf.pad(REFLECT) of the image (graph G), 0: zero (TF SAME)
    int tpw;              // output tiles per workgroup, side by side along W
    // generated input (layers fed by a 1-channel image): x is a one-value-per-pixel tensor d (pitch ldx) and the Cin-channel
    // input the depthwise stage sees is act(d * gen_a[c] + gen_t[c]) -- never written to memory
    const float* gen_a;
    const float* gen_t;
    int gen_act;
    // second output of the DUAL instances (emd_sep3x3_dual_f32): y2 = relu6(x * W2 * scale_b + shift_b), a 1x1 conv of the block's
    // INPUT -- the decoder's residual projection (denoiser.py:359/:371/:383), which reads the same tensor as the separable conv
    const uint16_t* W2hi;
    const uint16_t* W2lo;
    float* y2;
    const float* scale_b;
    const float* shift_b;
    int N2, ldy2;
    int out_split;        // y is a split32 tensor (pitch ldy 4-byte units; N % 32 == 0): the consumer is a split32 GEMM
    long long* stamps;    // dev hook: per-workgroup phase cycle sums (NULL otherwise)
    int nt;               // outputs leave with non-temporal stores (they are not re-read by this launch: keep L2 for the patch halos)
    int ablate;           // dev: sep_pipe phase ablation bits (0 in every product launch)
    int xcd;              // workgroup -> tile map that gives each XCD (workgroup id mod 8) one contiguous run of tiles
};

// sep_pipe.hip: true when the LDS-DMA pipelined kernel covers the launch (stride 1, split-bf16, no generated input, W % 32 == 0)
bool sep_pipe_covers(const SepParams& p, int precision);
// sep_pipe.hip: launches it; p.N2 > 0 selects the two-output (DUAL) instances
int sep_pipe_launch(const SepParams& p, int B, hipStream_t st);
// sep_pipe2.hip: the software-pipelined form (8 x 32 tiles, stride 1): sep_pipe_launch hands it the shapes the rule below gives it
bool sep_pipe2_covers(const SepParams& p);
int sep_pipe2_launch(const SepParams& p, int B, hipStream_t st);

}  // namespace emd
