// The two per-channel steps of the training-mode batch-norm chain (bn_train.hip: the algebra is stated there), as device functions, so that
// the kernels that finish a reduction can run them in the same launch (round 4: bn_stats_final + bn_train_fold_kernel and chan_reduce_final +
// bn_bwd_prep_kernel were ~1 400 launches of 4-10 us per training step, every one a link in its stream's dependent chain).  The
// statements are bn_train_fold_kernel's / bn_bwd_prep_kernel's own (those kernels call these functions): same bits either way.
#pragma once

namespace emd {

struct BnFoldArgs {       // device copy of emd_bn_train_fold_t (include/emdenoise.h); omd = float32(1 - decay)
    const float *gamma1, *beta1, *gamma2, *beta2, *bias;
    float eps, omd;
    float *scale, *shift, *rstd1, *rstd2, *mm1, *mv1, *mm2, *mv2;
};

struct BnPrepArgs {       // device copy of emd_bn_bwd_prep_t
    const float *gamma1, *gamma2, *rstd1, *rstd2;
    float eps;
    float *K, *m1, *m2, *dgamma1, *dgamma2, *dbeta2;
};

// entry i = image * C + c of the [image][C] vectors (i = c for batch statistics); moving: update the moving statistics (image 0 only)
__device__ __forceinline__ void bn_train_fold_one(const BnFoldArgs& a, int i, int c, bool moving, float mu, float v, float n) {
    const float r1 = rsqrtf(v + a.eps);
    const float bessel = n > 1.f ? n / (n - 1.f) : 1.f;
    a.rstd1[i] = r1;
    if (a.gamma1) {  // BN1 (gamma1, beta1) then BN2 (gamma2, beta2)
        const float g1 = a.gamma1[c];
        const float var2 = g1 * g1 * v * r1 * r1;
        const float r2 = rsqrtf(var2 + a.eps);
        a.rstd2[i] = r2;
        const float sc = g1 * a.gamma2[c] * r1 * r2;
        a.scale[i] = sc;
        a.shift[i] = a.beta2[c] - mu * sc;
        if (moving && a.mm1) {
            a.mm1[c] -= (a.mm1[c] - mu) * a.omd;
            a.mv1[c] -= (a.mv1[c] - v * bessel) * a.omd;
            a.mm2[c] -= (a.mm2[c] - a.beta1[c]) * a.omd;
            a.mv2[c] -= (a.mv2[c] - var2 * bessel) * a.omd;
        }
    } else {       // a single BN (gamma2, beta2) after conv + bias
        const float sc = a.gamma2[c] * r1;
        a.scale[i] = sc;
        a.shift[i] = a.beta2[c] - mu * sc;
        if (moving && a.mm2) {
            a.mm2[c] -= (a.mm2[c] - (mu + (a.bias ? a.bias[c] : 0.f))) * a.omd;
            a.mv2[c] -= (a.mv2[c] - v * bessel) * a.omd;
        }
    }
}

// parameter gradients ACCUMULATE (several towers / micro-batches add into one gradient set: float atomics)
__device__ __forceinline__ void bn_bwd_prep_one(const BnPrepArgs& a, int i, int c, float sv, float tv, float inv_n) {
    const float r1 = a.rstd1[i];
    a.m1[i] = sv * inv_n;
    atomicAdd(a.dbeta2 + c, sv);
    if (a.gamma1) {
        const float g1 = a.gamma1[c], g2 = a.gamma2[c], r2 = a.rstd2[i];
        const float aa = g1 * r2;
        const float e2 = a.eps * r2 * r2;
        a.K[i] = g1 * g2 * r1 * r2;
        a.m2[i] = r1 * tv * inv_n * (aa * aa + e2);
        atomicAdd(a.dgamma2 + c, aa * tv);
        atomicAdd(a.dgamma1 + c, g2 * r2 * e2 * tv);
    } else {
        a.K[i] = a.gamma2[c] * r1;
        a.m2[i] = r1 * tv * inv_n;
        atomicAdd(a.dgamma2 + c, tv);
    }
}

}  // namespace emd
