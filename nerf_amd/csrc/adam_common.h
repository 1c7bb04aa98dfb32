// The Adam update shared by the streaming kernel (optim_kernels.hip) and the binned table update (table_update.hip).
// Same update rule as torch.optim.Adam's single-tensor path as configured at train.py:84-88
// (L2 weight decay folded into the gradient, no amsgrad).
#pragma once
#include <math.h>

#include "common.h"

struct AdamK {
    float lr_over_bc1;   // lr / (1 - beta1^t)
    float inv_sqrt_bc2;  // 1 / sqrt(1 - beta2^t)
    float beta1, beta2, one_m_beta1, one_m_beta2, eps, wd, gscale;
};

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamK& k) {
    float gr = g * k.gscale;
    if (k.wd != 0.f) gr = fmaf(k.wd, p, gr);
    // A NaN / Inf gradient must never reach p, m, v (it would stay in the moments for good): the entry keeps its state.
    // The whole-step guard (skip flag: non-finite loss, non-finite compositing gradients) is GradScaler's rule
    // (trainer.py:134-136); this is the safety net under it for anything that overflows further down the backward pass.
    if (!(fabsf(gr) <= 3.402823466e38f)) return;
    m = m + k.one_m_beta1 * (gr - m);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(k.one_m_beta2 * gr, gr, v * k.beta2);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.lr_over_bc1 * (m / denom);
}

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam4(f4& p, const f4& g, f4& m, f4& v, const AdamK& k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = p[e], b = g[e], c = m[e], d = v[e];
        adam1(a, b, c, d, k);
        p[e] = a; m[e] = c; v[e] = d;
    }
}


static inline AdamK make_adam_k(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                float grad_scale) {
    AdamK k;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    k.lr_over_bc1 = (float)((double)lr / bc1);
    k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    k.beta1 = beta1; k.beta2 = beta2;
    k.one_m_beta1 = (float)(1.0 - (double)beta1);
    k.one_m_beta2 = (float)(1.0 - (double)beta2);
    k.eps = eps; k.wd = weight_decay; k.gscale = grad_scale;
    return k;
}
