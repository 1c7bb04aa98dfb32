// Dense Adam for gfx950, one pass over (param, grad, exp_avg, exp_avg_sq): a pure HBM stream of
// 16 B read + 12 B written per parameter (+4 B when the gradient is cleared in the same pass).
// Same update rule as torch.optim.Adam's single-tensor path as configured at train.py:84-88
// (L2 weight decay folded into the gradient, no amsgrad).
#include "common.h"

namespace {

struct AdamK {
    float lr_over_bc1;   // lr / (1 - beta1^t)
    float inv_sqrt_bc2;  // 1 / sqrt(1 - beta2^t)
    float beta1, beta2, one_m_beta1, one_m_beta2, eps, wd, gscale;
};

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamK& k) {
    float gr = g * k.gscale;
    if (k.wd != 0.f) gr = fmaf(k.wd, p, gr);
    m = m + k.one_m_beta1 * (gr - m);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(k.one_m_beta2 * gr, gr, v * k.beta2);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.lr_over_bc1 * (m / denom);
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, AdamK k) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* p4 = (float4*)p; float4* g4 = (float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, k);
        adam1(pp.y, gg.y, mm.y, vv.y, k);
        adam1(pp.z, gg.z, mm.z, vv.z, k);
        adam1(pp.w, gg.w, mm.w, vv.w, k);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (ZERO) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t tail = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < n) {
        float pp = p[tail], gg = g[tail], mm = m[tail], vv = v[tail];
        adam1(pp, gg, mm, vv, k);
        p[tail] = pp; m[tail] = mm; v[tail] = vv;
        if (ZERO) g[tail] = 0.f;
    }
}

}  // namespace

extern "C" int nerf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                              int zero_grad, void* stream) {
    NERF_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
    NERF_REQUIRE(count >= 0 && step >= 1, "bad count / step");
    NERF_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                 "buffers must be 16-byte aligned");
    if (count == 0) return 0;
    AdamK k;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    k.lr_over_bc1 = (float)((double)lr / bc1);
    k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    k.beta1 = beta1; k.beta2 = beta2;
    k.one_m_beta1 = (float)(1.0 - (double)beta1);
    k.one_m_beta2 = (float)(1.0 - (double)beta2);
    k.eps = eps; k.wd = weight_decay; k.gscale = grad_scale;
    int64_t want = ceil_div64(ceil_div64(count, 4), 256);
    if (want < 1) want = 1;
    const unsigned nb = (unsigned)(want < (int64_t)kNumCU * 8 ? want : (int64_t)kNumCU * 8);
    if (zero_grad) adam_kernel<true><<<nb, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, count, k);
    else adam_kernel<false><<<nb, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, count, k);
    return nerf_launch_status("nerf_adam_step");
}
