// Dense Adam for gfx950, one pass over (param, grad, exp_avg, exp_avg_sq): a pure HBM stream of
// 16 B read + 12 B written per parameter (+4 B when the gradient is cleared in the same pass).
// Same update rule as torch.optim.Adam's single-tensor path as configured at train.py:84-88
// (L2 weight decay folded into the gradient, no amsgrad).
#include "adam_common.h"

namespace {

// Two independent 16-byte quads per lane in flight, non-temporal loads/stores (every byte is touched
// exactly once per step, so nothing is worth keeping in L2 / Infinity Cache), and a grid of many
// short blocks rather than 2048 persistent ones: measured 2.45 ms vs 2.80 ms on the 400 M-parameter table.
// `skip` (nullable): a device flag written earlier on the stream (nerf_losses' sums[5]: the step's loss was not finite).
// When it is non-zero the pass only clears the gradient: parameters and moments stay as they are, which is what
// GradScaler.step does with a non-finite gradient at trainer.py:134-136.
// HALF: the new parameters are also written, rounded to nearest even, into `ph` (the half-precision gather image of a
// hash table slice, precision float16): same index as p.
template <bool ZERO, bool HALF = false>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, AdamK k,
                                                   const double* __restrict__ skip, uint16_t* __restrict__ ph_ = nullptr) {
    _Float16* ph = (_Float16*)ph_;      // (rocprofv3 cannot demangle a _Float16* parameter)
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    auto half4 = [](const f4& a) { h16x4 o; o.x = (_Float16)a.x; o.y = (_Float16)a.y; o.z = (_Float16)a.z; o.w = (_Float16)a.w; return o; };
    const int64_t n4 = n >> 2;
    if (skip && *skip != 0.0) {
        if (ZERO) {
            const f4 z4 = {0.f, 0.f, 0.f, 0.f};
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
                ((f4*)g)[i] = z4;
            const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
            if (t < n) g[t] = 0.f;
        }
        return;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    f4* p4 = (f4*)p; f4* g4 = (f4*)g; f4* m4 = (f4*)m; f4* v4 = (f4*)v;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        const int64_t i1 = i0 + stride;
        const bool two = i1 < n4;
        f4 pa = __builtin_nontemporal_load(p4 + i0), ga = __builtin_nontemporal_load(g4 + i0);
        f4 ma = __builtin_nontemporal_load(m4 + i0), va = __builtin_nontemporal_load(v4 + i0);
        f4 pb = zero, gb = zero, mb = zero, vb = zero;
        if (two) {
            pb = __builtin_nontemporal_load(p4 + i1); gb = __builtin_nontemporal_load(g4 + i1);
            mb = __builtin_nontemporal_load(m4 + i1); vb = __builtin_nontemporal_load(v4 + i1);
        }
        adam4(pa, ga, ma, va, k);
        __builtin_nontemporal_store(pa, p4 + i0); __builtin_nontemporal_store(ma, m4 + i0);
        __builtin_nontemporal_store(va, v4 + i0);
        if (HALF) ((h16x4*)ph)[i0] = half4(pa);
        if (ZERO) __builtin_nontemporal_store(zero, g4 + i0);
        if (two) {
            adam4(pb, gb, mb, vb, k);
            __builtin_nontemporal_store(pb, p4 + i1); __builtin_nontemporal_store(mb, m4 + i1);
            __builtin_nontemporal_store(vb, v4 + i1);
            if (HALF) ((h16x4*)ph)[i1] = half4(pb);
            if (ZERO) __builtin_nontemporal_store(zero, g4 + i1);
        }
    }
    const int64_t tail = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < n) {
        float pp = p[tail], gg = g[tail], mm = m[tail], vv = v[tail];
        adam1(pp, gg, mm, vv, k);
        p[tail] = pp; m[tail] = mm; v[tail] = vv;
        if (HALF) ph[tail] = (_Float16)pp;
        if (ZERO) g[tail] = 0.f;
    }
}

// fp32 master table -> half-precision gather copy (round to nearest even; the reference's `precision: float16`,
// nets.py:31-36,49: tinycudann keeps fp32 master parameters and gathers from their fp16 image).  A pure stream:
// 4 B read + 2 B written per parameter.
__global__ __launch_bounds__(256) void table_to_half_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst_, int64_t n) {
    _Float16* dst = (_Float16*)dst_;
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f4 a = __builtin_nontemporal_load((const f4*)src + i);
        h16x4 o;
        o.x = (_Float16)a.x; o.y = (_Float16)a.y; o.z = (_Float16)a.z; o.w = (_Float16)a.w;
        ((h16x4*)dst)[i] = o;
    }
    const int64_t tail = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < n) dst[tail] = (_Float16)src[tail];
}

}  // namespace

extern "C" int nerf_table_to_half(const float* table, void* table_f16, int64_t count, void* stream) {
    NERF_REQUIRE(count >= 0, "negative count");
    if (count == 0) return 0;           // (an empty tensor has no address)
    NERF_REQUIRE(table && table_f16, "null pointer");
    NERF_REQUIRE((((uintptr_t)table) & 15) == 0 && (((uintptr_t)table_f16) & 7) == 0, "table must be 16-byte, its copy 8-byte aligned");
    int64_t nb = ((count >> 2) + 255) / 256;
    nb = nb < 1 ? 1 : (nb > 65536 ? 65536 : nb);
    table_to_half_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(table, (uint16_t*)table_f16, count);
    return nerf_launch_status("nerf_table_to_half");
}

static int adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step,
                     float grad_scale, int zero_grad, const double* skip_flag, _Float16* param_f16, void* stream) {
    NERF_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
    NERF_REQUIRE((((uintptr_t)param_f16) & 7) == 0, "the half-precision image must be 8-byte aligned");
    NERF_REQUIRE(count >= 0 && step >= 1, "bad count / step");
    NERF_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                 "buffers must be 16-byte aligned");
    if (count == 0) return 0;
    const AdamK k = make_adam_k(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    int64_t want = ceil_div64(ceil_div64(count, 8), 256);
    if (want < 1) want = 1;
    const unsigned nb = (unsigned)(want < 65536 ? want : 65536);
    hipStream_t st = (hipStream_t)stream;
    if (param_f16) {
        if (zero_grad) adam_kernel<true, true><<<nb, 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, count, k, skip_flag, (uint16_t*)param_f16);
        else adam_kernel<false, true><<<nb, 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, count, k, skip_flag, (uint16_t*)param_f16);
    } else if (zero_grad) adam_kernel<true><<<nb, 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, count, k, skip_flag);
    else adam_kernel<false><<<nb, 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, count, k, skip_flag);
    return nerf_launch_status("nerf_adam_step");
}

extern "C" int nerf_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                                      float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                      float grad_scale, int zero_grad, const double* skip_flag, void* stream) {
    return adam_step(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad,
                     skip_flag, nullptr, stream);
}

extern "C" int nerf_adam_step_f16(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                  int zero_grad, const double* skip_flag, void* param_f16, void* stream) {
    NERF_REQUIRE(param_f16 || count == 0, "null pointer");
    return adam_step(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad,
                     skip_flag, (_Float16*)param_f16, stream);
}

extern "C" int nerf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                              int zero_grad, void* stream) {
    return nerf_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, weight_decay, step,
                                  grad_scale, zero_grad, nullptr, stream);
}
