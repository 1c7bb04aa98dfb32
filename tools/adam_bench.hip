// Micro-benchmark of dense-Adam streaming variants (development tool, not shipped).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct AdamK { float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, one_m_beta1, one_m_beta2, eps, wd, gscale; };
__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamK& k) {
    float gr = g * k.gscale;
    if (k.wd != 0.f) gr = fmaf(k.wd, p, gr);
    m = m + k.one_m_beta1 * (gr - m);
    v = fmaf(k.one_m_beta2 * gr, gr, v * k.beta2);
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.lr_over_bc1 * (m / denom);
}
__device__ __forceinline__ void adam4(f4& p, f4& g, f4& m, f4& v, const AdamK& k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { float a = p[e], b = g[e], c = m[e], d = v[e]; adam1(a, b, c, d, k); p[e] = a; m[e] = c; v[e] = d; }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void adam_v(f4* __restrict__ p4, f4* __restrict__ g4, f4* __restrict__ m4, f4* __restrict__ v4, int64_t n4, AdamK k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
        f4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t i = i0 + u * stride; if (i < n4) {
            if (NT) { pp[u] = __builtin_nontemporal_load(p4 + i); gg[u] = __builtin_nontemporal_load(g4 + i); mm[u] = __builtin_nontemporal_load(m4 + i); vv[u] = __builtin_nontemporal_load(v4 + i); }
            else { pp[u] = p4[i]; gg[u] = g4[i]; mm[u] = m4[i]; vv[u] = v4[i]; } } }
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t i = i0 + u * stride; if (i < n4) {
            adam4(pp[u], gg[u], mm[u], vv[u], k);
            const f4 z = {0.f, 0.f, 0.f, 0.f};
            if (NT) { __builtin_nontemporal_store(pp[u], p4 + i); __builtin_nontemporal_store(mm[u], m4 + i); __builtin_nontemporal_store(vv[u], v4 + i); __builtin_nontemporal_store(z, g4 + i); }
            else { p4[i] = pp[u]; m4[i] = mm[u]; v4[i] = vv[u]; g4[i] = z; } } }
    }
}
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
int main() {
    const int64_t n = 400265136; const int64_t n4 = n / 4;
    float *p, *g, *m, *v;
    CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
    CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
    AdamK k = {2e-2f, 1.f, 0.9f, 0.99f, 0.1f, 0.01f, 1e-15f, 0.f, 1.f};
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize()); float best = 1e9;
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; }
        printf("%-34s %.3f ms  %.0f GB/s\n", name, best, bytes / (best * 1e6));
    };
    const double B = 32.0 * n;
    for (int nb : {1024, 2048, 4096, 8192, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "U1 plain   grid %d", nb); run(nm, B, [&] { adam_v<1, false><<<nb, 256>>>((f4*)p, (f4*)g, (f4*)m, (f4*)v, n4, k); });
        snprintf(nm, 64, "U2 plain   grid %d", nb); run(nm, B, [&] { adam_v<2, false><<<nb, 256>>>((f4*)p, (f4*)g, (f4*)m, (f4*)v, n4, k); });
        snprintf(nm, 64, "U1 nontemp grid %d", nb); run(nm, B, [&] { adam_v<1, true><<<nb, 256>>>((f4*)p, (f4*)g, (f4*)m, (f4*)v, n4, k); });
        snprintf(nm, 64, "U2 nontemp grid %d", nb); run(nm, B, [&] { adam_v<2, true><<<nb, 256>>>((f4*)p, (f4*)g, (f4*)m, (f4*)v, n4, k); });
    }
    run("copy p->g (8 B/elt) grid 2048", 8.0 * n, [&] { copy4<<<2048, 256>>>((float4*)p, (float4*)g, n4); });
    run("copy p->g grid 16384", 8.0 * n, [&] { copy4<<<16384, 256>>>((float4*)p, (float4*)g, n4); });
    return 0;
}
