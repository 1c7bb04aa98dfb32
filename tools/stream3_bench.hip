// Micro-benchmark: how fast can three arrays (p, m, v; 1.6 GB each) be read and rewritten in place on MI355X, and what
// does the access structure cost?  (development tool, not shipped; hipcc --offload-arch=gfx950 -O3 -o tools/stream3_bench tools/stream3_bench.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 upd(f4 p, f4 m, f4 v) { return p * 0.999f + m * 0.001f + v * 1e-6f; }

// V: workgroup per block of BLK quads per array (contiguous), T threads, all loads first
template <int T, int Q, bool NT>
__global__ __launch_bounds__(T) void rmw_block(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v, int64_t n4) {
    const int64_t base = (int64_t)blockIdx.x * (T * Q);
    f4 pq[Q], mq[Q], vq[Q];
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const int64_t i = base + threadIdx.x + u * T;
        if (i < n4) {
            if (NT) { pq[u] = __builtin_nontemporal_load(p + i); mq[u] = __builtin_nontemporal_load(m + i); vq[u] = __builtin_nontemporal_load(v + i); }
            else { pq[u] = p[i]; mq[u] = m[i]; vq[u] = v[i]; }
        }
    }
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const int64_t i = base + threadIdx.x + u * T;
        if (i < n4) {
            const f4 a = upd(pq[u], mq[u], vq[u]), b = upd(mq[u], vq[u], pq[u]), c = upd(vq[u], pq[u], mq[u]);
            if (NT) { __builtin_nontemporal_store(a, p + i); __builtin_nontemporal_store(b, m + i); __builtin_nontemporal_store(c, v + i); }
            else { p[i] = a; m[i] = b; v[i] = c; }
        }
    }
}

struct AdamK { float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, one_m_beta1, one_m_beta2, eps, wd, gscale; };
template <bool FAST>
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamK& k) {
    float gr = g * k.gscale;
    if (k.wd != 0.f) gr = fmaf(k.wd, p, gr);
    m = m + k.one_m_beta1 * (gr - m);
    v = fmaf(k.one_m_beta2 * gr, gr, v * k.beta2);
    if (FAST) {
        const float denom = __builtin_amdgcn_sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
        p = p - k.lr_over_bc1 * (m * __builtin_amdgcn_rcpf(denom));
    } else {
        const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
        p = p - k.lr_over_bc1 * (m / denom);
    }
}
// the update kernel's shape with the real Adam arithmetic, optional LDS gradient + barriers
template <int T, int Q, bool FAST, bool LDS>
__global__ __launch_bounds__(T) void adam_block(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v, int64_t n4, AdamK k) {
    __shared__ f4 acc4[LDS ? T * Q : 1];
    const int64_t base = (int64_t)blockIdx.x * (T * Q);
    f4 pq[Q], mq[Q], vq[Q];
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const int64_t i = base + threadIdx.x + u * T;
        if (i < n4) { pq[u] = __builtin_nontemporal_load(p + i); mq[u] = __builtin_nontemporal_load(m + i); vq[u] = __builtin_nontemporal_load(v + i); }
    }
    if (LDS) {
        const f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < Q; ++u) acc4[threadIdx.x + u * T] = z;
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier();
        atomicAdd((float*)acc4 + ((threadIdx.x * 37 + blockIdx.x) % (T * Q * 4)), 1e-3f);
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const int64_t i = base + threadIdx.x + u * T;
        if (i < n4) {
            f4 gq = {1e-3f, -2e-3f, 0.f, 0.f};
            if (LDS) gq = acc4[threadIdx.x + u * T];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float a = pq[u][e], c = mq[u][e], d = vq[u][e]; adam1<FAST>(a, gq[e], c, d, k); pq[u][e] = a; mq[u][e] = c; vq[u][e] = d; }
            __builtin_nontemporal_store(pq[u], p + i); __builtin_nontemporal_store(mq[u], m + i); __builtin_nontemporal_store(vq[u], v + i);
        }
    }
}

// grid-stride, U quads per array in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void rmw_stride(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
        f4 pq[U], mq[U], vq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = i0 + u * stride; if (i < n4) {
            if (NT) { pq[u] = __builtin_nontemporal_load(p + i); mq[u] = __builtin_nontemporal_load(m + i); vq[u] = __builtin_nontemporal_load(v + i); }
            else { pq[u] = p[i]; mq[u] = m[i]; vq[u] = v[i]; } } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = i0 + u * stride; if (i < n4) {
            const f4 a = upd(pq[u], mq[u], vq[u]), b = upd(mq[u], vq[u], pq[u]), c = upd(vq[u], pq[u], mq[u]);
            if (NT) { __builtin_nontemporal_store(a, p + i); __builtin_nontemporal_store(b, m + i); __builtin_nontemporal_store(c, v + i); }
            else { p[i] = a; m[i] = b; v[i] = c; } } }
    }
}

// one array read and rewritten in place (is it the three streams or the in-place pattern?)
template <bool NT>
__global__ __launch_bounds__(256) void rmw_one(f4* __restrict__ p, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 a = NT ? __builtin_nontemporal_load(p + i) : p[i];
        a = a * 0.999f;
        if (NT) __builtin_nontemporal_store(a, p + i); else p[i] = a;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy4(const f4* __restrict__ a, f4* __restrict__ b, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i); else b[i] = a[i];
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void read_only(const f4* __restrict__ a, float* out, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    f4 s = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) s += NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}
template <bool NT>
__global__ __launch_bounds__(256) void write_only(f4* __restrict__ a, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const f4 z = {1.f, 2.f, 3.f, 4.f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { if (NT) __builtin_nontemporal_store(z, a + i); else a[i] = z; }
}

int main() {
    const int64_t n = 399603232; const int64_t n4 = n / 4;
    float *p, *m, *v, *o;
    CK(hipMalloc(&p, n * 4 + (4 << 20))); CK(hipMalloc(&m, n * 4 + (4 << 20))); CK(hipMalloc(&v, n * 4 + (4 << 20))); CK(hipMalloc(&o, 256));
    CK(hipMemset(p, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize()); float best = 1e9, sum = 0;
        for (int r = 0; r < 6; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms; }
        printf("%-44s best %.3f ms  avg %.3f ms  %.0f GB/s (best)\n", name, best, sum / 6, bytes / (best * 1e6)); fflush(stdout);
    };
    const double B3 = 24.0 * n;
    run("copy p->m plain grid 4096", 8.0 * n, [&] { copy4<false><<<4096, 256>>>((f4*)p, (f4*)m, n4); });
    run("copy p->m nt grid 4096", 8.0 * n, [&] { copy4<true><<<4096, 256>>>((f4*)p, (f4*)m, n4); });
    run("copy p->m nt grid 65536", 8.0 * n, [&] { copy4<true><<<65536, 256>>>((f4*)p, (f4*)m, n4); });
    run("read p plain grid 4096", 4.0 * n, [&] { read_only<false><<<4096, 256>>>((f4*)p, o, n4); });
    run("read p nt grid 4096", 4.0 * n, [&] { read_only<true><<<4096, 256>>>((f4*)p, o, n4); });
    run("write p plain grid 4096", 4.0 * n, [&] { write_only<false><<<4096, 256>>>((f4*)p, n4); });
    run("write p nt grid 4096", 4.0 * n, [&] { write_only<true><<<4096, 256>>>((f4*)p, n4); });
    run("rmw one array plain grid 4096", 8.0 * n, [&] { rmw_one<false><<<4096, 256>>>((f4*)p, n4); });
    run("rmw one array nt grid 4096", 8.0 * n, [&] { rmw_one<true><<<4096, 256>>>((f4*)p, n4); });
    run("rmw one array nt grid 65536", 8.0 * n, [&] { rmw_one<true><<<65536, 256>>>((f4*)p, n4); });
    for (int nb : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "rmw3 stride U1 nt grid %d", nb); run(nm, B3, [&] { rmw_stride<1, true><<<nb, 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
        snprintf(nm, 64, "rmw3 stride U2 nt grid %d", nb); run(nm, B3, [&] { rmw_stride<2, true><<<nb, 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
        snprintf(nm, 64, "rmw3 stride U4 nt grid %d", nb); run(nm, B3, [&] { rmw_stride<4, true><<<nb, 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
        snprintf(nm, 64, "rmw3 stride U2 plain grid %d", nb); run(nm, B3, [&] { rmw_stride<2, false><<<nb, 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    }
    run("rmw3 block 1024x4 nt (the update kernel's)", B3, [&] { rmw_block<1024, 4, true><<<(unsigned)((n4 + 4095) / 4096), 1024>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 1024x2 nt", B3, [&] { rmw_block<1024, 2, true><<<(unsigned)((n4 + 2047) / 2048), 1024>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 1024x1 nt", B3, [&] { rmw_block<1024, 1, true><<<(unsigned)((n4 + 1023) / 1024), 1024>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 256x4 nt", B3, [&] { rmw_block<256, 4, true><<<(unsigned)((n4 + 1023) / 1024), 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 256x2 nt", B3, [&] { rmw_block<256, 2, true><<<(unsigned)((n4 + 511) / 512), 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 256x1 nt", B3, [&] { rmw_block<256, 1, true><<<(unsigned)((n4 + 255) / 256), 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 512x2 nt", B3, [&] { rmw_block<512, 2, true><<<(unsigned)((n4 + 1023) / 1024), 512>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    run("rmw3 block 256x2 plain", B3, [&] { rmw_block<256, 2, false><<<(unsigned)((n4 + 511) / 512), 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    AdamK k = {2e-2f, 1.f, 0.9f, 0.99f, 0.1f, 0.01f, 1e-15f, 0.f, 1.f};
    const unsigned nb1 = (unsigned)((n4 + 1023) / 1024);
    run("adam block 256x4 ieee", B3, [&] { adam_block<256, 4, false, false><<<nb1, 256>>>((f4*)p, (f4*)m, (f4*)v, n4, k); });
    run("adam block 256x4 fast rcp/sqrt", B3, [&] { adam_block<256, 4, true, false><<<nb1, 256>>>((f4*)p, (f4*)m, (f4*)v, n4, k); });
    run("adam block 256x4 ieee + LDS grad + 2 barriers", B3, [&] { adam_block<256, 4, false, true><<<nb1, 256>>>((f4*)p, (f4*)m, (f4*)v, n4, k); });
    run("adam block 256x4 fast + LDS grad + 2 barriers", B3, [&] { adam_block<256, 4, true, true><<<nb1, 256>>>((f4*)p, (f4*)m, (f4*)v, n4, k); });
    run("adam block 256x2 ieee + LDS", B3, [&] { adam_block<256, 2, false, true><<<(unsigned)((n4 + 511) / 512), 256>>>((f4*)p, (f4*)m, (f4*)v, n4, k); });
    run("adam block 1024x4 ieee + LDS", B3, [&] { adam_block<1024, 4, false, true><<<(unsigned)((n4 + 4095) / 4096), 1024>>>((f4*)p, (f4*)m, (f4*)v, n4, k); });
    run("adam block 256x4 ieee + LDS, arrays offset by 64 B", B3, [&] { adam_block<256, 4, false, true><<<nb1, 256>>>((f4*)(p + 16), (f4*)(m + 16), (f4*)(v + 16), n4 - 8, k); });
    run("adam block 256x4 ieee + LDS, arrays offset by 2647616 B", B3, [&] { adam_block<256, 4, false, true><<<nb1, 256>>>((f4*)(p + 661904), (f4*)(m + 661904), (f4*)(v + 661904), n4 - 165476 - 8, k); });
    run("rmw3 block 256x4 nt (again)", B3, [&] { rmw_block<256, 4, true><<<nb1, 256>>>((f4*)p, (f4*)m, (f4*)v, n4); });
    return 0;
}
