// Micro-benchmark of hash-table gradient scatter variants (development tool, not shipped).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -I nerf_amd/csrc tools/scatter_bench.hip nerf_amd/csrc/api_common.cpp -o tools/scatter_bench
#include <vector>
#include <random>
#include "field_common.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// V0: thread per (point, level), point-major
__global__ void v0(GridDev g, const float* x01, const float* dout, int64_t M, float* dtable) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * g.L) return;
    const int64_t p = i / g.L; const int l = (int)(i - p * g.L);
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    const Corners c = level_corners(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]);
    for (int k = 0; k < 8; ++k) for (int f = 0; f < 2; ++f) atomicAdd(dtable + (size_t)c.idx[k] * 2 + f, c.w[k] * dout[i * 2 + f]);
}

__device__ __forceinline__ void one_corner(const LevelP& lp, float x, float y, float z, int k, uint32_t& idx, float& w) {
    const float px = fmaf(x, lp.scale, 0.5f), py = fmaf(y, lp.scale, 0.5f), pz = fmaf(z, lp.scale, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const uint32_t cx = (uint32_t)(int)fx + (k & 1), cy = (uint32_t)(int)fy + ((k >> 1) & 1), cz = (uint32_t)(int)fz + ((k >> 2) & 1);
    w = ((k & 1) ? wx : 1.f - wx) * ((k & 2) ? wy : 1.f - wy) * ((k & 4) ? wz : 1.f - wz);
    uint32_t dense = cx + cy * lp.res + cz * (lp.res * lp.res);
    if (dense >= lp.entries) dense %= lp.entries;
    const uint32_t hash = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (lp.entries - 1u);
    idx = lp.offset + (lp.hashed ? hash : dense);
}

// V1: thread per (level, point, corner, feature), level-major: a wave = 4 consecutive points of one level
__global__ void v1(GridDev g, const float* x01, const float* dout, int64_t M, float* dtable) {
    const int l = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t >> 4;
    if (p >= M) return;
    const int k = (t >> 1) & 7, f = t & 1;
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    uint32_t idx; float w;
    one_corner(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2], k, idx, w);
    atomicAdd(dtable + (size_t)idx * 2 + f, w * dout[p * (g.L * 2) + l * 2 + f]);
}

// V2: thread per (level, point, corner), two atomics per thread, level-major
__global__ void v2(GridDev g, const float* x01, const float* dout, int64_t M, float* dtable) {
    const int l = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t >> 3;
    if (p >= M) return;
    const int k = t & 7;
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    uint32_t idx; float w;
    one_corner(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2], k, idx, w);
    const float2 d = *(const float2*)(dout + p * (g.L * 2) + l * 2);
    atomicAdd(dtable + (size_t)idx * 2, w * d.x);
    atomicAdd(dtable + (size_t)idx * 2 + 1, w * d.y);
}

// V3: like V1 but point-major (a wave = 4 (point, level) pairs of consecutive levels)
__global__ void v3(GridDev g, const float* x01, const float* dout, int64_t M, float* dtable) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pl = t >> 4;
    if (pl >= M * g.L) return;
    const int64_t p = pl / g.L; const int l = (int)(pl - p * g.L);
    const int k = (t >> 1) & 7, f = t & 1;
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    uint32_t idx; float w;
    one_corner(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2], k, idx, w);
    atomicAdd(dtable + (size_t)idx * 2 + f, w * dout[pl * 2 + f]);
}

// V4: thread per (level, point), level-major (V0's work per thread, tcnn's launch shape)
__global__ void v4(GridDev g, const float* x01, const float* dout, int64_t M, float* dtable) {
    const int l = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    const Corners c = level_corners(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]);
    const float2 d = *(const float2*)(dout + p * (g.L * 2) + l * 2);
    for (int k = 0; k < 8; ++k) { atomicAdd(dtable + (size_t)c.idx[k] * 2, c.w[k] * d.x); atomicAdd(dtable + (size_t)c.idx[k] * 2 + 1, c.w[k] * d.y); }
}

// gather counterparts (forward encode): G0 thread per (level, point) level-major; G1 thread per (point, level)
__global__ void g0(GridDev g, const float* table, const float* x01, int64_t M, float* out) {
    const int l = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    const Corners c = level_corners(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]);
    float feat[2]; gather_level<2>(table, c, feat);
    *(float2*)(out + p * (g.L * 2) + l * 2) = make_float2(feat[0], feat[1]);
}
__global__ void g1(GridDev g, const float* table, const float* x01, int64_t M, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * g.L) return;
    const int64_t p = i / g.L; const int l = (int)(i - p * g.L);
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    const Corners c = level_corners(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]);
    float feat[2]; gather_level<2>(table, c, feat);
    *(float2*)(out + i * 2) = make_float2(feat[0], feat[1]);
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 524288;
    const int coherent = argc > 2 ? atoi(argv[2]) : 1;  // 1: points along rays (128 per ray), 0: iid uniform
    NerfGridCfg cfg; int64_t total = nerf_grid_cfg_init(&cfg, 20, 2, 24, 16, 1.3819f);
    GridDev g = make_grid_dev(&cfg);
    std::vector<float> hx(M * 3), hd(M * 40);
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0.f, 1.f);
    if (coherent == 2) {   // rays through a ball of radius 0.34 around the cube centre (what the bench scene produces)
        for (int64_t r = 0; r < M / 128; ++r) {
            float o[3], e[3];
            for (int c = 0; c < 3; ++c) { o[c] = 0.5f + 0.68f * (U(rng) - 0.5f); e[c] = 0.5f + 0.68f * (U(rng) - 0.5f); }
            for (int s = 0; s < 128; ++s) for (int c = 0; c < 3; ++c) hx[(r * 128 + s) * 3 + c] = o[c] + (e[c] - o[c]) * (s + U(rng)) / 128.f;
        }
    } else if (coherent) {
        for (int64_t r = 0; r < M / 128; ++r) {
            float o[3] = {U(rng), U(rng), U(rng)}, e[3] = {U(rng), U(rng), U(rng)};
            for (int s = 0; s < 128; ++s) for (int c = 0; c < 3; ++c) hx[(r * 128 + s) * 3 + c] = o[c] + (e[c] - o[c]) * (s + U(rng)) / 128.f;
        }
    } else for (auto& v : hx) v = U(rng);
    for (auto& v : hd) v = U(rng) - 0.5f;
    float *dx, *dd, *dt, *dout;
    CK(hipMalloc(&dx, M * 12)); CK(hipMalloc(&dd, M * 160)); CK(hipMalloc(&dt, total * 8)); CK(hipMalloc(&dout, M * 160));
    CK(hipMemcpy(dx, hx.data(), M * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd.data(), M * 160, hipMemcpyHostToDevice));
    CK(hipMemset(dt, 0, total * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0;
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms; }
        printf("%-28s best %.3f ms  avg %.3f ms  (%.1f G corner-feature ops/s)\n", name, best, sum / 5, M * 320.0 / (best * 1e6));
    };
    printf("M = %ld points, %s, table %.2f GB\n", (long)M, coherent ? "ray-coherent" : "iid", total * 8 / 1e9);
    run("v0 pt-major thread/(p,l)", [&] { v0<<<(unsigned)((M * 20 + 255) / 256), 256>>>(g, dx, dd, M, dt); });
    run("v4 lvl-major thread/(l,p)", [&] { v4<<<dim3((unsigned)((M + 255) / 256), 20), 256>>>(g, dx, dd, M, dt); });
    run("v1 lvl-major thread/(l,p,k,f)", [&] { v1<<<dim3((unsigned)((M * 16 + 255) / 256), 20), 256>>>(g, dx, dd, M, dt); });
    run("v2 lvl-major thread/(l,p,k)", [&] { v2<<<dim3((unsigned)((M * 8 + 255) / 256), 20), 256>>>(g, dx, dd, M, dt); });
    run("v3 pt-major thread/(p,l,k,f)", [&] { v3<<<(unsigned)((M * 20 * 16 + 255) / 256), 256>>>(g, dx, dd, M, dt); });
    run("g0 gather lvl-major", [&] { g0<<<dim3((unsigned)((M + 255) / 256), 20), 256>>>(g, dt, dx, M, dout); });
    run("g1 gather pt-major", [&] { g1<<<(unsigned)((M * 20 + 255) / 256), 256>>>(g, dt, dx, M, dout); });
    return 0;
}
