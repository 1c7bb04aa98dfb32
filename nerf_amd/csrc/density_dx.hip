// d(sigma)/d(x) of NeRFNetwork.density (nets.py:148-164) with respect to its input positions: what autograd + tcnn
// deliver when the reference's pose alignment differentiates model.density through the sample points
// (allign/trainer.py:174-208).  One thread per point: the chain per point is tiny (2 x 40x64 FMAs and two gather
// passes), the cost is the random 8-byte gathers, exactly as in the forward.
//   pass 1: features level by level -> hidden pre-activations (64 registers)
//   sigma' = W2[0,:] . relu(hidden);  dh = dsigma * [sigma' > 0] * W2[0,:] * [hidden > 0]
//   pass 2: per level dfeat = W1[:, level]^T dh, then d(feature)/d(pos) from the trilinear weights' derivative.
#include "field_common.h"

int nerf_check_cfg(const char* fn, const NerfGridCfg* grid, const NerfNetCfg* net, bool with_color);
int nerf_check_src(const char* fn, const NerfPointSrc* src, bool with_color);

namespace {

template <int F, bool HALF = false>      // HALF: `table` is the half-precision gather copy (gather_level)
__global__ __launch_bounds__(256) void density_dx_kernel(GridDev g, const float* __restrict__ table,
                                                         const float* __restrict__ sigma_w, NerfPointSrc src,
                                                         const float* __restrict__ dsigma, float* __restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LF = g.L * F;
    float* W1 = lds;                 // [in][64]: transposed so that a thread walks the 64 units of one input
    float* W2 = lds + LF * kHid;     // row 0 of layer 2 (the sigma row)
    for (int i = threadIdx.x; i < LF * kHid; i += blockDim.x) {
        const int o = i / LF, k = i - o * LF;
        W1[k * kHid + o] = sigma_w[i];
    }
    for (int i = threadIdx.x; i < kHid; i += blockDim.x) W2[i] = sigma_w[kHid * LF + i];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= src.n_points) return;
    const PointIn q = fetch_point(src, p, false);
    float hid[kHid];
#pragma unroll
    for (int j = 0; j < kHid; ++j) hid[j] = 0.f;
    for (int l = 0; l < g.L; ++l) {
        LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
        const Corners c = level_corners(lp, q.x, q.y, q.z);
        float feat[F];
        gather_level<F, HALF>(table, c, feat);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float* w = W1 + (l * F + f) * kHid;
#pragma unroll
            for (int j = 0; j < kHid; ++j) hid[j] = fmaf(w[j], feat[f], hid[j]);
        }
    }
    float pre = 0.f;
#pragma unroll
    for (int j = 0; j < kHid; ++j) pre = fmaf(W2[j], fmaxf(hid[j], 0.f), pre);
    const float ds = pre > 0.f ? dsigma[p] : 0.f;
#pragma unroll
    for (int j = 0; j < kHid; ++j) hid[j] = hid[j] > 0.f ? ds * W2[j] : 0.f;   // now d(loss)/d(hidden pre-activation)
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int l = 0; l < g.L; ++l) {
        float df[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float* w = W1 + (l * F + f) * kHid;
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < kHid; ++j) a = fmaf(w[j], hid[j], a);
            df[f] = a;
        }
        const float sc = g.scale[l];
        LevelP lp = {sc, g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
        const Corners c = level_corners(lp, q.x, q.y, q.z);
        const float px = fmaf(q.x, sc, 0.5f), py = fmaf(q.y, sc, 0.5f), pz = fmaf(q.z, sc, 0.5f);
        const float wx = px - floorf(px), wy = py - floorf(py), wz = pz - floorf(pz);
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const size_t e = (size_t)c.idx[k] * F + f;
                v = fmaf(df[f], HALF ? (float)((const _Float16*)table)[e] : table[e], v);
            }
            const float fx = (k & 1) ? wx : 1.f - wx, fy = (k & 2) ? wy : 1.f - wy, fz = (k & 4) ? wz : 1.f - wz;
            ax += ((k & 1) ? v : -v) * fy * fz;
            ay += ((k & 2) ? v : -v) * fx * fz;
            az += ((k & 4) ? v : -v) * fx * fy;
        }
        gx = fmaf(ax, sc, gx);
        gy = fmaf(ay, sc, gy);
        gz = fmaf(az, sc, gz);
    }
    const float inv = 1.f / (2.f * src.outer_bound);      // u = (x + ob) / (2 ob)
    dx[p * 3 + 0] = gx * inv;
    dx[p * 3 + 1] = gy * inv;
    dx[p * 3 + 2] = gz * inv;
}

}  // namespace

template <bool HALF>
static int density_bwd_input(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table,
                             const float* sigma_params, const NerfPointSrc* src, const float* dsigma, float* dx,
                             void* stream) {
    int rc = nerf_check_cfg("nerf_density_bwd_input", grid, net, false);
    if (rc) return rc;
    rc = nerf_check_src("nerf_density_bwd_input", src, false);
    if (rc) return rc;
    NERF_REQUIRE(src->mode == 0, "points mode only: the gradient is taken w.r.t. the positions passed in");
    NERF_REQUIRE(table && sigma_params && dsigma && dx, "null pointer");
    if (src->n_points == 0) return 0;
    const GridDev g = make_grid_dev(grid);
    const int LF = grid->n_levels * grid->n_features;
    const int lds_bytes = (LF * kHid + kHid) * (int)sizeof(float);
    const unsigned nb = (unsigned)ceil_div64(src->n_points, 256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(FF)                                                                                                       \
    do {                                                                                                                 \
        auto kern = density_dx_kernel<FF, HALF>;                                                                             \
        static NerfLdsAttrCache lds_cache = {};   /* per instantiation */    \
        if (int e = nerf_ensure_dynamic_lds((const void*)kern, lds_bytes, &lds_cache, "density_bwd_input")) return e;    \
        kern<<<nb, 256, lds_bytes, st>>>(g, table, sigma_params, *src, dsigma, dx);                                      \
    } while (0)
    switch (grid->n_features) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    return nerf_launch_status("nerf_density_bwd_input");
}

extern "C" int nerf_density_bwd_input(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table,
                                      const float* sigma_params, const NerfPointSrc* src, const float* dsigma, float* dx,
                                      void* stream) {
    return density_bwd_input<false>(grid, net, table, sigma_params, src, dsigma, dx, stream);
}

extern "C" int nerf_density_bwd_input_f16(const NerfGridCfg* grid, const NerfNetCfg* net, const void* table_f16,
                                          const float* sigma_params, const NerfPointSrc* src, const float* dsigma, float* dx,
                                          void* stream) {
    return density_bwd_input<true>(grid, net, (const float*)table_f16, sigma_params, src, dsigma, dx, stream);
}
