// Per-ray kernels of the render path for gfx950 (MI355X): ray generation, sample points and
// scene contraction, coarse-pass weights, PDF re-weighting, inverse-CDF resampling, alpha
// compositing (forward + backward) and the fused training losses.
//
// All of them are HBM-bandwidth-bound streaming kernels: one 64-lane wavefront owns one ray,
// a row goes from HBM straight into the registers of the lanes that own its samples (coalesced
// 8 / 16-byte pieces, no LDS staging, no workgroup barriers), and the along-ray recurrences
// (transmittance product, CDF sum, suffix sums) are wavefront prefix scans carried in fp64 and
// rounded to fp32 per element -- exactly how the reference's CPU cumsum/cumprod behave
// (SURVEY.md F6), which is what makes the resampled indices bit-exact.
//
// FP contraction is OFF in this file: the reference evaluates u*(range)+base etc. as separate
// roundings and the index parity depends on it.  fmaf is written out where a fused op is wanted.
#pragma clang fp contract(off)

#include <stdlib.h>
#include <initializer_list>
#include <type_traits>
#include "common.h"
#include "philox.h"

namespace {

constexpr int kRaysPerBlock = 4;  // 256 threads, one wavefront per ray
constexpr int kMaxS = NERF_MAX_SAMPLES;

// ------------------------------------------------------------------------------------------
// wavefront scans (fp64) on DPP: row shifts inside the 16-lane rows, then the two row broadcasts of GFX9 (lane 15 of a
// row into the next row, lane 31 into rows 2 and 3).  Plain VALU moves: a __shfl_up ladder is six ds_bpermute round
// trips through the LDS crossbar per scan, and these kernels are a chain of such scans.
// ------------------------------------------------------------------------------------------
// every lane: the value of the lane the DPP control names, `old` where that lane does not exist or the row is masked
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const uint64_t o = __builtin_bit_cast(uint64_t, old), s = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)s, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(s >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 0xf, 0xf, false));
}

constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143, kWaveShl1 = 0x130, kWaveShr1 = 0x138;

__device__ __forceinline__ double wave_incl_sum(double v) {
    v += dpp_f64<kRowShr1, 0xf>(0.0, v);
    v += dpp_f64<kRowShr2, 0xf>(0.0, v);
    v += dpp_f64<kRowShr4, 0xf>(0.0, v);
    v += dpp_f64<kRowShr8, 0xf>(0.0, v);
    v += dpp_f64<kRowBcast15, 0xa>(0.0, v);
    v += dpp_f64<kRowBcast31, 0xc>(0.0, v);
    return v;
}

__device__ __forceinline__ double wave_incl_prod(double v) {
    v *= dpp_f64<kRowShr1, 0xf>(1.0, v);
    v *= dpp_f64<kRowShr2, 0xf>(1.0, v);
    v *= dpp_f64<kRowShr4, 0xf>(1.0, v);
    v *= dpp_f64<kRowShr8, 0xf>(1.0, v);
    v *= dpp_f64<kRowBcast15, 0xa>(1.0, v);
    v *= dpp_f64<kRowBcast31, 0xc>(1.0, v);
    return v;
}

__device__ __forceinline__ double lane63(double v) {
    const uint64_t s = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(s >> 32), 63);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ double wave_excl_prod(double v, int) { return dpp_f64<kWaveShr1, 0xf>(1.0, wave_incl_prod(v)); }
__device__ __forceinline__ double wave_excl_sum(double v, int) { return dpp_f64<kWaveShr1, 0xf>(0.0, wave_incl_sum(v)); }
__device__ __forceinline__ double wave_sum(double v) { return lane63(wave_incl_sum(v)); }

// exclusive sum over the lanes ABOVE this one (suffix) = total - inclusive prefix
__device__ __forceinline__ double wave_excl_suffix_sum(double v, int) {
    const double incl = wave_incl_sum(v);
    return lane63(incl) - incl;
}

// the value of the next / previous lane (`edge` for lane 63 / lane 0)
__device__ __forceinline__ float next_lane(float v, float edge) { return dpp_f32<kWaveShl1>(edge, v); }
__device__ __forceinline__ float prev_lane(float v, float edge) { return dpp_f32<kWaveShr1>(edge, v); }

// ------------------------------------------------------------------------------------------
// R1: ray generation (renderer.py:38-48).  One thread per ray.
// ------------------------------------------------------------------------------------------
__global__ void raygen_kernel(const int64_t* __restrict__ h, const int64_t* __restrict__ w, const float* __restrict__ K,
                              const float* __restrict__ E, int64_t n, float* __restrict__ rays_o,
                              float* __restrict__ rays_d) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* k = K + r * 9;
    const float* e = E + r * 16;
    // inverse of the 3x3 intrinsics by cofactors, carried in fp64 and rounded once
    double a = k[0], b = k[1], c = k[2], d = k[3], ee = k[4], f = k[5], g = k[6], hh = k[7], i = k[8];
    double A = ee * i - f * hh, B = -(d * i - f * g), C = d * hh - ee * g;
    double det = a * A + b * B + c * C;
    double id = 1.0 / det;
    float inv[9] = {(float)(A * id), (float)(-(b * i - c * hh) * id), (float)((b * f - c * ee) * id),
                    (float)(B * id), (float)((a * i - c * g) * id),   (float)(-(a * f - c * d) * id),
                    (float)(C * id), (float)(-(a * hh - b * g) * id), (float)((a * ee - b * d) * id)};
    float px = (float)w[r] + 0.5f, py = (float)h[r] + 0.5f, pz = 1.0f;
    float cx = fmaf(inv[2], pz, fmaf(inv[1], py, inv[0] * px));
    float cy = fmaf(inv[5], pz, fmaf(inv[4], py, inv[3] * px));
    float cz = fmaf(inv[8], pz, fmaf(inv[7], py, inv[6] * px));
    rays_d[r * 3 + 0] = fmaf(e[2], cz, fmaf(e[1], cy, e[0] * cx));
    rays_d[r * 3 + 1] = fmaf(e[6], cz, fmaf(e[5], cy, e[4] * cx));
    rays_d[r * 3 + 2] = fmaf(e[10], cz, fmaf(e[9], cy, e[8] * cx));
    rays_o[r * 3 + 0] = e[3];
    rays_o[r * 3 + 1] = e[7];
    rays_o[r * 3 + 2] = e[11];
}

// ------------------------------------------------------------------------------------------
// R3 + R4: sample points and contraction.  One thread per (ray, sample).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void contract_point(float& x, float& y, float& z, float inner, float outer) {
    float r = sqrtf(x * x + y * y + z * z);
    if (r > inner) {
        float s = (outer - (outer - inner) / r) / r;
        x *= s;
        y *= s;
        z *= s;
    }
}

__global__ void sample_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                     const float* __restrict__ zv, int64_t z_stride, int64_t n_rays, int S, float inner,
                                     float outer, float* __restrict__ xyz, float* __restrict__ s_xyz,
                                     float* __restrict__ dirs) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_rays * S) return;
    int64_t r = p / S;
    int s = (int)(p - r * S);
    float dx = rays_d[r * 3], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
    float z = zv[r * z_stride + s];
    float x = rays_o[r * 3] + dx * z, y = rays_o[r * 3 + 1] + dy * z, zz = rays_o[r * 3 + 2] + dz * z;
    if (xyz) {
        xyz[p * 3] = x;
        xyz[p * 3 + 1] = y;
        xyz[p * 3 + 2] = zz;
    }
    if (s_xyz) {
        contract_point(x, y, zz, inner, outer);
        s_xyz[p * 3] = x;
        s_xyz[p * 3 + 1] = y;
        s_xyz[p * 3 + 2] = zz;
    }
    if (dirs && s == 0) {
        float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        dirs[r * 3] = dx / nrm;
        dirs[r * 3 + 1] = dy / nrm;
        dirs[r * 3 + 2] = dz / nrm;
    }
}

__global__ void contract_kernel(const float* __restrict__ xyz, int64_t n, float inner, float outer,
                                float* __restrict__ out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float x = xyz[p * 3], y = xyz[p * 3 + 1], z = xyz[p * 3 + 2];
    contract_point(x, y, z, inner, outer);
    out[p * 3] = x;
    out[p * 3 + 1] = y;
    out[p * 3 + 2] = z;
}

// ------------------------------------------------------------------------------------------
// Per-ray kernels.  One wavefront owns one ray; lane `lane` owns the C contiguous samples [lane*C, lane*C + C) of the
// ray's rows, C = 1, 2, 4 or 8 (the smallest with 64*C >= S, a template parameter: no code for samples a launch does
// not have).  A row goes from HBM straight into the registers of the lanes that own it, as 4 / 8 / 16-byte pieces of
// one contiguous run per wave instruction (VEC: S a multiple of W = min(C, 4) and every row base 4W-byte aligned;
// else 4-byte pieces), neighbours across a lane boundary come from DPP shuffles, results leave the same way.  Only the
// resampling stage needs random access to a row (the CDF search): it keeps two rows per ray in LDS, sized by the
// launch's S.  The waves of a workgroup share nothing, so there are no workgroup barriers and a dead tail wave exits.
// ------------------------------------------------------------------------------------------
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int C> struct VecW { static constexpr int W = C < 4 ? C : 4; };

// v[0..N) = row[first .. first + N) where the index is below `limit`, else 0
template <int N, bool VEC, int W>
__device__ __forceinline__ void load_run(float (&v)[N], const float* __restrict__ row, int first, int limit) {
    if constexpr (VEC && W == 4) {
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            vf4 t = {0.f, 0.f, 0.f, 0.f};
            if (first + j < limit) t = *(const vf4*)(row + first + j);
            v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
        }
    } else if constexpr (VEC && W == 2) {
#pragma unroll
        for (int j = 0; j < N; j += 2) {
            vf2 t = {0.f, 0.f};
            if (first + j < limit) t = *(const vf2*)(row + first + j);
            v[j] = t.x; v[j + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = first + j < limit ? row[first + j] : 0.f;
    }
}

template <int N, bool VEC, int W>
__device__ __forceinline__ void store_run(float* __restrict__ row, const float (&v)[N], int first, int limit) {
    if constexpr (VEC && W == 4) {
#pragma unroll
        for (int j = 0; j < N; j += 4)
            if (first + j < limit) { vf4 t = {v[j], v[j + 1], v[j + 2], v[j + 3]}; *(vf4*)(row + first + j) = t; }
    } else if constexpr (VEC && W == 2) {
#pragma unroll
        for (int j = 0; j < N; j += 2)
            if (first + j < limit) { vf2 t = {v[j], v[j + 1]}; *(vf2*)(row + first + j) = t; }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (first + j < limit) row[first + j] = v[j];
    }
}

// LDS traffic of ONE wave: its own earlier writes are visible to all its lanes once the compiler keeps the order
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct SegCfg {
    int n_seg;
    int start[NERF_MAX_SEGMENTS + 1];
    float scale[NERF_MAX_SEGMENTS];
};

// renderer.py:73-79 / :154-160: alpha_i = 1 - exp(-sigma_i delta_i), T_i = prod_{j<i}(1 - alpha_j) (fp64 product, as the
// CPU cumprod accumulates), w_i = alpha_i T_i.  delta in log10 depth, 0 for the last sample.
template <int C>
struct Trans {
    float alpha[C], delta[C], wgt[C];
    double om_local[C];     // product of (1 - alpha) over the chunk's earlier samples
    double before;          // ... over the samples of all earlier lanes
};

template <int C>
__device__ __forceinline__ void transmittance(Trans<C>& t, const float (&sig)[C], const float (&zl)[C], int S, int lane) {
    const float zl_next_lane = next_lane(zl[0], zl[0]);
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        t.alpha[c] = 0.f;
        t.delta[c] = 0.f;
        t.om_local[c] = prod;
        if (i < S) {
            const float zn = c + 1 < C ? zl[c + 1 < C ? c + 1 : c] : zl_next_lane;
            const float d = (i < S - 1) ? (zn - zl[c]) : 0.f;
            const float a = 1.f - expf(-sig[c] * d);
            t.alpha[c] = a;
            t.delta[c] = d;
            prod *= (double)(1.f - a);
        }
    }
    t.before = wave_excl_prod(prod, lane);
#pragma unroll
    for (int c = 0; c < C; ++c) t.wgt[c] = (lane * C + c < S) ? t.alpha[c] * (float)(t.before * t.om_local[c]) : 0.f;
}

// R7 (renderer.py:132-142): w[C] -> pdf[C]
template <int C>
__device__ __forceinline__ void stage_reweight(const float (&w)[C], float (&pdf)[C], int S, int lane, float alpha_over_s,
                                               const SegCfg& seg) {
    const float w_prev_lane = prev_lane(w[C - 1], 0.f);        // lane 0: the zero padding of :133
    const float w_next_lane = next_lane(w[0], 0.f);
    float v[C];
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        v[c] = 0.f;
        if (i < S) {
            const float prev = c > 0 ? w[c > 0 ? c - 1 : 0] : w_prev_lane;
            float next = c + 1 < C ? w[c + 1 < C ? c + 1 : c] : w_next_lane;
            if (i + 1 >= S) next = 0.f;
            float x = 0.5f * (fmaxf(prev, w[c]) + fmaxf(w[c], next));
            x = x + alpha_over_s;
            float sc = 1.f;
            for (int k = 0; k < seg.n_seg; ++k)
                if (i >= seg.start[k] && i < seg.start[k + 1]) sc = seg.scale[k];
            x = x * sc;
            v[c] = x;
            local += (double)x;
        }
    }
    const float total = (float)wave_sum(local);
#pragma unroll
    for (int c = 0; c < C; ++c) pdf[c] = (lane * C + c < S) ? v[c] / total : 0.f;
}

// R8 (renderer.py:87-118, :165): pdf[C], zl[C] -> fine depths.  cdf_s / zl_s: this wave's two LDS rows of S floats.
template <int C, class NoiseFn>
__device__ __forceinline__ void stage_sample_pdf(const float (&pdf)[C], const float (&zl)[C], float* cdf_s, float* zl_s, int S,
                                                 int lane, NoiseFn noise, const float* __restrict__ lin, int Sf,
                                                 float* __restrict__ zlog_f, float* __restrict__ z_f,
                                                 int64_t* __restrict__ inds, bool pair_stores) {
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) local += (double)pdf[c];
    double run = wave_excl_sum(local, lane);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        run += (double)pdf[c];
        if (i < S) { cdf_s[i] = (float)run; zl_s[i] = zl[c]; }
    }
    wave_lds_sync();
    const float* cdf = cdf_s;
    const float c1 = cdf[1], range = cdf[S - 2] - c1, fSf = (float)Sf;
    const int n_knots = S - 3;  // cdf[1 .. S-3]
    // a lane takes the fine samples 2 lane, 2 lane + 1 (+ 128, ...): two neighbours share one Philox block and leave as
    // one 8-byte store
    for (int k0 = 2 * lane; k0 < Sf; k0 += 2 * kWave) {
        float nz[2];
        noise(k0, nz);
        float zf[2];
        int idx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + j < Sf ? k0 + j : k0;
            float u = lin[k] + nz[j] / fSf;
            u = u * range;
            u = u + c1;
            int lo = 0, hi = n_knots;
            while (lo < hi) {  // searchsorted(..., right=True): first knot strictly greater than u
                int mid = (lo + hi) >> 1;
                if (cdf[1 + mid] <= u) lo = mid + 1; else hi = mid;
            }
            idx[j] = lo + 1;
            float below = cdf[idx[j] - 1], above = cdf[idx[j]];
            float t = (u - below) / (above - below);
            float mb = (zl_s[idx[j] - 1] + zl_s[idx[j]]) / 2.f, ma = (zl_s[idx[j]] + zl_s[idx[j] + 1]) / 2.f;
            zf[j] = mb + (ma - mb) * t;
        }
        const bool two = k0 + 1 < Sf;
        if (two && pair_stores) {
            vf2 a = {zf[0], zf[1]};
            *(vf2*)(zlog_f + k0) = a;
            if (z_f) { vf2 b = {powf(10.f, zf[0]), powf(10.f, zf[1])}; *(vf2*)(z_f + k0) = b; }
        } else {
            zlog_f[k0] = zf[0];
            if (z_f) z_f[k0] = powf(10.f, zf[0]);
            if (two) { zlog_f[k0 + 1] = zf[1]; if (z_f) z_f[k0 + 1] = powf(10.f, zf[1]); }
        }
        if (inds) { inds[k0] = idx[0]; if (two) inds[k0 + 1] = idx[1]; }
    }
}

// mode 0: sigma -> weights ; 1: weights -> pdf ; 2: pdf -> fine ; 3: sigma -> fine (fused)
template <int MODE, int C, bool VEC>
__global__ __launch_bounds__(256) void ray_sampling_kernel(const float* __restrict__ in, const float* __restrict__ zlog,
                                                           int64_t zlog_stride, const float* __restrict__ noise,
                                                           const float* __restrict__ lin, int64_t n_rays, int S, int Sf,
                                                           float alpha_over_s, SegCfg seg, float* __restrict__ out_a,
                                                           float* __restrict__ out_b, int64_t* __restrict__ inds,
                                                           uint64_t seed, uint64_t counter, uint64_t first_ray = 0) {
    extern __shared__ float s_rows[];       // MODE 2 / 3: [kRaysPerBlock][2][S]
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    if (ray >= n_rays) return;
    float a[C], zl[C], w[C], pdf[C];
    load_run<C, VEC, W>(a, in + ray * S, lane * C, S);              // sigma (0, 3) / weights (1) / pdf (2)
    if (MODE != 1) load_run<C, VEC, W>(zl, zlog + ray * zlog_stride, lane * C, S);
    if (MODE == 0 || MODE == 3) {
        Trans<C> t;
        transmittance(t, a, zl, S, lane);
#pragma unroll
        for (int c = 0; c < C; ++c) w[c] = t.wgt[c];
        if (MODE == 0) { store_run<C, VEC, W>(out_a + ray * S, w, lane * C, S); return; }
    }
    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) w[c] = a[c];
    }
    if (MODE == 1 || MODE == 3) {
        stage_reweight<C>(w, pdf, S, lane, alpha_over_s, seg);
        if (MODE == 1) { store_run<C, VEC, W>(out_a + ray * S, pdf, lane * C, S); return; }
    }
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c) pdf[c] = a[c];
    }
    // stratification noise (torch.rand of renderer.py:102): the caller's array, or element ray * Sf + k of the Philox stream
    // (seed, counter) -- the same numbers nerf_uniform_noise writes out
    float* cdf_s = s_rows + (size_t)wv * 2 * S;
    float* zl_s = cdf_s + S;
    // (8-byte stores of sample pairs need even rows at 8-byte aligned bases)
    const bool pair_stores = (Sf & 1) == 0 && ((((uintptr_t)out_a) | ((uintptr_t)out_b)) & 7) == 0;
    if (noise) {
        const float* nrow = noise + ray * Sf;
        stage_sample_pdf<C>(pdf, zl, cdf_s, zl_s, S, lane,
                            [&](int k, float (&nz)[2]) { nz[0] = nrow[k]; nz[1] = k + 1 < Sf ? nrow[k + 1] : 0.f; }, lin, Sf,
                            out_a + ray * Sf, out_b ? out_b + ray * Sf : nullptr, inds ? inds + ray * Sf : nullptr, pair_stores);
    } else {
        // (first_ray: this launch covers rays [first_ray, first_ray + n_rays) of the batch the noise stream belongs to)
        const uint64_t e0 = (first_ray + (uint64_t)ray) * (uint64_t)Sf;
        stage_sample_pdf<C>(pdf, zl, cdf_s, zl_s, S, lane,
                            [&](int k, float (&nz)[2]) { philox_uniform2(seed, counter, e0 + (uint64_t)k, nz[0], nz[1]); }, lin, Sf,
                            out_a + ray * Sf, out_b ? out_b + ray * Sf : nullptr, inds ? inds + ray * Sf : nullptr, pair_stores);
    }
}

// ------------------------------------------------------------------------------------------
// R10: alpha compositing (renderer.py:71-85, 181-182).
// ------------------------------------------------------------------------------------------
template <int C, bool VEC>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                            const float* __restrict__ z, const float* __restrict__ zlog,
                                                            const float* __restrict__ bg, int64_t n_rays, int S,
                                                            float zlog_lo, float zlog_range, float* __restrict__ image,
                                                            float* __restrict__ invdepth, float* __restrict__ weights,
                                                            float* __restrict__ zlog_s) {
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    if (ray >= n_rays) return;
    float sig[C], zl[C], zz[C], col[3 * C];
    load_run<C, VEC, W>(sig, sigma + ray * S, lane * C, S);
    load_run<C, VEC, W>(zl, zlog + ray * S, lane * C, S);
    load_run<C, VEC, W>(zz, z + ray * S, lane * C, S);
    load_run<3 * C, VEC, W>(col, rgb + ray * S * 3, lane * 3 * C, 3 * S);
    Trans<C> t;
    transmittance(t, sig, zl, S, lane);
    double ar = 0, ag = 0, ab = 0, ainv = 0, aw = 0;
    float ss[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        ss[c] = 0.f;
        if (lane * C + c < S) {
            const float wgt = t.wgt[c];
            ar += (double)(wgt * col[c * 3]);
            ag += (double)(wgt * col[c * 3 + 1]);
            ab += (double)(wgt * col[c * 3 + 2]);
            ainv += (double)(wgt / zz[c]);
            aw += (double)wgt;
            ss[c] = (zl[c] - zlog_lo) / zlog_range;
        }
    }
    ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ainv = wave_sum(ainv); aw = wave_sum(aw);
    store_run<C, VEC, W>(weights + ray * S, t.wgt, lane * C, S);
    if (zlog_s) store_run<C, VEC, W>(zlog_s + ray * S, ss, lane * C, S);
    if (lane == 0) {
        float rest = 1.f - (float)aw;
        image[ray * 3] = (float)ar + rest * bg[0];
        image[ray * 3 + 1] = (float)ag + rest * bg[1];
        image[ray * 3 + 2] = (float)ab + rest * bg[2];
        invdepth[ray] = (float)ainv;
    }
}

// Backward: w_i = a_i T_i, T_i = prod_{j<i}(1-a_j), a_i = 1-exp(-sigma_i delta_i).
//   g_i       = dL/dw_i + dL/dimage . (c_i - bg)
//   dL/dsig_i = delta_i * ( g_i T_{i+1} - sum_{k>i} g_k w_k )
//   dL/dc_i   = w_i * dL/dimage
// g[C] in, ds[C] / dc[3C] out; returns true when a gradient of this lane is not finite
template <int C>
__device__ __forceinline__ bool composite_grads(const Trans<C>& t, const float (&g)[C], float gi0, float gi1, float gi2, int S,
                                                int lane, float (&ds)[C], float (&dc)[3 * C]) {
    double gw_local = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (lane * C + c < S) gw_local += (double)g[c] * (double)t.wgt[c];
    const double after = wave_excl_suffix_sum(gw_local, lane);  // sum over lanes above
    double suffix = after;                                      // walk the chunk from its last sample down
    bool bad = false;
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        ds[c] = 0.f;
        if (lane * C + c < S) {
            const double t_next = t.before * t.om_local[c] * (double)(1.f - t.alpha[c]);
            ds[c] = t.delta[c] * (float)((double)g[c] * t_next - suffix);
            suffix += (double)g[c] * (double)t.wgt[c];
        }
        dc[c * 3] = t.wgt[c] * gi0;
        dc[c * 3 + 1] = t.wgt[c] * gi1;
        dc[c * 3 + 2] = t.wgt[c] * gi2;
        bad |= !(fabsf(ds[c]) <= 3.402823466e38f) || !(fabsf(dc[c * 3]) <= 3.402823466e38f) ||
               !(fabsf(dc[c * 3 + 1]) <= 3.402823466e38f) || !(fabsf(dc[c * 3 + 2]) <= 3.402823466e38f);
    }
    return bad;
}

template <int C, bool VEC>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                            const float* __restrict__ zlog, const float* __restrict__ bg,
                                                            const float* __restrict__ dimage,
                                                            const float* __restrict__ dweights, int64_t n_rays, int S,
                                                            float* __restrict__ dsigma, float* __restrict__ drgb) {
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    if (ray >= n_rays) return;
    float sig[C], zl[C], dw[C], col[3 * C];
    load_run<C, VEC, W>(sig, sigma + ray * S, lane * C, S);
    load_run<C, VEC, W>(zl, zlog + ray * S, lane * C, S);
    load_run<3 * C, VEC, W>(col, rgb + ray * S * 3, lane * 3 * C, 3 * S);
    if (dweights) load_run<C, VEC, W>(dw, dweights + ray * S, lane * C, S);
    else {
#pragma unroll
        for (int c = 0; c < C; ++c) dw[c] = 0.f;
    }
    const float gi0 = dimage[ray * 3], gi1 = dimage[ray * 3 + 1], gi2 = dimage[ray * 3 + 2];
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    Trans<C> t;
    transmittance(t, sig, zl, S, lane);
    float g[C], ds[C], dc[3 * C];
#pragma unroll
    for (int c = 0; c < C; ++c)
        g[c] = (lane * C + c < S) ? dw[c] + gi0 * (col[c * 3] - b0) + gi1 * (col[c * 3 + 1] - b1) + gi2 * (col[c * 3 + 2] - b2) : 0.f;
    composite_grads<C>(t, g, gi0, gi1, gi2, S, lane, ds, dc);
    store_run<C, VEC, W>(dsigma + ray * S, ds, lane * C, S);
    store_run<3 * C, VEC, W>(drgb + ray * S * 3, dc, lane * 3 * C, 3 * S);
}

// ------------------------------------------------------------------------------------------
// R11: losses (helpers.py:21-32, trainer.py:153-169) and their gradients.
// ------------------------------------------------------------------------------------------
// Huber(delta = 0.1) and squared error of one colour channel (lanes 0..2 of a ray's wave)
__device__ __forceinline__ void huber(float d, double& l_rgb, double& l_mse, float& grad) {
    const float ad = fabsf(d);
    const float hd = 0.1f;
    l_rgb = ad <= hd ? 0.5 * (double)d * d : (double)hd * ((double)ad - 0.5 * hd);
    l_mse = (double)d * d;
    grad = ad <= hd ? d : (d > 0 ? hd : -hd);
}

// sum_ij w_i w_j |s_i - s_j| = 2 sum_i w_i (s_i W_<i - WS_<i) for non-decreasing s, and sum_i w_i (1 - s_i); gdw (nullable):
// d(distortion)/d(w_i)
template <int C, bool GRAD>
__device__ __forceinline__ void dist_depth(const float (&w)[C], const float (&s)[C], int S, int lane, double& r_dist,
                                           double& r_depth, float (&gdw)[C]) {
    double lw = 0, lws = 0;
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (lane * C + c < S) { lw += (double)w[c]; lws += (double)w[c] * (double)s[c]; }
    double pw = wave_excl_sum(lw, lane), pws = wave_excl_sum(lws, lane);
    double tw = 0, tws = 0;
    if (GRAD) { tw = wave_sum(lw); tws = wave_sum(lws); }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        gdw[c] = 0.f;
        if (lane * C + c < S) {
            const double wi = w[c], si = s[c];
            r_dist += 2.0 * wi * (si * pw - pws);
            if (GRAD) {
                const double aw = tw - pw - wi, aws = tws - pws - wi * si;  // strictly after i
                gdw[c] = (float)(2.0 * (si * pw - pws + aws - si * aw));
            }
            r_depth += wi * (1.0 - si);
            pw += wi;
            pws += wi * si;
        }
    }
}

// four per-wave partial sums -> one row per workgroup (s_part: [kRaysPerBlock][4] doubles of LDS)
__device__ __forceinline__ void loss_partials(double (*s_part)[4], double l_rgb, double l_dist, double l_depth, double l_mse,
                                              double inv_n, int S, int wv, int lane) {
    l_rgb = wave_sum(l_rgb); l_mse = wave_sum(l_mse); l_dist = wave_sum(l_dist); l_depth = wave_sum(l_depth);
    if (lane == 0) {
        s_part[wv][0] = l_rgb * inv_n / 3.0;
        s_part[wv][1] = l_dist * inv_n;
        s_part[wv][2] = l_depth * inv_n / (double)S;
        s_part[wv][3] = l_mse * inv_n / 3.0;
    }
    __syncthreads();
}

// sorted form, rows in registers.  A workgroup walks the rays with a grid-sized stride and adds its five totals once at
// the end: the fp64 atomics of all workgroups land on one 64-byte line and serialise there (4 ns each).
template <int C, bool VEC>
__global__ __launch_bounds__(256) void losses_sorted_kernel(const float* __restrict__ image, const float* __restrict__ gt,
                                                            const float* __restrict__ weights, const float* __restrict__ zs,
                                                            int64_t n_rays, int S, float lam_dist, float lam_depth,
                                                            float grad_scale, double* __restrict__ sums,
                                                            float* __restrict__ dimage, float* __restrict__ dweights) {
    __shared__ double s_part[kRaysPerBlock][4];
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double l_rgb = 0, l_mse = 0, l_dist = 0, l_depth = 0;
    const double inv_n = 1.0 / (double)n_rays;
    for (int64_t base = (int64_t)blockIdx.x * kRaysPerBlock; base < n_rays; base += (int64_t)gridDim.x * kRaysPerBlock) {
        const int64_t ray = base + wv;
        if (ray >= n_rays) continue;
        float w[C], s[C], gdw[C];
        load_run<C, VEC, W>(w, weights + ray * S, lane * C, S);
        load_run<C, VEC, W>(s, zs + ray * S, lane * C, S);
        if (lane < 3) {
            double r_rgb, r_mse;
            float gr;
            huber(image[ray * 3 + lane] - gt[ray * 3 + lane], r_rgb, r_mse, gr);
            l_rgb += r_rgb;
            l_mse += r_mse;
            if (dimage) dimage[ray * 3 + lane] = grad_scale * gr * (float)(inv_n / 3.0);
        }
        double r_dist = 0, r_depth = 0;
        dist_depth<C, true>(w, s, S, lane, r_dist, r_depth, gdw);
        l_dist += r_dist;
        l_depth += r_depth;
        if (dweights) {
            const float cd = grad_scale * lam_dist * (float)inv_n;
            const float cz = grad_scale * lam_depth * (float)(inv_n / (double)S);
            float o[C];
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = cd * gdw[c] + cz * (1.f - s[c]);
            store_run<C, VEC, W>(dweights + ray * S, o, lane * C, S);
        }
    }
    loss_partials(s_part, l_rgb, l_dist, l_depth, l_mse, inv_n, S, wv, lane);
    if (threadIdx.x < 4) {
        double v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        atomicAdd(&sums[threadIdx.x], v);
    } else if (threadIdx.x == 4) {   // the training loss itself (trainer.py:168), so that the host needs no arithmetic
        double v = 0;
        for (int k = 0; k < 4; ++k) v += s_part[k][0] + (double)lam_dist * s_part[k][1] + (double)lam_depth * s_part[k][2];
        atomicAdd(&sums[4], v);
        // non-finite guard (the role GradScaler plays at trainer.py:134-136): a NaN / Inf anywhere in image, weights or
        // zlog_s reaches this workgroup's total; the guarded optimizer entry points drop the step when sums[5] != 0
        if (!(fabs(v) <= 1.7976931348623157e308)) atomicAdd(&sums[5], 1.0);
    }
}

// The general form: colour-only calls (weights == nullptr: psnr, criterion_rgb) and the O(S^2) distortion sum for rows
// that are not sorted (helpers.py:23-25 as written); rows staged in LDS, kMaxS samples at most.
constexpr int kMaxC = kMaxS / kWave;
__device__ __forceinline__ void load_row(float* dst, const float* src, int n, int lane) {
    for (int i = lane; i < n; i += kWave) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void losses_kernel(const float* __restrict__ image, const float* __restrict__ gt,
                                                     const float* __restrict__ weights, const float* __restrict__ zs,
                                                     int64_t n_rays, int S, float lam_dist, float lam_depth,
                                                     float grad_scale, double* __restrict__ sums,
                                                     float* __restrict__ dimage, float* __restrict__ dweights) {
    __shared__ float s_w[kRaysPerBlock][kMaxS];
    __shared__ float s_s[kRaysPerBlock][kMaxS];
    __shared__ double s_part[kRaysPerBlock][4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C = (S + kWave - 1) / kWave;
    const bool need_ws = weights != nullptr;
    const float *w = s_w[wv], *s = s_s[wv];
    double l_rgb = 0, l_mse = 0, l_dist = 0, l_depth = 0;
    const double inv_n = 1.0 / (double)n_rays;
    for (int64_t base = (int64_t)blockIdx.x * kRaysPerBlock; base < n_rays; base += (int64_t)gridDim.x * kRaysPerBlock) {
        int64_t ray = base + wv;
        const bool live = ray < n_rays;
        if (!live) ray = n_rays - 1;
        __syncthreads();        // the previous round's readers of this wave's rows are done
        if (need_ws) {
            load_row(s_w[wv], weights + ray * S, S, lane);
            load_row(s_s[wv], zs + ray * S, S, lane);
        }
        __syncthreads();
        double r_rgb = 0, r_mse = 0, r_dist = 0, r_depth = 0;
        if (lane < 3) {
            float gr;
            huber(image[ray * 3 + lane] - gt[ray * 3 + lane], r_rgb, r_mse, gr);
            if (dimage && live) dimage[ray * 3 + lane] = grad_scale * gr * (float)(inv_n / 3.0);
        }
        if (need_ws) {
            float gdw[kMaxC];
#pragma unroll
            for (int c = 0; c < kMaxC; ++c) {
                int i = lane * C + c;
                gdw[c] = 0.f;
                if (c < C && i < S) {
                    double acc = 0;
                    for (int j = 0; j < S; ++j) acc += (double)w[j] * fabs((double)s[i] - (double)s[j]);
                    r_dist += (double)w[i] * acc;
                    gdw[c] = (float)(2.0 * acc);
                    r_depth += (double)w[i] * (1.0 - (double)s[i]);
                }
            }
            if (dweights && live) {
                const float cd = grad_scale * lam_dist * (float)inv_n;
                const float cz = grad_scale * lam_depth * (float)(inv_n / (double)S);
#pragma unroll
                for (int c = 0; c < kMaxC; ++c) {
                    int i = lane * C + c;
                    if (c < C && i < S) dweights[ray * S + i] = cd * gdw[c] + cz * (1.f - s[i]);
                }
            }
        }
        if (live) { l_rgb += r_rgb; l_mse += r_mse; l_dist += r_dist; l_depth += r_depth; }
    }
    loss_partials(s_part, l_rgb, l_dist, l_depth, l_mse, inv_n, S, wv, lane);
    if (threadIdx.x < 4) {
        double v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        atomicAdd(&sums[threadIdx.x], v);
    } else if (threadIdx.x == 4) {
        double v = 0;
        for (int k = 0; k < 4; ++k) v += s_part[k][0] + (double)lam_dist * s_part[k][1] + (double)lam_depth * s_part[k][2];
        atomicAdd(&sums[4], v);
        if (!(fabs(v) <= 1.7976931348623157e308)) atomicAdd(&sums[5], 1.0);
    }
}

// ------------------------------------------------------------------------------------------
// R10 + R11 fused for the training step: compositing, the three losses and (second kernel) the backward of both.
// The arithmetic is composite_fwd_kernel's, losses_sorted_kernel's and composite_bwd_kernel's, operation for
// operation -- the same image, weights and gradients bit for bit; only the order in which the workgroups add their loss
// totals differs.  What goes away: the weights / zlog_s round trip through HBM between the kernels, dimage / dweights
// altogether, and five small launches (zeroing, fp64 -> fp32 copy, two multiplies by d(loss), the separate compositing).
// ------------------------------------------------------------------------------------------
template <int C, bool VEC>
__global__ __launch_bounds__(256) void composite_loss_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ z, const float* __restrict__ zlog,
    const float* __restrict__ bg, const float* __restrict__ gt, int64_t n_rays, int S, float zlog_lo, float zlog_range,
    float* __restrict__ image, float* __restrict__ invdepth, float* __restrict__ weights,
    float* __restrict__ zlog_s, double* __restrict__ partials) {
    __shared__ double s_part[kRaysPerBlock][4];
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double inv_n = 1.0 / (double)n_rays;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    double l_rgb = 0, l_mse = 0, l_dist = 0, l_depth = 0;
    for (int64_t base = (int64_t)blockIdx.x * kRaysPerBlock; base < n_rays; base += (int64_t)gridDim.x * kRaysPerBlock) {
        const int64_t ray = base + wv;
        if (ray >= n_rays) continue;
        float sig[C], zl[C], zz[C], col[3 * C];
        load_run<C, VEC, W>(sig, sigma + ray * S, lane * C, S);
        load_run<C, VEC, W>(zl, zlog + ray * S, lane * C, S);
        load_run<C, VEC, W>(zz, z + ray * S, lane * C, S);
        load_run<3 * C, VEC, W>(col, rgb + ray * S * 3, lane * 3 * C, 3 * S);
        // ---- composite_fwd_kernel ----
        Trans<C> t;
        transmittance(t, sig, zl, S, lane);
        double ar = 0, ag = 0, ab = 0, ainv = 0, aw = 0;
        float ss[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            ss[c] = 0.f;
            if (lane * C + c < S) {
                const float wgt = t.wgt[c];
                ss[c] = (zl[c] - zlog_lo) / zlog_range;
                ar += (double)(wgt * col[c * 3]);
                ag += (double)(wgt * col[c * 3 + 1]);
                ab += (double)(wgt * col[c * 3 + 2]);
                ainv += (double)(wgt / zz[c]);
                aw += (double)wgt;
            }
        }
        ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ainv = wave_sum(ainv); aw = wave_sum(aw);
        const float rest = 1.f - (float)aw;
        const float im0 = (float)ar + rest * b0, im1 = (float)ag + rest * b1, im2 = (float)ab + rest * b2;
        store_run<C, VEC, W>(weights + ray * S, t.wgt, lane * C, S);
        if (zlog_s) store_run<C, VEC, W>(zlog_s + ray * S, ss, lane * C, S);
        if (lane == 0) {
            image[ray * 3] = im0; image[ray * 3 + 1] = im1; image[ray * 3 + 2] = im2;
            invdepth[ray] = (float)ainv;
        }
        // ---- losses_sorted_kernel on the same registers ----
        if (lane < 3) {
            const float im = lane == 0 ? im0 : (lane == 1 ? im1 : im2);
            double r_rgb, r_mse;
            float gr;
            huber(im - gt[ray * 3 + lane], r_rgb, r_mse, gr);
            l_rgb += r_rgb;
            l_mse += r_mse;
        }
        double r_dist = 0, r_depth = 0;
        float unused[C];
        dist_depth<C, false>(t.wgt, ss, S, lane, r_dist, r_depth, unused);
        l_dist += r_dist;
        l_depth += r_depth;
    }
    loss_partials(s_part, l_rgb, l_dist, l_depth, l_mse, inv_n, S, wv, lane);
    // one row of four partial sums per workgroup, plain stores: composite_loss_finish_kernel adds the rows up in a fixed
    // order (no same-line fp64 atomics from a thousand workgroups, no device-scope fence, and a reproducible total)
    if (threadIdx.x < 4)
        partials[(int64_t)blockIdx.x * 4 + threadIdx.x] =
            s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
}

// sums[0..5] and loss_out[0..5] from the per-workgroup rows: one workgroup, a fixed-shape tree
__global__ __launch_bounds__(256) void composite_loss_finish_kernel(const double* __restrict__ partials, int n_rows,
                                                                    float lam_dist, float lam_depth,
                                                                    double* __restrict__ sums, float* __restrict__ loss_out) {
    __shared__ double s_acc[256][4];
    double a[4] = {0, 0, 0, 0};
    for (int r = threadIdx.x; r < n_rows; r += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += partials[(int64_t)r * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s_acc[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step)
#pragma unroll
            for (int k = 0; k < 4; ++k) s_acc[threadIdx.x][k] += s_acc[threadIdx.x + step][k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double t0 = s_acc[0][0], t1 = s_acc[0][1], t2 = s_acc[0][2], t3 = s_acc[0][3];
        const double total = t0 + (double)lam_dist * t1 + (double)lam_depth * t2;
        const bool bad = !(fabs(total) <= 1.7976931348623157e308);
        sums[0] = t0; sums[1] = t1; sums[2] = t2; sums[3] = t3; sums[4] = total; sums[5] = bad ? 1.0 : 0.0;
        loss_out[0] = (float)t0; loss_out[1] = (float)t1; loss_out[2] = (float)t2; loss_out[3] = (float)t3;
        loss_out[4] = (float)total; loss_out[5] = bad ? 1.f : 0.f;
    }
}

// nonfinite (nullable): the step's non-finite flag (sums[5] of the forward call); raised when a gradient this kernel
// writes is NaN / Inf although the loss was finite (GradScaler.step skips on non-finite GRADIENTS, trainer.py:134-136)
template <int C, bool VEC>
__global__ __launch_bounds__(256) void composite_loss_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ zlog, const float* __restrict__ bg,
    const float* __restrict__ gt, const float* __restrict__ image, const float* __restrict__ dout, int64_t n_rays, int S,
    float zlog_lo, float zlog_range, float lam_dist, float lam_depth, float* __restrict__ dsigma, float* __restrict__ drgb,
    double* __restrict__ nonfinite) {
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    if (ray >= n_rays) return;
    float sig[C], zl[C], col[3 * C];
    load_run<C, VEC, W>(sig, sigma + ray * S, lane * C, S);
    load_run<C, VEC, W>(zl, zlog + ray * S, lane * C, S);
    load_run<3 * C, VEC, W>(col, rgb + ray * S * 3, lane * 3 * C, 3 * S);
    const float up = dout ? dout[0] : 1.f;
    const double inv_n = 1.0 / (double)n_rays;
    // d(loss)/d(image): Huber gradient, as losses_sorted_kernel writes it (grad_scale = 1), times the upstream gradient
    float gi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double r_rgb, r_mse;
        float gr;
        huber(image[ray * 3 + k] - gt[ray * 3 + k], r_rgb, r_mse, gr);
        gi[k] = (1.0f * gr * (float)(inv_n / 3.0)) * up;
    }
    const float gi0 = gi[0], gi1 = gi[1], gi2 = gi[2];
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    Trans<C> t;
    transmittance(t, sig, zl, S, lane);
    float ss[C], gdw[C], g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ss[c] = (lane * C + c < S) ? (zl[c] - zlog_lo) / zlog_range : 0.f;
    // d(loss)/d(weights): losses_sorted_kernel's form
    double r_dist = 0, r_depth = 0;
    dist_depth<C, true>(t.wgt, ss, S, lane, r_dist, r_depth, gdw);
    const float cd = 1.0f * lam_dist * (float)inv_n;
    const float cz = 1.0f * lam_depth * (float)(inv_n / (double)S);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        g[c] = 0.f;
        if (lane * C + c < S) {
            const float dw = (cd * gdw[c] + cz * (1.f - ss[c])) * up;
            g[c] = dw + gi0 * (col[c * 3] - b0) + gi1 * (col[c * 3 + 1] - b1) + gi2 * (col[c * 3 + 2] - b2);
        }
    }
    // ---- composite_bwd_kernel from here ----
    float ds[C], dc[3 * C];
    const bool bad = composite_grads<C>(t, g, gi0, gi1, gi2, S, lane, ds, dc);
    store_run<C, VEC, W>(dsigma + ray * S, ds, lane * C, S);
    store_run<3 * C, VEC, W>(drgb + ray * S * 3, dc, lane * 3 * C, 3 * S);
    if (bad && nonfinite) *nonfinite = 1.0;     // (every writer stores the same value)
}

// inference.py:97-105: one-hot of the first sample whose running weight sum reaches thresh
template <int C, bool VEC>
__global__ __launch_bounds__(256) void cum_thresh_kernel(const float* __restrict__ weights, int64_t n_rays, int S,
                                                         float thresh, float* __restrict__ out) {
    constexpr int W = VecW<C>::W;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    if (ray >= n_rays) return;
    float w[C];
    load_run<C, VEC, W>(w, weights + ray * S, lane * C, S);
    double local = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) local += (double)w[c];
    double run = wave_excl_sum(local, lane);
    int first = S;  // first index with (float)cumsum >= thresh in this lane's chunk
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane * C + c;
        if (i < S) {
            run += (double)w[c];
            if ((float)run >= thresh && first == S) first = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, kWave));
    float o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = (lane * C + c == first) ? 1.f : 0.f;
    store_run<C, VEC, W>(out + ray * S, o, lane * C, S);
}

// Launch shape of a per-ray kernel: samples per lane and whether the rows can move as vectors
struct RowShape { int C; bool vec; };
static RowShape row_shape(int S, int64_t extra_stride, std::initializer_list<const void*> rows) {
    RowShape r;
    r.C = S <= 64 ? 1 : (S <= 128 ? 2 : (S <= 256 ? 4 : 8));
    const int W = r.C < 4 ? r.C : 4;
    r.vec = W > 1 && S % W == 0 && extra_stride % W == 0;
    for (const void* p : rows)
        if (p && (((uintptr_t)p) & (uintptr_t)(4 * W - 1))) r.vec = false;
    return r;
}

// fn(std::integral_constant<int, C>, std::bool_constant<VEC>) launches the instantiation for the shape
template <class Fn>
static void row_dispatch(RowShape s, Fn&& fn) {
    using std::integral_constant;
    switch (s.C) {
        case 1: fn(integral_constant<int, 1>{}, std::false_type{}); break;
        case 2: if (s.vec) fn(integral_constant<int, 2>{}, std::true_type{}); else fn(integral_constant<int, 2>{}, std::false_type{}); break;
        case 4: if (s.vec) fn(integral_constant<int, 4>{}, std::true_type{}); else fn(integral_constant<int, 4>{}, std::false_type{}); break;
        default: if (s.vec) fn(integral_constant<int, 8>{}, std::true_type{}); else fn(integral_constant<int, 8>{}, std::false_type{}); break;
    }
}
#define ROW_C decltype(c_)::value
#define ROW_V decltype(v_)::value

__global__ void uniform_noise_kernel(uint64_t seed, uint64_t counter, int64_t n, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = philox_uniform(seed, counter, (uint64_t)e);
}

__global__ void zero_doubles_kernel(double* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.0;
}

inline hipStream_t S_(void* s) { return (hipStream_t)s; }

int check_ray_args(int64_t n_rays, int S) {
    if (n_rays <= 0) return 1;
    if (S < 1 || S > kMaxS) return 2;
    return 0;
}

}  // namespace

extern "C" {

int nerf_raygen(const int64_t* h, const int64_t* w, const float* K, const float* E, int64_t n_rays, float* rays_o,
                float* rays_d, void* stream) {
    NERF_REQUIRE(h && w && K && E && rays_o && rays_d, "null pointer");
    NERF_REQUIRE(n_rays >= 0, "negative ray count");
    if (n_rays == 0) return 0;
    raygen_kernel<<<(unsigned)ceil_div64(n_rays, 256), 256, 0, S_(stream)>>>(h, w, K, E, n_rays, rays_o, rays_d);
    return nerf_launch_status("nerf_raygen");
}

int nerf_sample_points(const float* rays_o, const float* rays_d, const float* z, int64_t z_stride, int64_t n_rays,
                       int n_samples, float inner_bound, float outer_bound, float* xyz, float* s_xyz, float* dirs,
                       void* stream) {
    NERF_REQUIRE(rays_o && rays_d && z, "null pointer");
    NERF_REQUIRE(n_rays >= 0 && n_samples >= 1, "bad sizes");
    if (n_rays == 0) return 0;
    int64_t m = n_rays * n_samples;
    sample_points_kernel<<<(unsigned)ceil_div64(m, 256), 256, 0, S_(stream)>>>(rays_o, rays_d, z, z_stride, n_rays, n_samples,
                                                                             inner_bound, outer_bound, xyz, s_xyz, dirs);
    return nerf_launch_status("nerf_sample_points");
}

int nerf_contract(const float* xyz, int64_t n_points, float inner_bound, float outer_bound, float* out, void* stream) {
    NERF_REQUIRE(xyz && out, "null pointer");
    NERF_REQUIRE(n_points >= 0, "negative count");
    if (n_points == 0) return 0;
    contract_kernel<<<(unsigned)ceil_div64(n_points, 256), 256, 0, S_(stream)>>>(xyz, n_points, inner_bound, outer_bound, out);
    return nerf_launch_status("nerf_contract");
}

static int fill_segments(SegCfg& seg, const int32_t* steps, int n_segments, int S, const char** err) {
    if (n_segments < 1 || n_segments > NERF_MAX_SEGMENTS) { *err = "segment count out of range"; return 1; }
    int total = 0;
    for (int k = 0; k < n_segments; ++k) total += steps[k];
    if (total != S) { *err = "sum(steps) != n_samples"; return 1; }
    seg.n_seg = n_segments;
    int c = 0;
    for (int k = 0; k < n_segments; ++k) {
        seg.start[k] = c;
        seg.scale[k] = (float)((double)steps[k] / (double)total);
        c += steps[k];
    }
    seg.start[n_segments] = c;
    return 0;
}

int nerf_coarse_weights(const float* sigma, const float* zlog, int64_t zlog_stride, int64_t n_rays, int n_samples,
                        float* weights, void* stream) {
    NERF_REQUIRE(sigma && zlog && weights, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    SegCfg seg = {};
    row_dispatch(row_shape(n_samples, zlog_stride, {sigma, zlog, weights}), [&](auto c_, auto v_) {
        ray_sampling_kernel<0, ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
            sigma, zlog, zlog_stride, nullptr, nullptr, n_rays, n_samples, 0, 0.f, seg, weights, nullptr, nullptr, 0, 0);
    });
    return nerf_launch_status("nerf_coarse_weights");
}

int nerf_reweight(const float* weights, int64_t n_rays, int n_samples, float alpha, const int32_t* steps, int n_segments,
                  float* pdf, void* stream) {
    NERF_REQUIRE(weights && steps && pdf, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    SegCfg seg = {};
    const char* err = nullptr;
    if (fill_segments(seg, steps, n_segments, n_samples, &err)) { nerf_set_error("nerf_reweight: %s", err); return NERF_EINVAL; }
    row_dispatch(row_shape(n_samples, 0, {weights, pdf}), [&](auto c_, auto v_) {
        ray_sampling_kernel<1, ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
            weights, nullptr, 0, nullptr, nullptr, n_rays, n_samples, 0, (float)((double)alpha / (double)n_samples), seg, pdf,
            nullptr, nullptr, 0, 0);
    });
    return nerf_launch_status("nerf_reweight");
}

int nerf_sample_pdf(const float* zlog, int64_t zlog_stride, const float* pdf, const float* noise, const float* lin,
                    int64_t n_rays, int n_samples, int n_importance, float* zlog_fine, float* z_fine, int64_t* inds,
                    void* stream) {
    NERF_REQUIRE(zlog && pdf && noise && lin && zlog_fine, "null pointer");
    NERF_REQUIRE(n_samples >= 4 && n_samples <= kMaxS, "n_samples must be in [4, 512]");
    NERF_REQUIRE(n_importance >= 1, "n_importance must be positive");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    SegCfg seg = {};
    const size_t lds = (size_t)kRaysPerBlock * 2 * n_samples * sizeof(float);
    row_dispatch(row_shape(n_samples, zlog_stride, {pdf, zlog}), [&](auto c_, auto v_) {
        ray_sampling_kernel<2, ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, lds, S_(stream)>>>(
            pdf, zlog, zlog_stride, noise, lin, n_rays, n_samples, n_importance, 0.f, seg, zlog_fine, z_fine, inds, 0, 0);
    });
    return nerf_launch_status("nerf_sample_pdf");
}

static int coarse_to_fine(const char* fn, const float* sigma, const float* zlog, int64_t zlog_stride, const float* noise,
                          uint64_t seed, uint64_t counter, const float* lin, int64_t n_rays, int n_samples, int n_importance,
                          float alpha, const int32_t* steps, int n_segments, float* zlog_fine, float* z_fine, void* stream,
                          int64_t first_ray = 0) {
    if (first_ray < 0) { nerf_set_error("%s: negative first_ray", fn); return NERF_EINVAL; }
    if (!(sigma && zlog && lin && steps && zlog_fine)) { nerf_set_error("%s: null pointer", fn); return NERF_EINVAL; }
    if (n_samples < 4 || n_samples > kMaxS) { nerf_set_error("%s: n_samples must be in [4, 512]", fn); return NERF_EINVAL; }
    if (n_importance < 1) { nerf_set_error("%s: n_importance must be positive", fn); return NERF_EINVAL; }
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    SegCfg seg = {};
    const char* err = nullptr;
    if (fill_segments(seg, steps, n_segments, n_samples, &err)) { nerf_set_error("%s: %s", fn, err); return NERF_EINVAL; }
    const size_t lds = (size_t)kRaysPerBlock * 2 * n_samples * sizeof(float);
    row_dispatch(row_shape(n_samples, zlog_stride, {sigma, zlog}), [&](auto c_, auto v_) {
        ray_sampling_kernel<3, ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, lds, S_(stream)>>>(
            sigma, zlog, zlog_stride, noise, lin, n_rays, n_samples, n_importance,
            (float)((double)alpha / (double)n_samples), seg, zlog_fine, z_fine, nullptr, seed, counter, (uint64_t)first_ray);
    });
    return nerf_launch_status(fn);
}

int nerf_coarse_to_fine(const float* sigma, const float* zlog, int64_t zlog_stride, const float* noise, const float* lin,
                        int64_t n_rays, int n_samples, int n_importance, float alpha, const int32_t* steps, int n_segments,
                        float* zlog_fine, float* z_fine, void* stream) {
    NERF_REQUIRE(noise, "null pointer");
    return coarse_to_fine("nerf_coarse_to_fine", sigma, zlog, zlog_stride, noise, 0, 0, lin, n_rays, n_samples, n_importance,
                          alpha, steps, n_segments, zlog_fine, z_fine, stream);
}

int nerf_coarse_to_fine_rng(const float* sigma, const float* zlog, int64_t zlog_stride, uint64_t seed, uint64_t counter,
                            const float* lin, int64_t n_rays, int n_samples, int n_importance, float alpha,
                            const int32_t* steps, int n_segments, float* zlog_fine, float* z_fine, void* stream) {
    return coarse_to_fine("nerf_coarse_to_fine_rng", sigma, zlog, zlog_stride, nullptr, seed, counter, lin, n_rays, n_samples,
                          n_importance, alpha, steps, n_segments, zlog_fine, z_fine, stream);
}

int nerf_coarse_to_fine_rng_at(const float* sigma, const float* zlog, int64_t zlog_stride, uint64_t seed, uint64_t counter,
                               int64_t first_ray, const float* lin, int64_t n_rays, int n_samples, int n_importance, float alpha,
                               const int32_t* steps, int n_segments, float* zlog_fine, float* z_fine, void* stream) {
    return coarse_to_fine("nerf_coarse_to_fine_rng_at", sigma, zlog, zlog_stride, nullptr, seed, counter, lin, n_rays, n_samples,
                          n_importance, alpha, steps, n_segments, zlog_fine, z_fine, stream, first_ray);
}

int nerf_uniform_noise(uint64_t seed, uint64_t counter, int64_t n, float* out, void* stream) {
    NERF_REQUIRE(out || n == 0, "null pointer");
    NERF_REQUIRE(n >= 0, "negative count");
    if (n == 0) return 0;
    uniform_noise_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, S_(stream)>>>(seed, counter, n, out);
    return nerf_launch_status("nerf_uniform_noise");
}

int nerf_composite_fwd(const float* sigma, const float* rgb, const float* z, const float* zlog, const float* bg,
                       int64_t n_rays, int n_samples, float zlog_lo, float zlog_range, float* image, float* invdepth,
                       float* weights, float* zlog_s, void* stream) {
    NERF_REQUIRE(sigma && rgb && z && zlog && bg && image && invdepth && weights, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    row_dispatch(row_shape(n_samples, 0, {sigma, rgb, z, zlog, weights, zlog_s}), [&](auto c_, auto v_) {
        composite_fwd_kernel<ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
            sigma, rgb, z, zlog, bg, n_rays, n_samples, zlog_lo, zlog_range, image, invdepth, weights, zlog_s);
    });
    return nerf_launch_status("nerf_composite_fwd");
}

int nerf_composite_bwd(const float* sigma, const float* rgb, const float* zlog, const float* bg, const float* dimage,
                       const float* dweights, int64_t n_rays, int n_samples, float* dsigma, float* drgb, void* stream) {
    NERF_REQUIRE(sigma && rgb && zlog && bg && dimage && dsigma && drgb, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    row_dispatch(row_shape(n_samples, 0, {sigma, rgb, zlog, dweights, dsigma, drgb}), [&](auto c_, auto v_) {
        composite_bwd_kernel<ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
            sigma, rgb, zlog, bg, dimage, dweights, n_rays, n_samples, dsigma, drgb);
    });
    return nerf_launch_status("nerf_composite_bwd");
}

int nerf_losses(const float* image, const float* gt, const float* weights, const float* zlog_s, int64_t n_rays,
                int n_samples, float lam_dist, float lam_depth, float grad_scale, int sorted, double* sums, float* dimage,
                float* dweights, void* stream) {
    NERF_REQUIRE(image && gt && sums, "null pointer");
    NERF_REQUIRE((weights == nullptr) == (zlog_s == nullptr), "weights and zlog_s go together");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    NERF_REQUIRE(n_rays > 0, "need at least one ray");
    zero_doubles_kernel<<<1, 64, 0, S_(stream)>>>(sums, 6);
    const int64_t loss_blocks = ceil_div64(n_rays, kRaysPerBlock);
    static const int loss_cap = getenv("NERF_LOSS_BLOCKS") ? atoi(getenv("NERF_LOSS_BLOCKS")) : kNumCU;
    const unsigned n_wg = (unsigned)(loss_blocks < loss_cap ? loss_blocks : loss_cap);
    if (weights && sorted) {
        row_dispatch(row_shape(n_samples, 0, {weights, zlog_s, dweights}), [&](auto c_, auto v_) {
            losses_sorted_kernel<ROW_C, ROW_V><<<n_wg, 256, 0, S_(stream)>>>(image, gt, weights, zlog_s, n_rays, n_samples, lam_dist,
                                                                          lam_depth, grad_scale, sums, dimage, dweights);
        });
    } else {
        losses_kernel<<<n_wg, 256, 0, S_(stream)>>>(image, gt, weights, zlog_s, n_rays, n_samples, lam_dist, lam_depth,
                                                    grad_scale, sums, dimage, dweights);
    }
    return nerf_launch_status("nerf_losses");
}

int nerf_composite_loss_fwd(const float* sigma, const float* rgb, const float* z, const float* zlog, const float* bg,
                            const float* gt, int64_t n_rays, int n_samples, float zlog_lo, float zlog_range, float lam_dist,
                            float lam_depth, float* image, float* invdepth, float* weights, float* zlog_s, double* sums,
                            float* loss_out, void* stream) {
    NERF_REQUIRE(sigma && rgb && z && zlog && bg && gt && image && invdepth && weights && sums && loss_out, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    NERF_REQUIRE(n_rays > 0, "need at least one ray");
    NERF_REQUIRE((((uintptr_t)sums) & 7) == 0, "sums must be 8-byte aligned");
    const int64_t blocks = ceil_div64(n_rays, kRaysPerBlock);
    const int n_wg = (int)(blocks < NERF_LOSS_MAX_ROWS ? blocks : NERF_LOSS_MAX_ROWS);
    row_dispatch(row_shape(n_samples, 0, {sigma, rgb, z, zlog, weights, zlog_s}), [&](auto c_, auto v_) {
        composite_loss_fwd_kernel<ROW_C, ROW_V><<<(unsigned)n_wg, 256, 0, S_(stream)>>>(
            sigma, rgb, z, zlog, bg, gt, n_rays, n_samples, zlog_lo, zlog_range, image, invdepth, weights, zlog_s, sums + 8);
    });
    composite_loss_finish_kernel<<<1, 256, 0, S_(stream)>>>(sums + 8, n_wg, lam_dist, lam_depth, sums, loss_out);
    return nerf_launch_status("nerf_composite_loss_fwd");
}

int nerf_composite_loss_bwd(const float* sigma, const float* rgb, const float* zlog, const float* bg, const float* gt,
                            const float* image, const float* dout, int64_t n_rays, int n_samples, float zlog_lo,
                            float zlog_range, float lam_dist, float lam_depth, float* dsigma, float* drgb,
                            double* nonfinite_flag, void* stream) {
    NERF_REQUIRE(sigma && rgb && zlog && bg && gt && image && dsigma && drgb, "null pointer");
    NERF_REQUIRE((((uintptr_t)nonfinite_flag) & 7) == 0, "nonfinite_flag must be 8-byte aligned");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    row_dispatch(row_shape(n_samples, 0, {sigma, rgb, zlog, dsigma, drgb}), [&](auto c_, auto v_) {
        composite_loss_bwd_kernel<ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
            sigma, rgb, zlog, bg, gt, image, dout, n_rays, n_samples, zlog_lo, zlog_range, lam_dist, lam_depth, dsigma, drgb,
            nonfinite_flag);
    });
    return nerf_launch_status("nerf_composite_loss_bwd");
}

int nerf_cum_weights_thresh(const float* weights, int64_t n_rays, int n_samples, float thresh, float* out, void* stream) {
    NERF_REQUIRE(weights && out, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    row_dispatch(row_shape(n_samples, 0, {weights, out}), [&](auto c_, auto v_) {
        cum_thresh_kernel<ROW_C, ROW_V><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(weights, n_rays,
                                                                                                          n_samples, thresh, out);
    });
    return nerf_launch_status("nerf_cum_weights_thresh");
}

}  // extern "C"
