// Per-ray kernels of the render path for gfx950 (MI355X): ray generation, sample points and
// scene contraction, coarse-pass weights, PDF re-weighting, inverse-CDF resampling, alpha
// compositing (forward + backward) and the fused training losses.
//
// All of them are HBM-bandwidth-bound streaming kernels: one 64-lane wavefront owns one ray,
// rows are staged through LDS with coalesced loads, and the along-ray recurrences
// (transmittance product, CDF sum, suffix sums) are wavefront prefix scans carried in fp64 and
// rounded to fp32 per element -- exactly how the reference's CPU cumsum/cumprod behave
// (SURVEY.md F6), which is what makes the resampled indices bit-exact.
//
// FP contraction is OFF in this file: the reference evaluates u*(range)+base etc. as separate
// roundings and the index parity depends on it.  fmaf is written out where a fused op is wanted.
#pragma clang fp contract(off)

#include <stdlib.h>
#include "common.h"
#include "philox.h"

namespace {

constexpr int kRaysPerBlock = 4;  // 256 threads, one wavefront per ray
constexpr int kMaxS = NERF_MAX_SAMPLES;
constexpr int kMaxC = kMaxS / kWave;  // samples per lane, contiguous

// ------------------------------------------------------------------------------------------
// wavefront scans (fp64)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_excl_prod(double v, int lane) {
    double incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        double o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl *= o;
    }
    double e = __shfl_up(incl, 1, kWave);
    return lane == 0 ? 1.0 : e;
}

__device__ __forceinline__ double wave_excl_sum(double v, int lane) {
    double incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        double o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    double e = __shfl_up(incl, 1, kWave);
    return lane == 0 ? 0.0 : e;
}

// exclusive sum over the lanes ABOVE this one (suffix)
__device__ __forceinline__ double wave_excl_suffix_sum(double v, int lane) {
    double incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        double o = __shfl_down(incl, off, kWave);
        if (lane + off < kWave) incl += o;
    }
    double e = __shfl_down(incl, 1, kWave);
    return lane == kWave - 1 ? 0.0 : e;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__device__ __forceinline__ void load_row(float* dst, const float* src, int n, int lane) {
    for (int i = lane; i < n; i += kWave) dst[i] = src[i];
}

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
// Per-ray stages on LDS rows.  Lane `lane` owns the contiguous samples [lane*C, lane*C + C).
// ------------------------------------------------------------------------------------------
struct SegCfg {
    int n_seg;
    int start[NERF_MAX_SEGMENTS + 1];
    float scale[NERF_MAX_SEGMENTS];
};

// R6 (renderer.py:154-160): sig[S], zl[S] in LDS -> wpad[0..S+1] in LDS (zero padded both ends).
__device__ __forceinline__ void stage_weights(const float* sig, const float* zl, float* wpad, int S, int C, int lane) {
    float alpha[kMaxC];
    double om_local[kMaxC];
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        alpha[c] = 0.f;
        om_local[c] = prod;
        if (c < C && i < S) {
            float delta = (i < S - 1) ? (zl[i + 1] - zl[i]) : 0.f;
            float a = 1.f - expf(-sig[i] * delta);
            alpha[c] = a;
            prod *= (double)(1.f - a);
        }
    }
    double before = wave_excl_prod(prod, lane);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) wpad[i + 1] = alpha[c] * (float)(before * om_local[c]);
    }
    if (lane == 0) {
        wpad[0] = 0.f;
        wpad[S + 1] = 0.f;
    }
}

// R7 (renderer.py:132-142): wpad -> pdf[S]
__device__ __forceinline__ void stage_reweight(const float* wpad, float* pdf, int S, int C, int lane, float alpha_over_s,
                                               const SegCfg& seg) {
    float v[kMaxC];
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        v[c] = 0.f;
        if (c < C && i < S) {
            float x = 0.5f * (fmaxf(wpad[i], wpad[i + 1]) + fmaxf(wpad[i + 1], wpad[i + 2]));
            x = x + alpha_over_s;
            float sc = 1.f;
            for (int k = 0; k < seg.n_seg; ++k)
                if (i >= seg.start[k] && i < seg.start[k + 1]) sc = seg.scale[k];
            x = x * sc;
            v[c] = x;
            local += (double)x;
        }
    }
    float total = (float)wave_sum(local);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) pdf[i] = v[c] / total;
    }
}

// R8 (renderer.py:87-118, :165): pdf[S] (LDS, overwritten by the cdf), zl[S] -> fine depths
template <class NoiseFn>
__device__ __forceinline__ void stage_sample_pdf(float* pdf_cdf, const float* zl, int S, int C, int lane,
                                                 NoiseFn noise, const float* __restrict__ lin, int Sf,
                                                 float* __restrict__ zlog_f, float* __restrict__ z_f,
                                                 int64_t* __restrict__ inds, bool live) {
    float p[kMaxC];
    double local = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        p[c] = (c < C && i < S) ? pdf_cdf[i] : 0.f;
        local += (double)p[c];
    }
    double run = wave_excl_sum(local, lane);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        run += (double)p[c];
        if (c < C && i < S) pdf_cdf[i] = (float)run;
    }
    __syncthreads();
    const float* cdf = pdf_cdf;
    const float c1 = cdf[1], range = cdf[S - 2] - c1, fSf = (float)Sf;
    const int n_knots = S - 3;  // cdf[1 .. S-3]
    for (int k = lane; k < Sf; k += kWave) {
        float u = lin[k] + noise(k) / fSf;
        u = u * range;
        u = u + c1;
        int lo = 0, hi = n_knots;
        while (lo < hi) {  // searchsorted(..., right=True): first knot strictly greater than u
            int mid = (lo + hi) >> 1;
            if (cdf[1 + mid] <= u) lo = mid + 1; else hi = mid;
        }
        int idx = lo + 1;
        float below = cdf[idx - 1], above = cdf[idx];
        float t = (u - below) / (above - below);
        float mb = (zl[idx - 1] + zl[idx]) / 2.f, ma = (zl[idx] + zl[idx + 1]) / 2.f;
        float zf = mb + (ma - mb) * t;
        if (!live) continue;
        zlog_f[k] = zf;
        if (z_f) z_f[k] = powf(10.f, zf);
        if (inds) inds[k] = idx;
    }
}

// mode 0: sigma -> weights ; 1: weights -> pdf ; 2: pdf -> fine ; 3: sigma -> fine (fused)
template <int MODE>
__global__ __launch_bounds__(256) void ray_sampling_kernel(const float* __restrict__ in, const float* __restrict__ zlog,
                                                           int64_t zlog_stride, const float* __restrict__ noise,
                                                           const float* __restrict__ lin, int64_t n_rays, int S, int Sf,
                                                           float alpha_over_s, SegCfg seg, float* __restrict__ out_a,
                                                           float* __restrict__ out_b, int64_t* __restrict__ inds,
                                                           uint64_t seed, uint64_t counter) {
    __shared__ float s_w[kRaysPerBlock][kMaxS + 2];
    __shared__ float s_p[kRaysPerBlock][kMaxS];
    __shared__ float s_z[kRaysPerBlock][kMaxS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool live = ray < n_rays;
    if (!live) ray = n_rays - 1;  // keep every wave on the barriers; stores are masked below
    const int C = (S + kWave - 1) / kWave;
    float* wpad = s_w[wv];
    float* pp = s_p[wv];
    float* zl = s_z[wv];
    if (MODE != 1) load_row(zl, zlog + ray * zlog_stride, S, lane);
    if (MODE == 0 || MODE == 3) load_row(pp, in + ray * S, S, lane);  // sigma
    if (MODE == 1) {
        load_row(wpad + 1, in + ray * S, S, lane);
        if (lane == 0) { wpad[0] = 0.f; wpad[S + 1] = 0.f; }
    }
    if (MODE == 2) load_row(pp, in + ray * S, S, lane);  // pdf
    __syncthreads();
    if (MODE == 0 || MODE == 3) {
        stage_weights(pp, zl, wpad, S, C, lane);
        __syncthreads();
        if (MODE == 0) {
            if (live) for (int i = lane; i < S; i += kWave) out_a[ray * S + i] = wpad[i + 1];
            return;
        }
    }
    if (MODE == 1 || MODE == 3) {
        stage_reweight(wpad, pp, S, C, lane, alpha_over_s, seg);
        __syncthreads();
        if (MODE == 1) {
            if (live) for (int i = lane; i < S; i += kWave) out_a[ray * S + i] = pp[i];
            return;
        }
    }
    // MODE 2 / 3.  Dead waves (tail block) run the stage on a clamped ray with stores masked.
    // stratification noise (torch.rand of renderer.py:102): the caller's array, or element ray * Sf + k of the Philox stream
    // (seed, counter) -- the same numbers nerf_uniform_noise writes out
    if (noise) {
        const float* nz = noise + ray * Sf;
        stage_sample_pdf(pp, zl, S, C, lane, [&](int k) { return nz[k]; }, lin, Sf, out_a + ray * Sf,
                         out_b ? out_b + ray * Sf : nullptr, inds ? inds + ray * Sf : nullptr, live);
    } else {
        const uint64_t e0 = (uint64_t)ray * (uint64_t)Sf;
        stage_sample_pdf(pp, zl, S, C, lane, [&](int k) { return philox_uniform(seed, counter, e0 + (uint64_t)k); }, lin, Sf,
                         out_a + ray * Sf, out_b ? out_b + ray * Sf : nullptr, inds ? inds + ray * Sf : nullptr, live);
    }
}

// ------------------------------------------------------------------------------------------
// R10: alpha compositing (renderer.py:71-85, 181-182), one wavefront per ray.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                            const float* __restrict__ z, const float* __restrict__ zlog,
                                                            const float* __restrict__ bg, int64_t n_rays, int S,
                                                            float zlog_lo, float zlog_range, float* __restrict__ image,
                                                            float* __restrict__ invdepth, float* __restrict__ weights,
                                                            float* __restrict__ zlog_s) {
    __shared__ float s_sig[kRaysPerBlock][kMaxS];
    __shared__ float s_zl[kRaysPerBlock][kMaxS];
    __shared__ float s_z[kRaysPerBlock][kMaxS];
    __shared__ float s_rgb[kRaysPerBlock][kMaxS * 3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool live = ray < n_rays;
    if (!live) ray = n_rays - 1;
    const int C = (S + kWave - 1) / kWave;
    load_row(s_sig[wv], sigma + ray * S, S, lane);
    load_row(s_zl[wv], zlog + ray * S, S, lane);
    load_row(s_z[wv], z + ray * S, S, lane);
    load_row(s_rgb[wv], rgb + ray * S * 3, S * 3, lane);
    __syncthreads();
    const float *sig = s_sig[wv], *zl = s_zl[wv], *zz = s_z[wv], *col = s_rgb[wv];
    float alpha[kMaxC];
    double om_local[kMaxC];
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        alpha[c] = 0.f;
        om_local[c] = prod;
        if (c < C && i < S) {
            float delta = (i < S - 1) ? (zl[i + 1] - zl[i]) : 0.f;
            float a = 1.f - expf(-sig[i] * delta);
            alpha[c] = a;
            prod *= (double)(1.f - a);
        }
    }
    double before = wave_excl_prod(prod, lane);
    double ar = 0, ag = 0, ab = 0, ainv = 0, aw = 0;
    float* wrow = s_sig[wv];  // reuse: each lane overwrites only its own chunk
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) {
            float wgt = alpha[c] * (float)(before * om_local[c]);
            ar += (double)(wgt * col[i * 3]);
            ag += (double)(wgt * col[i * 3 + 1]);
            ab += (double)(wgt * col[i * 3 + 2]);
            ainv += (double)(wgt / zz[i]);
            aw += (double)wgt;
            wrow[i] = wgt;
        }
    }
    ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ainv = wave_sum(ainv); aw = wave_sum(aw);
    __syncthreads();
    if (!live) return;
    for (int i = lane; i < S; i += kWave) {
        weights[ray * S + i] = wrow[i];
        if (zlog_s) zlog_s[ray * S + i] = (zl[i] - zlog_lo) / zlog_range;
    }
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
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                            const float* __restrict__ zlog, const float* __restrict__ bg,
                                                            const float* __restrict__ dimage,
                                                            const float* __restrict__ dweights, int64_t n_rays, int S,
                                                            float* __restrict__ dsigma, float* __restrict__ drgb) {
    __shared__ float s_sig[kRaysPerBlock][kMaxS];
    __shared__ float s_zl[kRaysPerBlock][kMaxS];
    __shared__ float s_dw[kRaysPerBlock][kMaxS];
    __shared__ float s_rgb[kRaysPerBlock][kMaxS * 3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool live = ray < n_rays;
    if (!live) ray = n_rays - 1;
    const int C = (S + kWave - 1) / kWave;
    load_row(s_sig[wv], sigma + ray * S, S, lane);
    load_row(s_zl[wv], zlog + ray * S, S, lane);
    if (dweights) load_row(s_dw[wv], dweights + ray * S, S, lane);
    else for (int i = lane; i < S; i += kWave) s_dw[wv][i] = 0.f;
    load_row(s_rgb[wv], rgb + ray * S * 3, S * 3, lane);
    __syncthreads();
    const float *sig = s_sig[wv], *zl = s_zl[wv], *dw = s_dw[wv];
    float* col = s_rgb[wv];
    const float gi0 = dimage[ray * 3], gi1 = dimage[ray * 3 + 1], gi2 = dimage[ray * 3 + 2];
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float alpha[kMaxC], delta[kMaxC];
    double om_local[kMaxC];
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        alpha[c] = 0.f;
        delta[c] = 0.f;
        om_local[c] = prod;
        if (c < C && i < S) {
            float d = (i < S - 1) ? (zl[i + 1] - zl[i]) : 0.f;
            float a = 1.f - expf(-sig[i] * d);
            alpha[c] = a;
            delta[c] = d;
            prod *= (double)(1.f - a);
        }
    }
    double before = wave_excl_prod(prod, lane);
    float wgt[kMaxC], g[kMaxC];
    double gw_local = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        wgt[c] = 0.f;
        g[c] = 0.f;
        if (c < C && i < S) {
            wgt[c] = alpha[c] * (float)(before * om_local[c]);
            g[c] = dw[i] + gi0 * (col[i * 3] - b0) + gi1 * (col[i * 3 + 1] - b1) + gi2 * (col[i * 3 + 2] - b2);
            gw_local += (double)g[c] * (double)wgt[c];
        }
    }
    double after = wave_excl_suffix_sum(gw_local, lane);  // sum over lanes above
    // walk the chunk from its last sample down, carrying the suffix sum
    double suffix = after;
    float ds[kMaxC];
#pragma unroll
    for (int c = kMaxC - 1; c >= 0; --c) {
        int i = lane * C + c;
        ds[c] = 0.f;
        if (c < C && i < S) {
            double t_next = before * om_local[c] * (double)(1.f - alpha[c]);
            ds[c] = delta[c] * (float)((double)g[c] * t_next - suffix);
            suffix += (double)g[c] * (double)wgt[c];
        }
    }
    __syncthreads();  // all reads of col[] done before it is overwritten with drgb
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) {
            s_sig[wv][i] = ds[c];
            col[i * 3] = wgt[c] * gi0;
            col[i * 3 + 1] = wgt[c] * gi1;
            col[i * 3 + 2] = wgt[c] * gi2;
        }
    }
    __syncthreads();
    if (!live) return;
    for (int i = lane; i < S; i += kWave) dsigma[ray * S + i] = s_sig[wv][i];
    for (int i = lane; i < S * 3; i += kWave) drgb[ray * S * 3 + i] = col[i];
}

// ------------------------------------------------------------------------------------------
// R11: losses (helpers.py:21-32, trainer.py:153-169) and their gradients, one wavefront per ray.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void losses_kernel(const float* __restrict__ image, const float* __restrict__ gt,
                                                     const float* __restrict__ weights, const float* __restrict__ zs,
                                                     int64_t n_rays, int S, float lam_dist, float lam_depth,
                                                     float grad_scale, int sorted, double* __restrict__ sums,
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
    // A workgroup walks the rays with a grid-sized stride and adds its five totals once at the end: the fp64 atomics of
    // all workgroups land on one 64-byte line and serialise there (4 ns each: 20 us with a workgroup per four rays).
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
    // Huber(delta = 0.1), mean over N*3
    if (lane < 3) {
        float d = image[ray * 3 + lane] - gt[ray * 3 + lane];
        float ad = fabsf(d);
        const float hd = 0.1f;
        r_rgb = ad <= hd ? 0.5 * (double)d * d : (double)hd * ((double)ad - 0.5 * hd);
        r_mse = (double)d * d;
        if (dimage && live) {
            float gr = ad <= hd ? d : (d > 0 ? hd : -hd);
            dimage[ray * 3 + lane] = grad_scale * gr * (float)(inv_n / 3.0);
        }
    }
    if (need_ws) {
        float gdw[kMaxC];
        if (sorted) {
            // sum_ij w_i w_j |s_i - s_j| = 2 sum_i w_i (s_i W_<i - WS_<i)  for non-decreasing s
            double lw = 0, lws = 0;
#pragma unroll
            for (int c = 0; c < kMaxC; ++c) {
                int i = lane * C + c;
                if (c < C && i < S) { lw += (double)w[i]; lws += (double)w[i] * (double)s[i]; }
            }
            double pw = wave_excl_sum(lw, lane), pws = wave_excl_sum(lws, lane);
            double tw = wave_sum(lw), tws = wave_sum(lws);
#pragma unroll
            for (int c = 0; c < kMaxC; ++c) {
                int i = lane * C + c;
                gdw[c] = 0.f;
                if (c < C && i < S) {
                    double wi = w[i], si = s[i];
                    r_dist += 2.0 * wi * (si * pw - pws);
                    double aw = tw - pw - wi, aws = tws - pws - wi * si;  // strictly after i
                    gdw[c] = (float)(2.0 * (si * pw - pws + aws - si * aw));
                    r_depth += wi * (1.0 - si);
                    pw += wi;
                    pws += wi * si;
                }
            }
        } else {
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
    l_rgb = wave_sum(l_rgb); l_mse = wave_sum(l_mse); l_dist = wave_sum(l_dist); l_depth = wave_sum(l_depth);
    if (lane == 0) {
        s_part[wv][0] = l_rgb * inv_n / 3.0;
        s_part[wv][1] = l_dist * inv_n;
        s_part[wv][2] = l_depth * inv_n / (double)S;
        s_part[wv][3] = l_mse * inv_n / 3.0;
    }
    __syncthreads();
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

// ------------------------------------------------------------------------------------------
// R10 + R11 fused for the training step: compositing, the three losses and (second kernel) the backward of both.
// The arithmetic is composite_fwd_kernel's, losses_kernel's (sorted form) and composite_bwd_kernel's, operation for
// operation -- the same image, weights and gradients bit for bit; only the order in which the workgroups add their loss
// totals differs.  What goes away: the weights / zlog_s round trip through HBM between the kernels, dimage / dweights
// altogether, and five small launches (zeroing, fp64 -> fp32 copy, two multiplies by d(loss), the separate compositing).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_loss_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ z, const float* __restrict__ zlog,
    const float* __restrict__ bg, const float* __restrict__ gt, int64_t n_rays, int S, float zlog_lo, float zlog_range,
    float* __restrict__ image, float* __restrict__ invdepth, float* __restrict__ weights,
    float* __restrict__ zlog_s, double* __restrict__ partials) {
    __shared__ float s_sig[kRaysPerBlock][kMaxS];
    __shared__ float s_zl[kRaysPerBlock][kMaxS];
    __shared__ float s_z[kRaysPerBlock][kMaxS];
    __shared__ float s_rgb[kRaysPerBlock][kMaxS * 3];
    __shared__ double s_part[kRaysPerBlock][4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C = (S + kWave - 1) / kWave;
    const double inv_n = 1.0 / (double)n_rays;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    double l_rgb = 0, l_mse = 0, l_dist = 0, l_depth = 0;
    for (int64_t base = (int64_t)blockIdx.x * kRaysPerBlock; base < n_rays; base += (int64_t)gridDim.x * kRaysPerBlock) {
        int64_t ray = base + wv;
        const bool live = ray < n_rays;
        if (!live) ray = n_rays - 1;
        __syncthreads();        // the previous round's readers of this wave's rows are done
        load_row(s_sig[wv], sigma + ray * S, S, lane);
        load_row(s_zl[wv], zlog + ray * S, S, lane);
        load_row(s_z[wv], z + ray * S, S, lane);
        load_row(s_rgb[wv], rgb + ray * S * 3, S * 3, lane);
        __syncthreads();
        const float *sig = s_sig[wv], *zl = s_zl[wv], *zz = s_z[wv], *col = s_rgb[wv];
        // ---- composite_fwd_kernel ----
        float alpha[kMaxC];
        double om_local[kMaxC];
        double prod = 1.0;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            int i = lane * C + c;
            alpha[c] = 0.f;
            om_local[c] = prod;
            if (c < C && i < S) {
                float delta = (i < S - 1) ? (zl[i + 1] - zl[i]) : 0.f;
                float a = 1.f - expf(-sig[i] * delta);
                alpha[c] = a;
                prod *= (double)(1.f - a);
            }
        }
        double before = wave_excl_prod(prod, lane);
        double ar = 0, ag = 0, ab = 0, ainv = 0, aw = 0;
        float wgt[kMaxC], ss[kMaxC];
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            int i = lane * C + c;
            wgt[c] = 0.f;
            ss[c] = 0.f;
            if (c < C && i < S) {
                wgt[c] = alpha[c] * (float)(before * om_local[c]);
                ss[c] = (zl[i] - zlog_lo) / zlog_range;
                ar += (double)(wgt[c] * col[i * 3]);
                ag += (double)(wgt[c] * col[i * 3 + 1]);
                ab += (double)(wgt[c] * col[i * 3 + 2]);
                ainv += (double)(wgt[c] / zz[i]);
                aw += (double)wgt[c];
            }
        }
        ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ainv = wave_sum(ainv); aw = wave_sum(aw);
        const float rest = 1.f - (float)aw;
        const float im0 = (float)ar + rest * b0, im1 = (float)ag + rest * b1, im2 = (float)ab + rest * b2;
        if (live) {
#pragma unroll
            for (int c = 0; c < kMaxC; ++c) {
                int i = lane * C + c;
                if (c < C && i < S) {
                    weights[ray * S + i] = wgt[c];
                    if (zlog_s) zlog_s[ray * S + i] = ss[c];
                }
            }
            if (lane == 0) {
                image[ray * 3] = im0; image[ray * 3 + 1] = im1; image[ray * 3 + 2] = im2;
                invdepth[ray] = (float)ainv;
            }
        }
        // ---- losses_kernel (sorted form) on the same registers ----
        double r_rgb = 0, r_mse = 0, r_dist = 0, r_depth = 0;
        if (lane < 3) {
            const float im = lane == 0 ? im0 : (lane == 1 ? im1 : im2);
            float d = im - gt[ray * 3 + lane];
            float ad = fabsf(d);
            const float hd = 0.1f;
            r_rgb = ad <= hd ? 0.5 * (double)d * d : (double)hd * ((double)ad - 0.5 * hd);
            r_mse = (double)d * d;
        }
        double lw = 0, lws = 0;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            int i = lane * C + c;
            if (c < C && i < S) { lw += (double)wgt[c]; lws += (double)wgt[c] * (double)ss[c]; }
        }
        double pw = wave_excl_sum(lw, lane), pws = wave_excl_sum(lws, lane);
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) {
            int i = lane * C + c;
            if (c < C && i < S) {
                double wi = wgt[c], si = ss[c];
                r_dist += 2.0 * wi * (si * pw - pws);
                r_depth += wi * (1.0 - si);
                pw += wi;
                pws += wi * si;
            }
        }
        if (live) { l_rgb += r_rgb; l_mse += r_mse; l_dist += r_dist; l_depth += r_depth; }
    }
    l_rgb = wave_sum(l_rgb); l_mse = wave_sum(l_mse); l_dist = wave_sum(l_dist); l_depth = wave_sum(l_depth);
    if (lane == 0) {
        s_part[wv][0] = l_rgb * inv_n / 3.0;
        s_part[wv][1] = l_dist * inv_n;
        s_part[wv][2] = l_depth * inv_n / (double)S;
        s_part[wv][3] = l_mse * inv_n / 3.0;
    }
    __syncthreads();
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

__global__ __launch_bounds__(256) void composite_loss_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ rgb, const float* __restrict__ zlog, const float* __restrict__ bg,
    const float* __restrict__ gt, const float* __restrict__ image, const float* __restrict__ dout, int64_t n_rays, int S,
    float zlog_lo, float zlog_range, float lam_dist, float lam_depth, float* __restrict__ dsigma, float* __restrict__ drgb) {
    __shared__ float s_sig[kRaysPerBlock][kMaxS];
    __shared__ float s_zl[kRaysPerBlock][kMaxS];
    __shared__ float s_rgb[kRaysPerBlock][kMaxS * 3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool live = ray < n_rays;
    if (!live) ray = n_rays - 1;
    const int C = (S + kWave - 1) / kWave;
    load_row(s_sig[wv], sigma + ray * S, S, lane);
    load_row(s_zl[wv], zlog + ray * S, S, lane);
    load_row(s_rgb[wv], rgb + ray * S * 3, S * 3, lane);
    __syncthreads();
    const float *sig = s_sig[wv], *zl = s_zl[wv];
    float* col = s_rgb[wv];
    const float up = dout ? dout[0] : 1.f;
    const double inv_n = 1.0 / (double)n_rays;
    // d(loss)/d(image): Huber gradient, as losses_kernel writes it (grad_scale = 1), times the upstream gradient
    float gi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float d = image[ray * 3 + k] - gt[ray * 3 + k];
        float ad = fabsf(d);
        const float hd = 0.1f;
        float gr = ad <= hd ? d : (d > 0 ? hd : -hd);
        gi[k] = (1.0f * gr * (float)(inv_n / 3.0)) * up;
    }
    const float gi0 = gi[0], gi1 = gi[1], gi2 = gi[2];
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float alpha[kMaxC], delta[kMaxC];
    double om_local[kMaxC];
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        alpha[c] = 0.f;
        delta[c] = 0.f;
        om_local[c] = prod;
        if (c < C && i < S) {
            float d = (i < S - 1) ? (zl[i + 1] - zl[i]) : 0.f;
            float a = 1.f - expf(-sig[i] * d);
            alpha[c] = a;
            delta[c] = d;
            prod *= (double)(1.f - a);
        }
    }
    double before = wave_excl_prod(prod, lane);
    float wgt[kMaxC], ss[kMaxC], g[kMaxC];
    double lw = 0, lws = 0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        wgt[c] = 0.f;
        ss[c] = 0.f;
        if (c < C && i < S) {
            wgt[c] = alpha[c] * (float)(before * om_local[c]);
            ss[c] = (zl[i] - zlog_lo) / zlog_range;
            lw += (double)wgt[c];
            lws += (double)wgt[c] * (double)ss[c];
        }
    }
    // d(loss)/d(weights): losses_kernel's sorted form
    double pw = wave_excl_sum(lw, lane), pws = wave_excl_sum(lws, lane);
    const double tw = wave_sum(lw), tws = wave_sum(lws);
    const float cd = 1.0f * lam_dist * (float)inv_n;
    const float cz = 1.0f * lam_depth * (float)(inv_n / (double)S);
    double gw_local = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        g[c] = 0.f;
        if (c < C && i < S) {
            const double wi = wgt[c], si = ss[c];
            const double aw = tw - pw - wi, aws = tws - pws - wi * si;      // strictly after i
            const float gdw = (float)(2.0 * (si * pw - pws + aws - si * aw));
            const float dw = (cd * gdw + cz * (1.f - ss[c])) * up;
            pw += wi;
            pws += wi * si;
            g[c] = dw + gi0 * (col[i * 3] - b0) + gi1 * (col[i * 3 + 1] - b1) + gi2 * (col[i * 3 + 2] - b2);
            gw_local += (double)g[c] * (double)wgt[c];
        }
    }
    // ---- composite_bwd_kernel from here ----
    double after = wave_excl_suffix_sum(gw_local, lane);
    double suffix = after;
    float ds[kMaxC];
#pragma unroll
    for (int c = kMaxC - 1; c >= 0; --c) {
        int i = lane * C + c;
        ds[c] = 0.f;
        if (c < C && i < S) {
            double t_next = before * om_local[c] * (double)(1.f - alpha[c]);
            ds[c] = delta[c] * (float)((double)g[c] * t_next - suffix);
            suffix += (double)g[c] * (double)wgt[c];
        }
    }
    __syncthreads();  // all reads of col[] done before it is overwritten with drgb
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) {
            s_sig[wv][i] = ds[c];
            col[i * 3] = wgt[c] * gi0;
            col[i * 3 + 1] = wgt[c] * gi1;
            col[i * 3 + 2] = wgt[c] * gi2;
        }
    }
    __syncthreads();
    if (!live) return;
    for (int i = lane; i < S; i += kWave) dsigma[ray * S + i] = s_sig[wv][i];
    for (int i = lane; i < S * 3; i += kWave) drgb[ray * S * 3 + i] = col[i];
}

// inference.py:97-105: one-hot of the first sample whose running weight sum reaches thresh
__global__ __launch_bounds__(256) void cum_thresh_kernel(const float* __restrict__ weights, int64_t n_rays, int S,
                                                         float thresh, float* __restrict__ out) {
    __shared__ float s_w[kRaysPerBlock][kMaxS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    const bool live = ray < n_rays;
    if (!live) ray = n_rays - 1;
    const int C = (S + kWave - 1) / kWave;
    load_row(s_w[wv], weights + ray * S, S, lane);
    __syncthreads();
    double local = 0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) local += (double)s_w[wv][i];
    }
    double run = wave_excl_sum(local, lane);
    int first = S;  // first index with (float)cumsum >= thresh in this lane's chunk
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) {
        int i = lane * C + c;
        if (c < C && i < S) {
            run += (double)s_w[wv][i];
            if ((float)run >= thresh && first == S) first = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, kWave));
    if (!live) return;
    for (int i = lane; i < S; i += kWave) out[ray * S + i] = (i == first) ? 1.f : 0.f;
}

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
    ray_sampling_kernel<0><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        sigma, zlog, zlog_stride, nullptr, nullptr, n_rays, n_samples, 0, 0.f, seg, weights, nullptr, nullptr, 0, 0);
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
    ray_sampling_kernel<1><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        weights, nullptr, 0, nullptr, nullptr, n_rays, n_samples, 0, (float)((double)alpha / (double)n_samples), seg, pdf,
        nullptr, nullptr, 0, 0);
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
    ray_sampling_kernel<2><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        pdf, zlog, zlog_stride, noise, lin, n_rays, n_samples, n_importance, 0.f, seg, zlog_fine, z_fine, inds, 0, 0);
    return nerf_launch_status("nerf_sample_pdf");
}

static int coarse_to_fine(const char* fn, const float* sigma, const float* zlog, int64_t zlog_stride, const float* noise,
                          uint64_t seed, uint64_t counter, const float* lin, int64_t n_rays, int n_samples, int n_importance,
                          float alpha, const int32_t* steps, int n_segments, float* zlog_fine, float* z_fine, void* stream) {
    if (!(sigma && zlog && lin && steps && zlog_fine)) { nerf_set_error("%s: null pointer", fn); return NERF_EINVAL; }
    if (n_samples < 4 || n_samples > kMaxS) { nerf_set_error("%s: n_samples must be in [4, 512]", fn); return NERF_EINVAL; }
    if (n_importance < 1) { nerf_set_error("%s: n_importance must be positive", fn); return NERF_EINVAL; }
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    SegCfg seg = {};
    const char* err = nullptr;
    if (fill_segments(seg, steps, n_segments, n_samples, &err)) { nerf_set_error("%s: %s", fn, err); return NERF_EINVAL; }
    ray_sampling_kernel<3><<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        sigma, zlog, zlog_stride, noise, lin, n_rays, n_samples, n_importance, (float)((double)alpha / (double)n_samples), seg,
        zlog_fine, z_fine, nullptr, seed, counter);
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
    composite_fwd_kernel<<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        sigma, rgb, z, zlog, bg, n_rays, n_samples, zlog_lo, zlog_range, image, invdepth, weights, zlog_s);
    return nerf_launch_status("nerf_composite_fwd");
}

int nerf_composite_bwd(const float* sigma, const float* rgb, const float* zlog, const float* bg, const float* dimage,
                       const float* dweights, int64_t n_rays, int n_samples, float* dsigma, float* drgb, void* stream) {
    NERF_REQUIRE(sigma && rgb && zlog && bg && dimage && dsigma && drgb, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    composite_bwd_kernel<<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        sigma, rgb, zlog, bg, dimage, dweights, n_rays, n_samples, dsigma, drgb);
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
    losses_kernel<<<(unsigned)(loss_blocks < loss_cap ? loss_blocks : loss_cap), 256, 0, S_(stream)>>>(
        image, gt, weights, zlog_s, n_rays, n_samples, lam_dist, lam_depth, grad_scale, sorted, sums, dimage, dweights);
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
    composite_loss_fwd_kernel<<<(unsigned)n_wg, 256, 0, S_(stream)>>>(sigma, rgb, z, zlog, bg, gt, n_rays, n_samples, zlog_lo,
                                                                    zlog_range, image, invdepth, weights, zlog_s, sums + 8);
    composite_loss_finish_kernel<<<1, 256, 0, S_(stream)>>>(sums + 8, n_wg, lam_dist, lam_depth, sums, loss_out);
    return nerf_launch_status("nerf_composite_loss_fwd");
}

int nerf_composite_loss_bwd(const float* sigma, const float* rgb, const float* zlog, const float* bg, const float* gt,
                            const float* image, const float* dout, int64_t n_rays, int n_samples, float zlog_lo,
                            float zlog_range, float lam_dist, float lam_depth, float* dsigma, float* drgb, void* stream) {
    NERF_REQUIRE(sigma && rgb && zlog && bg && gt && image && dsigma && drgb, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    composite_loss_bwd_kernel<<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(
        sigma, rgb, zlog, bg, gt, image, dout, n_rays, n_samples, zlog_lo, zlog_range, lam_dist, lam_depth, dsigma, drgb);
    return nerf_launch_status("nerf_composite_loss_bwd");
}

int nerf_cum_weights_thresh(const float* weights, int64_t n_rays, int n_samples, float thresh, float* out, void* stream) {
    NERF_REQUIRE(weights && out, "null pointer");
    NERF_REQUIRE(check_ray_args(n_rays, n_samples) != 2, "n_samples must be in [1, 512]");
    if (n_rays <= 0) return n_rays == 0 ? 0 : NERF_EINVAL;
    cum_thresh_kernel<<<(unsigned)ceil_div64(n_rays, kRaysPerBlock), 256, 0, S_(stream)>>>(weights, n_rays, n_samples, thresh, out);
    return nerf_launch_status("nerf_cum_weights_thresh");
}

}  // extern "C"
