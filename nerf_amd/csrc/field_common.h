// Device building blocks of the fused field network on gfx950:
//   * multiresolution hash-grid gather (tiny-cuda-nn HashGrid semantics, nets.py:39-50),
//   * the bias-free 64-wide MLPs as chains of v_mfma_f32_32x32x2_f32 (exact fp32: the result is
//     bit-for-bit a k-ordered fmaf chain), weights resident in LDS, activations in registers.
//
// Register/lane layout ("P layout"): a wavefront owns a tile of 32 sample points.  Lane l holds
// point (l & 31); the two 32-lane halves h = l >> 5 split the K dimension of every product.  An
// MFMA D = A*B with A = W[out][k] (from LDS) and B = act[k][point] (from registers) leaves
// D[out][point] with the point on the lane and out = (r&3) + 8*(r>>2) + 4*h in register r --
// which is exactly the B operand of the next layer, so activations never leave the registers.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef NERF_CHAIN_PROBE
#define NERF_CHAIN_PROBE 0
#endif
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int kHid = 64;
constexpr int kGeoRows = 16;  // sigma net output rows: sigma + 15 geo features
constexpr int kTilePts = 32;

struct GridDev {
    int L, F, LH;  // LH = ceil(L/2): levels handled by each lane half
    float scale[NERF_MAX_LEVELS];
    uint32_t res[NERF_MAX_LEVELS];
    uint32_t entries[NERF_MAX_LEVELS];
    uint32_t hashed[NERF_MAX_LEVELS];
    uint32_t offset[NERF_MAX_LEVELS];
};

// LDS image of the network weights: every layer row-major [out][pitch] with an ODD pitch, which
// makes both the forward read (lanes walk `out`) and the backward read (lanes walk `in`)
// bank-conflict free for ds_read_b32.
// Pitches and offsets are COMPILE-TIME constants (sized for the widest supported inputs: 64 hash-grid
// features, 96 colour-net inputs): with a run-time pitch every transposed read needs its own address
// VGPR, hipcc hoists all of them out of the tile loop and the kernel drowns in spills; with constants
// they fold into the 16-bit immediate offset of ds_read.
constexpr int kPS1 = 65, kPS2 = 65, kPC1 = 97, kPC2 = 65, kPC3 = 65;
constexpr int kWS1 = 0;
constexpr int kWS2 = kWS1 + kHid * kPS1;
constexpr int kWC1 = kWS2 + kGeoRows * kPS2;
constexpr int kWC2 = kWC1 + kHid * kPC1;
constexpr int kWC3 = kWC2 + kHid * kPC2;
constexpr int kNetLdsSigma = kWC1;              // floats: sigma net only
constexpr int kNetLdsFull = kWC3 + 8 * kPC3;    // floats: both nets (colour layer 3: rows 3..7 stay zero, see layer_T_Nx64)

struct NetLds {
    int in_s;    // L*F
    int in_sp;   // 2*LH*F (internal, zero padded when L is odd)
    int in_c;    // 15 + sh_dim + latent
    int sh_dim, sh_half, lat, lat_half;
    int total;   // floats of LDS used by the weights
    int64_t n_img;  // rows of latent_emb: an image index outside [0, n_img) reads a zero latent row and writes no gradient
};

static inline GridDev make_grid_dev(const NerfGridCfg* g) {
    GridDev d = {};
    d.L = g->n_levels;
    d.F = g->n_features;
    d.LH = (g->n_levels + 1) / 2;
    for (int l = 0; l < g->n_levels; ++l) {
        d.scale[l] = g->scale[l];
        d.res[l] = g->res[l];
        d.entries[l] = g->entries[l];
        d.hashed[l] = g->hashed[l];
        d.offset[l] = (uint32_t)g->offset[l];
    }
    return d;
}

static inline NetLds make_net_lds(const NerfGridCfg* g, const NerfNetCfg* n, bool with_color, int64_t n_images = 0) {
    NetLds s = {};
    s.in_s = g->n_levels * g->n_features;
    s.in_sp = 2 * ((g->n_levels + 1) / 2) * g->n_features;
    s.sh_dim = n->sh_degree * n->sh_degree;
    s.sh_half = s.sh_dim / 2;
    s.lat = n->latent_dim;
    s.lat_half = s.lat / 2;
    s.in_c = n->geo_feat_dim + s.sh_dim + s.lat;
    s.total = with_color ? kNetLdsFull : kNetLdsSigma;
    s.n_img = n_images;
    return s;
}

// ------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// select AFTER an unconditional load: `c ? p[i] : 0` makes hipcc branch around the ds_read
// (s_and_saveexec + s_waitcnt lgkmcnt(0) per MFMA); addresses are clamped by the callers instead.
__device__ __forceinline__ float sel0(bool c, float v) {
    asm volatile("" : "+v"(v));   // the value is opaque here, so the load cannot be sunk into a branch
    return c ? v : 0.f;
}

__device__ __forceinline__ int krow(int t) { return (t & 3) + 8 * (t >> 2); }  // + 4*h: row of register t

// copy one layer [out][in] from global into LDS [out][pitch] (pad columns stay zero)
__device__ __forceinline__ void lds_load_layer(float* lds, int pitch, const float* __restrict__ w, int out, int in) {
    for (int i = threadIdx.x; i < out * in; i += blockDim.x) {
        int o = i / in, k = i - o * in;
        lds[o * pitch + k] = w[i];
    }
}

__device__ __forceinline__ void lds_load_net(float* lds, const NetLds& s, const float* __restrict__ sw,
                                             const float* __restrict__ cw, bool with_color) {
    for (int i = threadIdx.x; i < s.total; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    lds_load_layer(lds + kWS1, kPS1, sw, kHid, s.in_s);
    lds_load_layer(lds + kWS2, kPS2, sw + kHid * s.in_s, kGeoRows, kHid);
    if (with_color) {
        lds_load_layer(lds + kWC1, kPC1, cw, kHid, s.in_c);
        lds_load_layer(lds + kWC2, kPC2, cw + kHid * s.in_c, kHid, kHid);
        lds_load_layer(lds + kWC3, kPC3, cw + kHid * s.in_c + kHid * kHid, 3, kHid);
    }
    __syncthreads();
}

struct PointIn {
    float x, y, z;     // normalised to [0,1]^3 (nets.py:117)
    float ux, uy, uz;  // before contraction (aux['xyzs'], renderer.py:191)
    float dx, dy, dz;  // unit view direction
    int64_t n;         // image index
};

// Every operation is an explicitly rounded intrinsic (IEEE add / mul / div / sqrt, nothing the compiler may contract,
// reassociate or turn into a reciprocal multiply): the kernels that visit a point -- forward gather, gradient scatter,
// and above all the count and fill passes of the binned table update, which must agree on the table entry of every
// corner -- all derive bit-identical coordinates from the same inputs.  (With plain `/` the LDS-sorted fill kernel
// and the count kernel disagreed in the last bit of x for ~3 % of the contracted points.)
__device__ __forceinline__ PointIn fetch_point(const NerfPointSrc& s, int64_t p, bool want_dir) {
    PointIn q;
    float x, y, z;
    q.dx = q.dy = q.dz = 0.f;
    q.n = 0;
    if (s.mode == 0) {
        x = s.x[p * 3]; y = s.x[p * 3 + 1]; z = s.x[p * 3 + 2];
        q.ux = x; q.uy = y; q.uz = z;
        if (want_dir) {
            q.dx = s.d[p * 3]; q.dy = s.d[p * 3 + 1]; q.dz = s.d[p * 3 + 2];
            q.n = s.n ? s.n[p] : 0;
        }
    } else {
        const int64_t r = (int64_t)((uint64_t)p / (uint32_t)s.n_samples);
        const int smp = (int)(p - r * s.n_samples);
        const float rx = s.rays_d[r * 3], ry = s.rays_d[r * 3 + 1], rz = s.rays_d[r * 3 + 2];
        const float zv = s.z[r * s.z_stride + smp];
        x = __fadd_rn(s.rays_o[r * 3], __fmul_rn(rx, zv));
        y = __fadd_rn(s.rays_o[r * 3 + 1], __fmul_rn(ry, zv));
        z = __fadd_rn(s.rays_o[r * 3 + 2], __fmul_rn(rz, zv));
        q.ux = x; q.uy = y; q.uz = z;
        const float rr = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
        if (rr > s.inner_bound) {  // renderer.py:64-68
            const float sc = __fdiv_rn(__fsub_rn(s.outer_bound, __fdiv_rn(__fsub_rn(s.outer_bound, s.inner_bound), rr)), rr);
            x = __fmul_rn(x, sc); y = __fmul_rn(y, sc); z = __fmul_rn(z, sc);
        }
        if (want_dir) {
            const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz)));
            q.dx = __fdiv_rn(rx, nrm); q.dy = __fdiv_rn(ry, nrm); q.dz = __fdiv_rn(rz, nrm);
            q.n = s.n ? s.n[r] : 0;
        }
    }
    const float two_ob = __fmul_rn(2.f, s.outer_bound);
    q.x = __fdiv_rn(__fadd_rn(x, s.outer_bound), two_ob);
    q.y = __fdiv_rn(__fadd_rn(y, s.outer_bound), two_ob);
    q.z = __fdiv_rn(__fadd_rn(z, s.outer_bound), two_ob);
    return q;
}

// this lane's half of the latent row (up to 24 floats, zero padded): 16-byte loads when the half row is a whole
// number of 16-byte quads at a 16-byte aligned offset (latent 48: 24 floats = 6 quads), else guarded scalars
__device__ __forceinline__ void load_latent_half(float (&lat)[24], const float* __restrict__ lrow, int lat_half) {
    if (lat_half == 24 && (((uintptr_t)lrow) & 15) == 0) {
        const f32x4* l4 = (const f32x4*)lrow;
#pragma unroll
        for (int qd = 0; qd < 6; ++qd) {
            const f32x4 v = l4[qd];
#pragma unroll
            for (int e = 0; e < 4; ++e) lat[4 * qd + e] = v[e];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 24; ++t) lat[t] = t < lat_half ? lrow[t] : 0.f;
    }
}

struct LevelP {
    float scale;
    uint32_t res, entries, hashed, offset;
    bool active;
};

// level handled by lane half h at step j: h*LH + j (inactive when past the last level)
__device__ __forceinline__ LevelP level_params(const GridDev& g, int j, int h) {
    const int l0 = j, l1 = min(g.LH + j, g.L - 1);
    LevelP p;
    p.scale = h ? g.scale[l1] : g.scale[l0];
    p.res = h ? g.res[l1] : g.res[l0];
    p.entries = h ? g.entries[l1] : g.entries[l0];
    p.hashed = h ? g.hashed[l1] : g.hashed[l0];
    p.offset = h ? g.offset[l1] : g.offset[l0];
    p.active = h ? (g.LH + j < g.L) : true;
    return p;
}

struct Corners {
    uint32_t idx[8];
    float w[8];
};

__device__ __forceinline__ Corners level_corners(const LevelP& lp, float x, float y, float z) {
    const float px = fmaf(x, lp.scale, 0.5f), py = fmaf(y, lp.scale, 0.5f), pz = fmaf(z, lp.scale, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const float wx = __fsub_rn(px, fx), wy = __fsub_rn(py, fy), wz = __fsub_rn(pz, fz);
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
    const float ax[2] = {__fsub_rn(1.f, wx), wx}, ay[2] = {__fsub_rn(1.f, wy), wy}, az[2] = {__fsub_rn(1.f, wz), wz};
    Corners c;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t cx = ix + (k & 1), cy = iy + ((k >> 1) & 1), cz = iz + ((k >> 2) & 1);
        c.w[k] = __fmul_rn(__fmul_rn(ax[k & 1], ay[(k >> 1) & 1]), az[(k >> 2) & 1]);
        uint32_t dense = cx + cy * lp.res + cz * (lp.res * lp.res);
#if defined(NERF_BRICK_PROBE)
        // TIMING PROBE ONLY (tools/ab builds, never shipped: not a bijection at the level's edges): the big dense levels
        // addressed in 4 x 4 x 4 bricks, to measure what a brick-ordered table layout would save in the forward gathers
        if (!lp.hashed && lp.entries > (1u << 20)) {
            const uint32_t nbr = (lp.res + 3u) >> 2;
            dense = ((((cz >> 2) * nbr + (cy >> 2)) * nbr + (cx >> 2)) << 6) | ((cz & 3u) << 4) | ((cy & 3u) << 2) | (cx & 3u);
        }
#endif
        if (dense >= lp.entries) dense %= lp.entries;
        const uint32_t hash = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (lp.entries - 1u);
        c.idx[k] = lp.offset + (lp.hashed ? hash : dense);
    }
    return c;
}

// HALF: `table` is the half-precision gather copy of the table (the reference's `precision: float16`, nets.py:31-36,49:
// tinycudann gathers from fp16 parameters beside an fp32 master): entries of F _Float16, widened on load, blended in fp32.
template <int F, bool HALF = false>
__device__ __forceinline__ void gather_level(const float* __restrict__ table, const Corners& c, float (&out)[F]) {
    float v[8][F];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (HALF) {
            typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            const _Float16* p = (const _Float16*)table + (size_t)c.idx[k] * F;
            if (F == 1) v[k][0] = (float)p[0];
            if (F == 2) { const h16x2 t = *(const h16x2*)p; v[k][0] = (float)t.x; v[k][F - 1] = (float)t.y; }
            if (F == 4) { const h16x4 t = *(const h16x4*)p; v[k][0] = (float)t.x; v[k][1 % F] = (float)t.y; v[k][2 % F] = (float)t.z; v[k][3 % F] = (float)t.w; }
            continue;
        }
        const float* p = table + (size_t)c.idx[k] * F;
        if (F == 1) v[k][0] = p[0];
        if (F == 2) { f32x2 t = *(const f32x2*)p; v[k][0] = t.x; v[k][1] = t.y; }
        if (F == 4) { f32x4 t = *(const f32x4*)p; v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w; }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) out[f] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int f = 0; f < F; ++f) out[f] = fmaf(c.w[k], v[k][f], out[f]);
}

// 16 real spherical harmonics of degree < 4 (tiny-cuda-nn SphericalHarmonics; input in [0,1])
__device__ __forceinline__ void sh16(float d0, float d1, float d2, float (&o)[16]) {
    const float x = (d0 + 1.f) / 2.f * 2.f - 1.f, y = (d1 + 1.f) / 2.f * 2.f - 1.f, z = (d2 + 1.f) / 2.f * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.f - 5.f * z2);
    o[12] = 0.3731763325901154f * z * (5.f * z2 - 3.f);
    o[13] = 0.45704579946446572f * x * (1.f - 5.f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.f * y2);
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
    return v;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    return v;
}

// Software-pipelined MFMA chain: acc[n] += A(step, n) * B(step), step = 0 .. STEPS-1.
// A comes from LDS, B from registers.  hipcc otherwise emits {ds_read; s_waitcnt lgkmcnt(0); mfma} per
// step, which exposes the LDS latency on every 64-cycle MFMA when only one wave lives on the SIMD;
// here the reads of group g+1 are issued (in program order) before the MFMAs of group g.
template <int STEPS, int NT, class AFn, class BFn>
__device__ __forceinline__ void mfma_chain(f32x16 (&acc)[NT], AFn a_at, BFn b_at) {
    constexpr int G = 4;
    float buf[2][G * NT];
#pragma unroll
    for (int s = 0; s < G; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n)
            if (s < STEPS) buf[0][s * NT + n] = a_at(s, n);
#pragma unroll
    for (int g = 0; g * G < STEPS; ++g) {
#pragma unroll
        for (int s = 0; s < G; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if ((g + 1) * G + s < STEPS) buf[(g + 1) & 1][s * NT + n] = a_at((g + 1) * G + s, n);
        // nothing crosses: the scheduler otherwise sinks every read to right before its MFMA again (shortest live range)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < G; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if (g * G + s < STEPS) {
#if NERF_CHAIN_PROBE
                    // timing probe only (wrong values): what the chains would cost at 3/8 of their matrix-core cycles, every LDS read kept
                    if ((g * G + s) % 8 < 3) acc[n] = MFMA32(buf[g & 1][s * NT + n], b_at(g * G + s), acc[n]);
                    else asm volatile("" :: "v"(buf[g & 1][s * NT + n]));
#else
                    acc[n] = MFMA32(buf[g & 1][s * NT + n], b_at(g * G + s), acc[n]);
#endif
                }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// k index (before the +4h of the lane half) of chain step s over a 64-row P-layout pair
__device__ __forceinline__ constexpr int krow64(int s) { return s < 16 ? krow(s) : 32 + krow(s - 16); }

// out(64) = W[64][64] * in(64): forward through one hidden->hidden layer (P layout in and out)
template <int pitch>
__device__ __forceinline__ void layer_64x64(const float* w, int l31, int h, const f32x16& i0,
                                            const f32x16& i1, f32x16& o0, f32x16& o1) {
    const float* r0 = w + l31 * pitch + 4 * h;
    const float* r1 = r0 + 32 * pitch;
    f32x16 acc[2] = {zero16(), zero16()};
    mfma_chain<32, 2>(acc, [&](int s, int n) { return (n ? r1 : r0)[krow64(s)]; },
                      [&](int s) { return s < 16 ? i0[s & 15] : i1[s & 15]; });
    o0 = acc[0];
    o1 = acc[1];
}

// out(rows < n_out <= 32) = W[n_out][64] * in(64)
template <int pitch>
__device__ __forceinline__ f32x16 layer_64xN(const float* w, int n_out, int l31, int h, const f32x16& i0,
                                             const f32x16& i1) {
    // lanes past n_out read row 0 again: their output rows are copies of row 0 that no caller consumes (a select on
    // the loaded value would pin every ds_read to an s_waitcnt right behind it)
    const float* r0 = w + (l31 < n_out ? l31 : 0) * pitch + 4 * h;
    f32x16 acc[1] = {zero16()};
    mfma_chain<32, 1>(acc, [&](int s, int) { return r0[krow64(s)]; },
                      [&](int s) { return s < 16 ? i0[s & 15] : i1[s & 15]; });
    return acc[0];
}

// d_in(64) = W^T * d_out where d_out has n_out live rows held in registers t < NSTEPS
// (n_out = 16 -> NSTEPS 8, n_out = 3 -> NSTEPS 3).  The caller passes d with ZEROS in its rows >= n_out, and the LDS
// image holds zero rows up to krow(NSTEPS-1) + 4 (kNetLdsFull), so neither operand needs a select.
template <int NSTEPS, int pitch>
__device__ __forceinline__ void layer_T_Nx64(const float* w, int n_out, int l31, int h, const f32x16& d,
                                             f32x16& o0, f32x16& o1) {
    (void)n_out;
    const float* r = w + 4 * h * pitch + l31;
    f32x16 acc[2] = {zero16(), zero16()};
    mfma_chain<NSTEPS, 2>(acc, [&](int s, int n) { return r[krow(s) * pitch + 32 * n]; }, [&](int s) { return d[s & 15]; });
    o0 = acc[0];
    o1 = acc[1];
}

// d_in(64) = W[64][64]^T * d_out(64)
template <int pitch>
__device__ __forceinline__ void layer_T_64x64(const float* w, int l31, int h, const f32x16& d0,
                                              const f32x16& d1, f32x16& o0, f32x16& o1) {
    const float* r = w + 4 * h * pitch + l31;
    f32x16 acc[2] = {zero16(), zero16()};
    mfma_chain<32, 2>(acc, [&](int s, int n) { return r[krow64(s) * pitch + 32 * n]; },
                      [&](int s) { return s < 16 ? d0[s & 15] : d1[s & 15]; });
    o0 = acc[0];
    o1 = acc[1];
}

// Colour layer 1 on [geo(15) | SH | latent]: 8 + 8 + 24 k-steps, the two lane halves splitting every
// group (inactive steps of smaller configurations multiply by zero).  o2 = sigma-net output rows.
// REF: the reference's shape (SH degree 4, latent 48) with its step -> input map as compile-time offsets.
template <bool REF = false>
__device__ __forceinline__ void color_layer1(const float* lds, const NetLds& s, int l31, int h, const f32x16& o2,
                                             const float (&sh)[16], const float (&lat)[24], f32x16& c0, f32x16& c1) {
    if (REF) {
        // in(st, h) = krow(st) - 1 + 4h | 15 + (st-8) + 8h | 31 + (st-16) + 24h: one base pointer per segment carries the
        // lane half, the rest folds into the immediate offset of the ds_read.  (st = 0, h = 0 is the sigma row: it reads
        // the pad float in front of the row and multiplies it by the zero below.)
        const float* wg = lds + kWC1 + l31 * kPC1 + 4 * h;
        const float* ws = lds + kWC1 + l31 * kPC1 + 8 * h;
        const float* wl = lds + kWC1 + l31 * kPC1 + 24 * h;
        f32x16 acc[2] = {zero16(), zero16()};
        mfma_chain<40, 2>(
            acc,
            [&](int st, int n) {
                const int o = n * 32 * kPC1;
                if (st < 8) return wg[o + krow(st) - 1];
                if (st < 16) return ws[o + 15 + (st - 8)];
                return wl[o + 31 + (st - 16)];
            },
            [&](int st) {
                if (st < 8) return (krow(st) + 4 * h == 0) ? 0.f : o2[st & 7];
                if (st < 16) return h ? sh[st & 15] : sh[(st - 8) & 15];
                return lat[(st - 16) % 24];
            });
        c0 = acc[0];
        c1 = acc[1];
        return;
    }
    const float* wc0 = lds + kWC1 + l31 * kPC1;
    const float* wc1 = wc0 + 32 * kPC1;
    const int sh_base = 15 + h * s.sh_half, lat_base = 15 + s.sh_dim + h * s.lat_half;
    f32x16 acc[2] = {zero16(), zero16()};
    mfma_chain<40, 2>(
        acc,
        [&](int st, int n) {
            int in;
            if (st < 8) { const int row = krow(st) + 4 * h; in = row == 0 ? 0 : row - 1; }
            else if (st < 16) in = (st - 8 < s.sh_half) ? sh_base + (st - 8) : 0;
            else in = (st - 16 < s.lat_half) ? lat_base + (st - 16) : 0;
            return (n ? wc1 : wc0)[in];
        },
        [&](int st) {
            if (st < 8) return (krow(st) + 4 * h == 0) ? 0.f : o2[st & 7];
            if (st < 16) {
                const int t = st - 8;
                const float v = s.sh_half == 2 ? (h ? sh[(2 + t) & 15] : sh[t]) : (h ? sh[(8 + t) & 15] : sh[t]);
                return t < s.sh_half ? v : 0.f;
            }
            return (st - 16 < s.lat_half) ? lat[(st - 16) % 24] : 0.f;
        });
    c0 = acc[0];
    c1 = acc[1];
}

#endif  // __HIPCC__
