// Fused field-network forward for gfx950: hash-grid encoding feeding the sigma MLP (and, for the
// full field, SH + latent + colour MLP) in one kernel.  The encoding is written straight into
// the MFMA B operand of the first layer; weights stay in LDS, activations in registers.
// Replaces NeRFNetwork.forward / .density (nets.py:109-164) and the tcnn objects they call.
#include <stdlib.h>

#include "field_common.h"

namespace {

#ifndef NERF_FWD_THREADS
#define NERF_FWD_THREADS 512   // 768 (three waves per SIMD, 167 VGPRs) measured slower: 0.669 vs 0.611 ms, more gathers in flight only thrash the L2
#endif
constexpr int kFwdMaxThreads = NERF_FWD_THREADS;

// One wavefront = 32 points x 2 lane halves; lane half h gathers levels [h*LH, h*LH + LH).
// FEATS: the hash-grid features come ready-made from xh_in (level-sharded table: the level owners encoded
// them and an all-to-all brought them here) instead of being gathered from `table`.
// HALF: `table` points at the half-precision gather copy (gather_level).
template <int F, bool COLOR, bool FEATS, bool HALF = false>
__global__ __launch_bounds__(kFwdMaxThreads) void field_fwd_kernel(GridDev g, NetLds s, const float* __restrict__ table,
                                                        const float* __restrict__ sigma_w,
                                                        const float* __restrict__ color_w,
                                                        const float* __restrict__ latent, NerfPointSrc src,
                                                        float* __restrict__ sigma_out, float* __restrict__ rgb_out,
                                                        float* __restrict__ xh_out, float* __restrict__ xyz_out,
                                                        const float* __restrict__ xh_in, int stage_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lds_load_net(lds, s, sigma_w, color_w, COLOR);

    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
    const int64_t M = src.n_points;
    const int64_t n_tiles = (M + kTilePts - 1) / kTilePts;
    const int LF = g.L * F;
    // x_hashtable rows of the tile are collected in a wave-private LDS area [32 points][L*F] (stage_off > 0) and leave as
    // whole 16-byte pieces of one contiguous 32-row block: written level by level straight from the gather loop, every
    // 160-byte row reached the L2 as twenty 8-byte pieces spread over the whole tile time and was evicted half filled
    // several times (366 MB of HBM writes for 98 MB of output, 0.09 ms of the kernel)
    float* stg = (!FEATS && xh_out && stage_off > 0) ? lds + stage_off + wave * (kTilePts * LF) : nullptr;

    for (int64_t tile = (int64_t)blockIdx.x * waves_per_block + wave; tile < n_tiles;
         tile += (int64_t)gridDim.x * waves_per_block) {
        int64_t p_raw = tile * kTilePts + l31;
        if (src.ray_block) {
            // traversal hint (nerf_hip.h): this tile is sample `smp` of the 32 rays of block `blk`
            // (ray_block = 32: one sample of 32 rays; 16 / 8: two / four consecutive samples of 16 / 8 rays)
            const int rb = src.ray_block, sp = kTilePts / rb;           // rays and samples of a tile
            const uint32_t tiles_per_blk = (uint32_t)(src.n_samples / sp);
            const int64_t blk = (int64_t)((uint64_t)tile / tiles_per_blk);
            const int smp = (int)(tile - blk * tiles_per_blk) * sp + l31 / rb;
            p_raw = (blk * rb + (l31 & (rb - 1))) * src.n_samples + smp;
        }
        const bool valid = p_raw < M;
        const int64_t p = valid ? p_raw : M - 1;
        PointIn q = {};
        if (!FEATS || COLOR) q = fetch_point(src, p, COLOR);
        float lat[24];   // this lane's half of the latent row, fetched under the hash-grid gathers
        if (COLOR) {
            // (the reference's latent_emb[n] raises for an index outside the table, nets.py:132: here such a point reads
            // a zero latent row instead of foreign memory)
            const bool n_ok = (uint64_t)q.n < (uint64_t)s.n_img;
            const float* lrow = latent + (n_ok ? q.n : 0) * s.lat + h * s.lat_half;
            load_latent_half(lat, lrow, s.lat_half);
            if (!n_ok) {
#pragma unroll
                for (int t = 0; t < 24; ++t) lat[t] = 0.f;
            }
        }

        // ---- sigma layer 1: encode a level, feed its F features as the next F k-steps ----
        f32x16 a0 = zero16(), a1 = zero16();
        const float* w1r0 = lds + kWS1 + l31 * kPS1 + h * g.LH * F;
        const float* w1r1 = w1r0 + 32 * kPS1;
        for (int j = 0; j < g.LH; ++j) {
            const LevelP lp = level_params(g, j, h);
            float feat[F];
            if (FEATS) {
                const float* fsrc = xh_in + p * LF + (lp.active ? h * g.LH + j : 0) * F;
                if (F == 2) { const f32x2 t = *(const f32x2*)fsrc; feat[0] = t.x; feat[F - 1] = t.y; }
                else if (F == 4) { const f32x4 t = *(const f32x4*)fsrc; feat[0] = t.x; feat[1 % F] = t.y; feat[2 % F] = t.z; feat[3 % F] = t.w; }
                else feat[0] = fsrc[0];
            } else {
                const Corners c = level_corners(lp, q.x, q.y, q.z);
                gather_level<F, HALF>(table, c, feat);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float b = lp.active ? feat[f] : 0.f;
                a0 = MFMA32(w1r0[j * F + f], b, a0);
                a1 = MFMA32(w1r1[j * F + f], b, a1);
            }
            if (!FEATS && stg) {
                if (lp.active) {
                    float* dst = stg + l31 * LF + (h * g.LH + j) * F;
                    if (F == 2) { f32x2 t; t.x = feat[0]; t.y = feat[F - 1]; *(f32x2*)dst = t; }
                    else {
#pragma unroll
                        for (int f = 0; f < F; ++f) dst[f] = feat[f];
                    }
                }
            } else if (!FEATS && xh_out && valid && lp.active) {
                float* dst = xh_out + p * LF + (h * g.LH + j) * F;
                if (F == 2) { f32x2 t; t.x = feat[0]; t.y = feat[F - 1]; *(f32x2*)dst = t; }
                else if (F == 4) { f32x4 t; t.x = feat[0]; t.y = feat[1 % F]; t.z = feat[2 % F]; t.w = feat[3 % F]; *(f32x4*)dst = t; }
                else dst[0] = feat[0];
            }
        }
        if (!FEATS && stg) {
            // (LDS instructions of one wave complete in issue order: the rows written above are visible to the reads here,
            // and the next tile's writes come after these reads)
            __builtin_amdgcn_wave_barrier();
            const int64_t first = tile * kTilePts;
            const int n_rows = (int)(M - first < kTilePts ? M - first : kTilePts);
            const int n_floats = n_rows * LF;
            float* gdst = xh_out + first * LF;
            if ((LF & 3) == 0 && ((((uintptr_t)xh_out) & 15) == 0)) {
                for (int i = lane * 4; i < n_floats; i += 4 * kWave) *(f32x4*)(gdst + i) = *(const f32x4*)(stg + i);
            } else {
                for (int i = lane; i < n_floats; i += kWave) gdst[i] = stg[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
        a0 = relu16(a0);
        a1 = relu16(a1);
        // ---- sigma layer 2: 64 -> 16 rows (row 0 = sigma, rows 1..15 = geo features) ----
        const f32x16 o2 = layer_64xN<kPS2>(lds + kWS2, kGeoRows, l31, h, a0, a1);
        if (valid && h == 0) sigma_out[p] = fmaxf(o2[0], 0.f);
        if (!COLOR) continue;

        // ---- colour layer 1: inputs [geo(15) | SH | latent], K split between the halves ----
        f32x16 c0, c1;
        {
            float sh[16];
            sh16(q.dx, q.dy, q.dz, sh);
            color_layer1(lds, s, l31, h, o2, sh, lat, c0, c1);
        }
        c0 = relu16(c0);
        c1 = relu16(c1);
        f32x16 e0, e1;
        layer_64x64<kPC2>(lds + kWC2, l31, h, c0, c1, e0, e1);
        e0 = relu16(e0);
        e1 = relu16(e1);
        const f32x16 o3 = layer_64xN<kPC3>(lds + kWC3, 3, l31, h, e0, e1);
        if (valid && h == 0) {
            rgb_out[p * 3 + 0] = 1.f / (1.f + expf(-o3[0]));
            rgb_out[p * 3 + 1] = 1.f / (1.f + expf(-o3[1]));
            rgb_out[p * 3 + 2] = 1.f / (1.f + expf(-o3[2]));
            if (xyz_out) {
                xyz_out[p * 3 + 0] = q.ux;
                xyz_out[p * 3 + 1] = q.uy;
                xyz_out[p * 3 + 2] = q.uz;
            }
        }
    }
}

// ---- stand-alone encodings (what tcnn.Encoding does when called on its own) ----
template <int F, bool HALF = false>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(GridDev g, const float* __restrict__ table,
                                                           const float* __restrict__ x01, int64_t M,
                                                           float* __restrict__ out) {
    // one thread per (point, level); consecutive threads walk the levels of one point
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * g.L) return;
    const int64_t p = i / g.L;
    const int l = (int)(i - p * g.L);
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    const Corners c = level_corners(lp, x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]);
    float feat[F];
    gather_level<F, HALF>(table, c, feat);
#pragma unroll
    for (int f = 0; f < F; ++f) out[i * F + f] = feat[f];
}

// Encoding of a level subset for any point source: one thread per (point, level slot).
template <int F, bool HALF = false>
__global__ __launch_bounds__(256) void encode_levels_kernel(GridDev g, const float* __restrict__ table, NerfPointSrc src,
                                                            NerfLevelSet set, float* __restrict__ out, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= src.n_points * set.count) return;
    const int64_t p = i / set.count;
    const int slot = (int)(i - p * set.count);
    const int l = set.level[slot];
    const PointIn q = fetch_point(src, p, false);
    LevelP lp = {g.scale[l], g.res[l], g.entries[l], g.hashed[l], g.offset[l], true};
    const Corners c = level_corners(lp, q.x, q.y, q.z);
    float feat[F];
    gather_level<F, HALF>(table, c, feat);
    float* dst = out + p * stride + slot * F;
    if (F == 2 && !(stride & 1) && !(((uintptr_t)out) & 7)) { f32x2 t; t.x = feat[0]; t.y = feat[F - 1]; *(f32x2*)dst = t; }
    else {
#pragma unroll
        for (int f = 0; f < F; ++f) dst[f] = feat[f];
    }
}

// Hash-table gradient scatter.  fp32 atomics execute at the memory side at a fixed rate of 64-byte
// requests (~20 G/s chip-wide), so the cost is the number of distinct 64-B lines per wave
// instruction.  Lanes are therefore laid out as (point, corner, feature): the 16 lanes of one
// (point, level) touch ~4-5 lines (corner pairs along x and the F features of an entry share a
// line) instead of one line per lane -- measured 3.4x faster than one thread per (point, level).
// grid.y = level, so the level parameters are scalar.
// Level slot blockIdx.y handles level plan.level[y]; its F gradient columns sit at dfeat[p*stride + plan.col[y]].
struct ScatterPlan {
    int count;
    int level[NERF_MAX_LEVELS];
    int col[NERF_MAX_LEVELS];
};

template <int F>
__global__ __launch_bounds__(256) void scatter_kernel(GridDev g, NerfPointSrc src, const float* __restrict__ x01,
                                                      const float* __restrict__ dfeat, int64_t M,
                                                      float* __restrict__ dtable, ScatterPlan plan, int64_t stride) {
    const int l = plan.level[blockIdx.y];
    const int col0 = plan.col[blockIdx.y];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t / (8 * F);
    if (p >= M) return;
    const int k = (int)(t / F) & 7, f = (int)(t % F);
    float x, y, z;
    if (x01) { x = x01[p * 3]; y = x01[p * 3 + 1]; z = x01[p * 3 + 2]; }
    else { const PointIn q = fetch_point(src, p, false); x = q.x; y = q.y; z = q.z; }
    const float sc = g.scale[l];
    const uint32_t res = g.res[l], entries = g.entries[l];
    const float px = fmaf(x, sc, 0.5f), py = fmaf(y, sc, 0.5f), pz = fmaf(z, sc, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const uint32_t cx = (uint32_t)(int)fx + (k & 1), cy = (uint32_t)(int)fy + ((k >> 1) & 1), cz = (uint32_t)(int)fz + ((k >> 2) & 1);
    const float w = __fmul_rn(__fmul_rn((k & 1) ? wx : 1.f - wx, (k & 2) ? wy : 1.f - wy), (k & 4) ? wz : 1.f - wz);
    uint32_t idx;
    if (g.hashed[l]) idx = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (entries - 1u);
    else { idx = cx + cy * res + cz * (res * res); if (idx >= entries) idx %= entries; }
    atomicAdd(dtable + ((size_t)g.offset[l] + idx) * F + f, w * dfeat[p * stride + col0 + f]);
}

// Small dense levels (the coarse end of the pyramid: 4 K .. 64 K entries) receive millions of atomics on a
// few hundred cache lines; at the memory side those serialise (level 0 alone cost as much as six hashed
// levels).  Here every workgroup accumulates into a private copy of (a chunk of) the level in LDS with
// ds_add_f32 and flushes the copy once with global atomics.  grid = (blocks per chunk, chunks);
// a chunk is `chunk_entries` consecutive entries of the level.
constexpr int kSmallLdsFloats = 32768;   // 128 KB of the 160 KB LDS
constexpr int kSmallMaxChunks = 16;
constexpr int kSmallThreads = 1024;

// One launch can take several levels (blockIdx.z picks one; blocks past a level's own grid leave at once): the smallest
// levels, one chunk or two each, then run side by side instead of as three launches in a row -- which matters when they
// have to wait for whole CUs behind the table update pass and the step ends with whatever of them is still queued.
constexpr int kSmallSetMax = 8;
struct SmallSet {
    int n;
    int level[kSmallSetMax], col[kSmallSetMax], chunk_entries[kSmallSetMax], chunks[kSmallSetMax], per_chunk[kSmallSetMax],
        run[kSmallSetMax];
};

template <int F>
__global__ __launch_bounds__(kSmallThreads) void scatter_small_kernel(GridDev g, NerfPointSrc src,
                                                                      const float* __restrict__ x01,
                                                                      const float* __restrict__ dfeat, int64_t M,
                                                                      float* __restrict__ dtable, SmallSet set,
                                                                      int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int z = blockIdx.z, l = set.level[z], col0 = set.col[z], chunk_entries = set.chunk_entries[z], run = set.run[z];
    const uint32_t nbx = (uint32_t)set.per_chunk[z];        // this level's blocks per chunk (the launch may be wider)
    if (blockIdx.x >= nbx || blockIdx.y >= (uint32_t)set.chunks[z]) return;
    constexpr int transpose = 1;
    const uint32_t res = g.res[l], entries = g.entries[l], res2 = res * res;
    const uint32_t lo = blockIdx.y * (uint32_t)chunk_entries;
    const uint32_t n_here = min((uint32_t)chunk_entries, entries - lo);
    for (uint32_t i = threadIdx.x; i < n_here * F; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const float sc = g.scale[l];
    if (run > 1) {
        // Rays mode, coarse level: a thread walks `run` consecutive samples of ONE ray and sums the corner updates of a
        // cell in registers for as long as the samples stay in that cell (13 samples per cell on level 0, 3-4 on level
        // 4): one round of 8 x F LDS atomics per visited cell instead of per sample.  Consecutive lanes take the same
        // run of consecutive rays -- unrelated pixels, so their cells rarely coincide.
        const int64_t S = src.n_samples, n_rays = M / S, runs_per_ray = S / run;
        const int64_t total = n_rays * runs_per_ray;
        for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (int64_t)nbx * blockDim.x) {
            const int64_t rblk = (int64_t)((uint64_t)v / (uint64_t)n_rays), ray = v - rblk * n_rays;
            const int64_t p0 = ray * S + rblk * run;
            uint32_t cx0 = 0xffffffffu, cy0 = 0, cz0 = 0;
            float sum[8][F];
            bool open = false;
            for (int j = 0; j <= run; ++j) {
                uint32_t ix = 0xfffffffeu, iy = 0, iz = 0;
                float wx = 0.f, wy = 0.f, wz = 0.f, d[F];
                bool have = false;      // a sample whose cell can touch this chunk (its gradient is only read then)
                if (j < run) {
                    const PointIn q = fetch_point(src, p0 + j, false);
                    const float px = fmaf(q.x, sc, 0.5f), py = fmaf(q.y, sc, 0.5f), pz = fmaf(q.z, sc, 0.5f);
                    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
                    wx = px - fx; wy = py - fy; wz = pz - fz;
                    ix = (uint32_t)(int)fx; iy = (uint32_t)(int)fy; iz = (uint32_t)(int)fz;
                    // in-range cells put their 8 corners in [iz*res2, (iz+3)*res2): skip those whose z slab misses this
                    // chunk (not near the wrap-around of the dense index at the top of the level, nor for stray points)
                    have = !(ix < res && iy < res && (uint64_t)(iz + 3) * res2 <= entries &&
                             ((iz + 3) * res2 <= lo || iz * res2 >= lo + n_here));
                    if (have) {
#pragma unroll
                        for (int f = 0; f < F; ++f) d[f] = dfeat[(p0 + j) * stride + col0 + f];
                    }
                }
                if (open && (!have || ix != cx0 || iy != cy0 || iz != cz0)) {      // leave the cell: flush its 8 corners
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        uint32_t idx = (cx0 + (k & 1)) + (cy0 + ((k >> 1) & 1)) * res + (cz0 + ((k >> 2) & 1)) * res2;
                        if (idx >= entries) idx %= entries;
                        const uint32_t rel = idx - lo;
                        if (rel < n_here) {
#pragma unroll
                            for (int f = 0; f < F; ++f) atomicAdd(acc + rel * F + f, sum[k][f]);
                        }
                    }
                    open = false;
                }
                if (have) {
                    if (!open) {
                        cx0 = ix; cy0 = iy; cz0 = iz; open = true;
#pragma unroll
                        for (int k = 0; k < 8; ++k)
#pragma unroll
                            for (int f = 0; f < F; ++f) sum[k][f] = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float w = __fmul_rn(__fmul_rn((k & 1) ? wx : 1.f - wx, (k & 2) ? wy : 1.f - wy), (k & 4) ? wz : 1.f - wz);
#pragma unroll
                        for (int f = 0; f < F; ++f) sum[k][f] += w * d[f];
                    }
                }
            }
        }
    } else {
    // Neighbouring samples of one ray fall into the same coarse cell, and ds_add_f32 serialises lanes that
    // hit one address: walk the points "transposed" (consecutive lanes = the same sample of consecutive rays).
    // Measured better than the natural order on every level this kernel takes (also the chunked ones).
    const int64_t cols = src.mode == 1 ? src.n_samples : 64;
    const int64_t rows = M / cols, body = transpose ? rows * cols : 0;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < M; v += (int64_t)nbx * blockDim.x) {
        int64_t p = v;
        if (v < body) { const int64_t c = (int64_t)((uint64_t)v / (uint64_t)rows); p = (v - c * rows) * cols + c; }
        float x, y, z;
        if (x01) { x = x01[p * 3]; y = x01[p * 3 + 1]; z = x01[p * 3 + 2]; }
        else { const PointIn q = fetch_point(src, p, false); x = q.x; y = q.y; z = q.z; }
        const float px = fmaf(x, sc, 0.5f), py = fmaf(y, sc, 0.5f), pz = fmaf(z, sc, 0.5f);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        const float wx = px - fx, wy = py - fy, wz = pz - fz;
        const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
        // in-range points put their 8 corners in [iz*res2, (iz+3)*res2): skip those whose z slab misses this
        // chunk (not near the wrap-around of the dense index at the top of the level, nor for stray points)
        if (ix < res && iy < res && (uint64_t)(iz + 3) * res2 <= entries &&
            ((iz + 3) * res2 <= lo || iz * res2 >= lo + n_here)) continue;
        float d[F];
#pragma unroll
        for (int f = 0; f < F; ++f) d[f] = dfeat[p * stride + col0 + f];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t cx = ix + (k & 1), cy = iy + ((k >> 1) & 1), cz = iz + ((k >> 2) & 1);
            const float w = __fmul_rn(__fmul_rn((k & 1) ? wx : 1.f - wx, (k & 2) ? wy : 1.f - wy), (k & 4) ? wz : 1.f - wz);
            uint32_t idx = cx + cy * res + cz * res2;
            if (idx >= entries) idx %= entries;
            const uint32_t rel = idx - lo;
            if (rel < n_here) {
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(acc + rel * F + f, w * d[f]);
            }
        }
    }
    }
    __syncthreads();
    // flush; every block starts its sweep at a different place so that the blocks do not chase each other
    const uint32_t n = n_here * F;
    const uint32_t start = (uint32_t)(((uint64_t)blockIdx.x * n) / nbx);
    float* dst = dtable + ((size_t)g.offset[l] + lo) * F;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t j = i + start;
        if (j >= n) j -= n;
        const float v = acc[j];
        if (v != 0.f) atomicAdd(dst + j, v);
    }
}

__global__ void sh_fwd_kernel(const float* __restrict__ d01, int64_t M, int dim, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    float sh[16];
    // sh16 expects unit directions and re-applies the (d+1)/2 map; undo it here
    sh16(d01[p * 3] * 2.f - 1.f, d01[p * 3 + 1] * 2.f - 1.f, d01[p * 3 + 2] * 2.f - 1.f, sh);
    for (int k = 0; k < dim; ++k) out[p * dim + k] = sh[k];
}

// Generic bias-free MLP, one thread per row (stand-alone tcnn.Network calls; not the hot path).
constexpr int kMlpMaxWidth = 128;
struct Dims9 { int v[9]; };
__global__ void mlp_fwd_kernel(const float* __restrict__ params, Dims9 d, int n_layers, const float* __restrict__ x,
                               int64_t rows, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float a[kMlpMaxWidth], b[kMlpMaxWidth];
    const int n_in = d.v[0];
    for (int k = 0; k < n_in; ++k) a[k] = x[r * n_in + k];
    const float* w = params;
    for (int l = 0; l < n_layers; ++l) {
        const int in = d.v[l], o = d.v[l + 1];
        for (int j = 0; j < o; ++j) {
            float acc = 0.f;
            for (int k = 0; k < in; ++k) acc = fmaf(w[j * in + k], a[k], acc);
            b[j] = (l + 1 < n_layers) ? fmaxf(acc, 0.f) : acc;
        }
        for (int j = 0; j < o; ++j) a[j] = b[j];
        w += in * o;
    }
    const int n_out = d.v[n_layers];
    for (int j = 0; j < n_out; ++j) out[r * n_out + j] = a[j];
}

}  // namespace

int nerf_check_cfg(const char* fn, const NerfGridCfg* grid, const NerfNetCfg* net, bool with_color) {
    if (!grid || !net) { nerf_set_error("%s: null configuration", fn); return NERF_EINVAL; }
    if (grid->n_levels < 1 || grid->n_levels > NERF_MAX_LEVELS || grid->total_entries >= (1ull << 32)) {
        nerf_set_error("%s: bad grid configuration", fn); return NERF_EINVAL;
    }
    if (grid->n_features != 1 && grid->n_features != 2 && grid->n_features != 4) {
        nerf_set_error("%s: unsupported: n_features_per_level must be 1, 2 or 4", fn); return NERF_EUNSUPPORTED;
    }
    if (net->hidden_dim != kHid || net->sigma_layers != 2 || net->geo_feat_dim != kGeoRows - 1) {
        nerf_set_error("%s: unsupported: fused kernels need hidden_dim 64, sigma_num_layers 2, geo_feat_dim 15", fn);
        return NERF_EUNSUPPORTED;
    }
    if (with_color && (net->color_layers != 3 || (net->sh_degree != 0 && net->sh_degree != 2 && net->sh_degree != 4) ||
                       net->latent_dim < 0 || (net->latent_dim & 1))) {
        nerf_set_error("%s: unsupported: fused kernels need color_num_layers 3, SH degree 0/2/4, even latent dim", fn);
        return NERF_EUNSUPPORTED;
    }
    return 0;
}

int nerf_check_src(const char* fn, const NerfPointSrc* src, bool with_color) {
    if (!src) { nerf_set_error("%s: null point source", fn); return NERF_EINVAL; }
    if (src->n_points < 0) { nerf_set_error("%s: negative point count", fn); return NERF_EINVAL; }
    if (src->mode == 0) {
        if (!src->x || (with_color && !src->d)) { nerf_set_error("%s: points mode needs x (and d)", fn); return NERF_EINVAL; }
    } else if (src->mode == 1) {
        if (!src->rays_o || !src->rays_d || !src->z || src->n_samples < 1 || src->n_points % src->n_samples) {
            nerf_set_error("%s: rays mode needs rays_o, rays_d, z and n_points = rays * n_samples", fn); return NERF_EINVAL;
        }
    } else { nerf_set_error("%s: unknown point-source mode", fn); return NERF_EINVAL; }
    if (src->ray_block != 0) {
        const int rb = src->ray_block;
        if ((rb != 8 && rb != 16 && rb != 32) || src->mode != 1 || src->n_points % ((int64_t)rb * src->n_samples) ||
            src->n_samples % (kTilePts / rb)) {
            nerf_set_error("%s: ray_block must be 0, or 8 / 16 / 32 in rays mode with a multiple of that many rays and a "
                           "multiple of 4 / 2 / 1 samples", fn);
            return NERF_EINVAL;
        }
    }
    if (!(src->outer_bound > 0.f)) { nerf_set_error("%s: outer_bound must be positive", fn); return NERF_EINVAL; }
    return 0;
}

// The forward kernels walk the tiles with a grid-sized stride, so the grid is what the chip holds at once: `per_cu`
// resident workgroups per CU (registers and LDS, from the occupancy query) -- a larger grid runs as a full round of
// workgroups followed by a partly filled one.
static unsigned fwd_grid(int64_t n_points, int per_cu, int waves_per_block, bool color, bool coherent) {
    const int64_t tiles = ceil_div64(n_points, kTilePts);
    static const int forced = getenv("NERF_FWD_BLOCKS_PER_CU") ? atoi(getenv("NERF_FWD_BLOCKS_PER_CU")) : 0;
    static const int forced_d = getenv("NERF_DENSITY_BLOCKS_PER_CU") ? atoi(getenv("NERF_DENSITY_BLOCKS_PER_CU")) : 0;
    // two workgroups (eight waves) per CU measured best for both kernels: the density kernel could hold five, but runs
    // 0.307 ms at two against 0.325 ms at five and 0.341 ms with a grid of eight per CU (more waves only thrash the L2
    // with more random gathers in flight); NERF_FWD_BLOCKS_PER_CU / NERF_DENSITY_BLOCKS_PER_CU override for tuning
    // (pixel-ordered rays, NerfPointSrc.ray_block: the density pass gains from a third workgroup, 71.7 -> 69.3 ms per
    // 800 x 800 image; the full field does not: 1 / 2 / 3 / 4 per CU = 89.6 / 71.7 / 77.4 / 72.9 ms)
    const int most = (!color && coherent) ? 3 : 2;
    if (per_cu > most) per_cu = most;
    if (color && forced > 0) per_cu = forced;
    if (!color && forced_d > 0) per_cu = forced_d;
    const int64_t cap = (int64_t)kNumCU * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    const int64_t want = ceil_div64(tiles, waves_per_block);
    return (unsigned)(want < cap ? want : cap);
}

template <typename K>
static int resident_blocks_per_cu(K kern, int threads, int lds_bytes) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, (size_t)lds_bytes) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = lds_bytes > 0 ? (160 * 1024) / lds_bytes : 8;      // LDS bound only
    }
    return n;
}

template <bool COLOR, bool FEATS, bool HALF = false>
static int launch_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sw, const float* cw,
                      const float* latent, int64_t n_images, const NerfPointSrc* src, float* sigma, float* rgb, float* xh,
                      float* xyz, const float* xh_in, void* stream) {
    const GridDev g = make_grid_dev(grid);
    const NetLds s = make_net_lds(grid, net, COLOR, n_images);
    int lds_bytes = s.total * (int)sizeof(float);
    // feature rows staged per wave (see the kernel): one 512-thread workgroup per CU instead of two of 256 threads, so
    // that the weights are held once and 8 x [32][L*F] floats fit beside them
    int threads = 256, stage_off = 0;
    const int stage_wave_floats = kTilePts * grid->n_levels * grid->n_features;
    if (!FEATS && xh && src->n_points >= (int64_t)kNumCU * 512 / 2) {
        const int off = (s.total + 3) & ~3;
        const int big = kFwdMaxThreads / 64;
        const int need = (off + big * stage_wave_floats) * (int)sizeof(float);
        if (need <= 112 * 1024) { threads = kFwdMaxThreads; stage_off = off; lds_bytes = need; }
    }
    const int waves = threads / 64;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(FF)                                                                                                     \
    do {                                                                                                               \
        auto kern = field_fwd_kernel<FF, COLOR, FEATS, HALF>;                                                               \
        static NerfLdsAttrCache lds_cache = {};   /* per instantiation */    \
        if (int e = nerf_ensure_dynamic_lds((const void*)kern, lds_bytes, &lds_cache, "field_fwd")) return e;    \
        static int per_cu = 0, per_cu_lds = -1;     /* per instantiation; the net shape rarely changes */              \
        if (per_cu_lds != lds_bytes) { per_cu = resident_blocks_per_cu(kern, threads, lds_bytes); per_cu_lds = lds_bytes; } \
        const unsigned nb = fwd_grid(src->n_points, threads > 256 ? (per_cu > 1 ? 1 : per_cu) : per_cu, waves, COLOR, src->ray_block != 0);   \
        kern<<<nb, threads, lds_bytes, st>>>(g, s, table, sw, cw, latent, *src, sigma, rgb, xh, xyz, xh_in, stage_off); \
    } while (0)
    switch (grid->n_features) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    return nerf_launch_status(COLOR ? "nerf_field_fwd" : "nerf_density_fwd");
}

template <bool HALF>
static int density_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                       const NerfPointSrc* src, float* sigma, float* x_hashtable, void* stream) {
    int rc = nerf_check_cfg("nerf_density_fwd", grid, net, false);
    if (rc) return rc;
    rc = nerf_check_src("nerf_density_fwd", src, false);
    if (rc) return rc;
    NERF_REQUIRE(table && sigma_params && sigma, "null pointer");
    NERF_REQUIRE(!(src->ray_block && x_hashtable), "ray_block is for passes that keep no x_hashtable");
    if (src->n_points == 0) return 0;
    return launch_fwd<false, false, HALF>(grid, net, table, sigma_params, nullptr, nullptr, 0, src, sigma, nullptr, x_hashtable,
                                          nullptr, nullptr, stream);
}

template <bool HALF>
static int field_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                     const float* color_params, const float* latent_emb, int64_t n_images, const NerfPointSrc* src,
                     float* sigma, float* rgb, float* x_hashtable, float* xyz, void* stream) {
    int rc = nerf_check_cfg("nerf_field_fwd", grid, net, true);
    if (rc) return rc;
    rc = nerf_check_src("nerf_field_fwd", src, true);
    if (rc) return rc;
    NERF_REQUIRE(table && sigma_params && color_params && sigma && rgb, "null pointer");
    NERF_REQUIRE(net->latent_dim == 0 || (latent_emb && n_images > 0), "latent embedding missing");
    NERF_REQUIRE(!(src->ray_block && x_hashtable), "ray_block is for passes that keep no x_hashtable");
    if (src->n_points == 0) return 0;
    return launch_fwd<true, false, HALF>(grid, net, table, sigma_params, color_params, latent_emb, n_images, src, sigma, rgb,
                                         x_hashtable, xyz, nullptr, stream);
}

template <bool HALF>
static int hashgrid_fwd(const NerfGridCfg* grid, const float* table, const float* x01, int64_t n_points, float* out,
                        void* stream) {
    NERF_REQUIRE(grid && table && x01 && out, "null pointer");
    NERF_REQUIRE(n_points >= 0 && grid->total_entries < (1ull << 32), "bad sizes");
    if (n_points == 0) return 0;
    const GridDev g = make_grid_dev(grid);
    const unsigned nb = (unsigned)ceil_div64(n_points * g.L, 256);
    hipStream_t st = (hipStream_t)stream;
    switch (grid->n_features) {
        case 1: hashgrid_fwd_kernel<1, HALF><<<nb, 256, 0, st>>>(g, table, x01, n_points, out); break;
        case 2: hashgrid_fwd_kernel<2, HALF><<<nb, 256, 0, st>>>(g, table, x01, n_points, out); break;
        case 4: hashgrid_fwd_kernel<4, HALF><<<nb, 256, 0, st>>>(g, table, x01, n_points, out); break;
        default: nerf_set_error("nerf_hashgrid_fwd: unsupported n_features"); return NERF_EUNSUPPORTED;
    }
    return nerf_launch_status("nerf_hashgrid_fwd");
}

extern "C" {

int nerf_density_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                     const NerfPointSrc* src, float* sigma, float* x_hashtable, void* stream) {
    return density_fwd<false>(grid, net, table, sigma_params, src, sigma, x_hashtable, stream);
}

int nerf_field_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                   const float* color_params, const float* latent_emb, int64_t n_images, const NerfPointSrc* src,
                   float* sigma, float* rgb, float* x_hashtable, float* xyz, void* stream) {
    return field_fwd<false>(grid, net, table, sigma_params, color_params, latent_emb, n_images, src, sigma, rgb, x_hashtable,
                            xyz, stream);
}

int nerf_hashgrid_fwd(const NerfGridCfg* grid, const float* table, const float* x01, int64_t n_points, float* out,
                      void* stream) {
    return hashgrid_fwd<false>(grid, table, x01, n_points, out, stream);
}

/* ---- the same three with the half-precision gather copy of the table (precision: float16) ---- */
int nerf_density_fwd_f16(const NerfGridCfg* grid, const NerfNetCfg* net, const void* table_f16, const float* sigma_params,
                         const NerfPointSrc* src, float* sigma, float* x_hashtable, void* stream) {
    return density_fwd<true>(grid, net, (const float*)table_f16, sigma_params, src, sigma, x_hashtable, stream);
}

int nerf_field_fwd_f16(const NerfGridCfg* grid, const NerfNetCfg* net, const void* table_f16, const float* sigma_params,
                       const float* color_params, const float* latent_emb, int64_t n_images, const NerfPointSrc* src,
                       float* sigma, float* rgb, float* x_hashtable, float* xyz, void* stream) {
    return field_fwd<true>(grid, net, (const float*)table_f16, sigma_params, color_params, latent_emb, n_images, src, sigma,
                           rgb, x_hashtable, xyz, stream);
}

int nerf_hashgrid_fwd_f16(const NerfGridCfg* grid, const void* table_f16, const float* x01, int64_t n_points, float* out,
                          void* stream) {
    return hashgrid_fwd<true>(grid, (const float*)table_f16, x01, n_points, out, stream);
}

static NerfLevelSet level_range(int begin, int count) {
    NerfLevelSet set = {};
    set.count = count;
    for (int i = 0; i < count; ++i) set.level[i] = begin + i;
    return set;
}

static int check_level_set(const char* fn, const NerfGridCfg* grid, const NerfLevelSet* set) {
    if (!set || set->count < 0 || set->count > grid->n_levels) { nerf_set_error("%s: bad level set", fn); return NERF_EINVAL; }
    for (int i = 0; i < set->count; ++i)
        if (set->level[i] < 0 || set->level[i] >= grid->n_levels) { nerf_set_error("%s: level out of range", fn); return NERF_EINVAL; }
    return 0;
}

static int launch_scatter(const NerfGridCfg* grid, const NerfPointSrc* src, const float* x01, const float* dfeat,
                          int64_t n_points, float* dtable, const NerfLevelSet& set, int64_t stride, int col0, void* stream) {
    const GridDev g = make_grid_dev(grid);
    NerfPointSrc s = {};
    if (src) s = *src;
    hipStream_t st = (hipStream_t)stream;
    const int F = grid->n_features;
    if (F != 1 && F != 2 && F != 4) { nerf_set_error("scatter: unsupported n_features"); return NERF_EUNSUPPORTED; }
    // small dense levels go through the LDS-privatised kernel (worth it once there are enough points to
    // amortise the flush), everything else through the lane-grouped global-atomic kernel
    ScatterPlan plan = {};
    // (largest level first: the chunked levels need whole CUs -- 128 KB of LDS per workgroup -- and starve once the
    // table update pass fills every CU with its 16 KB workgroups; the smallest level fits beside those)
    static const int small_order = getenv("NERF_SMALL_ORDER") ? atoi(getenv("NERF_SMALL_ORDER")) : 1;
    static const bool merge_small = getenv("NERF_SMALL_MERGE") ? atoi(getenv("NERF_SMALL_MERGE")) != 0 : true;
    SmallSet merged = {};
    auto launch_small = [&](const SmallSet& ss) -> int {
        int gx = 1, gy = 1, lds_bytes = 0;
        for (int k = 0; k < ss.n; ++k) {
            gx = ss.per_chunk[k] > gx ? ss.per_chunk[k] : gx;
            gy = ss.chunks[k] > gy ? ss.chunks[k] : gy;
            const int b = ss.chunk_entries[k] * F * (int)sizeof(float);
            lds_bytes = b > lds_bytes ? b : lds_bytes;
        }
        const dim3 nb((unsigned)gx, (unsigned)gy, (unsigned)ss.n);
#define LAUNCH_SMALL(FF)                                                                                                  \
        do {                                                                                                              \
            auto kern = scatter_small_kernel<FF>;                                                                         \
            static NerfLdsAttrCache lds_cache = {};   /* per instantiation */    \
            if (int e = nerf_ensure_dynamic_lds((const void*)kern, lds_bytes, &lds_cache, "scatter")) return e;    \
            kern<<<nb, kSmallThreads, lds_bytes, st>>>(g, s, x01, dfeat, n_points, dtable, ss, stride);                   \
        } while (0)
        switch (F) {
            case 1: LAUNCH_SMALL(1); break;
            case 2: LAUNCH_SMALL(2); break;
            default: LAUNCH_SMALL(4); break;
        }
#undef LAUNCH_SMALL
        return 0;
    };
    for (int ii = 0; ii < set.count; ++ii) {
        const int i = small_order ? set.count - 1 - ii : ii;
        const int l = set.level[i], col = col0 + i * F;
        const int64_t floats = (int64_t)grid->entries[l] * F;
        const bool small = !grid->hashed[l] && floats <= (int64_t)kSmallLdsFloats * kSmallMaxChunks && n_points >= 65536;
        if (!small) {
            plan.level[plan.count] = l;
            plan.col[plan.count++] = col;
            continue;
        }
        // rays mode: runs of consecutive samples per thread (scatter_small_kernel); NERF_SMALL_RUN=1 turns it off
        static const int run_max = getenv("NERF_SMALL_RUN") ? atoi(getenv("NERF_SMALL_RUN")) : 16;
        int run = 1;        // the longest power-of-two run that divides the ray and still leaves a thread for every lane
        if (!x01 && s.mode == 1)
            while (run * 2 <= run_max && s.n_samples % (run * 2) == 0 && n_points / (run * 2) >= (int64_t)kNumCU * kSmallThreads)
                run *= 2;
        const int chunks = (int)ceil_div64(floats, kSmallLdsFloats);
        const int chunk_entries = (int)ceil_div64(grid->entries[l], chunks);
        int64_t per_chunk = kNumCU / chunks;
        const int64_t want = ceil_div64(n_points, kSmallThreads);
        if (per_chunk > want) per_chunk = want;
        if (per_chunk < 1) per_chunk = 1;
        // the levels of one or two chunks share a launch (issued after the chunked ones), the others get their own
        SmallSet one = {};
        SmallSet& dst = (chunks <= 2 && merged.n < kSmallSetMax && merge_small) ? merged : one;
        const int k = dst.n++;
        dst.level[k] = l; dst.col[k] = col; dst.chunk_entries[k] = chunk_entries; dst.chunks[k] = chunks;
        dst.per_chunk[k] = (int)per_chunk; dst.run[k] = run;
        if (&dst == &one) { const int rc = launch_small(one); if (rc) return rc; }
    }
    if (merged.n > 0) { const int rc = launch_small(merged); if (rc) return rc; }
    if (plan.count > 0) {
        const dim3 nb((unsigned)ceil_div64(n_points * 8 * F, 256), (unsigned)plan.count);
        switch (F) {
            case 1: scatter_kernel<1><<<nb, 256, 0, st>>>(g, s, x01, dfeat, n_points, dtable, plan, stride); break;
            case 2: scatter_kernel<2><<<nb, 256, 0, st>>>(g, s, x01, dfeat, n_points, dtable, plan, stride); break;
            default: scatter_kernel<4><<<nb, 256, 0, st>>>(g, s, x01, dfeat, n_points, dtable, plan, stride); break;
        }
    }
    return nerf_launch_status("nerf_hashgrid_scatter");
}

int nerf_hashgrid_bwd(const NerfGridCfg* grid, const float* x01, const float* dout, int64_t n_points, float* dtable,
                      void* stream) {
    NERF_REQUIRE(grid && x01 && dout && dtable, "null pointer");
    NERF_REQUIRE(n_points >= 0 && grid->total_entries < (1ull << 32), "bad sizes");
    if (n_points == 0) return 0;
    return launch_scatter(grid, nullptr, x01, dout, n_points, dtable, level_range(0, grid->n_levels),
                          (int64_t)grid->n_levels * grid->n_features, 0, stream);
}

int nerf_sh_fwd(const float* d01, int64_t n_points, int degree, float* out, void* stream) {
    NERF_REQUIRE(d01 && out, "null pointer");
    NERF_REQUIRE(degree >= 1 && degree <= 4 && n_points >= 0, "degree must be 1..4");
    if (n_points == 0) return 0;
    sh_fwd_kernel<<<(unsigned)ceil_div64(n_points, 256), 256, 0, (hipStream_t)stream>>>(d01, n_points, degree * degree, out);
    return nerf_launch_status("nerf_sh_fwd");
}

int nerf_mlp_fwd(const float* params, const int32_t* dims, int n_layers, const float* x, int64_t n_rows, float* out,
                 void* stream) {
    NERF_REQUIRE(params && dims && x && out, "null pointer");
    NERF_REQUIRE(n_layers >= 1 && n_layers <= 8 && n_rows >= 0, "bad sizes");
    for (int i = 0; i <= n_layers; ++i) NERF_REQUIRE(dims[i] >= 1 && dims[i] <= kMlpMaxWidth, "layer width must be in [1,128]");
    if (n_rows == 0) return 0;
    Dims9 d = {};
    for (int i = 0; i <= n_layers; ++i) d.v[i] = dims[i];
    mlp_fwd_kernel<<<(unsigned)ceil_div64(n_rows, 64), 64, 0, (hipStream_t)stream>>>(params, d, n_layers, x, n_rows, out);
    return nerf_launch_status("nerf_mlp_fwd");
}

}  // extern "C"

extern "C" int nerf_hashgrid_scatter(const NerfGridCfg* grid, const NerfPointSrc* src, const float* dfeat, float* dtable,
                                     int level_begin, int level_count, void* stream) {
    NERF_REQUIRE(grid && dfeat && dtable, "null pointer");
    NERF_REQUIRE(grid->total_entries < (1ull << 32), "bad grid");
    int rc = nerf_check_src("nerf_hashgrid_scatter", src, false);
    if (rc) return rc;
    NERF_REQUIRE(level_begin >= 0 && level_count >= 0 && level_begin + level_count <= grid->n_levels, "bad level range");
    if (src->n_points == 0 || level_count == 0) return 0;
    return launch_scatter(grid, src, nullptr, dfeat, src->n_points, dtable, level_range(level_begin, level_count),
                          (int64_t)grid->n_levels * grid->n_features, level_begin * grid->n_features, stream);
}

// ---- level-sharded table: the two halves of the fused calls, cut at the feature matrix ----
template <bool HALF>
static int encode_levels(const NerfGridCfg* grid, const float* table, const NerfPointSrc* src,
                         const NerfLevelSet* levels, float* out, int64_t out_stride, void* stream) {
    NERF_REQUIRE(grid && table && out, "null pointer");
    NERF_REQUIRE(grid->total_entries < (1ull << 32), "bad grid");
    int rc = nerf_check_src("nerf_hashgrid_encode_levels", src, false);
    if (rc) return rc;
    rc = check_level_set("nerf_hashgrid_encode_levels", grid, levels);
    if (rc) return rc;
    NERF_REQUIRE(out_stride >= (int64_t)levels->count * grid->n_features, "out_stride smaller than count*F");
    if (src->n_points == 0 || levels->count == 0) return 0;
    const GridDev g = make_grid_dev(grid);
    const unsigned nb = (unsigned)ceil_div64(src->n_points * levels->count, 256);
    hipStream_t st = (hipStream_t)stream;
    switch (grid->n_features) {
        case 1: encode_levels_kernel<1, HALF><<<nb, 256, 0, st>>>(g, table, *src, *levels, out, out_stride); break;
        case 2: encode_levels_kernel<2, HALF><<<nb, 256, 0, st>>>(g, table, *src, *levels, out, out_stride); break;
        case 4: encode_levels_kernel<4, HALF><<<nb, 256, 0, st>>>(g, table, *src, *levels, out, out_stride); break;
        default: nerf_set_error("nerf_hashgrid_encode_levels: unsupported n_features"); return NERF_EUNSUPPORTED;
    }
    return nerf_launch_status("nerf_hashgrid_encode_levels");
}

extern "C" int nerf_hashgrid_encode_levels(const NerfGridCfg* grid, const float* table, const NerfPointSrc* src,
                                           const NerfLevelSet* levels, float* out, int64_t out_stride, void* stream) {
    return encode_levels<false>(grid, table, src, levels, out, out_stride, stream);
}

extern "C" int nerf_hashgrid_encode_levels_f16(const NerfGridCfg* grid, const void* table_f16, const NerfPointSrc* src,
                                               const NerfLevelSet* levels, float* out, int64_t out_stride, void* stream) {
    return encode_levels<true>(grid, (const float*)table_f16, src, levels, out, out_stride, stream);
}

extern "C" int nerf_hashgrid_scatter_levels(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels,
                                            const float* dfeat, int64_t dfeat_stride, float* dtable, void* stream) {
    NERF_REQUIRE(grid && dfeat && dtable, "null pointer");
    NERF_REQUIRE(grid->total_entries < (1ull << 32), "bad grid");
    int rc = nerf_check_src("nerf_hashgrid_scatter_levels", src, false);
    if (rc) return rc;
    rc = check_level_set("nerf_hashgrid_scatter_levels", grid, levels);
    if (rc) return rc;
    NERF_REQUIRE(dfeat_stride >= (int64_t)levels->count * grid->n_features, "dfeat_stride smaller than count*F");
    if (src->n_points == 0 || levels->count == 0) return 0;
    return launch_scatter(grid, src, nullptr, dfeat, src->n_points, dtable, *levels, dfeat_stride, 0, stream);
}

extern "C" int nerf_density_from_features(const NerfGridCfg* grid, const NerfNetCfg* net, const float* sigma_params,
                                          const float* x_hashtable, int64_t n_points, float* sigma, void* stream) {
    int rc = nerf_check_cfg("nerf_density_from_features", grid, net, false);
    if (rc) return rc;
    NERF_REQUIRE(sigma_params && x_hashtable && sigma && n_points >= 0, "null pointer / negative count");
    if (n_points == 0) return 0;
    NerfPointSrc src = {};
    src.n_points = n_points;
    src.outer_bound = 1.f;
    return launch_fwd<false, true>(grid, net, nullptr, sigma_params, nullptr, nullptr, 0, &src, sigma, nullptr, nullptr,
                                   nullptr, x_hashtable, stream);
}

extern "C" int nerf_field_from_features(const NerfGridCfg* grid, const NerfNetCfg* net, const float* sigma_params,
                                        const float* color_params, const float* latent_emb, int64_t n_images,
                                        const NerfPointSrc* src, const float* x_hashtable, float* sigma, float* rgb,
                                        float* xyz, void* stream) {
    int rc = nerf_check_cfg("nerf_field_from_features", grid, net, true);
    if (rc) return rc;
    rc = nerf_check_src("nerf_field_from_features", src, true);
    if (rc) return rc;
    NERF_REQUIRE(sigma_params && color_params && x_hashtable && sigma && rgb, "null pointer");
    NERF_REQUIRE(net->latent_dim == 0 || (latent_emb && n_images > 0), "latent embedding missing");
    if (src->n_points == 0) return 0;
    return launch_fwd<true, true>(grid, net, nullptr, sigma_params, color_params, latent_emb, n_images, src, sigma, rgb,
                                  nullptr, xyz, x_hashtable, stream);
}

namespace {
__global__ __launch_bounds__(256) void feature_shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int64_t n_rows, int n_cols, NerfColMap map, int direction) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_cols) return;
    const int64_t p = i / n_cols;
    const int c = (int)(i - p * n_cols);
    const int64_t flat = map.offset[c] + p * map.stride[c];
    if (direction == 0) dst[i] = src[flat];
    else dst[flat] = src[i];
}
}  // namespace

extern "C" int nerf_feature_shuffle(const float* src, float* dst, int64_t n_rows, int n_cols, const NerfColMap* map,
                                    int direction, void* stream) {
    NERF_REQUIRE(src && dst && map, "null pointer");
    NERF_REQUIRE(n_rows >= 0 && n_cols >= 1 && n_cols <= NERF_MAX_FEATURE_COLS, "bad sizes");
    NERF_REQUIRE(direction == 0 || direction == 1, "direction must be 0 or 1");
    if (n_rows == 0) return 0;
    feature_shuffle_kernel<<<(unsigned)ceil_div64(n_rows * n_cols, 256), 256, 0, (hipStream_t)stream>>>(src, dst, n_rows, n_cols,
                                                                                                     *map, direction);
    return nerf_launch_status("nerf_feature_shuffle");
}
