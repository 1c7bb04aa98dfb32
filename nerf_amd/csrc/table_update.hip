// Binned hash-table update for gfx950: gradient scatter and Adam in ONE pass over the table.
//
// The classic pair -- atomic scatter of d(features) into a dense gradient array (tcnn HashGrid backward), then a
// dense Adam pass (train.py:84-88) -- moves 32 B per table parameter through HBM (p, g, m, v read; p, m, v written;
// g cleared) and pays the memory-side atomic rate for every corner update.  Here the corner updates of a step are
// first BINNED by table partition (a counting sort: count, scan, fill -- no global atomics, a deterministic number of
// items), then every partition (16 K floats, cut at multiples of 64 KB from the table's base) is summed in LDS with
// ds_add_f32 by four workgroups, a quarter each, which stream p, m, v through the Adam update with the gradient read
// from LDS: 24 B per parameter, no gradient array, no memory-side atomics.  The update rule is exactly nerf_adam_step's (adam_common.h); only the summation order of
// the corner contributions differs (as it does between any two runs of the atomic scatter).
//
// Used for the levels that are large enough to fill many partitions (the hashed levels and the big dense ones); the
// small dense levels keep the LDS-privatised scatter + nerf_adam_step (a level smaller than a partition would put
// every item of the step into one bin).
#include <stdlib.h>

#include "adam_common.h"
#include "field_common.h"

int nerf_check_src(const char* fn, const NerfPointSrc* src, bool with_color);   // field_fwd.hip

namespace {

#ifndef NERF_PART_FLOATS
#define NERF_PART_FLOATS 16384
#endif
constexpr int kPartFloats = NERF_PART_FLOATS;   // floats of one table partition (64 KB of LDS in the update kernel)
#ifndef NERF_BIN_CHUNK
#define NERF_BIN_CHUNK 16384
#endif
constexpr int kBinChunkPts = NERF_BIN_CHUNK;   // points per workgroup of the count / fill passes
#ifndef NERF_BIN_THREADS
#define NERF_BIN_THREADS 1024
#endif
constexpr int kBinThreads = NERF_BIN_THREADS;      // threads of a count / scan / fill workgroup (512: two fill workgroups per CU, 0.63 vs 0.55 ms -- DESIGN.md "tried")
#ifndef NERF_MAX_BINS
#define NERF_MAX_BINS 8192
#endif
constexpr int kMaxBinsPerLevel = NERF_MAX_BINS;    // LDS histogram of the count / fill passes: 32 KB
static_assert(kPartFloats % 4 == 0 && kPartFloats <= 65536, "an item's partition-local offset is stored as uint16");
static_assert(kMaxBinsPerLevel <= 65536, "the fill pass packs (bin, offset) into one 32-bit word");

struct BinPlan {
    int count;
    int level[NERF_MAX_LEVELS];
    int col[NERF_MAX_LEVELS];                    // first d(features) column of the level
    uint32_t nbins[NERF_MAX_LEVELS];
    // Partitions are cut at multiples of kPartFloats of the ABSOLUTE float index in the table, not of the level's own:
    // shift = (first float of the level) % kPartFloats, a level-local float x lies in partition (x + shift) / kPartFloats
    // at offset (x + shift) % kPartFloats.  The p, m, v chunks of the update pass then start at multiples of 16 KB from
    // the tensors' bases whatever the level offsets are (the same stream runs 1.64 ms aligned, 1.75 ms at config_base's
    // offset of level 5, tools/stream3_bench.hip); a level's first and last partition may be short.
    uint32_t shift[NERF_MAX_LEVELS];
    uint32_t part_base[NERF_MAX_LEVELS + 1];     // running sum of nbins: workgroup id of the update kernel -> level slot
    int64_t hist_off[NERF_MAX_LEVELS];           // in uint32 units: the slot's [workgroup][bin] histogram / offsets
    int64_t item_off[NERF_MAX_LEVELS];           // in items: the slot's 8 * n_points items
};

// An item = one corner update = (float offset inside its partition: uint16, F gradient values), kept as two arrays.
// (One 16-byte record per item halves the scattered store requests of the fill pass but was not faster, 0.79 vs
// 0.84 ms for 63 M items, and the update kernel then reads 1.0 GB instead of 0.63 GB: 2.18 vs 2.07 ms.)
struct BinLayout {       // byte offsets into the caller's workspace
    int64_t hist, bin_start, item_idx, item_val, total;
    int nwg;
};

__device__ __forceinline__ LevelP plan_level(const GridDev& g, int l) {
    LevelP lp;
    lp.scale = g.scale[l];
    lp.res = g.res[l];
    lp.entries = g.entries[l];
    lp.hashed = g.hashed[l];
    lp.offset = 0;          // level-relative entry indices
    lp.active = true;
    return lp;
}

// pass 1: per workgroup (a chunk of points of one level) the number of items that fall into every partition
template <int F>
__global__ __launch_bounds__(kBinThreads) void bin_count_kernel(GridDev g, NerfPointSrc src, BinPlan plan, int64_t M,
                                                                int nwg, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[kMaxBinsPerLevel];
    const int slot = blockIdx.y, l = plan.level[slot];
    const uint32_t nb = plan.nbins[slot];
    for (uint32_t i = threadIdx.x; i < nb; i += kBinThreads) lh[i] = 0u;
    __syncthreads();
    const LevelP lp = plan_level(g, l);
    const uint32_t shift = plan.shift[slot];
    const int64_t p0 = (int64_t)blockIdx.x * kBinChunkPts;
    const int64_t p1 = p0 + kBinChunkPts < M ? p0 + kBinChunkPts : M;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kBinThreads) {
        const PointIn q = fetch_point(src, p, false);
        const Corners c = level_corners(lp, q.x, q.y, q.z);
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&lh[(c.idx[k] * (uint32_t)F + shift) / kPartFloats], 1u);
    }
    __syncthreads();
    uint32_t* dst = hist + plan.hist_off[slot] + (int64_t)blockIdx.x * nb;      // [workgroup][bin]: coalesced
    for (uint32_t i = threadIdx.x; i < nb; i += kBinThreads) dst[i] = lh[i];
}

// pass 2: turn a slot's [workgroup][bin] counts (in place) into where every workgroup writes its items of every bin:
// offset(bin, wg) = items of all earlier bins + items of this bin from earlier workgroups.  bin_start[b] = first item
// of bin b, bin_start[nbins] = 8 * n_points.  One workgroup per slot; thread t takes bins t, t + 1024, ... so that
// every pass over the workgroups is a coalesced row read.
__global__ __launch_bounds__(kBinThreads) void bin_scan_kernel(BinPlan plan, int nwg, uint32_t* __restrict__ hist,
                                                               uint32_t* __restrict__ bin_start) {
    __shared__ uint32_t part[kBinThreads];
    __shared__ uint32_t carry_s;
    const int slot = blockIdx.x;
    const uint32_t nb = plan.nbins[slot];
    uint32_t* h = hist + plan.hist_off[slot];
    uint32_t* bs = bin_start + plan.part_base[slot] + slot;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += kBinThreads) {
        const uint32_t b = b0 + threadIdx.x;
        uint32_t tot = 0;
        if (b < nb) {       // eight independent row reads in flight: the column walk is latency-bound otherwise
            int w = 0;
            for (; w + 8 <= nwg; w += 8) {
                uint32_t t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = h[(int64_t)(w + k) * nb + b];
#pragma unroll
                for (int k = 0; k < 8; ++k) tot += t[k];
            }
            for (; w < nwg; ++w) tot += h[(int64_t)w * nb + b];
        }
        part[threadIdx.x] = tot;
        __syncthreads();
        for (int off = 1; off < kBinThreads; off <<= 1) {      // Hillis-Steele inclusive scan of this round's bin totals
            const uint32_t add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        uint32_t run = carry + part[threadIdx.x] - tot;         // first item of bin b
        if (b < nb) {
            bs[b] = run;
            int w = 0;
            for (; w + 8 <= nwg; w += 8) {
                uint32_t t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = h[(int64_t)(w + k) * nb + b];
#pragma unroll
                for (int k = 0; k < 8; ++k) { h[(int64_t)(w + k) * nb + b] = run; run += t[k]; }
            }
            for (; w < nwg; ++w) {
                const uint32_t c = h[(int64_t)w * nb + b];
                h[(int64_t)w * nb + b] = run;
                run += c;
            }
        }
        __syncthreads();
        if (threadIdx.x == kBinThreads - 1) carry_s = carry + part[kBinThreads - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) bs[nb] = carry_s;
}

// pass 3: write every item (partition-local float offset, F gradient values) to its place in its bin
template <int F>
__global__ __launch_bounds__(kBinThreads) void bin_fill_kernel(GridDev g, NerfPointSrc src, BinPlan plan, int64_t M, int chunk0,
                                                               const uint32_t* __restrict__ offs,
                                                               const float* __restrict__ dfeat, int64_t stride,
                                                               uint16_t* __restrict__ item_idx, float* __restrict__ item_val) {
    __shared__ uint32_t cur[kMaxBinsPerLevel];
    const int slot = blockIdx.y, l = plan.level[slot];
    const uint32_t nb = plan.nbins[slot];
    const int64_t chunk = (int64_t)blockIdx.x + chunk0;         // (a launch may cover a range of the chunks only)
    const uint32_t* o = offs + plan.hist_off[slot] + chunk * nb;
    for (uint32_t i = threadIdx.x; i < nb; i += kBinThreads) cur[i] = o[i];
    __syncthreads();
    const LevelP lp = plan_level(g, l);
    const int col = plan.col[slot];
    uint16_t* ii = item_idx + plan.item_off[slot];
    float* iv = item_val + plan.item_off[slot] * F;
    const int64_t p0 = chunk * kBinChunkPts;
    const int64_t p1 = p0 + kBinChunkPts < M ? p0 + kBinChunkPts : M;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kBinThreads) {
        const PointIn q = fetch_point(src, p, false);
        const Corners c = level_corners(lp, q.x, q.y, q.z);
        float d[F];
        const float* dp = dfeat + p * stride + col;
        if (F == 2 && ((((uintptr_t)dp) & 7) == 0)) { const f32x2 t = *(const f32x2*)dp; d[0] = t.x; d[F - 1] = t.y; }
        else {
#pragma unroll
            for (int f = 0; f < F; ++f) d[f] = dp[f];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t fl = c.idx[k] * (uint32_t)F + plan.shift[slot];
            const uint32_t pos = atomicAdd(&cur[fl / kPartFloats], 1u);
            ii[pos] = (uint16_t)(fl % kPartFloats);
            if (F == 2) { f32x2 t; t.x = c.w[k] * d[0]; t.y = c.w[k] * d[F - 1]; *(f32x2*)(iv + (int64_t)pos * 2) = t; }
            else {
#pragma unroll
                for (int f = 0; f < F; ++f) iv[(int64_t)pos * F + f] = c.w[k] * d[f];
            }
        }
    }
}

// pass 3, sorted: the same placement, but every round of kSortPts points (8 * kSortPts items) is first sorted by bin
// in LDS (counting sort inside the workgroup) and then written out in bin order, so that the items a round adds to one
// bin leave as one contiguous run per wave instruction instead of as single 2- and 8-byte stores at random times
// (the plain kernel writes 2.0 GB for 0.63 GB of items -- partially filled lines are evicted and rewritten).
// Dynamic LDS: cur[nb] | cnt[nb] | start[nb] | wave sums | staged keys (u32: bin, offset), values.
constexpr int kSortPts = kBinThreads;       // one point per thread and round
#ifndef NERF_FILL_ABL
#define NERF_FILL_ABL 0      // timing ablations of the fill pass (tools/fill_ablation.sh): never in a shipped build
#endif

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t t = __shfl_up(v, off, kWave);
        if (lane >= off) v += t;
    }
    return v;
}

template <int F>
__global__ __launch_bounds__(kBinThreads) void bin_fill_sorted_kernel(GridDev g, NerfPointSrc src, BinPlan plan, int64_t M,
                                                                      int chunk0, uint32_t nb_max,
                                                                      const uint32_t* __restrict__ offs,
                                                                      const float* __restrict__ dfeat, int64_t stride,
                                                                      uint16_t* __restrict__ item_idx,
                                                                      float* __restrict__ item_val) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    constexpr int kItems = 8 * kSortPts;
    uint32_t* cur = lds_u;
    uint32_t* cnt = cur + nb_max;
    uint32_t* start = cnt + nb_max;
    uint32_t* wsum = start + nb_max;                      // kBinThreads / kWave wave totals
    // staged items: one packed word (bin << 16 | offset in the partition) and the F values; the scattered LDS stores of
    // this staging are what the pass is bound by (16 waves share one LDS pipe), so an item is two stores, not four: its
    // place in global memory is rebuilt at write-out time from the word (cur[bin] + position in the bin's run)
    uint32_t* st_key = wsum + kBinThreads / kWave;
    float* st_val = (float*)(st_key + kItems);
    const int slot = blockIdx.y, l = plan.level[slot];
    const uint32_t nb = plan.nbins[slot];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t chunk = (int64_t)blockIdx.x + chunk0;         // (a launch may cover a range of the chunks only)
    const uint32_t* o = offs + plan.hist_off[slot] + chunk * nb;
    for (uint32_t i = threadIdx.x; i < nb; i += kBinThreads) { cur[i] = o[i]; cnt[i] = 0u; }
    __syncthreads();
    // From here on the barriers order LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): nothing in global memory is
    // shared between the threads of this kernel, and a __syncthreads() would make every round wait until its scattered
    // item stores -- and the next round's point / gradient loads, requested a round ahead -- have completed.
    // (asm with a "memory" clobber: the two builtins are IntrNoMem for LLVM, which could then move LDS accesses across them)
    auto lds_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const LevelP lp = plan_level(g, l);
    const int col = plan.col[slot];
    uint16_t* ii = item_idx + plan.item_off[slot];
    float* iv = item_val + plan.item_off[slot] * F;
    const int64_t p0 = chunk * kBinChunkPts;
    const int64_t p1 = p0 + kBinChunkPts < M ? p0 + kBinChunkPts : M;
    const uint32_t per = (nb + kBinThreads - 1) / kBinThreads;     // consecutive bins per thread in the scan
    // this thread's point of the coming round: position and gradient are requested one round ahead
    float nx = 0.f, ny = 0.f, nz = 0.f, nd[F];
#pragma unroll
    for (int f = 0; f < F; ++f) nd[f] = 0.f;
    auto request = [&](int64_t pp) {
        if (pp < p1) {
            const PointIn q = fetch_point(src, pp, false);
            nx = q.x; ny = q.y; nz = q.z;
            const float* dp = dfeat + pp * stride + col;
#pragma unroll
            for (int f = 0; f < F; ++f) nd[f] = dp[f];
        }
    };
    request(p0 + threadIdx.x);
    for (int64_t base = p0; base < p1; base += kSortPts) {
        const int64_t p = base + threadIdx.x;
        const bool live = p < p1;
        uint32_t bin[8], rank[8], loc[8];
        float val[8][F];
        const float qx = nx, qy = ny, qz = nz;
        float d[F];
#pragma unroll
        for (int f = 0; f < F; ++f) d[f] = nd[f];
        request(p + kSortPts);
        if (live) {
            const Corners c = level_corners(lp, qx, qy, qz);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t fl = c.idx[k] * (uint32_t)F + plan.shift[slot];
                bin[k] = fl / kPartFloats;
                loc[k] = fl % kPartFloats;
                rank[k] = NERF_FILL_ABL >= 4 ? (uint32_t)k : atomicAdd(&cnt[bin[k]], 1u);
#pragma unroll
                for (int f = 0; f < F; ++f) val[k][f] = c.w[k] * d[f];
            }
        }
        lds_sync();
        if (NERF_FILL_ABL >= 3) continue;
        // exclusive scan of cnt[0..nb) -> start
        uint32_t mine = 0;
        for (uint32_t j = 0; j < per; ++j) { const uint32_t b = threadIdx.x * per + j; if (b < nb) mine += cnt[b]; }
        const uint32_t incl = wave_incl_scan(mine, lane);
        if (lane == kWave - 1) wsum[wave] = incl;
        lds_sync();
        uint32_t run = incl - mine;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t b = threadIdx.x * per + j;
            if (b < nb) { start[b] = run; run += cnt[b]; }
        }
        lds_sync();
        if (live && NERF_FILL_ABL < 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t s_ = start[bin[k]] + rank[k];
                st_key[s_] = (bin[k] << 16) | loc[k];
                if (F == 2) { f32x2 t; t.x = val[k][0]; t.y = val[k][F - 1]; *(f32x2*)(st_val + s_ * 2) = t; }
                else {
#pragma unroll
                    for (int f = 0; f < F; ++f) st_val[s_ * F + f] = val[k][f];
                }
            }
        }
        lds_sync();
        const int64_t n_live = NERF_FILL_ABL >= 1 ? 0 : (p1 - base < kSortPts ? p1 - base : kSortPts) * 8;
        for (int s_ = threadIdx.x; s_ < n_live; s_ += kBinThreads) {
            const uint32_t key = st_key[s_], b_ = key >> 16;
            const uint32_t dst = cur[b_] + ((uint32_t)s_ - start[b_]);
            ii[dst] = (uint16_t)(key & 0xFFFFu);
            if (F == 2) *(f32x2*)(iv + (int64_t)dst * 2) = *(const f32x2*)(st_val + s_ * 2);
            else {
#pragma unroll
                for (int f = 0; f < F; ++f) iv[(int64_t)dst * F + f] = st_val[s_ * F + f];
            }
        }
        lds_sync();          // the write-out above reads cur[] and start[]
        for (uint32_t i = threadIdx.x; i < nb; i += kBinThreads) { cur[i] += cnt[i]; cnt[i] = 0u; }
        lds_sync();
    }
}

// pass 4: FOUR 256-thread workgroups per partition, each owning a quarter of it (4096 floats, 16 KB of LDS): sum the
// quarter's share of the bin into LDS, then Adam over the quarter with the gradient from LDS.
// What the memory system rewards here is many small workgroups in flight: the bare p, m, v stream runs at 5.9 TB/s from
// 256-thread workgroups of 16 KB per array against 5.0 TB/s from 1024-thread ones holding a whole 64 KB partition
// (tools/stream3_bench.hip), and with 16 KB of LDS a CU holds seven workgroups, so the short LDS phase of one hides under
// the streams of the others.  Every quarter scans the whole bin (2- and 8-byte loads, mostly L2 hits after the first
// quarter's pass) and keeps the items that fall into its quarter.  Block b takes quarter b % 4 of partition b / 4:
// giving the four quarters of a partition to ONE XCD (block indices equal modulo 8), so that the bin is fetched from HBM
// once, measured SLOWER (2.00 vs 1.92 ms on one box): the stream wants consecutive blocks on consecutive addresses.
// (Measured on one box, update pass alone: quarters 1.86 ms; halves 1.96; eighths 1.89; quarters with every item's
// values requested together with its index, before the quarter test, 1.90.)
#ifndef NERF_UPD_VARIANT
#define NERF_UPD_VARIANT 0      // 1: timing variant of the update pass (tools/ab builds only)
#endif
constexpr int kSubFloats = 4096;
constexpr int kSubs = kPartFloats / kSubFloats;
constexpr int kSplitThreads = 256;

// HALF: the new parameters are also written, rounded to nearest even, into the half-precision gather image of the table
// (`ph`, same indexing as p; precision float16, nerf_hip.h): 26 B per parameter instead of 24 + a conversion pass of 6.
template <int F, bool HALF = false>
__global__ __launch_bounds__(kSplitThreads) void table_update_kernel(GridDev g, BinPlan plan,
                                                                     const uint32_t* __restrict__ bin_start,
                                                                     const uint16_t* __restrict__ item_idx,
                                                                     const float* __restrict__ item_val,
                                                                     int64_t n_items_level, float* __restrict__ p,
                                                                     float* __restrict__ m, float* __restrict__ v,
                                                                     AdamK k, const double* __restrict__ skip,
                                                                     uint16_t* __restrict__ ph = nullptr) {   // (_Float16 bits: rocprofv3 cannot demangle a _Float16* parameter)
    static_assert(kSubFloats % (4 * kSplitThreads) == 0, "whole quads per thread");
    __shared__ __attribute__((aligned(16))) float acc[kSubFloats];
    if (skip && *skip != 0.0) return;       // non-finite loss this step (nerf_losses sums[5]): the step is dropped
    constexpr int T = kSplitThreads, kQuads = kSubFloats / 4 / T;
    const uint32_t part = blockIdx.x / kSubs, sub = blockIdx.x % kSubs;
    // level slot of the partition: a count of independent compares (no chain of dependent kernel-argument loads)
    int slot = 0;
#pragma unroll
    for (int s_ = 1; s_ < NERF_MAX_LEVELS; ++s_) slot += (s_ < plan.count && part >= plan.part_base[s_]) ? 1 : 0;
    const uint32_t bin = part - plan.part_base[slot];
    const int l = plan.level[slot];
    const int64_t level_floats = (int64_t)g.entries[l] * F;
    // level-local float index of this quarter's first LDS word (negative in the short first partition of a level)
    const int64_t f0 = (int64_t)bin * kPartFloats - (int64_t)plan.shift[slot] + (int64_t)sub * kSubFloats;
    if (f0 >= level_floats || f0 + kSubFloats <= 0) return;
    const int q_lo = f0 < 0 ? (int)((-f0) >> 2) : 0;                       // quads [q_lo, q_hi) of the quarter exist
    const int q_hi = (int)((level_floats - f0 < kSubFloats ? level_floats - f0 : kSubFloats) >> 2);
    const int tid = threadIdx.x;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4* acc4 = (f4*)acc;
    // this thread's p, m, v: requested first, they arrive while the bin is summed
    const int64_t base4 = ((int64_t)g.offset[l] * F + f0) >> 2;            // (arithmetic shift: f0 may be negative)
    f4 pq[kQuads], mq[kQuads], vq[kQuads];
    auto request_pmv = [&] {
#pragma unroll
        for (int u = 0; u < kQuads; ++u) {
            const int i = tid + u * T;
            const int64_t a = base4 + ((i >= q_lo && i < q_hi) ? i : q_lo);
            pq[u] = __builtin_nontemporal_load((const f4*)p + a);
            mq[u] = __builtin_nontemporal_load((const f4*)m + a);
            vq[u] = __builtin_nontemporal_load((const f4*)v + a);
        }
    };
#if NERF_UPD_VARIANT == 0
    request_pmv();
#endif
#pragma unroll
    for (int u = 0; u < kQuads; ++u) acc4[tid + u * T] = zero;
    // lgkmcnt(0) only: the loads above stay in flight across the barrier ("memory": a compiler-level fence for the LDS
    // accesses on both sides, which the IntrNoMem builtins would not be)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const uint32_t* bs = bin_start + plan.part_base[slot] + slot;
    // (clamped: a workspace that was not filled for these levels / this point count must not turn into wild reads)
    const uint32_t n_items = (uint32_t)n_items_level;
    uint32_t b0 = 0, b1 = 0;            // no points: a plain Adam step with a zero gradient, the workspace is not read
    if (n_items) { b0 = bs[bin] < n_items ? bs[bin] : n_items; b1 = bs[bin + 1] < n_items ? bs[bin + 1] : n_items; }
    const int64_t i0 = plan.item_off[slot] + b0, i1 = plan.item_off[slot] + (b1 > b0 ? b1 : b0);
    const int lo = (int)sub * kSubFloats;
    auto add_item = [&](int64_t i, int idx) {
        const int li = idx - lo;
        if ((unsigned)li <= (unsigned)(kSubFloats - F)) {
            if (F == 2) {
                const f32x2 t = *(const f32x2*)(item_val + i * 2);
                atomicAdd(acc + li, t.x);
                atomicAdd(acc + li + 1, t.y);
            } else {
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(acc + li + f, item_val[i * F + f]);
            }
        }
    };
    // the bin's indices are scanned eight at a time (16-byte loads, two per thread in flight); the values of the few items
    // that fall into this workgroup's share of the partition are fetched afterwards
    const int64_t a0 = (i0 + 7) & ~(int64_t)7, a1 = i1 & ~(int64_t)7;
    if (a0 < a1) {
        for (int64_t i = i0 + tid; i < a0; i += T) add_item(i, item_idx[i]);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* iv4 = (const u32x4*)item_idx;
        int64_t gidx = (a0 >> 3) + tid;
        const int64_t gend = a1 >> 3;
        for (; gidx + T < gend; gidx += 2 * T) {
            const u32x4 va = iv4[gidx], vb = iv4[gidx + T];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                add_item(gidx * 8 + 2 * e, (int)(va[e] & 0xFFFFu));
                add_item(gidx * 8 + 2 * e + 1, (int)(va[e] >> 16));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                add_item((gidx + T) * 8 + 2 * e, (int)(vb[e] & 0xFFFFu));
                add_item((gidx + T) * 8 + 2 * e + 1, (int)(vb[e] >> 16));
            }
        }
        for (; gidx < gend; gidx += T) {
            const u32x4 va = iv4[gidx];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                add_item(gidx * 8 + 2 * e, (int)(va[e] & 0xFFFFu));
                add_item(gidx * 8 + 2 * e + 1, (int)(va[e] >> 16));
            }
        }
        for (int64_t i = a1 + tid; i < i1; i += T) add_item(i, item_idx[i]);
    } else {
        for (int64_t i = i0 + tid; i < i1; i += T) add_item(i, item_idx[i]);
    }
#if NERF_UPD_VARIANT == 1
    request_pmv();      // (timing variant: sum the bin first, then stream)
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        const int ii = tid + u * T;
        if (ii >= q_lo && ii < q_hi) {
            const f4 gq = acc4[ii];
            adam4(pq[u], gq, mq[u], vq[u], k);
            __builtin_nontemporal_store(pq[u], (f4*)p + base4 + ii);
            __builtin_nontemporal_store(mq[u], (f4*)m + base4 + ii);
            __builtin_nontemporal_store(vq[u], (f4*)v + base4 + ii);
            if (HALF) {
                typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                h16x4 o;
                o.x = (_Float16)pq[u].x; o.y = (_Float16)pq[u].y; o.z = (_Float16)pq[u].z; o.w = (_Float16)pq[u].w;
                ((h16x4*)ph)[base4 + ii] = o;
            }
        }
    }
}

int make_plan(const char* fn, const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points, const int32_t* cols,
              BinPlan* plan, BinLayout* lay) {
    if (!grid || !levels || levels->count < 1 || levels->count > grid->n_levels || levels->count > NERF_MAX_LEVELS) {
        nerf_set_error("%s: bad level set", fn); return NERF_EINVAL;
    }
    if (n_points < 0 || n_points >= (1ll << 28)) { nerf_set_error("%s: n_points must be in [0, 2^28)", fn); return NERF_EINVAL; }
    const int F = grid->n_features;
    if (F != 1 && F != 2 && F != 4) { nerf_set_error("%s: unsupported: n_features_per_level must be 1, 2 or 4", fn); return NERF_EUNSUPPORTED; }
    if (grid->total_entries >= (1ull << 32)) { nerf_set_error("%s: bad grid", fn); return NERF_EINVAL; }
    const int nwg = (int)ceil_div64(n_points, kBinChunkPts);
    *plan = BinPlan();
    plan->count = levels->count;
    int64_t hist = 0, items = 0;
    uint32_t parts = 0;
    for (int i = 0; i < levels->count; ++i) {
        const int l = levels->level[i];
        if (l < 0 || l >= grid->n_levels) { nerf_set_error("%s: level out of range", fn); return NERF_EINVAL; }
        for (int j = 0; j < i; ++j)
            if (levels->level[j] == l) { nerf_set_error("%s: level listed twice", fn); return NERF_EINVAL; }
        const int64_t floats = (int64_t)grid->entries[l] * F;
        const int64_t shift = ((int64_t)grid->offset[l] * F) % kPartFloats;
        const int64_t nb = ceil_div64(floats + shift, kPartFloats);
        if ((floats & 3) || (((int64_t)grid->offset[l] * F) & 3) || nb > kMaxBinsPerLevel ||
            (uint64_t)grid->entries[l] * F + kPartFloats >= (1ull << 32)) {
            nerf_set_error("%s: unsupported: level %d (needs a multiple of 4 floats at a 16-byte offset and at most %d "
                           "partitions of %d floats)", fn, l, kMaxBinsPerLevel, kPartFloats);
            return NERF_EUNSUPPORTED;
        }
        plan->level[i] = l;
        plan->col[i] = cols ? cols[i] : i * F;
        if (plan->col[i] < 0) { nerf_set_error("%s: negative gradient column", fn); return NERF_EINVAL; }
        plan->nbins[i] = (uint32_t)nb;
        plan->shift[i] = (uint32_t)shift;
        plan->part_base[i] = parts;
        plan->hist_off[i] = hist;
        plan->item_off[i] = items;
        parts += (uint32_t)nb;
        hist += nb * nwg;
        items += 8 * n_points;
    }
    plan->part_base[levels->count] = parts;
    auto up = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    lay->nwg = nwg;
    lay->hist = 0;
    lay->bin_start = up(hist * 4);
    lay->item_idx = lay->bin_start + up(((int64_t)parts + levels->count) * 4);
    lay->item_val = lay->item_idx + up(items * 2);
    lay->total = lay->item_val + up(items * F * 4);
    return 0;
}

}  // namespace

extern "C" int64_t nerf_table_bin_workspace_bytes(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points) {
    BinPlan plan;
    BinLayout lay;
    const int rc = make_plan("nerf_table_bin_workspace_bytes", grid, levels, n_points, nullptr, &plan, &lay);
    return rc ? (int64_t)(rc < 0 ? rc : -rc) : lay.total;
}

namespace {

// count + scan (which need only the points) and / or the fill (which needs the gradient rows) of nerf_table_bin_levels
int bin_passes(const char* fn, const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels, const float* dfeat,
               int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace, int64_t workspace_bytes, void* stream,
               bool do_count, bool do_fill, int64_t p_begin = 0, int64_t p_end = -1) {
    if (!grid || !workspace || (do_fill && !dfeat)) { nerf_set_error("%s: null pointer", fn); return NERF_EINVAL; }
    int rc = nerf_check_src(fn, src, false);
    if (rc) return rc;
    BinPlan plan;
    BinLayout lay;
    rc = make_plan(fn, grid, levels, src->n_points, dfeat_cols, &plan, &lay);
    if (rc) return rc;
    if (workspace_bytes < lay.total || (((uintptr_t)workspace) & 15) != 0) {
        nerf_set_error("%s: workspace too small or not 16-byte aligned", fn); return NERF_EINVAL;
    }
    if (src->n_points == 0) return 0;       // nothing to sort; nerf_table_update_binned(n_points = 0) reads no bins
    const int F = grid->n_features;
    if (do_fill)
        for (int i = 0; i < plan.count; ++i)
            if (dfeat_stride < plan.col[i] + F) { nerf_set_error("%s: dfeat_stride smaller than the last gradient column", fn); return NERF_EINVAL; }
    const GridDev g = make_grid_dev(grid);
    char* ws = (char*)workspace;
    uint32_t* hist = (uint32_t*)(ws + lay.hist);
    uint32_t* bin_start = (uint32_t*)(ws + lay.bin_start);
    uint16_t* item_idx = (uint16_t*)(ws + lay.item_idx);
    float* item_val = (float*)(ws + lay.item_val);
    hipStream_t st = (hipStream_t)stream;
    const dim3 nb((unsigned)lay.nwg, (unsigned)plan.count);
    const int64_t M = src->n_points;
    // the fill may cover a range of the points only (whole chunks): the rows of d(features) arrive group by group
    if (p_end < 0) p_end = M;
    if (p_begin < 0 || p_begin > p_end || p_end > M || p_begin % kBinChunkPts || (p_end % kBinChunkPts && p_end != M)) {
        nerf_set_error("%s: point range must be whole chunks of %d points", fn, kBinChunkPts); return NERF_EINVAL;
    }
    const int chunk0 = (int)(p_begin / kBinChunkPts);
    const dim3 nbf((unsigned)(ceil_div64(p_end, kBinChunkPts) - chunk0), (unsigned)plan.count);
    if (do_fill && nbf.x == 0) return 0;
    uint32_t nb_max = 0;
    for (int i = 0; i < plan.count; ++i) nb_max = plan.nbins[i] > nb_max ? plan.nbins[i] : nb_max;
    nb_max = (nb_max + 3u) & ~3u;            // keeps the staged arrays behind the counters 16-byte aligned
    // LDS of the sorted fill pass: 3 arrays of nb_max counters, 16 wave sums, 8 * kSortPts staged items
    const int sorted_lds = (int)((3 * (size_t)nb_max + kBinThreads / kWave) * 4 + (size_t)8 * kSortPts * (4 + 4 * F));
    const bool sorted = sorted_lds <= 150 * 1024 && !getenv("NERF_BIN_FILL_PLAIN");
#define BIN_PASSES(FF)                                                                                                 \
    do {                                                                                                               \
        if (do_count) {                                                                                                \
            bin_count_kernel<FF><<<nb, kBinThreads, 0, st>>>(g, *src, plan, M, lay.nwg, hist);                         \
            bin_scan_kernel<<<(unsigned)plan.count, kBinThreads, 0, st>>>(plan, lay.nwg, hist, bin_start);             \
        }                                                                                                              \
        if (do_fill && sorted) {                                                                                       \
            auto kern = bin_fill_sorted_kernel<FF>;                                                                    \
            static NerfLdsAttrCache lds_cache = {};   /* per instantiation */    \
            if (int e = nerf_ensure_dynamic_lds((const void*)kern, sorted_lds, &lds_cache, "table_bin_fill")) return e;    \
            kern<<<nbf, kBinThreads, sorted_lds, st>>>(g, *src, plan, M, chunk0, nb_max, hist, dfeat, dfeat_stride, item_idx, \
                                                       item_val);                                                      \
        } else if (do_fill) {                                                                                          \
            bin_fill_kernel<FF><<<nbf, kBinThreads, 0, st>>>(g, *src, plan, M, chunk0, hist, dfeat, dfeat_stride, item_idx, \
                                                             item_val);                                                \
        }                                                                                                              \
    } while (0)
    switch (F) {
        case 1: BIN_PASSES(1); break;
        case 2: BIN_PASSES(2); break;
        default: BIN_PASSES(4); break;
    }
#undef BIN_PASSES
    return nerf_launch_status(fn);
}

}  // namespace

extern "C" int nerf_table_bin_levels(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels,
                                     const float* dfeat, int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    return bin_passes("nerf_table_bin_levels", grid, src, levels, dfeat, dfeat_stride, dfeat_cols, workspace, workspace_bytes,
                      stream, true, true);
}

extern "C" int nerf_table_bin_count(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    return bin_passes("nerf_table_bin_count", grid, src, levels, nullptr, 0, nullptr, workspace, workspace_bytes, stream, true,
                      false);
}

extern "C" int nerf_table_bin_fill(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels,
                                   const float* dfeat, int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    return bin_passes("nerf_table_bin_fill", grid, src, levels, dfeat, dfeat_stride, dfeat_cols, workspace, workspace_bytes,
                      stream, false, true);
}

extern "C" int nerf_table_bin_fill_range(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels,
                                         const float* dfeat, int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace,
                                         int64_t workspace_bytes, int64_t p_begin, int64_t p_end, void* stream) {
    return bin_passes("nerf_table_bin_fill_range", grid, src, levels, dfeat, dfeat_stride, dfeat_cols, workspace,
                      workspace_bytes, stream, false, true, p_begin, p_end);
}

static int table_update_binned(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points,
                               const void* workspace, int64_t workspace_bytes, float* table, float* exp_avg,
                               float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float grad_scale, const double* skip_flag,
                               _Float16* table_f16, void* stream) {
    NERF_REQUIRE(grid && workspace && table && exp_avg && exp_avg_sq, "null pointer");
    NERF_REQUIRE((((uintptr_t)table_f16) & 7) == 0, "the half-precision image must be 8-byte aligned");
    NERF_REQUIRE(step >= 1, "bad step");
    NERF_REQUIRE((((uintptr_t)table | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)workspace) & 15) == 0,
                 "buffers must be 16-byte aligned");
    BinPlan plan;
    BinLayout lay;
    const int rc = make_plan("nerf_table_update_binned", grid, levels, n_points, nullptr, &plan, &lay);
    if (rc) return rc;
    NERF_REQUIRE(workspace_bytes >= lay.total, "workspace too small");
    const GridDev g = make_grid_dev(grid);
    const char* ws = (const char*)workspace;
    const uint32_t* bin_start = (const uint32_t*)(ws + lay.bin_start);
    const uint16_t* item_idx = (const uint16_t*)(ws + lay.item_idx);
    const float* item_val = (const float*)(ws + lay.item_val);
    const AdamK k = make_adam_k(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = plan.part_base[plan.count];
#define UPDATE(FF)                                                                                                     \
    do {                                                                                                               \
        if (table_f16)                                                                                                 \
            table_update_kernel<FF, true><<<nb * kSubs, kSplitThreads, 0, st>>>(g, plan, bin_start, item_idx, item_val, \
                                                                                8 * n_points, table, exp_avg, exp_avg_sq, k, \
                                                                                skip_flag, (uint16_t*)table_f16);       \
        else                                                                                                           \
            table_update_kernel<FF><<<nb * kSubs, kSplitThreads, 0, st>>>(g, plan, bin_start, item_idx, item_val,      \
                                                                          8 * n_points, table, exp_avg, exp_avg_sq, k, \
                                                                          skip_flag);                                  \
    } while (0)
    switch (grid->n_features) {
        case 1: UPDATE(1); break;
        case 2: UPDATE(2); break;
        default: UPDATE(4); break;
    }
#undef UPDATE
    return nerf_launch_status("nerf_table_update_binned");
}

extern "C" int nerf_table_update_binned_guarded(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points,
                                                const void* workspace, int64_t workspace_bytes, float* table, float* exp_avg,
                                                float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                                float weight_decay, int64_t step, float grad_scale, const double* skip_flag,
                                                void* stream) {
    return table_update_binned(grid, levels, n_points, workspace, workspace_bytes, table, exp_avg, exp_avg_sq, lr, beta1,
                               beta2, eps, weight_decay, step, grad_scale, skip_flag, nullptr, stream);
}

extern "C" int nerf_table_update_binned_f16(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points,
                                            const void* workspace, int64_t workspace_bytes, float* table, float* exp_avg,
                                            float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                            float weight_decay, int64_t step, float grad_scale, const double* skip_flag,
                                            void* table_f16, void* stream) {
    NERF_REQUIRE(table_f16, "null pointer");
    return table_update_binned(grid, levels, n_points, workspace, workspace_bytes, table, exp_avg, exp_avg_sq, lr, beta1,
                               beta2, eps, weight_decay, step, grad_scale, skip_flag, (_Float16*)table_f16, stream);
}

extern "C" int nerf_table_update_binned(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points,
                                        const void* workspace, int64_t workspace_bytes, float* table, float* exp_avg,
                                        float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        int64_t step, float grad_scale, void* stream) {
    return nerf_table_update_binned_guarded(grid, levels, n_points, workspace, workspace_bytes, table, exp_avg, exp_avg_sq, lr,
                                            beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}
