// Fused field-network backward for gfx950.  One 256-thread workgroup walks 128-point block
// tiles; each of its 4 wavefronts re-runs the forward for its own 32 points from the saved
// hash-grid features (x_hashtable), back-propagates through both MLPs with the same
// register-resident MFMA chains as the forward (A operand = W^T read from the same LDS image),
// writes d(loss)/d(features) out for the table-scatter launch that follows and reduces the
// latent gradient across the wavefront.  Weight gradients dW = dZ * H^T are summed over the 128 points
// on the matrix cores too: each (dZ, H) pair is staged once through LDS as [unit][point], and
// every wavefront owns a fixed subset of the 18 32x32 dW tiles in accumulators that persist
// for the whole launch (one atomic flush per workgroup at the end).
// Replaces autograd + tcnn backward under trainer.py:134.
#include "field_common.h"

int nerf_check_cfg(const char* fn, const NerfGridCfg* grid, const NerfNetCfg* net, bool with_color);
int nerf_check_src(const char* fn, const NerfPointSrc* src, bool with_color);

namespace {

constexpr int kBlockPts = 128;
#ifndef NERF_DW_BF16
#define NERF_DW_BF16 1
#endif
#ifndef NERF_DW_G
#define NERF_DW_G 1          // 16-byte operand pairs read ahead of the MFMAs of a dW tile
#endif
#ifndef NERF_BWD_XF_PREFETCH
#define NERF_BWD_XF_PREFETCH (NERF_DW_BF16 ? 0 : 1)     // 1: the feature half row a whole tile ahead (20 registers the packed staging cannot spare); 2: from the end of the previous tile; 0: at the tile's top
#endif
#if NERF_DW_BF16
// The weight-gradient products dW = dZ * H^T run on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the f32-input rate)
// with both operands split in two: every staged value v is kept as the pair (hi, lo) of bf16 halves, hi = v truncated to
// bf16, lo = (v - hi) rounded to bf16 (v - hi is exact in fp32), |v - hi - lo| <= 2^-16 |v|.  A staged word holds (hi, lo) of
// one point, so the 8 k-slots of a lane are 4 points x (hi, lo); one MFMA on the words as staged sums hi*hi + lo*lo over
// 8 points, a second one with A's halves swapped sums lo*hi + hi*lo: all four terms, fp32 accumulation, 2 x 32 cycles per
// 8 points against 4 x 64 for the f32-input MFMA.  The MLP chains themselves stay exact fp32; a weight
// gradient is a sum over 500 k points and tolerates 2^-16 per term (gradient bar 1e-3).
constexpr int kSP = kBlockPts + 4;  // staging pitch in words: rows stay 16-byte aligned, 16 rows x 4 words cover the 64 banks
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float split_pack(float v) {
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xffff0000u);     // exact in bf16
    const f32x2v pair = {hi, v - hi};
    return __builtin_bit_cast(float, __builtin_convertvector(pair, bf16x2v));      // v_cvt_pk_bf16_f32: low half hi, high half rne(lo)
}
#define STG(v) split_pack(v)

// acc += SA[a_tile rows][:] * SB[b_tile rows][:]^T over the 128 staged points
__device__ __forceinline__ f32x16 dw_tile(f32x16 acc, const float* SA, const float* SB, int a_tile, int b_tile, int n_out,
                                          int n_in, int l31, int h) {
    // rows past n_out / n_in are clamped to row 0: they land in accumulator entries flush_tile never writes
    const int ar = a_tile * 32 + l31, br = b_tile * 32 + l31;
    const u32x4* pa = (const u32x4*)(SA + (ar < n_out ? ar : 0) * kSP + 4 * h);     // lane half h: points 8 s + 4 h ...
    const u32x4* pb = (const u32x4*)(SB + (br < n_in ? br : 0) * kSP + 4 * h);
    constexpr int NS = kBlockPts / 8, G = NERF_DW_G, NG = NS / G;
    u32x4 abuf[2][G], bbuf[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) { abuf[0][i] = pa[2 * i]; bbuf[0][i] = pb[2 * i]; }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) {
#pragma unroll
            for (int i = 0; i < G; ++i) { abuf[(gi + 1) & 1][i] = pa[2 * ((gi + 1) * G + i)]; bbuf[(gi + 1) & 1][i] = pb[2 * ((gi + 1) * G + i)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const u32x4 a = abuf[gi & 1][i], b = bbuf[gi & 1][i];
            u32x4 asw;
#pragma unroll
            for (int e = 0; e < 4; ++e) asw[e] = __builtin_amdgcn_alignbit(a[e], a[e], 16);      // swap the two halves
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, asw), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
#else
constexpr int kSP = kBlockPts + 1;  // staging pitch (odd -> conflict-free row walks)
#define STG(v) (v)

// acc += SA[a_tile rows][:] * SB[b_tile rows][:]^T over the 128 staged points
__device__ __forceinline__ f32x16 dw_tile(f32x16 acc, const float* SA, const float* SB, int a_tile, int b_tile, int n_out,
                                          int n_in, int l31, int h) {
    // rows past n_out / n_in are clamped to row 0: they land in accumulator entries flush_tile never writes
    const int ar = a_tile * 32 + l31, br = b_tile * 32 + l31;
    const float* pa = SA + (ar < n_out ? ar : 0) * kSP + h;
    const float* pb = SB + (br < n_in ? br : 0) * kSP + h;
    constexpr int G = 4, NG = kBlockPts / 2 / G;
    float abuf[2][G], bbuf[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) { abuf[0][i] = pa[2 * i]; bbuf[0][i] = pb[2 * i]; }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) {
#pragma unroll
            for (int i = 0; i < G; ++i) { abuf[(gi + 1) & 1][i] = pa[2 * ((gi + 1) * G + i)]; bbuf[(gi + 1) & 1][i] = pb[2 * ((gi + 1) * G + i)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) acc = MFMA32(abuf[gi & 1][i], bbuf[gi & 1][i], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
#endif

// The barriers of the tile loop order LDS traffic only (the staging buffers SA / SB).  __syncthreads() also waits for vmcnt(0),
// i.e. for every global load in flight (the next tile's inputs, requested a tile ahead; the latent / feature rows re-read for
// staging); an LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier, -DNERF_BWD_LDS_BARRIER=1) measured the same, 0.503 vs
// 0.503 ms on one box: those loads have landed by the time a barrier is reached.  The plain form stays.
#ifndef NERF_BWD_LDS_BARRIER
#define NERF_BWD_LDS_BARRIER 0
#endif
__device__ __forceinline__ void lds_barrier() {
#if NERF_BWD_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// stage a 64-row P-layout activation pair as [unit][point]
__device__ __forceinline__ void stage64(float* S, int col, int h, const f32x16& v0, const f32x16& v1) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        S[(krow(t) + 4 * h) * kSP + col] = STG(v0[t]);
        S[(32 + krow(t) + 4 * h) * kSP + col] = STG(v1[t]);
#if NERF_DW_BF16
        if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // keep the packing temporaries of a few elements alive, not of all 32
#endif
    }
}

__device__ __forceinline__ void flush_tile(float* __restrict__ dw, const f32x16& acc, int a_tile, int b_tile, int n_out,
                                           int n_in, int l31, int h) {
    const int col = b_tile * 32 + l31;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int row = a_tile * 32 + krow(t) + 4 * h;
        if (row < n_out && col < n_in) atomicAdd(dw + row * n_in + col, acc[t]);
    }
}

// Sum over the 32 lanes of each wave half, result in every lane.  Four rotate-and-add steps inside the 16-lane DPP rows
// (plain VALU, no LDS crossbar) and one cross-row exchange; the __shfl_xor ladder this replaces cost 5 ds_bpermute
// round trips per value (9.5 k cycles per tile for the 24 latent gradients).
__device__ __forceinline__ float half_sum32(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v + __shfl_xor(v, 16, kWave);
}

// this lane's half row of hash-grid features: 16-byte loads when the layout allows (`vec`), else guarded scalars
template <int XF>
__device__ __forceinline__ void load_features(float (&xf)[XF], const float* __restrict__ xrow, int LHF, int LF, int h, bool vec) {
    if (vec) {
        const f32x4* x4 = (const f32x4*)xrow;
#pragma unroll
        for (int qd = 0; qd < XF / 4; ++qd) {
            const f32x4 v = x4[qd];
#pragma unroll
            for (int e = 0; e < 4; ++e) xf[4 * qd + e] = v[e];
        }
    } else {
#pragma unroll
        for (int t = 0; t < XF; ++t) xf[t] = (t < LHF && h * LHF + t < LF) ? xrow[t] : 0.f;
    }
}

__device__ __forceinline__ f32x16 relu_mask(const f32x16& d, const f32x16& act) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = act[i] > 0.f ? d[i] : 0.f;
    return r;
}

// REF: the reference's colour-net input (SH degree 4 + latent 48): its layer-1 input map is a compile-time function and
// the colour net's shape numbers are constants, which takes ~200 uniform values and shape tests out of the scalar registers
// (262 -> 67 SGPRs spilled into VGPR lanes; 0.520 -> 0.504 ms).  Making the grid's shape constant as well (exactly 2 * XF
// features: no SGPR spills at all) measured SLOWER, 0.558 ms: the allocator then spills 54 vector registers instead of 27.
template <int F, int XF, bool REF>
__global__ __launch_bounds__(256, 1) void field_bwd_kernel(GridDev g, NetLds s_in, const float* __restrict__ sigma_w,
                                                           const float* __restrict__ color_w,
                                                           const float* __restrict__ latent, NerfPointSrc src,
                                                           const float* __restrict__ xh, const float* __restrict__ dsigma,
                                                           const float* __restrict__ drgb, float* __restrict__ dfeat,
                                                           float* __restrict__ dsw, float* __restrict__ dcw,
                                                           float* __restrict__ dlatent, int sb_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    NetLds s = s_in;
    if (REF) { s.sh_dim = 16; s.sh_half = 8; s.lat = 48; s.lat_half = 24; s.in_c = 79; }
    lds_load_net(lds, s, sigma_w, color_w, true);
    float* SA = lds + s.total;
    float* SB = SA + kHid * kSP;

    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = wave * 32 + l31;
    const int64_t M = src.n_points;
    const int64_t n_btiles = (M + kBlockPts - 1) / kBlockPts;
    const int LF = g.L * F, LHF = g.LH * F;
    const int in_c_tiles = (s.in_c + 31) / 32;
    // every lane's half row of features / feature gradients is XF contiguous, 16-byte aligned floats (config_base: 20)
    const bool vec = (XF % 4 == 0) && LHF == XF && LF == 2 * XF;

    // persistent dW accumulators (roles depend on the wave, see the file comment)
    f32x16 acc_s1 = zero16(), acc_c2 = zero16(), acc_c1 = zero16(), acc_x = zero16(), acc_y = zero16();
    //   acc_s1: sigma layer 1 tile (wave>>1, wave&1)      acc_c2: colour layer 2 tile (wave>>1, wave&1)
    //   acc_c1: colour layer 1 tile id = wave             acc_x : waves 0,1: colour layer 1 tile id 4+wave
    //   acc_y : waves 0,1: sigma layer 2 tile (0, wave)            waves 2,3: colour layer 3 tile (0, wave-2)

    // A tile's inputs -- the point, its two upstream gradients, this lane's LH*F hash-grid features and its half of the
    // latent row -- are requested one tile AHEAD (right after the forward recompute of the previous tile has consumed its
    // own): the two dependent global round trips (ray -> image index -> latent row) cost ~7 % of a tile when they were
    // exposed at its top.  (Features and latent are fetched again, from L2, right before the two places that stage them
    // for the dW products: holding them across the whole tile costs scratch spills instead.)
    struct TileIn {
        bool valid, n_ok;
        int64_t p;
        PointIn q;
        float dsig, dr0, dr1, dr2;
        const float *xrow, *lrow;
        float xf[XF];
#if !NERF_DW_BF16
        float lat[24];
#endif
    };
    auto request = [&](int64_t bt, TileIn& t) {
        const int64_t p_raw = bt * kBlockPts + col;
        t.valid = p_raw < M;
        t.p = t.valid ? p_raw : M - 1;
        t.q = fetch_point(src, t.p, true);
        t.dsig = dsigma[t.p];
        t.dr0 = drgb[t.p * 3]; t.dr1 = drgb[t.p * 3 + 1]; t.dr2 = drgb[t.p * 3 + 2];
        t.xrow = xh + t.p * LF + h * LHF;
        // (an image index outside the latent table: zero latent row, no latent gradient -- as in the forward)
        t.n_ok = (uint64_t)t.q.n < (uint64_t)s.n_img;
        t.lrow = latent + (t.n_ok ? t.q.n : 0) * s.lat + h * s.lat_half;
#if NERF_BWD_XF_PREFETCH == 1
        load_features<XF>(t.xf, t.xrow, LHF, LF, h, vec);
#endif
#if !NERF_DW_BF16
        load_latent_half(t.lat, t.lrow, s.lat_half);
#endif
    };
    TileIn nxt;
    if ((int64_t)blockIdx.x < n_btiles) {
        request(blockIdx.x, nxt);
#if NERF_BWD_XF_PREFETCH == 2
        load_features<XF>(nxt.xf, nxt.xrow, LHF, LF, h, vec);
#endif
    }
    for (int64_t bt = blockIdx.x; bt < n_btiles; bt += gridDim.x) {
        const bool valid = nxt.valid, n_ok = nxt.n_ok;
        const int64_t p = nxt.p;
        const PointIn q = nxt.q;
        const float dsig = valid ? nxt.dsig : 0.f;
        const float dr0 = valid ? nxt.dr0 : 0.f, dr1 = valid ? nxt.dr1 : 0.f, dr2 = valid ? nxt.dr2 : 0.f;
        const float* xrow = nxt.xrow;
        const float* lrow = nxt.lrow;
        float xf[XF], lat[24];
#if NERF_BWD_XF_PREFETCH
#pragma unroll
        for (int t = 0; t < XF; ++t) xf[t] = nxt.xf[t];
#else
        load_features<XF>(xf, xrow, LHF, LF, h, vec);
#endif
#if NERF_DW_BF16
        // (the latent half row is requested here, not a tile ahead: 24 registers less across the whole tile, which the
        // packed staging needs; it is consumed only after the two sigma layers)
        load_latent_half(lat, lrow, s.lat_half);
#pragma unroll
        for (int t = 0; t < 24; ++t) lat[t] = n_ok ? lat[t] : 0.f;
#else
#pragma unroll
        for (int t = 0; t < 24; ++t) lat[t] = n_ok ? nxt.lat[t] : 0.f;
#endif
        // ------------------------- forward recompute -------------------------
        f32x16 a0, a1;
        {
            const float* w1r0 = lds + kWS1 + l31 * kPS1 + h * LHF;
            const float* w1r1 = w1r0 + 32 * kPS1;
            f32x16 acc[2] = {zero16(), zero16()};
            // steps past LH*F (generic XF = 32 build) multiply by xf = 0
            // (steps past LH*F of the generic XF = 32 build read the other half's weights or the zero pad: times xf = 0)
            mfma_chain<XF, 2>(acc, [&](int t, int n) { return (n ? w1r1 : w1r0)[t]; }, [&](int t) { return xf[t % XF]; });
            a0 = acc[0];
            a1 = acc[1];
        }
        a0 = relu16(a0);
        a1 = relu16(a1);
        const f32x16 o2 = layer_64xN<kPS2>(lds + kWS2, kGeoRows, l31, h, a0, a1);
        f32x16 c0, c1;
        {
            float sh[16];
            sh16(q.dx, q.dy, q.dz, sh);
            color_layer1<REF>(lds, s, l31, h, o2, sh, lat, c0, c1);
        }
        c0 = relu16(c0);
        c1 = relu16(c1);
        f32x16 e0, e1;
        layer_64x64<kPC2>(lds + kWC2, l31, h, c0, c1, e0, e1);
        e0 = relu16(e0);
        e1 = relu16(e1);
        const f32x16 o3 = layer_64xN<kPC3>(lds + kWC3, 3, l31, h, e0, e1);
        if (bt + gridDim.x < n_btiles) request(bt + gridDim.x, nxt);      // lands under the backward chains below

        // ------------------------- colour layer 3 -------------------------
        f32x16 d3 = zero16();
        if (h == 0) {
            const float r0 = 1.f / (1.f + expf(-o3[0])), r1 = 1.f / (1.f + expf(-o3[1])), r2 = 1.f / (1.f + expf(-o3[2]));
            d3[0] = dr0 * r0 * (1.f - r0);
            d3[1] = dr1 * r1 * (1.f - r1);
            d3[2] = dr2 * r2 * (1.f - r2);
        }
        lds_barrier();  // previous iteration's readers of SA/SB are done
        if (h == 0) {
            SA[0 * kSP + col] = STG(d3[0]);
            SA[1 * kSP + col] = STG(d3[1]);
            SA[2 * kSP + col] = STG(d3[2]);
        }
        stage64(SB, col, h, e0, e1);
        lds_barrier();
        if (wave >= 2) acc_y = dw_tile(acc_y, SA, SB, 0, wave - 2, 3, kHid, l31, h);
        f32x16 g0, g1;
        layer_T_Nx64<3, kPC3>(lds + kWC3, 3, l31, h, d3, g0, g1);
        g0 = relu_mask(g0, e0);
        g1 = relu_mask(g1, e1);
        // ------------------------- colour layer 2 -------------------------
        lds_barrier();
        stage64(SA, col, h, g0, g1);
        stage64(SB, col, h, c0, c1);
        lds_barrier();
        float lat2[24];
        load_latent_half(lat2, lrow, s.lat_half);      // lands under the dW MFMAs below
        if (!n_ok) {
#pragma unroll
            for (int t = 0; t < 24; ++t) lat2[t] = 0.f;
        }
        acc_c2 = dw_tile(acc_c2, SA, SB, wave >> 1, wave & 1, kHid, kHid, l31, h);
        {
            f32x16 t0, t1;
            layer_T_64x64<kPC2>(lds + kWC2, l31, h, g0, g1, t0, t1);
            g0 = relu_mask(t0, c0);
            g1 = relu_mask(t1, c1);
        }
        // ------------------------- colour layer 1 -------------------------
        lds_barrier();
        stage64(SA, col, h, g0, g1);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = krow(t) + 4 * h;
            if (row >= 1) SB[(row - 1) * kSP + col] = STG(o2[t]);
        }
        {
            float sh2[16];
            sh16(q.dx, q.dy, q.dz, sh2);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t < s.sh_half) {
                    float b = h ? sh2[8 + t] : sh2[t];
                    if (s.sh_half == 2) b = h ? sh2[2 + t] : sh2[t];
                    SB[(15 + h * s.sh_half + t) * kSP + col] = STG(b);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 24; ++t)
            if (t < s.lat_half) SB[(15 + s.sh_dim + h * s.lat_half + t) * kSP + col] = STG(lat2[t]);
        lds_barrier();
        acc_c1 = dw_tile(acc_c1, SA, SB, wave / in_c_tiles, wave % in_c_tiles, kHid, s.in_c, l31, h);
        if (wave < 2 && 4 + wave < 2 * in_c_tiles)
            acc_x = dw_tile(acc_x, SA, SB, (4 + wave) / in_c_tiles, (4 + wave) % in_c_tiles, kHid, s.in_c, l31, h);
        // d(colour input): tile G = [geo rows 0..15 | latent 0..15], tile L = latent 16..47
        f32x16 dG = zero16(), dL = zero16();
        {
            const int lat0 = 15 + s.sh_dim;
            const int inG = l31 < 16 ? l31 - 1 : lat0 + (l31 - 16);
            const bool liveG = l31 < 16 ? (l31 >= 1) : (l31 - 16 < s.lat);
            const int inL = lat0 + 16 + l31;
            const bool liveL = 16 + l31 < s.lat;
            const float* wG = lds + kWC1 + (liveG ? inG : 0);
            const float* wL = lds + kWC1 + (liveL ? inL : 0);
            f32x16 acc[2] = {zero16(), zero16()};
            mfma_chain<32, 2>(acc,
                              [&](int st, int n) {
                                  const int kk = (krow64(st) + 4 * h) * kPC1;
                                  return n ? wL[kk] : wG[kk];      // dead rows: never read back
                              },
                              [&](int st) { return st < 16 ? g0[st & 15] : g1[st & 15]; });
            dG = acc[0];
            dL = acc[1];
        }
        // latent gradient: rows 16..31 of tile G are latent 0..15, tile L holds latent 16..47
        if (s.lat > 0) {
            const int64_t n0 = __shfl(q.n, 0, kWave);
            const bool uniform = __all(q.n == n0);
            if (uniform) {
                if ((uint64_t)n0 < (uint64_t)s.n_img) {
#pragma unroll
                for (int t = 8; t < 16; ++t) {
                    const float v = half_sum32(dG[t]);
                    const int li = krow(t) + 4 * h - 16;
                    if (l31 == 0 && li < s.lat) atomicAdd(dlatent + n0 * s.lat + li, v);
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float v = half_sum32(dL[t]);
                    const int li = 16 + krow(t) + 4 * h;
                    if (l31 == 0 && li < s.lat) atomicAdd(dlatent + n0 * s.lat + li, v);
                }
                }
            } else if (valid && n_ok) {
#pragma unroll
                for (int t = 8; t < 16; ++t) {
                    const int li = krow(t) + 4 * h - 16;
                    if (li < s.lat) atomicAdd(dlatent + q.n * s.lat + li, dG[t]);
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int li = 16 + krow(t) + 4 * h;
                    if (li < s.lat) atomicAdd(dlatent + q.n * s.lat + li, dL[t]);
                }
            }
        }
        // ------------------------- sigma layer 2 -------------------------
        f32x16 d2 = zero16();
#pragma unroll
        for (int t = 0; t < 8; ++t) d2[t] = dG[t];
        if (h == 0) d2[0] = o2[0] > 0.f ? dsig : 0.f;
        lds_barrier();
#pragma unroll
        for (int t = 0; t < 8; ++t) SA[(krow(t) + 4 * h) * kSP + col] = STG(d2[t]);
        stage64(SB, col, h, a0, a1);
        lds_barrier();
        float xf2[XF];
        load_features<XF>(xf2, xrow, LHF, LF, h, vec);
        if (wave < 2) acc_y = dw_tile(acc_y, SA, SB, 0, wave, kGeoRows, kHid, l31, h);
        layer_T_Nx64<8, kPS2>(lds + kWS2, kGeoRows, l31, h, d2, g0, g1);
        g0 = relu_mask(g0, a0);
        g1 = relu_mask(g1, a1);
        // ------------------------- sigma layer 1 -------------------------
        lds_barrier();
        stage64(SA, col, h, g0, g1);
#pragma unroll
        for (int t = 0; t < XF; ++t)
            if (t < LHF) SB[(h * LHF + t) * kSP + col] = STG(xf2[t]);
        lds_barrier();
        acc_s1 = dw_tile(acc_s1, SA, SB, wave >> 1, wave & 1, kHid, s.in_sp, l31, h);
#if NERF_BWD_XF_PREFETCH == 2
        // the next tile's feature half row is requested HERE, late in the tile (its registers are free only now): it lands
        // under the last chain and the stores below instead of at the top of the next tile
        if (bt + gridDim.x < n_btiles) load_features<XF>(nxt.xf, nxt.xrow, LHF, LF, h, vec);
#endif
        // d(features): row map chosen so that each lane receives the gradients of its OWN levels:
        // register t of tile T <-> internal feature h*LHF + 16*T + t
        f32x16 x0 = zero16(), x1 = zero16();
        {
            const int hh = (l31 >> 2) & 1, qq = (l31 & 3) + 4 * (l31 >> 3);
            const bool live0 = qq < LHF, live1 = 16 + qq < LHF;
            const float* w0 = lds + kWS1 + (live0 ? hh * LHF + qq : 0);
            const float* w1 = lds + kWS1 + (live1 ? hh * LHF + 16 + qq : 0);
            f32x16 acc[2] = {zero16(), zero16()};
            mfma_chain<32, 2>(acc,
                              [&](int st, int n) {
                                  const int kk = (krow64(st) + 4 * h) * kPS1;
                                  return n ? w1[kk] : w0[kk];      // dead rows: never stored
                              },
                              [&](int st) { return st < 16 ? g0[st & 15] : g1[st & 15]; });
            x0 = acc[0];
            x1 = acc[1];
        }
        // d(loss)/d(features) goes to HBM as [point][L*F]; the table scatter is a separate,
        // atomic-bound launch with its own lane mapping (see scatter_kernel in field_fwd.hip)
#ifndef NERF_BWD_ABL_NO_DFEAT      // (timing ablation, tools/ab builds only: what the strided stores of d(features) cost)
#define NERF_BWD_ABL_NO_DFEAT 0
#endif
        if (valid && !(NERF_BWD_ABL_NO_DFEAT && x0[0] != 12345.678f)) {
            float* dst = dfeat + p * LF + h * LHF;
            if (vec) {      // this lane's LHF gradients are 16-byte aligned and contiguous: 16-byte stores
                f32x4* d4 = (f32x4*)dst;
#pragma unroll
                for (int qd = 0; qd < XF / 4; ++qd) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (4 * qd + e < 16) ? x0[(4 * qd + e) & 15] : x1[(4 * qd + e - 16) & 15];
                    d4[qd] = v;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    if (t < LHF && h * LHF + t < LF) dst[t] = x0[t];
                    if (16 + t < LHF && h * LHF + 16 + t < LF) dst[16 + t] = x1[t];
                }
            }
        }
    }

    // ------------------------- flush the weight gradients -------------------------
    float* d_s1 = dsw;
    float* d_s2 = dsw + kHid * s.in_s;
    float* d_c1 = dcw;
    float* d_c2 = dcw + kHid * s.in_c;
    float* d_c3 = d_c2 + kHid * kHid;
    flush_tile(d_s1, acc_s1, wave >> 1, wave & 1, kHid, s.in_s, l31, h);
    flush_tile(d_c2, acc_c2, wave >> 1, wave & 1, kHid, kHid, l31, h);
    flush_tile(d_c1, acc_c1, wave / in_c_tiles, wave % in_c_tiles, kHid, s.in_c, l31, h);
    if (wave < 2) {
        if (4 + wave < 2 * in_c_tiles)
            flush_tile(d_c1, acc_x, (4 + wave) / in_c_tiles, (4 + wave) % in_c_tiles, kHid, s.in_c, l31, h);
        flush_tile(d_s2, acc_y, 0, wave, kGeoRows, kHid, l31, h);
    } else {
        flush_tile(d_c3, acc_y, 0, wave - 2, 3, kHid, l31, h);
    }
}

}  // namespace

extern "C" int nerf_field_bwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                              const float* color_params, const float* latent_emb, int64_t n_images, const NerfPointSrc* src,
                              const float* x_hashtable, const float* dsigma, const float* drgb, float* dfeat,
                              float* dtable, float* dsigma_params, float* dcolor_params, float* dlatent, void* stream) {
    (void)table;
    int rc = nerf_check_cfg("nerf_field_bwd", grid, net, true);
    if (rc) return rc;
    rc = nerf_check_src("nerf_field_bwd", src, true);
    if (rc) return rc;
    NERF_REQUIRE(sigma_params && color_params && x_hashtable && dsigma && drgb && dfeat && dsigma_params && dcolor_params,
                 "null pointer");
    NERF_REQUIRE(net->latent_dim == 0 || (latent_emb && dlatent && n_images > 0), "latent embedding missing");
    if (src->n_points == 0) return 0;
    const GridDev g = make_grid_dev(grid);
    const NetLds s = make_net_lds(grid, net, true, n_images);
    NERF_UNSUPPORTED(s.in_sp > 64, "more than 64 hash-grid features");
    NERF_UNSUPPORTED(s.in_c > 96 || s.lat > 48, "colour-net input wider than 96 or latent wider than 48");
    NERF_UNSUPPORTED(net->latent_dim > 0 && !latent_emb, "latent table missing");
    int sb_rows = s.in_sp > kHid ? s.in_sp : kHid;
    if (s.in_c > sb_rows) sb_rows = s.in_c;
    const int lds_bytes = (s.total + kHid * kSP + sb_rows * kSP) * (int)sizeof(float);
    NERF_UNSUPPORTED(lds_bytes > 160 * 1024, "network too large for LDS");
    const int64_t n_btiles = ceil_div64(src->n_points, kBlockPts);
    const unsigned nb = (unsigned)(n_btiles < kNumCU ? n_btiles : kNumCU);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH3(FF, XX, RR)                                                                                            \
    do {                                                                                                               \
        auto kern = field_bwd_kernel<FF, XX, (RR)>;                                                                      \
        static NerfLdsAttrCache lds_cache = {};   /* per instantiation */    \
        if (int e = nerf_ensure_dynamic_lds((const void*)kern, lds_bytes, &lds_cache, "field_bwd")) return e;    \
        kern<<<nb, 256, lds_bytes, st>>>(g, s, sigma_params, color_params, latent_emb, *src, x_hashtable, dsigma, drgb, \
                                         dfeat, dsigma_params, dcolor_params, dlatent, sb_rows);                       \
    } while (0)
    // the reference's colour-net input (SH degree 4 + latent 48) gets its layer-1 input map at compile time
    const bool ref_shape = s.sh_half == 8 && s.lat_half == 24;
#define LAUNCH2(FF, XX)                                  \
    do {                                                 \
        if (ref_shape) LAUNCH3(FF, XX, true);            \
        else LAUNCH3(FF, XX, false);                     \
    } while (0)
#define LAUNCH(FF)                                       \
    do {                                                 \
        if (s.in_sp / 2 <= 16) LAUNCH2(FF, 16);          \
        else if (s.in_sp / 2 <= 20) LAUNCH2(FF, 20);     \
        else LAUNCH2(FF, 32);                            \
    } while (0)
    switch (grid->n_features) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
#undef LAUNCH2
#undef LAUNCH3
    rc = nerf_launch_status("nerf_field_bwd");
    if (rc || !dtable) return rc;
    return nerf_hashgrid_scatter(grid, src, dfeat, dtable, 0, grid->n_levels, stream);
}
