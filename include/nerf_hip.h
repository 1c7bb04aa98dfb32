/*
 * nerf_hip.h -- C ABI of libnerf_hip.so: the MI355X (gfx950) kernels behind the per-ray
 * render-and-train hot path of caseypeat/NeRF.
 *
 * The reference has no native code of its own: its hot path is Python (renderer.py, nets.py,
 * helpers.py, trainer.py) calling stock ATen ops and the third-party CUDA package tinycudann.
 * Each entry point below names the reference lines (relative to the reference checkout) whose
 * arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library allocates
 *    nothing, keeps no pointer past return and never synchronises: it only enqueues work on
 *    `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *  - tensors are contiguous row-major fp32 unless stated; indices are int64;
 *  - return 0 on success, a negative NERF_E* code for an invalid argument, or a positive
 *    hipError_t; nerf_last_error() gives the thread-local message;
 *  - "nullable" outputs may be NULL and are then skipped.
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_ABI_VERSION 4
#define NERF_MAX_LEVELS 32
#define NERF_MAX_SEGMENTS 16
#define NERF_MAX_SAMPLES 512 /* per-ray kernels: one wavefront per ray, at most 8 samples per lane */

#define NERF_EINVAL (-1)   /* null pointer / bad size */
#define NERF_EUNSUPPORTED (-2) /* configuration outside what the fused kernels cover */

/* Multiresolution hash grid geometry (tiny-cuda-nn "HashGrid" as configured at nets.py:39-50). */
typedef struct NerfGridCfg {
    int32_t n_levels;
    int32_t n_features;        /* floats per table entry (1, 2 or 4) */
    int32_t log2_hashmap_size;
    int32_t base_resolution;
    float scale[NERF_MAX_LEVELS];      /* pos = fma(x, scale, 0.5) */
    uint32_t res[NERF_MAX_LEVELS];
    uint32_t entries[NERF_MAX_LEVELS]; /* table entries of the level */
    uint32_t hashed[NERF_MAX_LEVELS];  /* 1: spatial hash, 0: dense index */
    uint64_t offset[NERF_MAX_LEVELS];  /* first entry of the level in the flat table */
    uint64_t total_entries;
} NerfGridCfg;

/* Field network shape (nets.py:13-28 keyword arguments).  The fused kernels cover exactly the shape of
 * configs/config_base.yaml:78-100 plus the listed variations; any other value makes the fused entry points return
 * NERF_EUNSUPPORTED (nerf_sigma_param_count / nerf_color_param_count are plain size formulas and accept any shape). */
typedef struct NerfNetCfg {
    int32_t hidden_dim;    /* sigma_hidden_dim == color_hidden_dim, must be 64 */
    int32_t sigma_layers;  /* sigma_num_layers, must be 2 (in -> 64 -> 16) */
    int32_t geo_feat_dim;  /* must be 15 (sigma net emits 16 rows) */
    int32_t color_layers;  /* color_num_layers, must be 3 (in -> 64 -> 64 -> 3) */
    int32_t sh_degree;     /* encoding_dir_degree: 0, 2 or 4 */
    int32_t latent_dim;    /* latent_embedding_dim, even, 0..48 */
} NerfNetCfg;

/* Where the sample points of a launch come from.
 * mode 0 "points": x[M,3] are already contracted positions (what nets.py:109,148 receive),
 *                  d[M,3] unit directions and n[M] image indices (field calls only).
 * mode 1 "rays":   point (ray r, sample s) = rays_o[r] + rays_d[r] * z[r*z_stride + s]
 *                  (renderer.py:51-61), contracted in-kernel (renderer.py:64-68), direction
 *                  rays_d[r]/|rays_d[r]|, image index n[r]; z_stride = 0 shares one depth table. */
typedef struct NerfPointSrc {
    int32_t mode;
    int32_t n_samples;        /* S (mode 1) */
    int64_t n_points;         /* M = rays * S in mode 1 */
    const float* x;
    const float* d;
    const int64_t* n;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    int64_t z_stride;
    float inner_bound;
    float outer_bound;
    /* Traversal hint for the forward kernels (mode 1 only; 0 = off, else 32, 16 or 8, dividing the ray count): a
     * wavefront takes 1 (2, 4) consecutive sample indices of 32 (16, 8) consecutive rays instead of 32 consecutive
     * samples of one ray (n_samples must be a multiple of 1, 2, 4). Inputs and
     * outputs keep their [ray][sample] places, results are bit-identical; it pays when consecutive rays are
     * neighbouring pixels (inference.py:54-94), whose samples share hash-grid cells. Ignored by every other kernel. */
    int32_t ray_block;
    int32_t reserved_;
} NerfPointSrc;

/* A subset of hash-grid levels, for the level-sharded (multi-GPU) table: column block i of a
 * [*, count*F] feature matrix belongs to level level[i]. */
typedef struct NerfLevelSet {
    int32_t count;
    int32_t level[NERF_MAX_LEVELS];
} NerfLevelSet;

int nerf_abi_version(void);
const char* nerf_last_error(void);

/* Fill a NerfGridCfg from the encoding config (nets.py:41-48); returns total table entries. */
int64_t nerf_grid_cfg_init(NerfGridCfg* cfg, int n_levels, int n_features, int log2_hashmap_size,
                           int base_resolution, float per_level_scale);
/* Floats of flat MLP parameters for the two networks (nets.py:53-63, 96-106), no biases. */
int64_t nerf_sigma_param_count(const NerfGridCfg* grid, const NerfNetCfg* net);
int64_t nerf_color_param_count(const NerfNetCfg* net);

/* ---- training batch: loaders/camera_geometry_loader.py:112-117 (get_random_batch) + :91-108 + :68-88 ----
 * One launch draws `batch` rays (image n, row h, column w ~ uniform; Philox4x32-10, key = seed, counter = (ray,
 * `counter`): a batch is a pure function of (seed, counter)), gathers their calibration from the per-image stores
 * (`extrinsics` already re-centred as at :95) and forms the ground truth: uint8 / 255, RGBA blended over ONE random
 * background colour per batch (:72,81), written to color_bg[3].  images [n_images,H,W,channels] uint8, channels 3 or 4,
 * nullable (then rgb_gt is the loader's pseudo-colour ((h, w, h+w) % 256) / 255); depths [n_images,H,W] nullable with
 * depth [batch] nullable. */
int nerf_sample_batch(const uint8_t* images, const float* depths, const float* intrinsics /*[n_images,3,3]*/,
                      const float* extrinsics /*[n_images,4,4]*/, int64_t n_images, int H, int W, int channels,
                      int64_t batch, uint64_t seed, uint64_t counter, int64_t* n, int64_t* h, int64_t* w,
                      float* K /*[batch,3,3]*/, float* E /*[batch,4,4]*/, float* rgb_gt /*[batch,3]*/, float* color_bg /*[3]*/,
                      float* depth, void* stream);

/* ---- ray generation and sample points: renderer.py:38-48 (get_rays) ---- */
int nerf_raygen(const int64_t* h, const int64_t* w, const float* K /*[N,3,3]*/, const float* E /*[N,4,4]*/,
                int64_t n_rays, float* rays_o /*[N,3]*/, float* rays_d /*[N,3]*/, void* stream);
/* renderer.py:51-61 (get_sample_points) + :64-68 (mipnerf360_scale); all outputs nullable. */
int nerf_sample_points(const float* rays_o, const float* rays_d, const float* z, int64_t z_stride, int64_t n_rays,
                       int n_samples, float inner_bound, float outer_bound, float* xyz /*[N,S,3]*/,
                       float* s_xyz /*[N,S,3]*/, float* dirs /*[N,3]*/, void* stream);
/* renderer.py:64-68 on arbitrary points. */
int nerf_contract(const float* xyz, int64_t n_points, float inner_bound, float outer_bound, float* out, void* stream);

/* ---- encodings as stand-alone operators (what tcnn.Encoding does at nets.py:118,129,154) ---- */
int nerf_hashgrid_fwd(const NerfGridCfg* grid, const float* table, const float* x01 /*[M,3]*/, int64_t n_points,
                      float* out /*[M, L*F]*/, void* stream);
int nerf_hashgrid_bwd(const NerfGridCfg* grid, const float* x01, const float* dout /*[M, L*F]*/, int64_t n_points,
                      float* dtable /* accumulated into */, void* stream);
int nerf_sh_fwd(const float* d01 /*[M,3] in [0,1]*/, int64_t n_points, int degree, float* out /*[M,degree^2]*/, void* stream);
/* Bias-free MLP (tcnn.Network at nets.py:120,135,155): dims[0]=n_in, dims[n_layers]=n_out. */
int nerf_mlp_fwd(const float* params, const int32_t* dims, int n_layers, const float* x, int64_t n_rows, float* out,
                 void* stream);

/* Backward of nerf_mlp_fwd (what autograd + tcnn.Network do when the two networks are called as separate modules).
 * dparams is ACCUMULATED into; dx [n_rows, dims[0]] is nullable; workspace: nerf_mlp_bwd_workspace_floats floats. */
int64_t nerf_mlp_bwd_workspace_floats(const int32_t* dims, int n_layers, int64_t n_rows);
int nerf_mlp_bwd(const float* params, const int32_t* dims, int n_layers, const float* x, const float* dout, int64_t n_rows,
                 float* dx, float* dparams, float* workspace, int64_t workspace_floats, void* stream);

/* nerf_mlp_bwd without a workspace argument.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The
 * workspace is a static 18 MiB device array inside the library (every GPU the library runs on holds its own copy
 * until the process ends); the call allocates and frees nothing.  n_rows is limited by that array: the floats
 * nerf_mlp_bwd_workspace_floats reports must not exceed 4 718 592 (9162 rows for 79 -> 64 -> 64 -> 3), else
 * NERF_EINVAL with a message naming the limit.  All calls on one GPU share the array: calls that overlap on one GPU
 * (two streams, two threads) race.  Same results as nerf_mlp_bwd.
 * The FIRST call on a device looks the array's address up (hipGetSymbolAddress, which may load the code object): that is
 * not stream-ordered and must not run inside a graph capture.  Make one warm-up call per device outside any capture;
 * n_rows = 0 (pointers may then be NULL) resolves the address and launches nothing. */
int nerf_mlp_bwd_noalloc(const float* params, const int32_t* dims, int n_layers, const float* x, const float* dout,
                         int64_t n_rows, float* dx, float* dparams, void* stream);

/* ---- fused field network ---- */
/* NeRFNetwork.density (nets.py:148-164): normalise, hash-grid encode, sigma MLP, relu. */
int nerf_density_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                     const NerfPointSrc* src, float* sigma /*[M]*/, float* x_hashtable /*[M,L*F] nullable*/, void* stream);
/* NeRFNetwork.forward (nets.py:109-146). rgb [M,3]; xyz [M,3] (uncontracted, mode 1 only) nullable.
 * An image index outside [0, n_images) (the reference's latent_emb[n] raises there, nets.py:132) is memory safe: the
 * point reads a zero latent row, and nerf_field_bwd adds nothing to dlatent for it. */
int nerf_field_fwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                   const float* color_params, const float* latent_emb /*[n_images, latent_dim]*/, int64_t n_images,
                   const NerfPointSrc* src, float* sigma, float* rgb, float* x_hashtable /*nullable*/,
                   float* xyz /*nullable*/, void* stream);
/* Backward of nerf_field_fwd (what autograd + tcnn do under trainer.py:134).  Gradients are
 * ACCUMULATED into dtable / dsigma_params / dcolor_params / dlatent (caller zeroes them).
 * dfeat [M, L*F] receives d(loss)/d(hash-grid features) (it is also the staging buffer between the
 * MLP-backward launch and the table-scatter launch); dtable may be NULL to skip the scatter.
 * Precision: the forward recompute, the back-propagated gradients, dfeat and dlatent are exact fp32 (f32-input MFMA);
 * the weight-gradient sums dW = sum over points of dZ * H^T run on the bf16 matrix cores with both operands split into
 * bf16 (hi, lo) halves, all four cross terms, fp32 accumulation: 2^-16 relative per term. */
int nerf_field_bwd(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                   const float* color_params, const float* latent_emb, int64_t n_images, const NerfPointSrc* src,
                   const float* x_hashtable, const float* dsigma /*[M]*/, const float* drgb /*[M,3]*/,
                   float* dfeat /*[M,L*F]*/, float* dtable, float* dsigma_params, float* dcolor_params,
                   float* dlatent, void* stream);

/* The table-scatter half of nerf_field_bwd on its own: dtable += scatter(dfeat) for the points of `src`
 * (the same trilinear weights as the forward gather; what tcnn's HashGrid backward does), restricted to
 * levels [level_begin, level_begin + level_count) so that a caller can pipeline per-level
 * scatter -> all-reduce -> optimizer step on separate streams. */
int nerf_hashgrid_scatter(const NerfGridCfg* grid, const NerfPointSrc* src, const float* dfeat /*[M,L*F]*/,
                          float* dtable, int level_begin, int level_count, void* stream);

/* d(loss)/d(x) of NeRFNetwork.density (nets.py:148-164) for dL/dsigma = dsigma: the input gradient autograd + tcnn
 * deliver when the pose alignment differentiates model.density through the sample points
 * (allign/trainer.py:174-208).  Points mode only; dx [M,3] is w.r.t. the positions in src->x. */
int nerf_density_bwd_input(const NerfGridCfg* grid, const NerfNetCfg* net, const float* table, const float* sigma_params,
                           const NerfPointSrc* src, const float* dsigma /*[M]*/, float* dx /*[M,3]*/, void* stream);

/* ---- level-sharded hash table (no counterpart in the single-GPU reference) ----
 * Under multi-GPU training every rank owns a subset of the hash-grid levels.  The owner evaluates
 * tcnn.Encoding (nets.py:118,154) for ITS levels on the sample points of all ranks, the per-point feature
 * rows travel by all-to-all, and the MLP half of nets.py:109-164 runs where the rays live.  These four entry
 * points are the two halves of nerf_density_fwd / nerf_field_fwd / nerf_hashgrid_scatter cut at the feature
 * matrix. */
/* out[p*out_stride + i*F + f] = feature f of level levels->level[i] at point p of `src` (either mode). */
int nerf_hashgrid_encode_levels(const NerfGridCfg* grid, const float* table, const NerfPointSrc* src,
                                const NerfLevelSet* levels, float* out, int64_t out_stride, void* stream);
/* dtable += scatter of dfeat[p*dfeat_stride + i*F + f] into level levels->level[i], points of `src`. */
int nerf_hashgrid_scatter_levels(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels,
                                 const float* dfeat, int64_t dfeat_stride, float* dtable, void* stream);
/* nets.py:155-161 from ready features: sigma = relu(sigma_net(x_hashtable)[:, 0]). */
int nerf_density_from_features(const NerfGridCfg* grid, const NerfNetCfg* net, const float* sigma_params,
                               const float* x_hashtable /*[M,L*F]*/, int64_t n_points, float* sigma, void* stream);
/* nets.py:120-146 from ready features; `src` still provides directions, image indices and xyz. */
int nerf_field_from_features(const NerfGridCfg* grid, const NerfNetCfg* net, const float* sigma_params,
                             const float* color_params, const float* latent_emb, int64_t n_images,
                             const NerfPointSrc* src, const float* x_hashtable /*[M,L*F]*/, float* sigma, float* rgb,
                             float* xyz /*nullable*/, void* stream);

/* Column map between a row-major [n_rows, n_cols] feature matrix and the flat buffers of the feature all-to-all
 * (one [n_rows, width_r] block per level owner): element (p, c) lives at flat[offset[c] + p*stride[c]]. */
#define NERF_MAX_FEATURE_COLS 128
typedef struct NerfColMap {
    int64_t offset[NERF_MAX_FEATURE_COLS];
    int32_t stride[NERF_MAX_FEATURE_COLS];
} NerfColMap;
/* direction 0: matrix[p*n_cols + c] = flat[offset[c] + p*stride[c]]   (assemble x_hashtable after the all-to-all)
 * direction 1: flat[offset[c] + p*stride[c]] = matrix[p*n_cols + c]   (split d(features) before the all-to-all) */
int nerf_feature_shuffle(const float* src, float* dst, int64_t n_rows, int n_cols, const NerfColMap* map, int direction,
                         void* stream);

/* ---- half-precision gather copy of the table: `precision: float16` (nets.py:31-36, passed to tcnn.Encoding as dtype at
 * nets.py:49) ----
 * tinycudann keeps fp32 master parameters (what torch.optim.Adam updates, train.py:84-88) and evaluates the encoding
 * from their fp16 image.  Here the image is a second buffer of total_entries * n_features _Float16 that the caller
 * refreshes with nerf_table_to_half after every parameter update (4 B read + 2 B written per parameter), and the
 * `_f16` forms of the entry points that READ the table gather from it: entries are widened on load, the trilinear
 * blend and everything downstream stay fp32 (the reference blends and emits fp16: this path is at least as precise;
 * tinycudann is absent, so its fp16 arithmetic is unpinned -- the parity bar of these entry points is the oracle with
 * its table rounded to fp16, tests/test_gpu_half_table.py).  Arguments, errors and outputs are those of the
 * fp32 forms; gradients (nerf_field_bwd, the scatter / bin passes) never read the table and need no `_f16` form. */
int nerf_table_to_half(const float* table, void* table_f16, int64_t count /* floats */, void* stream);
int nerf_hashgrid_fwd_f16(const NerfGridCfg* grid, const void* table_f16, const float* x01 /*[M,3]*/, int64_t n_points,
                          float* out /*[M,L*F]*/, void* stream);
int nerf_density_fwd_f16(const NerfGridCfg* grid, const NerfNetCfg* net, const void* table_f16, const float* sigma_params,
                         const NerfPointSrc* src, float* sigma /*[M]*/, float* x_hashtable /*[M,L*F] nullable*/, void* stream);
int nerf_field_fwd_f16(const NerfGridCfg* grid, const NerfNetCfg* net, const void* table_f16, const float* sigma_params,
                       const float* color_params, const float* latent_emb, int64_t n_images, const NerfPointSrc* src,
                       float* sigma, float* rgb, float* x_hashtable, float* xyz, void* stream);
int nerf_density_bwd_input_f16(const NerfGridCfg* grid, const NerfNetCfg* net, const void* table_f16,
                               const float* sigma_params, const NerfPointSrc* src, const float* dsigma /*[M]*/,
                               float* dx /*[M,3]*/, void* stream);
int nerf_hashgrid_encode_levels_f16(const NerfGridCfg* grid, const void* table_f16, const NerfPointSrc* src,
                                    const NerfLevelSet* levels, float* out, int64_t out_stride, void* stream);
/* The optimizer passes that keep the image current themselves: nerf_adam_step_guarded / nerf_table_update_binned_guarded
 * (train.py:84-88, trainer.py:134-138) with one more store per parameter, the new value rounded to nearest even into
 * param_f16[i] / table_f16[i] (same index as the fp32 array; 8-byte aligned).  26 B per parameter instead of 24 B + a
 * nerf_table_to_half pass of 6 B.  A dropped step (skip_flag) leaves the image untouched as well. */
int nerf_adam_step_f16(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int64_t step, float grad_scale, int zero_grad,
                       const double* skip_flag /* nullable */, void* param_f16, void* stream);
int nerf_table_update_binned_f16(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points, const void* workspace,
                                 int64_t workspace_bytes, float* table, float* exp_avg, float* exp_avg_sq, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                 const double* skip_flag /* nullable */, void* table_f16, void* stream);

/* ---- hierarchical resampling ---- */
/* renderer.py:154-160: alpha = 1-exp(-sigma*delta), exclusive transmittance, weights. zlog has
 * row stride zlog_stride (0 = one shared [S] table). */
int nerf_coarse_weights(const float* sigma, const float* zlog, int64_t zlog_stride, int64_t n_rays, int n_samples,
                        float* weights, void* stream);
/* renderer.py:132-142 (reweight_the_weights). */
int nerf_reweight(const float* weights, int64_t n_rays, int n_samples, float alpha, const int32_t* steps, int n_segments,
                  float* pdf, void* stream);
/* renderer.py:87-118 (sample_pdf) + :165 (pow 10).  noise[N,Sf] replaces torch.rand of :102;
 * lin[Sf] is torch.linspace(0, 1-1/Sf, Sf) from the host.  inds (int64) nullable; z_fine nullable. */
int nerf_sample_pdf(const float* zlog, int64_t zlog_stride, const float* pdf, const float* noise, const float* lin,
                    int64_t n_rays, int n_samples, int n_importance, float* zlog_fine, float* z_fine, int64_t* inds,
                    void* stream);
/* renderer.py:154-165 in one launch: sigma -> weights -> pdf -> fine depths. */
int nerf_coarse_to_fine(const float* sigma, const float* zlog, int64_t zlog_stride, const float* noise, const float* lin,
                        int64_t n_rays, int n_samples, int n_importance, float alpha, const int32_t* steps, int n_segments,
                        float* zlog_fine, float* z_fine, void* stream);
/* nerf_coarse_to_fine with the stratification noise of renderer.py:102 (torch.rand) drawn INSIDE the kernel: element
 * ray * n_importance + k of the Philox4x32-10 uniform stream (seed, counter) -- 24-bit values in [0, 1), a pure function of
 * (seed, counter, element), so a step is reproducible whatever the launch shape.  nerf_uniform_noise writes the first n
 * elements of the same stream: nerf_coarse_to_fine on that array gives the same depths bit for bit (replay, tests). */
int nerf_coarse_to_fine_rng(const float* sigma, const float* zlog, int64_t zlog_stride, uint64_t seed, uint64_t counter,
                            const float* lin, int64_t n_rays, int n_samples, int n_importance, float alpha,
                            const int32_t* steps, int n_segments, float* zlog_fine, float* z_fine, void* stream);
/* The same for rays [first_ray, first_ray + n_rays) of a batch (sigma, zlog rows and outputs are those of the slice): the
 * noise is element (first_ray + ray) * n_importance + k of the stream, so a batch resampled slice by slice -- the ray
 * groups of the level-sharded step, which start each group's depth all-gather while the next group is still in its
 * coarse pass -- draws exactly what one launch over the whole batch draws. */
int nerf_coarse_to_fine_rng_at(const float* sigma, const float* zlog, int64_t zlog_stride, uint64_t seed, uint64_t counter,
                               int64_t first_ray, const float* lin, int64_t n_rays, int n_samples, int n_importance,
                               float alpha, const int32_t* steps, int n_segments, float* zlog_fine, float* z_fine,
                               void* stream);
int nerf_uniform_noise(uint64_t seed, uint64_t counter, int64_t n, float* out, void* stream);

/* ---- alpha compositing: renderer.py:71-85 (render_rays_log) + :181-182 ---- */
int nerf_composite_fwd(const float* sigma, const float* rgb, const float* z, const float* zlog, const float* bg /*[3]*/,
                       int64_t n_rays, int n_samples, float zlog_lo, float zlog_range, float* image /*[N,3]*/,
                       float* invdepth /*[N]*/, float* weights /*[N,S]*/, float* zlog_s /*[N,S] nullable*/, void* stream);
/* d(loss)/d(sigma, rgb) from d(loss)/d(image, weights); dweights nullable (treated as 0). */
int nerf_composite_bwd(const float* sigma, const float* rgb, const float* zlog, const float* bg, const float* dimage,
                       const float* dweights, int64_t n_rays, int n_samples, float* dsigma, float* drgb, void* stream);

/* ---- losses: helpers.py:21-32, trainer.py:153-169 ---- */
/* sums[0..5] (SIX doubles, zeroed by the call) receive loss_rgb, loss_dist, loss_depth, mse,
 * loss_rgb + lam_dist*loss_dist + lam_depth*loss_depth, and in sums[5] a non-finite flag: != 0 iff that total is
 * NaN / Inf (the guard torch.cuda.amp.GradScaler provides at trainer.py:134-136; see nerf_adam_step_guarded).
 * dimage / dweights (nullable) receive grad_scale * d(loss_rgb + lam_dist*loss_dist + lam_depth*loss_depth).
 * sorted != 0 asserts zlog_s rows are non-decreasing (O(S) distortion form), else O(S^2). */
int nerf_losses(const float* image, const float* gt, const float* weights, const float* zlog_s, int64_t n_rays,
                int n_samples, float lam_dist, float lam_depth, float grad_scale, int sorted, double* sums,
                float* dimage, float* dweights, void* stream);

/* ---- compositing + losses in one launch for the training step, and the backward of both in one more ----
 * nerf_composite_fwd followed by nerf_losses (sorted form, grad_scale 1) / the gradients nerf_losses would write followed
 * by nerf_composite_bwd, with the same arithmetic (renderer.py:71-85,180-182; helpers.py:21-32; trainer.py:153-169 and
 * the autograd of :134): image, weights, zlog_s, dsigma and drgb equal the separate entry points' bit for bit.
 * sums: NERF_LOSS_SUMS_DOUBLES doubles, written by the call: [0..4] as nerf_losses, [5] the non-finite flag of the total,
 * [8 ..] one row of four partial sums per workgroup (added up in a fixed order: the totals are reproducible run to run).
 * loss_out: six floats: loss_rgb, loss_dist, loss_depth, mse, the total and the flag.
 * dout (nullable = 1): device scalar, d(objective)/d(total loss) -- what autograd hands to backward(). */
#define NERF_LOSS_MAX_ROWS 4096
#define NERF_LOSS_SUMS_DOUBLES (8 + 4 * NERF_LOSS_MAX_ROWS)
int nerf_composite_loss_fwd(const float* sigma, const float* rgb, const float* z, const float* zlog, const float* bg,
                            const float* gt /*[N,3]*/, int64_t n_rays, int n_samples, float zlog_lo, float zlog_range,
                            float lam_dist, float lam_depth, float* image, float* invdepth, float* weights,
                            float* zlog_s /*nullable*/, double* sums /*[NERF_LOSS_SUMS_DOUBLES]*/, float* loss_out /*[6]*/,
                            void* stream);
/* nonfinite_flag (nullable): the step's flag, sums + 5 of the forward call.  The backward sets it to 1 when a gradient it
 * writes is NaN / Inf although the loss was finite (an Inf upstream gradient, an overflow): GradScaler.step skips on
 * non-finite GRADIENTS (trainer.py:134-136), and the guarded optimizer entry points read the flag after this kernel. */
int nerf_composite_loss_bwd(const float* sigma, const float* rgb, const float* zlog, const float* bg, const float* gt,
                            const float* image, const float* dout, int64_t n_rays, int n_samples, float zlog_lo,
                            float zlog_range, float lam_dist, float lam_depth, float* dsigma, float* drgb,
                            double* nonfinite_flag, void* stream);

/* ---- optimizer: torch.optim.Adam as configured at train.py:84-88 (dense, L2 weight decay) ---- */
/* step is the 1-based step count; when zero_grad != 0 the gradient is cleared in the same pass. */
int nerf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, float grad_scale, int zero_grad, void* stream);

/* nerf_adam_step with a device-side guard: skip_flag (nullable) points at a double written earlier on the stream
 * (nerf_losses' sums[5]); when it is non-zero the parameters and moments are left untouched and only the gradient is
 * cleared (zero_grad != 0) -- GradScaler.step's behaviour on a non-finite gradient (trainer.py:134-136), without a
 * host synchronisation. */
int nerf_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                           int zero_grad, const double* skip_flag, void* stream);

/* ---- binned table update: tcnn HashGrid backward + torch.optim.Adam (train.py:84-88) on the hash table in one pass ----
 * The same result as nerf_hashgrid_scatter_levels into a zeroed gradient followed by nerf_adam_step over the listed
 * levels (up to the summation order of the corner contributions), without a gradient array: the step's corner updates
 * are sorted into table partitions first (nerf_table_bin_levels: needs no optimizer hyper-parameters, so it can run
 * inside the backward pass), then every partition is summed in LDS and updated in place (nerf_table_update_binned:
 * 24 B of HBM traffic per parameter instead of 32 B, no memory-side atomics).  Levels must hold a multiple of 4 floats
 * at a 16-byte offset and at most 8192 partitions of 16 K floats (NERF_EUNSUPPORTED otherwise); meant for the hashed
 * and the large dense levels -- a level smaller than one partition works but serialises on one workgroup.
 * The caller owns the workspace; it must stay untouched between the two calls.  n_points == 0 is valid: nothing is
 * sorted and the update is Adam with a zero gradient (the workspace is not read). */
int64_t nerf_table_bin_workspace_bytes(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points); /* < 0: error */
/* dfeat_cols (host array of levels->count ints, nullable): first of the F gradient columns of levels->level[i] in a
 * row of dfeat; NULL = i*F (as in nerf_hashgrid_scatter_levels).  The full [M, L*F] matrix nerf_field_bwd writes
 * has level l at column l*F. */
int nerf_table_bin_levels(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels, const float* dfeat,
                          int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace, int64_t workspace_bytes,
                          void* stream);
/* The two halves of nerf_table_bin_levels: the count + scan passes need only the points (known once the fine depths are
 * sampled), the fill pass needs the gradient rows.  A caller may issue the first half early, e.g. on a side stream under
 * the field forward; the fill must see the SAME src, levels and workspace, untouched in between. */
int nerf_table_bin_count(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels, void* workspace,
                         int64_t workspace_bytes, void* stream);
int nerf_table_bin_fill(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels, const float* dfeat,
                        int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace, int64_t workspace_bytes,
                        void* stream);
/* The fill for the points [p_begin, p_end) of `src` only (whole chunks of 16 384 points; p_end may also be n_points):
 * under level sharding the rows of d(features) reach a level's owner ray group by ray group, and the rows of a group are
 * sorted into the partitions while the next group is still on the wire.  Every point must be filled exactly once before
 * nerf_table_update_binned; dfeat is the base of the full matrix (row = point index in `src`). */
int nerf_table_bin_fill_range(const NerfGridCfg* grid, const NerfPointSrc* src, const NerfLevelSet* levels, const float* dfeat,
                              int64_t dfeat_stride, const int32_t* dfeat_cols, void* workspace, int64_t workspace_bytes,
                              int64_t p_begin, int64_t p_end, void* stream);
/* table / exp_avg / exp_avg_sq: the WHOLE table's tensors (level offsets come from grid); only the listed levels change.
 * n_points and levels must be those of the nerf_table_bin_levels call that filled the workspace. */
int nerf_table_update_binned(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points, const void* workspace,
                             int64_t workspace_bytes, float* table, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);

/* The same with the guard of nerf_adam_step_guarded: a non-zero *skip_flag drops the whole update. */
int nerf_table_update_binned_guarded(const NerfGridCfg* grid, const NerfLevelSet* levels, int64_t n_points,
                                     const void* workspace, int64_t workspace_bytes, float* table, float* exp_avg,
                                     float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int64_t step, float grad_scale, const double* skip_flag, void* stream);

/* ---- inference helpers: inference.py:97-105 ---- */
int nerf_cum_weights_thresh(const float* weights, int64_t n_rays, int n_samples, float thresh, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
