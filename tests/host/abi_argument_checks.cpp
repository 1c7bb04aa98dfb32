// Host-side driver for the sanitizer build (make -C nerf_amd/csrc asan): calls every argument-validation path of
// libnerf_hip.so that returns BEFORE any kernel launch, with the library's host code compiled under
// -fsanitize=address,undefined.  No GPU is needed and none is touched: every call here is either rejected by its checks
// or is a documented no-op (zero rays / rows).  Exit code 0 = every call returned what its contract says and the
// sanitizers stayed silent.  Run by tests/test_host_sanitizers.py on the CPU box.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/nerf_hip.h"

static int g_fail = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { printf("FAIL %s:%d: %s   [%s]\n", __FILE__, __LINE__, #cond, nerf_last_error()); ++g_fail; } \
    } while (0)

// a non-null "device pointer" that is never dereferenced: every call that receives it is rejected before its launch
template <class T> static T* fake(uintptr_t a = 0x1000) { return (T*)a; }

int main() {
    EXPECT(nerf_abi_version() == NERF_ABI_VERSION);
    // ---- grid geometry (nets.py:39-50) ----
    NerfGridCfg g;
    EXPECT(nerf_grid_cfg_init(&g, 20, 2, 24, 16, 1.3819f) == 200132568);
    EXPECT(g.res[0] == 16 && g.res[8] == 213 && !g.hashed[8] && g.hashed[9] && g.entries[19] == (1u << 24));
    EXPECT(nerf_grid_cfg_init(nullptr, 20, 2, 24, 16, 1.3819f) == NERF_EINVAL);
    EXPECT(nerf_grid_cfg_init(&g, 0, 2, 24, 16, 1.3819f) == NERF_EINVAL);
    EXPECT(nerf_grid_cfg_init(&g, NERF_MAX_LEVELS + 1, 2, 24, 16, 1.3819f) == NERF_EINVAL);
    EXPECT(nerf_grid_cfg_init(&g, 20, 3, 24, 16, 1.3819f) == NERF_EINVAL);
    EXPECT(nerf_grid_cfg_init(&g, 20, 2, 31, 16, 1.3819f) == NERF_EINVAL);
    EXPECT(nerf_grid_cfg_init(&g, 20, 2, 24, 0, 1.3819f) == NERF_EINVAL);
    EXPECT(nerf_grid_cfg_init(&g, 20, 2, 24, 16, 0.5f) == NERF_EINVAL);
    EXPECT(strlen(nerf_last_error()) > 0);
    EXPECT(nerf_grid_cfg_init(&g, NERF_MAX_LEVELS, 4, 30, 16, 2.0f) > 0);       // the largest table the struct describes
    EXPECT(nerf_grid_cfg_init(&g, 20, 2, 24, 16, 1.3819f) == 200132568);
    NerfNetCfg net = {64, 2, 15, 3, 4, 48};
    EXPECT(nerf_sigma_param_count(&g, &net) == 40 * 64 + 64 * 16);
    EXPECT(nerf_color_param_count(&net) == 79 * 64 + 64 * 64 + 64 * 3);
    EXPECT(nerf_sigma_param_count(nullptr, &net) == NERF_EINVAL && nerf_color_param_count(nullptr) == NERF_EINVAL);

    // ---- fused field entry points: configuration and point-source checks ----
    float* T = fake<float>();
    NerfPointSrc pts = {};
    pts.mode = 0; pts.n_points = 64; pts.x = T; pts.d = T; pts.n = fake<int64_t>(); pts.outer_bound = 2.5f;
    NerfNetCfg bad_net = net;
    bad_net.hidden_dim = 32;
    EXPECT(nerf_density_fwd(&g, &bad_net, T, T, &pts, T, nullptr, nullptr) == NERF_EUNSUPPORTED);
    bad_net = net; bad_net.latent_dim = 47;
    EXPECT(nerf_field_fwd(&g, &bad_net, T, T, T, T, 5, &pts, T, T, nullptr, nullptr, nullptr) == NERF_EUNSUPPORTED);
    bad_net = net; bad_net.sh_degree = 3;
    EXPECT(nerf_field_fwd(&g, &bad_net, T, T, T, T, 5, &pts, T, T, nullptr, nullptr, nullptr) == NERF_EUNSUPPORTED);
    EXPECT(nerf_density_fwd(nullptr, &net, T, T, &pts, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_density_fwd(&g, &net, nullptr, T, &pts, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_density_fwd(&g, &net, T, T, nullptr, T, nullptr, nullptr) == NERF_EINVAL);
    NerfPointSrc s2 = pts;
    s2.n_points = -1;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == NERF_EINVAL);
    s2 = pts; s2.mode = 2;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == NERF_EINVAL);
    s2 = pts; s2.outer_bound = 0.f;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == NERF_EINVAL);
    s2 = pts; s2.ray_block = 32;                                                  // points mode takes no ray blocks
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == NERF_EINVAL);
    NerfPointSrc rays = {};
    rays.mode = 1; rays.n_samples = 40; rays.n_points = 40 * 96; rays.rays_o = T; rays.rays_d = T; rays.z = T;
    rays.z_stride = 40; rays.inner_bound = 2.f; rays.outer_bound = 2.5f; rays.n = fake<int64_t>();
    s2 = rays; s2.n_points = 40 * 96 + 1;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == NERF_EINVAL);
    s2 = rays; s2.ray_block = 12;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == NERF_EINVAL);
    s2 = rays; s2.ray_block = 32;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, T, nullptr) == NERF_EINVAL);  // ray blocks keep no x_hashtable
    s2 = rays; s2.n_points = 0;
    EXPECT(nerf_density_fwd(&g, &net, T, T, &s2, T, nullptr, nullptr) == 0);      // nothing to do
    EXPECT(nerf_field_fwd(&g, &net, T, T, T, nullptr, 0, &rays, T, T, nullptr, nullptr, nullptr) == NERF_EINVAL);   // latent missing
    EXPECT(nerf_field_bwd(&g, &net, T, T, T, T, 5, &rays, nullptr, T, T, T, nullptr, T, T, T, nullptr) != 0);
    EXPECT(nerf_density_bwd_input(&g, &net, T, T, &rays, T, T, nullptr) != 0);    // points mode only
    EXPECT(nerf_hashgrid_scatter(&g, &rays, T, T, 18, 5, nullptr) != 0);          // levels 18..22 of 20
    EXPECT(nerf_hashgrid_fwd(&g, nullptr, T, 10, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_sh_fwd(T, 10, 3, T, nullptr) != 0);
    // the half-precision table forms validate like the fp32 ones
    NerfNetCfg bad16 = net;
    bad16.hidden_dim = 32;
    EXPECT(nerf_density_fwd_f16(&g, &bad16, T, T, &pts, T, nullptr, nullptr) == NERF_EUNSUPPORTED);
    EXPECT(nerf_density_fwd_f16(&g, &net, nullptr, T, &pts, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_field_fwd_f16(&g, &net, T, T, T, nullptr, 0, &rays, T, T, nullptr, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_density_bwd_input_f16(&g, &net, T, T, &rays, T, T, nullptr) != 0);
    EXPECT(nerf_hashgrid_fwd_f16(&g, nullptr, T, 10, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_hashgrid_encode_levels_f16(&g, T, &pts, nullptr, T, 40, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_to_half(nullptr, T, 16, nullptr) == NERF_EINVAL && nerf_table_to_half(T, T, -1, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_to_half((const float*)T + 1, T, 16, nullptr) == NERF_EINVAL);         // misaligned
    EXPECT(nerf_table_to_half(T, T, 0, nullptr) == 0);
    EXPECT(nerf_adam_step_f16(T, T, T, T, 16, 1e-2f, 0.9f, 0.99f, 1e-15f, 0.f, 1, 1.f, 1, nullptr, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_adam_step_f16(T, T, T, T, 16, 1e-2f, 0.9f, 0.99f, 1e-15f, 0.f, 0, 1.f, 1, nullptr, T, nullptr) == NERF_EINVAL);   // step 0
    EXPECT(nerf_adam_step_f16(T, T, T, T, 16, 1e-2f, 0.9f, 0.99f, 1e-15f, 0.f, 1, 1.f, 1, nullptr, (char*)T + 2, nullptr) == NERF_EINVAL);

    // ---- level sets of the sharded table and of the binned update ----
    NerfLevelSet ls = {};
    ls.count = 2; ls.level[0] = 3; ls.level[1] = 3;
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, 64) < 0);                      // level listed twice
    ls.level[1] = 40;
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, 64) < 0);                      // level out of range
    ls.count = 0;
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, 64) < 0);
    ls.count = NERF_MAX_LEVELS + 1;
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, 64) < 0);
    ls.count = 2; ls.level[0] = 9; ls.level[1] = 19;
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, -1) < 0);
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, 1ll << 28) < 0);
    EXPECT(nerf_table_bin_workspace_bytes(&g, nullptr, 64) < 0 && nerf_table_bin_workspace_bytes(nullptr, &ls, 64) < 0);
    const int64_t ws_bytes = nerf_table_bin_workspace_bytes(&g, &ls, 524288);
    EXPECT(ws_bytes > 2 * 8 * 524288 * 10);                                       // 10 B per item at least
    EXPECT(nerf_table_bin_workspace_bytes(&g, &ls, 0) >= 0);
    NerfGridCfg g4;
    EXPECT(nerf_grid_cfg_init(&g4, 8, 4, 30, 16, 2.0f) > 0);
    NerfLevelSet big = {};
    big.count = 1; big.level[0] = 7;                                             // 2^30 entries x 4 floats: too many partitions
    EXPECT(nerf_table_bin_workspace_bytes(&g4, &big, 64) == -2 || nerf_table_bin_workspace_bytes(&g4, &big, 64) == NERF_EINVAL);
    void* ws = fake<void>(0x10000);
    rays.n_points = 524288; rays.n_samples = 128;
    EXPECT(nerf_table_bin_levels(&g, &rays, &ls, T, 4, nullptr, nullptr, ws_bytes, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_bin_levels(&g, &rays, &ls, nullptr, 4, nullptr, ws, ws_bytes, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_bin_levels(&g, &rays, &ls, T, 4, nullptr, ws, ws_bytes - 1, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_bin_levels(&g, &rays, &ls, T, 3, nullptr, ws, ws_bytes, nullptr) == NERF_EINVAL);      // stride < 2 levels x F
    EXPECT(nerf_table_bin_levels(&g, &rays, &ls, T, 4, nullptr, fake<void>(0x10004), ws_bytes, nullptr) == NERF_EINVAL);
    int32_t cols[2] = {0, -2};
    EXPECT(nerf_table_bin_levels(&g, &rays, &ls, T, 40, cols, ws, ws_bytes, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_bin_fill_range(&g, &rays, &ls, T, 4, nullptr, ws, ws_bytes, 100, 16384, nullptr) == NERF_EINVAL);   // not whole chunks
    EXPECT(nerf_table_bin_fill_range(&g, &rays, &ls, T, 4, nullptr, ws, ws_bytes, 16384, 0, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_bin_fill_range(&g, &rays, &ls, T, 4, nullptr, ws, ws_bytes, 0, 524288 + 16384, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_bin_fill_range(&g, &rays, &ls, T, 4, nullptr, ws, ws_bytes, 16384, 16384, nullptr) == 0);          // empty range
    EXPECT(nerf_table_update_binned(&g, &ls, 524288, ws, ws_bytes, T, T, nullptr, 1e-2f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_update_binned(&g, &ls, 524288, ws, ws_bytes, T, T, T, 1e-2f, .9f, .99f, 1e-15f, 0.f, 0, 1.f, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_update_binned(&g, &ls, 524288, ws, ws_bytes, fake<float>(0x1004), T, T, 1e-2f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_update_binned(&g, &ls, 524288, ws, ws_bytes - 1, T, T, T, 1e-2f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_update_binned_f16(&g, &ls, 524288, ws, ws_bytes, T, T, T, 1e-2f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, nullptr, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_table_update_binned_f16(&g, &ls, 524288, ws, ws_bytes - 1, T, T, T, 1e-2f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, nullptr, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_hashgrid_encode_levels(&g, T, &rays, nullptr, T, 4, nullptr) != 0);
    ls.level[1] = 25;
    EXPECT(nerf_hashgrid_scatter_levels(&g, &rays, &ls, T, 4, T, nullptr) != 0);
    NerfColMap map = {};
    EXPECT(nerf_feature_shuffle(T, T, 10, NERF_MAX_FEATURE_COLS + 1, &map, 0, nullptr) == NERF_EINVAL);
    EXPECT(nerf_feature_shuffle(T, T, 10, 40, &map, 2, nullptr) == NERF_EINVAL);
    EXPECT(nerf_feature_shuffle(T, T, 10, 40, nullptr, 0, nullptr) == NERF_EINVAL);

    // ---- per-ray kernels ----
    int32_t steps4[4] = {256, 128, 64, 32};
    std::vector<int32_t> many(NERF_MAX_SEGMENTS + 1, 1);
    int64_t* I = fake<int64_t>();
    EXPECT(nerf_raygen(I, I, T, T, 0, T, T, nullptr) == 0);
    EXPECT(nerf_raygen(I, I, T, T, -1, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_raygen(nullptr, I, T, T, 8, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_sample_points(T, T, T, 64, 8, 0, 2.f, 2.5f, T, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_contract(nullptr, 8, 2.f, 2.5f, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_coarse_weights(T, T, 0, 8, NERF_MAX_SAMPLES + 1, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_coarse_weights(T, T, 0, 0, 64, T, nullptr) == 0);
    EXPECT(nerf_reweight(T, 8, 480, 0.2f, steps4, 3, T, nullptr) == NERF_EINVAL);                 // sum(steps) != n_samples
    EXPECT(nerf_reweight(T, 8, NERF_MAX_SEGMENTS + 1, 0.2f, many.data(), NERF_MAX_SEGMENTS + 1, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_reweight(T, 8, 480, 0.2f, nullptr, 4, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_sample_pdf(T, 0, T, T, T, 8, 3, 128, T, T, nullptr, nullptr) == NERF_EINVAL);     // needs >= 4 coarse samples
    EXPECT(nerf_sample_pdf(T, 0, T, T, T, 8, 64, 0, T, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_coarse_to_fine(T, T, 0, nullptr, T, 8, 480, 128, 0.2f, steps4, 4, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_coarse_to_fine_rng(T, T, 0, 1, 2, T, 8, 481, 128, 0.2f, steps4, 4, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_coarse_to_fine_rng(T, T, 0, 1, 2, T, 0, 480, 128, 0.2f, steps4, 4, T, T, nullptr) == 0);
    EXPECT(nerf_coarse_to_fine_rng_at(T, T, 0, 1, 2, -1, T, 8, 480, 128, 0.2f, steps4, 4, T, T, nullptr) == NERF_EINVAL);   // first_ray < 0
    EXPECT(nerf_coarse_to_fine_rng_at(T, T, 0, 1, 2, 100, T, 0, 480, 128, 0.2f, steps4, 4, T, T, nullptr) == 0);
    EXPECT(nerf_uniform_noise(1, 2, -1, T, nullptr) == NERF_EINVAL && nerf_uniform_noise(1, 2, 0, nullptr, nullptr) == 0);
    EXPECT(nerf_composite_fwd(T, T, T, T, T, 8, 0, 0.f, 1.f, T, T, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_composite_fwd(T, T, T, T, T, 8, 128, 0.f, 1.f, T, T, nullptr, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_composite_bwd(T, T, T, T, nullptr, T, 8, 128, T, T, nullptr) == NERF_EINVAL);
    double* D = fake<double>();
    EXPECT(nerf_losses(T, T, T, nullptr, 8, 128, 1e-3f, 1e-1f, 1.f, 1, D, T, T, nullptr) == NERF_EINVAL);   // weights without zlog_s
    EXPECT(nerf_losses(T, T, T, T, 0, 128, 1e-3f, 1e-1f, 1.f, 1, D, T, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_composite_loss_fwd(T, T, T, T, T, T, 8, 128, 0.f, 1.f, 1e-3f, 1e-1f, T, T, T, T, fake<double>(0x1004), T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_composite_loss_bwd(T, T, T, T, T, T, nullptr, 8, 513, 0.f, 1.f, 1e-3f, 1e-1f, T, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_composite_loss_bwd(T, T, T, T, T, T, nullptr, 8, 128, 0.f, 1.f, 1e-3f, 1e-1f, T, T, fake<double>(0x1004), nullptr) == NERF_EINVAL);
    EXPECT(nerf_cum_weights_thresh(T, 0, 128, 0.4f, T, nullptr) == 0);
    EXPECT(nerf_cum_weights_thresh(nullptr, 8, 128, 0.4f, T, nullptr) == NERF_EINVAL);

    // ---- batch sampling, optimizer, generic MLP ----
    uint8_t* U = fake<uint8_t>();
    EXPECT(nerf_sample_batch(U, nullptr, T, T, 100, 800, 800, 2, 4096, 1, 2, I, I, I, T, T, T, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_sample_batch(U, nullptr, T, T, 0, 800, 800, 4, 4096, 1, 2, I, I, I, T, T, T, T, nullptr, nullptr) == NERF_EINVAL);
    EXPECT(nerf_sample_batch(U, nullptr, T, T, 100, 800, 800, 4, 4096, 1, 2, I, I, I, T, T, T, T, T, nullptr) == NERF_EINVAL);   // depth without a store
    EXPECT(nerf_adam_step(T, T, T, T, -1, 1e-3f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, 1, nullptr) == NERF_EINVAL);
    EXPECT(nerf_adam_step(T, T, T, T, 16, 1e-3f, .9f, .99f, 1e-15f, 0.f, 0, 1.f, 1, nullptr) == NERF_EINVAL);
    EXPECT(nerf_adam_step(fake<float>(0x1004), T, T, T, 16, 1e-3f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, 1, nullptr) == NERF_EINVAL);
    EXPECT(nerf_adam_step(T, T, T, T, 0, 1e-3f, .9f, .99f, 1e-15f, 0.f, 1, 1.f, 1, nullptr) == 0);
    int32_t dims[4] = {79, 64, 64, 3}, wide[3] = {40, 129, 16};
    EXPECT(nerf_mlp_bwd_workspace_floats(dims, 3, 1000) == 1000 * (64 + 64 + 3 + 3 * 128));
    EXPECT(nerf_mlp_fwd(T, wide, 2, T, 10, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_mlp_fwd(T, dims, 9, T, 10, T, nullptr) == NERF_EINVAL);
    EXPECT(nerf_mlp_bwd(T, dims, 3, T, T, 1000, T, T, T, 10, nullptr) == NERF_EINVAL);            // workspace too small
    EXPECT(nerf_mlp_bwd_noalloc(T, dims, 3, T, T, 20000, T, T, nullptr) == NERF_EINVAL);          // beyond the built-in buffer
    EXPECT(strstr(nerf_last_error(), "at most") != nullptr);
    EXPECT(nerf_mlp_bwd_noalloc(T, wide, 2, T, T, 10, T, T, nullptr) == NERF_EINVAL);

    printf(g_fail ? "%d check(s) failed\n" : "all argument checks passed under the sanitizers%.0d\n", g_fail);
    return g_fail ? 1 : 0;
}
