// Error reporting, ABI version and host-side configuration helpers of libnerf_hip.so.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nerf_hip.h"

static thread_local char g_err[512] = "";

extern "C" __attribute__((visibility("hidden"))) void nerf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int nerf_abi_version(void) { return NERF_ABI_VERSION; }
extern "C" const char* nerf_last_error(void) { return g_err; }

// Level table of the multiresolution hash grid (tiny-cuda-nn HashGrid as configured at
// nets.py:39-50): scale_l = 2^(l*log2(per_level_scale)) * base - 1 (fp32), res = ceil(scale)+1,
// entries = min(align8(res^3), 2^T); a level is hashed iff res^3 does not fit.
extern "C" int64_t nerf_grid_cfg_init(NerfGridCfg* cfg, int n_levels, int n_features, int log2_hashmap_size,
                                      int base_resolution, float per_level_scale) {
    if (!cfg || n_levels < 1 || n_levels > NERF_MAX_LEVELS || log2_hashmap_size < 3 || log2_hashmap_size > 30 ||
        (n_features != 1 && n_features != 2 && n_features != 4) || base_resolution < 1 || !(per_level_scale >= 1.f)) {
        nerf_set_error("nerf_grid_cfg_init: invalid configuration");
        return NERF_EINVAL;
    }
    memset(cfg, 0, sizeof(*cfg));
    cfg->n_levels = n_levels;
    cfg->n_features = n_features;
    cfg->log2_hashmap_size = log2_hashmap_size;
    cfg->base_resolution = base_resolution;
    const float log2_pls = (float)log2((double)per_level_scale);
    uint64_t offset = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float expo = (float)l * log2_pls;
        const float scale = (float)(exp2((double)expo) * (double)base_resolution - 1.0);
        uint64_t res = (uint64_t)ceilf(scale) + 1;
        uint64_t full = res * res * res;
        uint64_t entries = ((full + 7) / 8) * 8;
        const uint64_t cap = 1ull << log2_hashmap_size;
        if (entries > cap) entries = cap;
        cfg->scale[l] = scale;
        cfg->res[l] = (uint32_t)res;
        cfg->entries[l] = (uint32_t)entries;
        cfg->hashed[l] = full > entries ? 1u : 0u;
        cfg->offset[l] = offset;
        offset += entries;
    }
    cfg->total_entries = offset;
    return (int64_t)offset;
}

extern "C" int64_t nerf_sigma_param_count(const NerfGridCfg* grid, const NerfNetCfg* net) {
    if (!grid || !net) return NERF_EINVAL;
    int64_t in = (int64_t)grid->n_levels * grid->n_features, h = net->hidden_dim;
    return in * h + (int64_t)(net->sigma_layers - 2) * h * h + h * (1 + net->geo_feat_dim);
}

extern "C" int64_t nerf_color_param_count(const NerfNetCfg* net) {
    if (!net) return NERF_EINVAL;
    int64_t in = net->geo_feat_dim + (int64_t)net->sh_degree * net->sh_degree + net->latent_dim, h = net->hidden_dim;
    return in * h + (int64_t)(net->color_layers - 2) * h * h + h * 3;
}
