// Training-batch sampling on the GPU in ONE launch: what CameraGeometryLoader.get_random_batch does on the host with
// three torch.randint calls, fancy indexing of the image / calibration stores, the uint8 -> float ground-truth blend and
// a host-to-device copy per step (loaders/camera_geometry_loader.py:68-88, 91-117).  At a 4.5 ms training step the
// dozen small ATen kernels of the torch formulation of the same thing cost ~75 us of launches on the critical path.
//
// Randomness: Philox4x32-10 (Salmon et al. 2011), key = seed, counter = (ray index, batch counter): every ray's draws
// are a pure function of (seed, batch counter, ray index), so a batch is reproducible and independent of the launch
// shape.  Ray r takes image n = mulhi(x0, N), row h = mulhi(x1, H), column w = mulhi(x2, W); the batch's background
// colour (ONE colour per batch, camera_geometry_loader.py:72) comes from counter index `batch`.
#pragma clang fp contract(off)

#include "common.h"
#include "philox.h"

namespace {

__global__ __launch_bounds__(256) void sample_batch_kernel(const uint8_t* __restrict__ images, const float* __restrict__ depths,
                                                           const float* __restrict__ K_all, const float* __restrict__ E_all,
                                                           int64_t n_images, int H, int W, int channels, int64_t batch,
                                                           uint64_t seed, uint64_t counter, int64_t* __restrict__ n_out,
                                                           int64_t* __restrict__ h_out, int64_t* __restrict__ w_out,
                                                           float* __restrict__ K_out, float* __restrict__ E_out,
                                                           float* __restrict__ gt_out, float* __restrict__ bg_out,
                                                           float* __restrict__ depth_out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t c2 = (uint32_t)counter, c3 = (uint32_t)(counter >> 32);
    const U4 b = philox4x32_10((uint32_t)batch, (uint32_t)((uint64_t)batch >> 32), c2, c3, k0, k1);
    const float bg[3] = {u01(b.x), u01(b.y), u01(b.z)};
    if (r == 0) { bg_out[0] = bg[0]; bg_out[1] = bg[1]; bg_out[2] = bg[2]; }
    if (r >= batch) return;
    const U4 x = philox4x32_10((uint32_t)r, (uint32_t)((uint64_t)r >> 32), c2, c3, k0, k1);
    const int64_t n = (int64_t)(((uint64_t)x.x * (uint64_t)n_images) >> 32);
    const int64_t h = (int64_t)(((uint64_t)x.y * (uint64_t)H) >> 32);
    const int64_t w = (int64_t)(((uint64_t)x.z * (uint64_t)W) >> 32);
    n_out[r] = n; h_out[r] = h; w_out[r] = w;
    const float* Ks = K_all + n * 9;
    const float* Es = E_all + n * 16;
#pragma unroll
    for (int i = 0; i < 9; ++i) K_out[r * 9 + i] = Ks[i];
    const float4* E4 = (const float4*)Es;
#pragma unroll
    for (int i = 0; i < 4; ++i) ((float4*)(E_out + r * 16))[i] = E4[i];
    const int64_t pix = (n * H + h) * W + w;
    if (images) {
        // camera_geometry_loader.py:76-86: uint8 / 255, alpha blend over the batch's background colour
        const uint8_t* px = images + pix * channels;
        const float c0 = (float)px[0] / 255.f, c1 = (float)px[1] / 255.f, c2f = (float)px[2] / 255.f;
        if (channels == 4) {
            const float a = (float)px[3] / 255.f;
            gt_out[r * 3 + 0] = c0 * a + bg[0] * (1.f - a);
            gt_out[r * 3 + 1] = c1 * a + bg[1] * (1.f - a);
            gt_out[r * 3 + 2] = c2f * a + bg[2] * (1.f - a);
        } else {
            gt_out[r * 3 + 0] = c0; gt_out[r * 3 + 1] = c1; gt_out[r * 3 + 2] = c2f;
        }
    } else {    // no image store (throughput runs): the loader's deterministic pseudo-colours
        gt_out[r * 3 + 0] = (float)(h % 256) / 255.f;
        gt_out[r * 3 + 1] = (float)(w % 256) / 255.f;
        gt_out[r * 3 + 2] = (float)((h + w) % 256) / 255.f;
    }
    if (depth_out && depths) depth_out[r] = depths[pix];
}

}  // namespace

extern "C" int nerf_sample_batch(const uint8_t* images, const float* depths, const float* intrinsics, const float* extrinsics,
                                 int64_t n_images, int H, int W, int channels, int64_t batch, uint64_t seed, uint64_t counter,
                                 int64_t* n, int64_t* h, int64_t* w, float* K, float* E, float* rgb_gt, float* color_bg,
                                 float* depth, void* stream) {
    NERF_REQUIRE(intrinsics && extrinsics && n && h && w && K && E && rgb_gt && color_bg, "null pointer");
    NERF_REQUIRE(n_images > 0 && n_images < (1ll << 31) && H > 0 && W > 0 && batch >= 0, "bad sizes");
    NERF_REQUIRE(!images || channels == 3 || channels == 4, "images must be RGB or RGBA uint8");
    NERF_REQUIRE((((uintptr_t)extrinsics | (uintptr_t)E) & 15) == 0, "extrinsics must be 16-byte aligned");
    NERF_REQUIRE(!depth || depths, "depth output without a depth store");
    const unsigned nb = (unsigned)ceil_div64(batch > 0 ? batch : 1, 256);
    sample_batch_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(images, depths, intrinsics, extrinsics, n_images, H, W, channels,
                                                             batch, seed, counter, n, h, w, K, E, rgb_gt, color_bg, depth);
    return nerf_launch_status("nerf_sample_batch");
}
