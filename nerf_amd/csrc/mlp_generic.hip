// Stand-alone bias-free MLP backward: what autograd + tcnn.Network deliver when the reference's two networks are
// called as separate modules (nets.py:120,135,155).  NOT the hot path -- training goes through the fused field
// kernels -- but it completes the tinycudann-level boundary (Encoding fwd/bwd, Network fwd/bwd) for a maintainer who
// keeps the reference's own nets.py.  Plain, layer-by-layer, caller-provided workspace:
//   acts   [layer 1..n][rows][dims[layer]]   post-activation outputs of every layer (forward pass, recomputed here)
//   dz, g0, g1 [rows][128] each              dz of the current layer, ping-pong buffers for the gradient passed down
#include "common.h"

namespace {

constexpr int kMaxW = 128;
struct Dims9 { int v[9]; };

// one thread per row: forward through every layer, keeping all activations
__global__ void mlp_acts_kernel(const float* __restrict__ params, Dims9 d, int n_layers, const float* __restrict__ x,
                                int64_t rows, float* __restrict__ acts) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* in = x + r * d.v[0];
    const float* w = params;
    float* out = acts;
    for (int l = 0; l < n_layers; ++l) {
        const int ni = d.v[l], no = d.v[l + 1];
        float* o = out + r * no;
        for (int j = 0; j < no; ++j) {
            float acc = 0.f;
            for (int k = 0; k < ni; ++k) acc = fmaf(w[j * ni + k], in[k], acc);
            o[j] = (l + 1 < n_layers) ? fmaxf(acc, 0.f) : acc;
        }
        in = o;
        w += ni * no;
        out += rows * no;
    }
}

// dz = dout * relu'(act) (hidden layers), dh = dz * W   -- one thread per row
__global__ void mlp_bwd_layer_kernel(const float* __restrict__ w, int ni, int no, int is_last, const float* __restrict__ act,
                                     const float* __restrict__ dout, int64_t rows, float* __restrict__ dz,
                                     float* __restrict__ dh) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float z[kMaxW];
    for (int j = 0; j < no; ++j) {
        const float g = dout[r * no + j];
        z[j] = (is_last || act[r * no + j] > 0.f) ? g : 0.f;
        dz[r * no + j] = z[j];
    }
    if (!dh) return;
    for (int k = 0; k < ni; ++k) {
        float acc = 0.f;
        for (int j = 0; j < no; ++j) acc = fmaf(z[j], w[j * ni + k], acc);
        dh[r * ni + k] = acc;
    }
}

// dW[o][i] += sum_r dz[r][o] * h[r][i]: 16x16 threads own a 16x16 tile of dW, blockIdx.z walks row chunks
constexpr int kChunk = 2048;
__global__ __launch_bounds__(256) void mlp_dw_kernel(const float* __restrict__ dz, const float* __restrict__ h, int ni,
                                                     int no, int64_t rows, float* __restrict__ dw) {
    __shared__ float sz[64][17], sh[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int o0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
    const int64_t r0 = (int64_t)blockIdx.z * kChunk;
    const int64_t r1 = r0 + kChunk < rows ? r0 + kChunk : rows;
    float acc = 0.f;
    for (int64_t base = r0; base < r1; base += 64) {
        for (int t = threadIdx.x; t < 64 * 16; t += 256) {
            const int rr = t >> 4, c = t & 15;
            const int64_t r = base + rr;
            sz[rr][c] = (r < r1 && o0 + c < no) ? dz[r * no + o0 + c] : 0.f;
            sh[rr][c] = (r < r1 && i0 + c < ni) ? h[r * ni + i0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < 64; ++rr) acc = fmaf(sz[rr][ty], sh[rr][tx], acc);
        __syncthreads();
    }
    if (o0 + ty < no && i0 + tx < ni) atomicAdd(dw + (o0 + ty) * ni + i0 + tx, acc);
}

int check_dims(const int32_t* dims, int n_layers) {
    if (!dims || n_layers < 1 || n_layers > 8) return 1;
    for (int i = 0; i <= n_layers; ++i)
        if (dims[i] < 1 || dims[i] > kMaxW) return 1;
    return 0;
}

}  // namespace

extern "C" int64_t nerf_mlp_bwd_workspace_floats(const int32_t* dims, int n_layers, int64_t n_rows) {
    if (check_dims(dims, n_layers) || n_rows < 0) return -1;
    int64_t f = 0;
    for (int l = 1; l <= n_layers; ++l) f += n_rows * dims[l];
    return f + 3 * n_rows * kMaxW;
}

extern "C" int nerf_mlp_bwd(const float* params, const int32_t* dims, int n_layers, const float* x, const float* dout,
                            int64_t n_rows, float* dx, float* dparams, float* workspace, int64_t workspace_floats,
                            void* stream) {
    NERF_REQUIRE(params && x && dout && dparams && workspace, "null pointer");
    NERF_REQUIRE(!check_dims(dims, n_layers) && n_rows >= 0, "layer widths must be in [1,128], 1..8 layers");
    NERF_REQUIRE(workspace_floats >= nerf_mlp_bwd_workspace_floats(dims, n_layers, n_rows), "workspace too small");
    if (n_rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    Dims9 d = {};
    for (int i = 0; i <= n_layers; ++i) d.v[i] = dims[i];
    float* acts = workspace;
    int64_t acts_floats = 0;
    for (int l = 1; l <= n_layers; ++l) acts_floats += n_rows * dims[l];
    float* dz = workspace + acts_floats;
    float* g[2] = {dz + n_rows * kMaxW, dz + 2 * n_rows * kMaxW};
    const unsigned nb = (unsigned)ceil_div64(n_rows, 64);
    mlp_acts_kernel<<<nb, 64, 0, st>>>(params, d, n_layers, x, n_rows, acts);
    int64_t w_off[9], a_off[9];       // offsets of every layer's weights / activations
    int64_t wo = 0, ao = 0;
    for (int l = 0; l < n_layers; ++l) {
        w_off[l] = wo; a_off[l] = ao;
        wo += (int64_t)dims[l] * dims[l + 1];
        ao += n_rows * dims[l + 1];
    }
    const float* grad_in = dout;
    int cur = 0;
    for (int l = n_layers - 1; l >= 0; --l) {
        const int ni = dims[l], no = dims[l + 1];
        float* dh_l = (l == 0) ? dx : g[cur];             // dx nullable: the input gradient is optional
        mlp_bwd_layer_kernel<<<nb, 64, 0, st>>>(params + w_off[l], ni, no, l == n_layers - 1, acts + a_off[l], grad_in, n_rows,
                                                dz, dh_l);
        const float* h = (l == 0) ? x : acts + a_off[l - 1];
        const dim3 grid((unsigned)((ni + 15) / 16), (unsigned)((no + 15) / 16), (unsigned)ceil_div64(n_rows, kChunk));
        mlp_dw_kernel<<<grid, 256, 0, st>>>(dz, h, ni, no, n_rows, dparams + w_off[l]);
        grad_in = dh_l;
        cur ^= 1;
    }
    return nerf_launch_status("nerf_mlp_bwd");
}

// ---- the same call for callers that cannot allocate -------------------------------------------------------------------
// nerf_mlp_bwd needs a caller-provided workspace (activations of every layer + three [rows][128] gradient buffers).  A
// caller under HIP graph capture, in a tight serving loop or behind a harness that forbids hipMalloc has nowhere to get
// one from: this variant keeps the workspace in a file-scope device array instead.  Every GPU the library runs on holds
// its own copy of the array (18 MiB, until the process ends) at its own address; host code must never pass the array's
// NAME to a kernel (that is a host-side stand-in, not a device address): the address is looked up once per device with
// hipGetSymbolAddress and only that pointer is used.
namespace {
constexpr int64_t kScratchFloats = 4718592;       // 18 MiB: e.g. 9162 rows of the colour net (79 -> 64 -> 64 -> 3)
__device__ __align__(256) float scratch[kScratchFloats];
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};
}  // namespace

static int64_t noalloc_max_rows(const int32_t* dims, int n_layers) {
    int64_t per_row = 3 * kMaxW;
    for (int l = 1; l <= n_layers; ++l) per_row += dims[l];
    return kScratchFloats / per_row;
}

extern "C" int nerf_mlp_bwd_noalloc(const float* params, const int32_t* dims, int n_layers, const float* x, const float* dout,
                                    int64_t n_rows, float* dx, float* dparams, void* stream) {
    NERF_REQUIRE(!check_dims(dims, n_layers) && n_rows >= 0, "layer widths must be in [1,128], 1..8 layers");
    const int64_t need = nerf_mlp_bwd_workspace_floats(dims, n_layers, n_rows);
    if (need > kScratchFloats) {
        nerf_set_error("nerf_mlp_bwd_noalloc: %lld rows need %lld floats of scratch, the built-in buffer holds %lld (at most %lld "
                       "rows for these layer widths); use nerf_mlp_bwd with a workspace", (long long)n_rows, (long long)need,
                       (long long)kScratchFloats, (long long)noalloc_max_rows(dims, n_layers));
        return NERF_EINVAL;
    }
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        nerf_set_error("nerf_mlp_bwd_noalloc: hipGetDevice: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (dev < 0 || dev >= kMaxDevices) {
        nerf_set_error("nerf_mlp_bwd_noalloc: device index %d is outside the %d devices this entry point keeps a buffer "
                       "address for", dev, kMaxDevices);
        return NERF_EINVAL;
    }
    float* ws = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!ws) {      // first call on this device (it is the current one): this device's copy of the array
        e = hipGetSymbolAddress((void**)&ws, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ws) {
            nerf_set_error("nerf_mlp_bwd_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return e != hipSuccess ? (int)e : NERF_EINVAL;
        }
        __atomic_store_n(&g_scratch_ptr[dev], ws, __ATOMIC_RELEASE);
    }
    // The look-up above (and the lazy load of the code object behind it) is not stream-ordered and must not happen under
    // a HIP graph capture: a caller that captures makes ONE call per device beforehand -- n_rows = 0 is enough (pointers
    // may be NULL), it resolves the address and launches nothing.
    if (n_rows == 0) return 0;
    // (nothing in the workspace has to start at zero: the forward pass rewrites every activation, every layer rewrites dz
    // and the gradient ping-pong buffers before reading them)
    return nerf_mlp_bwd(params, dims, n_layers, x, dout, n_rows, dx, dparams, ws, kScratchFloats, stream);
}
