// Shared host-side helpers for libnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nerf_hip.h"

// (library-internal: hidden, so that the .so exports exactly what include/nerf_hip.h declares)
extern "C" __attribute__((visibility("hidden"))) void nerf_set_error(const char* fmt, ...);

#define NERF_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            nerf_set_error("%s: %s", __func__, msg);              \
            return NERF_EINVAL;                                   \
        }                                                         \
    } while (0)

#define NERF_UNSUPPORTED(cond, msg)                               \
    do {                                                          \
        if (cond) {                                               \
            nerf_set_error("%s: unsupported: %s", __func__, msg); \
            return NERF_EUNSUPPORTED;                             \
        }                                                         \
    } while (0)

static inline int nerf_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        nerf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel, device and size instead of on every launch: `cache` is
// a static of the call site (one per kernel instantiation) holding the largest size set so far on every device.
struct NerfLdsAttrCache { int bytes[16]; };
static inline int nerf_ensure_dynamic_lds(const void* kern, int bytes, NerfLdsAttrCache* cache, const char* what) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev >= 0 && cache->bytes[dev] >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        nerf_set_error("%s: LDS attribute: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    if (dev >= 0) cache->bytes[dev] = bytes;
    return 0;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;
constexpr int kNumCU = 256;
