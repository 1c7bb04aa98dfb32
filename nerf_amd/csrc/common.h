// Shared host-side helpers for libnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nerf_hip.h"

extern "C" void nerf_set_error(const char* fmt, ...);

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

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;
constexpr int kNumCU = 256;
