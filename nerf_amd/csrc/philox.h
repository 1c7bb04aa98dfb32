// Philox4x32-10 (Salmon et al. 2011): the counter-based generator behind the device-side draws of this library
// (nerf_sample_batch, nerf_coarse_to_fine_rng, nerf_uniform_noise): every value is a pure function of (key, counter).
#pragma once
#include <stdint.h>

namespace {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x & 0xFFFFFFu) * (1.0f / 16777216.0f); }   // [0, 1), 24 bits

// element e of the uniform [0,1) stream (seed, counter): Philox block e / 4, word e % 4 (24 random bits)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t counter, uint64_t e) {
    const uint64_t blk = e >> 2;
    const U4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)counter, (uint32_t)(counter >> 32),
                               (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w = (e & 3) == 0 ? r.x : ((e & 3) == 1 ? r.y : ((e & 3) == 2 ? r.z : r.w));
    return u01(w);
}

// elements e and e + 1 of the same stream: one Philox block when both lie in it (e even), else two
__device__ __forceinline__ void philox_uniform2(uint64_t seed, uint64_t counter, uint64_t e, float& a, float& b) {
    if ((e & 1) == 0) {
        const uint64_t blk = e >> 2;
        const U4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)counter, (uint32_t)(counter >> 32),
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
        const bool upper = (e & 2) != 0;
        a = u01(upper ? r.z : r.x);
        b = u01(upper ? r.w : r.y);
    } else {
        a = philox_uniform(seed, counter, e);
        b = philox_uniform(seed, counter, e + 1);
    }
}

}  // namespace