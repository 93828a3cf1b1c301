// Counter-based Gaussian noise for the kernels that draw it on the device (device code only).  Every kernel that
// draws noise goes through these two functions, so that equal (seed, counter) pairs give equal draws everywhere.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace danra {

// Philox4x32-10 (Salmon et al.), counter = (element quad, step offset), key = seed.
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                        uint32_t* r) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// Contraction off here too: every kernel that draws noise must round the draws identically.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
#pragma clang fp contract(off)
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

}  // namespace danra
