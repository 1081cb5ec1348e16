// Philox4x32-10 (Salmon et al., SC'11) -- device side of the stream defined in oracle/philox.py:
//   key = (seed lo, seed hi); counter = (index lo, index hi, block, stream);
//   u_i = ((w_i >> 8) + 0.5) * 2^-24;  (n0,n1) = BoxMuller(u0,u1), (n2,n3) = BoxMuller(u2,u3);
//   eps[index, 4*block + j] = n_j.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct Philox4 { uint32_t w[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    Philox4 out;
    out.w[0] = c0; out.w[1] = c1; out.w[2] = c2; out.w[3] = c3;
    return out;
}

__device__ __forceinline__ float philox_u01(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four standard normals for (index, block)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t index, uint32_t block, uint32_t stream,
                                               float (&n)[4]) {
    Philox4 p = philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), block, stream, (uint32_t)seed,
                              (uint32_t)(seed >> 32));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = philox_u01(p.w[2 * h]), u2 = philox_u01(p.w[2 * h + 1]);
        const float r = sqrtf(-2.0f * logf(u1));
        float s, c;
        sincosf(6.283185307179586f * u2, &s, &c);
        n[2 * h] = r * c;
        n[2 * h + 1] = r * s;
    }
}
