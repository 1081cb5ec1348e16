// One workgroup's share of component-ordered sampling x = mu_k + L_k eps (models/gmm.py:361-386, models/full_cov_gmm.py:36-39):
// the samples [256 chunk, 256 chunk + 256) of component k.  Device code shared by the sampling launch (sampling.hip) and by
// the density launches that carry the NEXT iteration's draw as extra workgroups (riders.h): any workgroup size >= 256 -- the
// (sample, Philox block) items and the stores are spread over all threads, the per-sample arithmetic runs on threads 0..255.
#pragma once
#include "common.h"
#include "philox.h"

// (mu_k, L_k) staged in LDS and read as broadcasts, eps and x in registers (DP = padded dimension, loops unrolled), the output
// tile leaves through LDS with coalesced stores.  LDS: D * D + D + 256 * (D | 1) floats at `sm`.
template <int DP>
__device__ __forceinline__ void sample_block(float* sm, int k, int chunk, int D, const float* __restrict__ means,
                                             const float* __restrict__ chols, const int32_t* __restrict__ offsets,
                                             uint64_t seed, uint64_t first_index, uint32_t stream_id,
                                             const float* __restrict__ eps_in, float* __restrict__ X,
                                             int32_t* __restrict__ mapping, int32_t mapping_base, int uniform_count) {
    const int nt = blockDim.x;
    // equal counts known to the caller (single-call iteration): no dependent load in front of everything else
    const int begin = uniform_count > 0 ? k * uniform_count : offsets[k];
    const int end = uniform_count > 0 ? begin + uniform_count : offsets[k + 1];
    const int base = begin + chunk * 256;
    if (base >= end) return;
    const int n_here = min(256, end - base);
    float* Ls = sm;                      // [D][D]
    float* mus = sm + D * D;             // [D]
    float* tile = mus + D;               // [256][ldx]
    const int ldx = D | 1;
    const int t = threadIdx.x;
    // (mu, L) are fetched into registers first and reach LDS after the random numbers are made: the loads (L2 round trips:
    // the previous iteration's update kernel wrote them on other CUs) overlap the Philox rounds
    constexpr int NL = (DP * DP + 255) / 256;
    float lreg[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) lreg[u] = (t < 256 && t + 256 * u < D * D) ? chols[(size_t)k * D * D + t + 256 * u] : 0.f;
    const float mreg = t < D ? means[(size_t)k * D + t] : 0.f;
    if (eps_in) {
        for (int e = t; e < n_here * D; e += nt) tile[(e / D) * ldx + (e % D)] = eps_in[(size_t)base * D + e];
    }
    const bool valid = t < n_here;
    if (!eps_in) {
        // the standard normals of the tile, four per Philox block (counter = sample index, block): the (sample, block) items are
        // spread over ALL threads -- a thread that made all of its sample's blocks itself spent 4 us of a 9 us launch in the
        // Philox rounds and the Box-Muller transforms (D = 20: five blocks a sample, 100 samples on 256 threads)
        constexpr int NB4 = (DP + 3) / 4;
        for (int item = t; item < n_here * NB4; item += nt) {
            const int smp = item / NB4, b = item - smp * NB4;
            float nn[4];
            philox_normal4(seed, first_index + (uint64_t)(base + smp), (uint32_t)b, stream_id, nn);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * b + j < D) tile[smp * ldx + 4 * b + j] = nn[j];
        }
    }
#pragma unroll
    for (int u = 0; u < NL; ++u)
        if (t < 256 && t + 256 * u < D * D) Ls[t + 256 * u] = lreg[u];
    if (t < D) mus[t] = mreg;
    __syncthreads();
    if constexpr (DP >= 32) {
        // X = mu + eps L^T as a matrix-core product (v_mfma_f32_16x16x4_f32): a wave owns 64 samples (four 16-row tiles of
        // eps), A = eps[16 mt + i][4 s + kk], B = L^T: B[kk][j] = L[16 nt + j][4 s + kk], k-steps beyond the diagonal block of
        // the lower-triangular L skipped.  (One lane per sample with the row of L broadcast from LDS is a chain of D^2 / 2
        // dependent multiply-adds on 100 of the 256 threads: 16 of the 21 us of the launch at D = 50.)
        typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
        constexpr int NT = (DP + 15) / 16, KS = (DP + 3) / 4;
        const int wave = t >> 6, lane = t & 63, i16 = lane & 15, kk = lane >> 4;
        const int s0 = 64 * wave;                      // first sample of this wave
        if (s0 < n_here) {
            sc_f32x4 acc[4][NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = sc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < KS; ++s4) {
                const int kcol = 4 * s4 + kk;
                float a[4], b[NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int smp = s0 + 16 * mt + i16;
                    a[mt] = (smp < n_here && kcol < D) ? tile[smp * ldx + kcol] : 0.f;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int row = 16 * nt + i16;
                    b[nt] = (row < D && kcol <= row) ? Ls[row * D + kcol] : 0.f;      // lower triangle only (and kcol < D)
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (4 * s4 > 16 * nt + 15) continue;                                // this k-step lies above the diagonal block
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
            }
            // every eps value of the wave's rows has been read (by this wave only): the results overwrite them in place.
            // D[i][j]: lane l, register r -> sample 16 mt + 4 (l >> 4) + r, dimension 16 nt + (l & 15)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int smp = s0 + 16 * mt + 4 * kk + r, dim = 16 * nt + i16;
                        if (smp < n_here && dim < D) tile[smp * ldx + dim] = acc[mt][nt][r] + mus[dim];
                    }
        }
        if (valid && mapping) mapping[base + t] = k + mapping_base;
    } else {
    float eps[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) eps[i] = (valid && i < D) ? tile[t * ldx + i] : 0.f;
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                float v = mus[i];
#pragma unroll
                for (int j = 0; j <= i; ++j) v = fmaf(Ls[i * D + j], eps[j], v);
                tile[t * ldx + i] = v;
            }
        }
        if (mapping) mapping[base + t] = k + mapping_base;
    }
    }
    __syncthreads();
    for (int e = t; e < n_here * D; e += nt) X[(size_t)base * D + e] = tile[(e / D) * ldx + (e % D)];
}
