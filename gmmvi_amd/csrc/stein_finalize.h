// stein_finalize_component: the last step of the Stein estimate for ONE component, executed by all threads of a workgroup of
// any size (a multiple of 64) -- shared by the stand-alone kernel (stein.hip) and by the update kernel that runs it as its
// prologue on the single-call path (update_kl.hip), so that both routes perform the same arithmetic in the same order:
// every element is summed by exactly one thread over the partials r = 0, 1, ... in eight interleaved chains.
#pragma once
#include "common.h"
#include "wave_reduce.h"

struct SteinSlab { const float* part; const float* part_m; int R; };      // [K][R][(D+1)^2] partial moment matrices, [K][R] maxima

inline size_t stein_finalize_lds_floats(int dp, int D, int R) {
    const size_t D1 = (size_t)D + 1, T = (size_t)dp * (dp - 1) / 2;
    const size_t work = dp >= GMMVI_MFMA_DENSITY_FROM_DP ? (size_t)dp * dp + (size_t)D * D : (size_t)dp + 2 * T;
    return D1 * D1 + (size_t)R + work;
}

// Sums the R per-range partials of A_k (each referred to its own maximum m_r) in fixed order, applies
// Sigma_k^-1 = L^-T L^-1 from the right to the D x D block, normalises, symmetrises, negates.  Small blocks (no L^-1
// fragments): row i of the result by substitution, h' L^T = t ascending over the rows of L, then h L = h' descending over its
// columns (L from the packed block: 1/diag, rows, columns, staged in LDS; the row in registers, loops unrolled for the padded
// dimension).  Blocks with fragments (DP >= 32): two triangular products with the explicit inverse on all threads.
// A: LDS, stein_finalize_lds_floats(DP, D, R) floats.  Ends with the results in global memory (no trailing barrier).
// A[(D+1)^2] <- sum over the R partials of component k, each referred to its own maximum m_r, rescaled to the common maximum
// (returned).  Every element is summed by exactly one thread over r = 0, 1, ... in eight interleaved chains, so the result
// does not depend on the number of threads.  scale_r: R floats of LDS.  Ends with a barrier.
__device__ __forceinline__ float stein_slab_sum(float* A, float* scale_r, int k, int D, int R, const float* __restrict__ part,
                                                const float* __restrict__ part_m) {
    const int nth = blockDim.x;
    const int D1 = D + 1;
    float M = -3.0e38f;
    for (int r = threadIdx.x & 63; r < R; r += 64) M = fmaxf(M, part_m[(size_t)k * R + r]);
    M = gmmvi_wave_max(M);
    for (int r = threadIdx.x; r < R; r += nth) scale_r[r] = __expf(part_m[(size_t)k * R + r] - M);
    __syncthreads();
    const float* pk = part + (size_t)k * R * (size_t)(D1 * D1);
    for (int e = threadIdx.x; e < D1 * D1; e += nth) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = 0;
        for (; r + 7 < R; r += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = fmaf(pk[(size_t)(r + u) * (D1 * D1) + e], scale_r[r + u], v[u]);
        }
        if (r < R) {
            // the last (partial) group of eight: all eight loads unconditional (clamped index, weight 0 beyond R) -- a run-time
            // trip count makes the compiler wait for every load separately (a few partials only at D = 50: 17 us for the sum)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = min(r + u, R - 1);
                const float sc = (r + u < R) ? scale_r[rr] : 0.f;
                v[u] = fmaf(pk[(size_t)rr * (D1 * D1) + e], sc, v[u]);
            }
        }
        A[e] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    __syncthreads();
    return M;
}

// normalisation of the summed moments: 1 / sum e (self-normalised; 0 over an empty set), exp(M) / N (plain importance
// weights), 1 / sum e = exp(M) / n_own for plain weights over the component's own samples
__device__ __forceinline__ float stein_moment_scale(float se, float M, int N, int flags) {
    const bool snis = (flags & GMMVI_SELF_NORMALIZED) != 0, own = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    return snis ? (se > 0.f ? 1.f / se : 0.f) : (own ? 1.f / se : __expf(M) / (float)N);
}

template <int DP>
__device__ __forceinline__ void stein_finalize_component(float* A, int k, int D, int R, int N, int flags,
                                                         const float* __restrict__ part, const float* __restrict__ part_m,
                                                         float* H_neg, float* g_neg, const float* __restrict__ packed) {
    using PK = Pack<DP>;
    const int nth = blockDim.x;
    const int D1 = D + 1;
    float* scale_r = A + D1 * D1;
    float* Wk = scale_r + R;                                               // work area
    const float M = stein_slab_sum(A, scale_r, k, D, R, part, part_m);
    if constexpr (PK::FRAGS) {
        // blocks that carry L^-1 (operand fragments, common.h): Sigma^-1 from the right as two triangular products spread over
        // the whole workgroup -- W = T L^-T, then W L^-1 -- instead of two substitution chains on D threads
        const float* Pk = packed + (size_t)k * PK::STRIDE;
        float* Li = Wk;                  // dense L^-1 [DP][DP] , then W [D][D]
        float* W = Wk + DP * DP;
        for (int e = threadIdx.x; e < DP * DP; e += nth) Li[e] = 0.f;
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * PK::NF; e += nth) {
            const int f = e >> 6, l = e & 63;
            int mt = 0, rem = f;
            for (;; ++mt) { const int nfm = PK::nf(mt); if (rem < nfm) break; rem -= nfm; }
            const int row = 16 * mt + (l & 15), col = 4 * rem + (l >> 4);        // the fragments cover 16 MT x 4 KS >= DP x DP
            if (row < DP && col < DP) Li[row * DP + col] = Pk[PK::FWD + e];
        }
        __syncthreads();
        for (int o = threadIdx.x; o < D * D; o += nth) {
            const int a = o / D, i = o - a * D;
            float w = 0.f;
            for (int j = 0; j <= i; ++j) w = fmaf(A[a * D1 + j], Li[i * DP + j], w);
            W[o] = w;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < D * D; o += nth) {
            const int a = o / D, i2 = o - a * D;
            float h = 0.f;
            for (int i = i2; i < D; ++i) h = fmaf(W[a * D + i], Li[i * DP + i2], h);
            A[a * D1 + i2] = h;
        }
        __syncthreads();
    } else {
        const float* Pk = packed + (size_t)k * PK::STRIDE;
        float* Lc = Wk;                  // as [1/diag (DP) | columns (T) | rows (T)]
        for (int e = threadIdx.x; e < DP + 2 * PK::T; e += nth)
            Lc[e] = (e < DP) ? Pk[PK::RD + e] : (e < DP + PK::T ? Pk[PK::LCOL + (e - DP)] : Pk[PK::LROW + (e - DP - PK::T)]);
        __syncthreads();
        if (threadIdx.x < D) {
            float* row = A + threadIdx.x * D1;
            float h[DP];
#pragma unroll
            for (int j = 0; j < DP; ++j) h[j] = (j < D) ? row[j] : 0.f;
#pragma unroll
            for (int j = 0; j < DP; ++j) {                                  // h' L^T = t
                float t = h[j];
#pragma unroll
                for (int m = 0; m < j; ++m) t = fmaf(-h[m], Lc[DP + PK::T + PK::rowofs(j) + m], t);      // padding: L = 0
                h[j] = t * Lc[j];
            }
#pragma unroll
            for (int j = DP - 1; j >= 0; --j) {                             // h L = h'
                float t = h[j];
#pragma unroll
                for (int m = j + 1; m < DP; ++m) t = fmaf(-h[m], Lc[DP + PK::colofs(j) + (m - j - 1)], t);
                h[j] = t * Lc[j];
            }
#pragma unroll
            for (int j = 0; j < DP; ++j)
                if (j < D) row[j] = h[j];
        }
        __syncthreads();
    }
    const bool snis = (flags & GMMVI_SELF_NORMALIZED) != 0;
    // A[i][j] = sum e g_i y_j, A[i][D] = sum e g_i, A[D][D] = sum e.
    // plain importance weights: 1/N * sum exp(ld - bg) v   (ng_estimator.py:146-152), Hessian not symmetrised
    // with only_use_own_samples the expectation runs over the component's own samples only (get_rewards_for_comp,
    // ng_estimator.py:110-118: weights exp(0) = 1, divisor = their number): sum e = n_own exp(-M) => exp(M) / n_own = 1 / sum e
    // self-normalised weights over an EMPTY own-sample set: every reduce_sum of the reference runs over nothing and returns
    // zeros (ng_estimator.py:171-188), the plain branch divides by the set's length (NaN, a rejected update)
    const float se = A[D * D1 + D];
    const float scale = stein_moment_scale(se, M, N, flags);
    for (int e = threadIdx.x; e < D * D; e += nth) {
        const int i = e / D, j = e % D;
        const float v = snis ? 0.5f * (A[i * D1 + j] + A[j * D1 + i]) : A[i * D1 + j];
        H_neg[(size_t)k * D * D + e] = -v * scale;
    }
    for (int i = threadIdx.x; i < D; i += nth) g_neg[(size_t)k * D + i] = -A[i * D1 + D] * scale;
}

