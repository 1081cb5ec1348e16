// Importance-weighted Stein estimate of the per-component expected gradient / Hessian
// (gmmvi_modules/ng_estimator.py:204-263, :171-188 self-normalised, :154-169 plain importance weights).
//
// For component k:  A_k = sum_n [g_n; 1] (e_kn [y_kn; 1])^T,  e_kn = exp(ld[k,n] - bg[n] - m),  g_n = grad log p~ - grad log q,
// y_kn = Sigma_k^-1 (x_n - mu_k).  The (D+1)x(D+1) matrix A_k carries sum e g y^T, sum e g (last column) and sum e
// (corner), so one contraction over the samples yields the Hessian, the gradient and the normaliser.
//
// Two kernels produce per-(component, 256-sample tile) partials of A_k; grid = (tiles, component chunks),
// 4 waves per workgroup, the x and g tiles of the workgroup fetched ONCE with coalesced 16-byte loads and staged through LDS
// (per-lane row loads of a row-major [N, D] array were the dominant stall of the first version):
//   * stein_wc_kernel (D <= 24, the production path): wave = component.  Every lane keeps its x row of the four 64-sample
//     sub-tiles in VGPRs; wave w walks the components w, w + 4, ... of the chunk and for each of them substitutes the four
//     sub-tiles (generated hand-scheduled scalar-fed asm for D = 10 / 20, subst_asm_gen.h), writes e * [y; 1] rows to its
//     private LDS tile and contracts with v_mfma_f32_32x32x2 (16x16x4 for D + 1 <= 16), rescaling the accumulators online
//     by the running maximum.  No cross-wave merge, no block barrier in the component loop.
//   * stein_partial_kernel (D > 24): wave = 64 samples, [g;1] A-fragments in VGPRs reused for every component of the chunk,
//     the four waves merged through LDS per component (own maximum per wave, fixed summation order).
// stein_finalize sums the slab in fixed order (bitwise reproducible), normalises, symmetrises and negates.
#include "common.h"
#include "blocked.h"
#include "wave_reduce.h"
#include "subst_asm_gen.h"
#include <cstdlib>
#include <type_traits>

#ifndef GMMVI_STEIN_BLOCKED_FROM_DP
#define GMMVI_STEIN_BLOCKED_FROM_DP 64     // padded dimension from which gmmvi_stein takes the blocked contractions (see below)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DP>
__device__ __forceinline__ void forward_subst_s(const float* __restrict__ P, const float (&x)[DP], float (&z)[DP]) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        float t = x[i] - P[PK::MU + i];
#pragma unroll
        for (int j = 0; j < i; ++j) t = fmaf(-P[PK::LROW + PK::rowofs(i) + j], z[j], t);
        z[i] = t * P[PK::RD + i];
    }
}

template <int DP>
__device__ __forceinline__ void backward_subst_s(const float* __restrict__ P, const float (&z)[DP], float (&y)[DP]) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = DP - 1; i >= 0; --i) {
        float t = z[i];
#pragma unroll
        for (int j = i + 1; j < DP; ++j) t = fmaf(-P[PK::LCOL + PK::colofs(i) + (j - i - 1)], y[j], t);
        y[i] = t * P[PK::RD + i];
    }
}

__device__ __forceinline__ float wave_max(float v) { return gmmvi_wave_max(v); }

template <int DP, int NB>
__global__ __launch_bounds__(256, 2) void stein_partial_kernel(int K, int D, int chunk, const float* __restrict__ packed,
                                                            const float* __restrict__ X, const float* __restrict__ TG,
                                                            const float* __restrict__ QG, int N,
                                                            const float* __restrict__ ld, const float* __restrict__ bg,
                                                            const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                            float* __restrict__ part, float* __restrict__ part_m) {
    using PK = Pack<DP>;
    constexpr int W = 32 * NB;         // padded width of [g;1] and [y;1]
    constexpr int LDW = W + 1;         // LDS row stride of the MFMA operand tiles
    extern __shared__ float sm[];
    __shared__ float sm_m[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int n_tiles = gridDim.x;
    const int D1 = D + 1;
    const int n0 = tile * 256;
    const int n_here = min(256, N - n0);
    // One [256][LDW] LDS image, used in turn as: staging of the x tile, the [g;1] rows (read once into the A fragments), then
    // the four waves' e*[y;1] tiles / the merge scratch (a separate G image doubled the LDS to 133 KB at W = 64: one
    // workgroup per CU).  Together with the (256, 2) launch bound two workgroups share a CU.
    float* Ys = sm;                                   // 4 x [64][LDW] rows e*[y;1;0...]; also staging / merge scratch
    float* Gs = sm;                                   // [256][LDW]  rows [g;1;0...] (until the A fragments are loaded)
    float* Yw = Ys + wave * 64 * LDW;
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;

    // ---- stage the x tile (coalesced) through Ys, keep this lane's row in registers -------------------------------
    const int ldx = D | 1;                            // odd stride: conflict-free row reads
    for (int e = tid; e < n_here * D; e += 256) Ys[(e / D) * ldx + (e % D)] = X[(size_t)n0 * D + e];
    __syncthreads();
    const int row = wave * 64 + lane;
    const bool valid = row < n_here;
    const int n = n0 + row;
    float x[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) x[i] = (valid && i < D) ? Ys[row * ldx + i] : 0.f;
    __syncthreads();
    // ---- stage the g tile (coalesced) as [g;1] rows, zero padding; pre-zero the padded columns of the Y tiles ----------
    for (int e = tid; e < 256 * W; e += 256) {
        const int r = e / W, c = e % W;
        float v = 0.f;
        if (r < n_here) {
            const size_t gi = (size_t)(n0 + r) * D + c;
            v = (c < D) ? TG[gi] - QG[gi] : (c == D ? 1.f : 0.f);       // g = grad log p~ - grad log q (:248)
        }
        Gs[r * LDW + c] = v;
    }
    __syncthreads();
    // A fragments of this wave's 64 samples: lane (col = l & 31, half = l >> 5), step s -> Gs[2s + half][col]
    const int col = lane & 31, half = lane >> 5;
    float af[NB][32];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int s = 0; s < 32; ++s) af[a][s] = Gs[(wave * 64 + 2 * s + half) * LDW + 32 * a + col];
    __syncthreads();                                  // the G rows are in registers: the image becomes the Y tiles
    for (int c = D1; c < W; ++c) Yw[lane * LDW + c] = 0.f;      // padded columns stay zero (columns <= D are rewritten per component)

    const int k_begin = blockIdx.y * chunk;
    const int k_end = min(K, k_begin + chunk);
    for (int k = k_begin; k < k_end; ++k) {
        const float* __restrict__ P = packed + (size_t)k * PK::STRIDE;
        float a_log = -3.0e38f;
        if (valid) {
            if (own_only) a_log = (mapping[n] + map_offset == k) ? 0.f : -3.0e38f;
            else a_log = ld[(size_t)k * N + n] - bg[n];
        }
        const float m_w = wave_max(a_log);
        const float e = (valid && a_log > -1.0e38f) ? __expf(a_log - m_w) : 0.f;

        float z[DP];
        forward_subst_s<DP>(P, x, z);                  // z form: L^-T is applied once per component in stein_finalize
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < D) Yw[lane * LDW + i] = e * z[i];
        Yw[lane * LDW + D] = e;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        f32x16 acc[NB][NB];
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[a][b][t] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            float bf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) bf[b] = Yw[(2 * s + half) * LDW + 32 * b + col];
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b], acc[a][b], 0, 0, 0);
        }

        // ---- merge the four waves: common maximum, fixed order ------------------------------------------------------
        if (lane == 0) sm_m[wave] = m_w;
        __syncthreads();                                  // also: every wave is done reading its Y tile
        const float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
        const float f = __expf(m_w - M);
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int i = 32 * a + (t & 3) + 8 * (t >> 2) + 4 * half;
                    const int j = 32 * b + col;
                    Yw[i * W + j] = acc[a][b][t] * f;      // W*W <= 64*LDW floats
                }
        __syncthreads();
        float* out = part + ((size_t)k * n_tiles + tile) * (size_t)(D1 * D1);
        for (int el = tid; el < D1 * D1; el += 256) {
            const int i = el / D1, j = el % D1;
            out[el] = (Ys[i * W + j] + Ys[64 * LDW + i * W + j]) + (Ys[2 * 64 * LDW + i * W + j] + Ys[3 * 64 * LDW + i * W + j]);
        }
        if (tid == 0) part_m[(size_t)k * n_tiles + tile] = M;
        __syncthreads();
        // the merge overwrote the padded columns of the Y tiles: restore the zeros for the next component
        for (int c = D1; c < W; ++c) Yw[lane * LDW + c] = 0.f;
    }
}

// DPZ > 0: the partials are in z form (see stein_wc_kernel) and DPZ is the padded dimension of the packed blocks; 0: y form
template <int DPZ>
__global__ __launch_bounds__(1024) void stein_finalize_kernel(int D, int R, int N, int flags, const float* __restrict__ part,
                                                             const float* __restrict__ part_m, float* __restrict__ H_neg,
                                                             float* __restrict__ g_neg, const float* __restrict__ packed_z) {
    extern __shared__ float A[];       // (D+1)^2, then R scale factors
    const int k = blockIdx.x;
    const int D1 = D + 1;
    float* scale_r = A + D1 * D1;
    float M = -3.0e38f;
    for (int r = threadIdx.x & 63; r < R; r += 64) M = fmaxf(M, part_m[(size_t)k * R + r]);
    M = wave_max(M);
    for (int r = threadIdx.x; r < R; r += blockDim.x) scale_r[r] = __expf(part_m[(size_t)k * R + r] - M);
    __syncthreads();
    // partial sums: the R partials of an element are split over G = blockDim / 256 thread groups (r = g, g + G, ...), eight
    // independent chains per thread keep the loads in flight; the groups are combined through LDS in fixed order
    const float* pk = part + (size_t)k * R * (size_t)(D1 * D1);
    const int G = blockDim.x >> 8, g = threadIdx.x >> 8, tl = threadIdx.x & 255;
    float* Ag = scale_r + R;                                              // [G][(D+1)^2]
    for (int e = tl; e < D1 * D1; e += 256) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = g;
        for (; r + 7 * G < R; r += 8 * G) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = fmaf(pk[(size_t)(r + u * G) * (D1 * D1) + e], scale_r[r + u * G], v[u]);
        }
        for (int u = 0; r < R; r += G, ++u) v[u & 7] = fmaf(pk[(size_t)r * (D1 * D1) + e], scale_r[r], v[u & 7]);
        Ag[g * D1 * D1 + e] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D1 * D1; e += blockDim.x) {
        float a = 0.f;
        for (int gg = 0; gg < G; ++gg) a += Ag[gg * D1 * D1 + e];
        A[e] = a;
    }
    __syncthreads();
    if constexpr (DPZ > 0) {
        // the partials hold sum e g z^T with z = L^-1 (x - mu): y = L^-T z, so row i of sum e g y^T is (row i) L^-1, i.e.
        // h L = b solved from the last column; L from the component's packed block (common.h Pack<DP>: 1/diag, columns)
        // staged in LDS, the row in registers, loops unrolled for the padded dimension
        using PK = Pack<DPZ>;
        const float* Pk = packed_z + (size_t)k * PK::STRIDE;
        float* Lc = Ag;                                // the group sums are consumed: reuse as [1/diag (DPZ) | columns (T)]
        for (int e = threadIdx.x; e < DPZ + PK::T; e += blockDim.x)
            Lc[e] = (e < DPZ) ? Pk[PK::RD + e] : Pk[PK::LCOL + (e - DPZ)];
        __syncthreads();
        if (threadIdx.x < D) {
            float* row = A + threadIdx.x * D1;
            float h[DPZ];
#pragma unroll
            for (int j = 0; j < DPZ; ++j) h[j] = (j < D) ? row[j] : 0.f;
#pragma unroll
            for (int j = DPZ - 1; j >= 0; --j) {
                float t = h[j];
#pragma unroll
                for (int m = j + 1; m < DPZ; ++m) t = fmaf(-h[m], Lc[DPZ + PK::colofs(j) + (m - j - 1)], t);   // padding: L = 0
                h[j] = t * Lc[j];
            }
#pragma unroll
            for (int j = 0; j < DPZ; ++j)
                if (j < D) row[j] = h[j];
        }
        __syncthreads();
    }
    const bool snis = (flags & GMMVI_SELF_NORMALIZED) != 0;
    // A[i][j] = sum e g_i y_j, A[i][D] = sum e g_i, A[D][D] = sum e.
    // plain importance weights: 1/N * sum exp(ld - bg) v   (ng_estimator.py:146-152), Hessian not symmetrised
    // with only_use_own_samples the expectation runs over the component's own samples only (get_rewards_for_comp,
    // ng_estimator.py:110-118: weights exp(0) = 1, divisor = their number): sum e = n_own exp(-M) => exp(M) / n_own = 1 / sum e
    const bool own = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    const float scale = (snis || own) ? 1.f / A[D * D1 + D] : __expf(M) / (float)N;
    for (int e = threadIdx.x; e < D * D; e += blockDim.x) {
        const int i = e / D, j = e % D;
        const float v = snis ? 0.5f * (A[i * D1 + j] + A[j * D1 + i]) : A[i * D1 + j];
        H_neg[(size_t)k * D * D + e] = -v * scale;
    }
    for (int i = threadIdx.x; i < D; i += blockDim.x) g_neg[(size_t)k * D + i] = -A[i * D1 + D] * scale;
}

template <int DPZ>
static int launch_stein_finalize_t(gmmvi_ctx* ctx, int K, int D, int R, int N, int flags, const float* part,
                                   const float* part_m, float* H_neg, float* g_neg, const float* packed_z) {
    const int D1 = D + 1;
    size_t floats = (size_t)5 * D1 * D1 + R;
    if (DPZ > 0 && (size_t)D1 * D1 + R + DPZ + Pack<(DPZ > 0 ? DPZ : 2)>::T > floats)
        floats = (size_t)D1 * D1 + R + DPZ + Pack<(DPZ > 0 ? DPZ : 2)>::T;
    const size_t shmem = floats * sizeof(float);
    static size_t attr = 64 * 1024;
    if (shmem > attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_finalize_kernel<DPZ>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr = shmem;
    }
    GMMVI_PROF(ctx, "stein_finalize");
    hipLaunchKernelGGL(stein_finalize_kernel<DPZ>, dim3(K), dim3(1024), shmem, ctx->stream, D, R, N, flags, part, part_m, H_neg,
                       g_neg, packed_z);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// packed_z != nullptr: partials in z form (wave-per-component kernel), L^-T applied here
static int launch_stein_finalize(gmmvi_ctx* ctx, int K, int D, int R, int N, int flags, const float* part,
                                 const float* part_m, float* H_neg, float* g_neg, const float* packed_z = nullptr) {
    if (packed_z == nullptr) return launch_stein_finalize_t<0>(ctx, K, D, R, N, flags, part, part_m, H_neg, g_neg, nullptr);
    switch (gmmvi_padded_dim(D)) {
#define GMMVI_FIN(DPV) case DPV: return launch_stein_finalize_t<DPV>(ctx, K, D, R, N, flags, part, part_m, H_neg, g_neg, packed_z)
        GMMVI_FIN(2); GMMVI_FIN(4); GMMVI_FIN(8); GMMVI_FIN(10); GMMVI_FIN(12); GMMVI_FIN(16); GMMVI_FIN(20); GMMVI_FIN(24);
        GMMVI_FIN(32); GMMVI_FIN(40); GMMVI_FIN(50); GMMVI_FIN(64);
#undef GMMVI_FIN
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "stein_finalize: unsupported dimension");
    }
}

// Row tile [rows, D] (contiguous in memory) -> LDS image with row stride ld, all loads of a thread in flight at once:
// 16-byte loads when the tile is 16-byte aligned, one division per float4; A minus B when B is given.
template <int VMAX>
__device__ __forceinline__ void stage_rows(const float* __restrict__ A, const float* __restrict__ B, int total, int D,
                                           float* dst, int ld, int tid) {
    const bool vec = ((reinterpret_cast<uintptr_t>(A) | (B ? reinterpret_cast<uintptr_t>(B) : 0)) & 15) == 0;
    if (vec) {
        const int nv = total >> 2;
        float4 v[VMAX];
#pragma unroll
        for (int u = 0; u < VMAX; ++u) {
            const int idx = tid + 256 * u;
            if (idx < nv) {
                v[u] = reinterpret_cast<const float4*>(A)[idx];
                if (B) {
                    const float4 b = reinterpret_cast<const float4*>(B)[idx];
                    v[u].x -= b.x; v[u].y -= b.y; v[u].z -= b.z; v[u].w -= b.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < VMAX; ++u) {
            const int idx = tid + 256 * u;
            if (idx < nv) {
                int r = (4 * idx) / D, c = (4 * idx) - r * D;
                const float vals[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dst[r * ld + c] = vals[q];
                    if (++c == D) { c = 0; ++r; }
                }
            }
        }
        for (int e = 4 * nv + tid; e < total; e += 256) dst[(e / D) * ld + (e % D)] = B ? A[e] - B[e] : A[e];
    } else {
        for (int e = tid; e < total; e += 256) dst[(e / D) * ld + (e % D)] = B ? A[e] - B[e] : A[e];
    }
}

// Wave-per-component form (D <= 24): the workgroup stages a 256-sample tile once; wave w then takes the components
// k_begin + w, + 4, ... of the chunk and runs over all four 64-sample sub-tiles itself, accumulating the augmented
// matrix in its MFMA registers with an online rescale of the running maximum.  Nothing is merged across waves and the
// component loop has no block barrier: one partial per (component, tile) leaves straight from the accumulators.  A
// component's block is read through the scalar cache four times in a row (once per sub-tile) instead of by four waves
// in four different places.  The kernel accumulates  sum_n e [g;1] [z;1]^T  with the FORWARD-substituted z = L^-1 (x - mu) only:
// y = L^-T z is linear in z with a per-component matrix, so L^-T is applied once per component to the finished sum
// (stein_finalize, z_form) instead of once per sample -- half the per-sample vector work.
// W = 32: v_mfma_f32_32x32x2 (D + 1 <= 32); W = 16: v_mfma_f32_16x16x4 for D + 1 <= 16 -- a quarter of the matrix-pipe time
// and half the LDS, the 32-wide tile is 88 % padding at D = 10.  D[i][j] of the 16x16x4 form: i = 4 (l / 16) + r, j = l % 16
// (probed: tools/probe/mfma_f32_16x16x4_layout.hip).
// The e * [y; 1] tile is private to the wave: ordering its LDS writes against the MFMA operand reads only needs the wave's own
// LDS queue drained (and the compiler kept from moving the accesses).  A workgroup-scope fence would also wait for every
// outstanding GLOBAL access -- the prefetched log-density rows, the previous component's partial stores -- ~1 us each time.
#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

template <int DP, int W>
__global__ __launch_bounds__(256, 2) void stein_wc_kernel(int K, int D, int chunk, const float* __restrict__ packed,
                                                          const float* __restrict__ X, const float* __restrict__ TG,
                                                          const float* __restrict__ QG, int N,
                                                          const float* __restrict__ ld, const float* __restrict__ bg,
                                                          const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                          float* __restrict__ part, float* __restrict__ part_m) {
    using PK = Pack<DP>;
    constexpr int LDW = W + 1;
    constexpr int NACC = W == 16 ? 4 : 16;
    using AccT = typename std::conditional<W == 16, f32x4, f32x16>::type;
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int n_tiles = gridDim.x;
    const int D1 = D + 1;
    const int n0 = tile * 256;
    const int n_here = min(256, N - n0);
    float* Gs = sm;                                   // [256][LDW]  rows [g;1;0...]
    float* Ys = sm + 256 * LDW;                       // 4 x [64][LDW] wave-private rows e*[y;1;0...]; first: x staging
    float* Yw = Ys + wave * 64 * LDW;
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;

    // ---- stage the x tile (coalesced); every lane keeps its row of each of the four sub-tiles ---------------------------
    const int ldx = D | 1;
    constexpr int VMAX = (DP + 3) / 4;
    stage_rows<VMAX>(X + (size_t)n0 * D, nullptr, n_here * D, D, Ys, ldx, tid);
    for (int e = tid; e < 256 * LDW; e += 256) Gs[e] = ((e % LDW) == D && e / LDW < n_here) ? 1.f : 0.f;   // [.;1] column
    __syncthreads();
    float x[4][DP];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < DP; ++i) x[t][i] = (64 * t + lane < n_here && i < D) ? Ys[(64 * t + lane) * ldx + i] : 0.f;
    __syncthreads();                                   // x rows are in registers: Ys may be reused
    // g = grad log p~ - grad log q (:248) into the first D columns of Gs
    stage_rows<VMAX>(TG + (size_t)n0 * D, QG + (size_t)n0 * D, n_here * D, D, Gs, LDW, tid);
    __syncthreads();
    for (int c = D1; c < W; ++c) Yw[lane * LDW + c] = 0.f;
    __syncthreads();

    const int col = lane & 31, half = lane >> 5;
    const int k_begin = blockIdx.y * chunk;
    const int k_end = min(K, k_begin + chunk);
    // per-sample constants of the four sub-tiles; the log-density row of the NEXT component is fetched while the
    // current one is being worked on (an L2 round trip per sub-tile would otherwise sit in front of every wave_max)
    float bgv[4], la[4];
    int mp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool v = 64 * t + lane < n_here;
        const int n = n0 + 64 * t + lane;
        bgv[t] = (v && !own_only) ? bg[n] : 0.f;
        mp[t] = (v && own_only) ? mapping[n] + map_offset : -1;
        la[t] = (v && !own_only && k_begin + wave < k_end) ? ld[(size_t)(k_begin + wave) * N + n] : 0.f;
    }
    for (int k = k_begin + wave; k < k_end; k += 4) {
        const float* __restrict__ P = packed + (size_t)k * PK::STRIDE;
        float la_next[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            la_next[t] = (64 * t + lane < n_here && !own_only && k + 4 < k_end) ? ld[(size_t)(k + 4) * N + n0 + 64 * t + lane] : 0.f;
        AccT acc;
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = 0.f;
        float M = -3.0e38f;
        if constexpr (SubstAsmPk<DP>::available) {
            // Sub-tiles substituted in PAIRS: two samples per lane as packed float2 registers, one v_pk_fma_f32 per element of
            // L for both (subst_asm_gen.h) -- the packed rate is the fp32 peak of the vector unit, the plain v_fma_f32 form
            // runs at half of it.
            gmmvi_f32x2 yp[DP];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if ((t & 1) == 0) {
#pragma unroll
                    for (int i = 0; i < DP; ++i) { yp[i].x = x[t][i]; yp[i].y = x[t + 1 < 4 ? t + 1 : 3][i]; }
                    SubstAsmPk<DP>::forward(P, yp);             // z = L^-1 (x - mu); L^-T is applied once, in stein_finalize
                }
                const bool vt = 64 * t + lane < n_here;
                const float a_t = !vt ? -3.0e38f : (own_only ? ((mp[t] == k) ? 0.f : -3.0e38f) : la[t] - bgv[t]);
                const float m_t = wave_max(a_t);
                const float Mn = fmaxf(M, m_t);
                const float f = __expf(M - Mn);                        // 1 when the maximum did not move, 0 at the start
                M = Mn;
#pragma unroll
                for (int r = 0; r < NACC; ++r) acc[r] *= f;
                const float e = (a_t > -1.0e38f) ? __expf(a_t - M) : 0.f;
#pragma unroll
                for (int i = 0; i < DP; ++i)
                    if (i < D) Yw[lane * LDW + i] = e * ((t & 1) ? yp[i].y : yp[i].x);
                Yw[lane * LDW + D] = e;
                WAVE_LDS_SYNC();
                constexpr int KS = W == 16 ? 4 : 2, NS = 64 / KS;
                const int fr = W == 16 ? (lane & 15) : col, fk = W == 16 ? (lane >> 4) : half;
                const float* Gt = Gs + (64 * t + fk) * LDW + fr;
                const float* Yt = Yw + fk * LDW + fr;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) {
                    if constexpr (W == 16) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Gt[KS * s2 * LDW], Yt[KS * s2 * LDW], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[KS * s2 * LDW], Yt[KS * s2 * LDW], acc, 0, 0, 0);
                }
                WAVE_LDS_SYNC();
            }
        } else {
        // Software pipeline over the four sub-tiles: the substitution of sub-tile t + 1 (VALU, scalar loads) is issued in the
            // same straight-line region as the 32 MFMAs of sub-tile t, so the matrix pipe works in the shadow of the vector
            // work of the same wave; the rescale by the running maximum sits between the regions and is branch-free.
            float yn[DP], a_n;
            {
                const bool v0 = lane < n_here;
                a_n = !v0 ? -3.0e38f : (own_only ? ((mp[0] == k) ? 0.f : -3.0e38f) : la[0] - bgv[0]);
                if constexpr (SubstAsm<DP>::available) {
#pragma unroll
                    for (int i = 0; i < DP; ++i) yn[i] = x[0][i];
                    SubstAsm<DP>::forward(P, yn);                          // hand-scheduled: double-buffered scalar feed
                } else {
                    forward_subst_s<DP>(P, x[0], yn);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // ---- region B: new running maximum, rescale, publish e * [y; 1] of sub-tile t ---------------------------------
                const float m_t = wave_max(a_n);
                const float Mn = fmaxf(M, m_t);
                const float f = __expf(M - Mn);                            // 1 when the maximum did not move, 0 at the start
                M = Mn;
#pragma unroll
                for (int r = 0; r < NACC; ++r) acc[r] *= f;
                const float e = (a_n > -1.0e38f) ? __expf(a_n - M) : 0.f;
#pragma unroll
                for (int i = 0; i < DP; ++i)
                    if (i < D) Yw[lane * LDW + i] = e * yn[i];
                Yw[lane * LDW + D] = e;
                WAVE_LDS_SYNC();
                // ---- region A: MFMAs of sub-tile t  ||  substitution of sub-tile t + 1 ---------------------------------------
                // operand fragments: 32x32x2 -> lane (row/col = l % 32, k = l / 32), 32 steps of 2 samples;
                //                    16x16x4 -> lane (row/col = l % 16, k = l / 16), 16 steps of 4 samples
                constexpr int KS = W == 16 ? 4 : 2, NS = 64 / KS;
                const int fr = W == 16 ? (lane & 15) : col, fk = W == 16 ? (lane >> 4) : half;
                const float* Gt = Gs + (64 * t + fk) * LDW + fr;
                const float* Yt = Yw + fk * LDW + fr;
                float ga[NS], yb[NS];
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) { ga[s2] = Gt[KS * s2 * LDW]; yb[s2] = Yt[KS * s2 * LDW]; }
                if (t < 3) {
                    const bool v1 = 64 * (t + 1) + lane < n_here;
                    a_n = !v1 ? -3.0e38f : (own_only ? ((mp[t + 1 < 4 ? t + 1 : 3] == k) ? 0.f : -3.0e38f)
                                                      : la[t + 1 < 4 ? t + 1 : 3] - bgv[t + 1 < 4 ? t + 1 : 3]);
                    if constexpr (SubstAsm<DP>::available) {
#pragma unroll
                        for (int i = 0; i < DP; ++i) yn[i] = x[t + 1 < 4 ? t + 1 : 3][i];
                        SubstAsm<DP>::forward(P, yn);
                    } else {
                        forward_subst_s<DP>(P, x[t + 1 < 4 ? t + 1 : 3], yn);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) {
                    if constexpr (W == 16) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s2], yb[s2], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[s2], yb[s2], acc, 0, 0, 0);
                }
                WAVE_LDS_SYNC();
            }
        }
        float* out = part + ((size_t)k * n_tiles + tile) * (size_t)(D1 * D1);
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int i = W == 16 ? 4 * (lane >> 4) + r : (r & 3) + 8 * (r >> 2) + 4 * half;
            const int j = W == 16 ? (lane & 15) : col;
            if (i < D1 && j < D1) out[i * D1 + j] = acc[r];
        }
        if (lane == 0) part_m[(size_t)k * n_tiles + tile] = M;
#pragma unroll
        for (int t = 0; t < 4; ++t) la[t] = la_next[t];
    }
}

template <int DP, int W>
static int launch_stein_wc(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                           const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping, int map_offset,
                           int flags, float* H_neg, float* g_neg) {
    constexpr int LDW = W + 1;
    const int D1 = D + 1;
    const int n_tiles = (N + 255) / 256;
    // components per workgroup: a multiple of the four waves, ~2 workgroups per CU in flight
    int chunk = (int)(((long)n_tiles * K + 2L * ctx->num_cus - 1) / (2L * ctx->num_cus));
    chunk = ((chunk + 3) / 4) * 4;
    if (chunk < 4) chunk = 4;
    if (chunk > 16) chunk = 16;
    static const int env_chunk = getenv("GMMVI_STEIN_CHUNK") ? atoi(getenv("GMMVI_STEIN_CHUNK")) : 0;
    if (env_chunk > 0) chunk = env_chunk;
    const int n_chunks = (K + chunk - 1) / chunk;
    const size_t part_floats = (size_t)K * n_tiles * D1 * D1;
    int rc = gmmvi_ws_reserve(ctx, (part_floats + (size_t)K * n_tiles) * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    float* part = (float*)ctx->ws;
    float* part_m = part + part_floats;
    const size_t shmem = (size_t)(256 + 4 * 64) * LDW * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_wc_kernel<DP, W>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    {
        GMMVI_PROF(ctx, "stein_partial");
        hipLaunchKernelGGL((stein_wc_kernel<DP, W>), dim3(n_tiles, n_chunks), dim3(256), shmem, ctx->stream, K, D, chunk,
                           packed, X, tgrad, qgrad, N, ld, bg, mapping, map_offset, flags, part, part_m);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    return launch_stein_finalize(ctx, K, D, n_tiles, N, flags, part, part_m, H_neg, g_neg, packed);   // z form
}


template <int DP, int NB>
static int launch_stein(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                        const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping, int map_offset,
                        int flags, float* H_neg, float* g_neg) {
    constexpr int LDW = 32 * NB + 1;
    const int D1 = D + 1;
    const int n_tiles = (N + 255) / 256;
    // components per workgroup: amortise the tile staging, keep >= ~4 workgroups per CU in flight
    int chunk = (int)(((long)n_tiles * K + 4L * ctx->num_cus - 1) / (4L * ctx->num_cus));
    if (chunk < 1) chunk = 1;
    if (chunk > 16) chunk = 16;
    const int n_chunks = (K + chunk - 1) / chunk;
    const size_t part_floats = (size_t)K * n_tiles * D1 * D1;
    const size_t need = (part_floats + (size_t)K * n_tiles) * sizeof(float);
    int rc = gmmvi_ws_reserve(ctx, need);
    if (rc != GMMVI_OK) return rc;
    float* part = (float*)ctx->ws;
    float* part_m = part + part_floats;
    const size_t shmem = (size_t)256 * LDW * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_partial_kernel<DP, NB>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    {
        GMMVI_PROF(ctx, "stein_partial");
        hipLaunchKernelGGL((stein_partial_kernel<DP, NB>), dim3(n_tiles, n_chunks), dim3(256), shmem, ctx->stream, K, D,
                           chunk, packed, X, tgrad, qgrad, N, ld, bg, mapping, map_offset, flags, part, part_m);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    return launch_stein_finalize(ctx, K, D, n_tiles, N, flags, part, part_m, H_neg, g_neg, packed);   // z form
}

extern "C" int gmmvi_stein(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N,
                           const float* ld_dev, const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev,
                           const int32_t* mapping_dev, int map_offset, int flags, float* H_neg_out_dev,
                           float* g_neg_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && (D < GMMVI_MAX_DIM || gmmvi_is_blocked_dim(D)) && N >= 1);
    GMMVI_ARG_CHECK(ctx, packed_dev && X_dev && qgrad_dev && tgrad_dev && H_neg_out_dev && g_neg_out_dev);
    if (flags & GMMVI_OWN_SAMPLES_ONLY) GMMVI_ARG_CHECK(ctx, mapping_dev != nullptr);
    else GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev);
    if (gmmvi_is_blocked_dim(D))
        return gmmvi_blocked_stein(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev, mapping_dev,
                                   map_offset, flags, H_neg_out_dev, g_neg_out_dev);
    const int dp = gmmvi_padded_dim(D);
    const bool two = (D + 1) > 32;
    // Padded dimension 64 below the blocked threshold (only when GMMVI_BLOCKED_ABOVE was raised): the blocked contractions on
    // L^-1 blocks rebuilt from the packed ones.  Up to 50 the tiled kernel wins since it runs two workgroups per CU (C3 shape:
    // 407 + 32 us against 29 + 188 + 257 + 18 us); GMMVI_STEIN_TILED=1 forces the tiled kernel.
    static const bool force_tiled_big = getenv("GMMVI_STEIN_TILED") != nullptr;
    if (dp >= GMMVI_STEIN_BLOCKED_FROM_DP && !force_tiled_big)
        return gmmvi_blocked_stein_from_register_pack(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev,
                                                      mapping_dev, map_offset, flags, H_neg_out_dev, g_neg_out_dev);
    // D <= 24: wave-per-component kernel (GMMVI_STEIN_TILED=1 selects the tiled kernel with its cross-wave merge)
    static const bool force_tiled = getenv("GMMVI_STEIN_TILED") != nullptr;
    if (!force_tiled && dp <= 24) {
        switch (dp) {
#define GMMVI_STEIN_WC(DPV, WV)                                                                                      \
    case DPV: return launch_stein_wc<DPV, WV>(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev,  \
                                              mapping_dev, map_offset, flags, H_neg_out_dev, g_neg_out_dev)
            GMMVI_STEIN_WC(2, 16); GMMVI_STEIN_WC(4, 16); GMMVI_STEIN_WC(8, 16); GMMVI_STEIN_WC(10, 16);
            GMMVI_STEIN_WC(12, 16);
            case 16:
                if (D + 1 <= 16)
                    return launch_stein_wc<16, 16>(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev,
                                                   mapping_dev, map_offset, flags, H_neg_out_dev, g_neg_out_dev);
                return launch_stein_wc<16, 32>(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev,
                                               mapping_dev, map_offset, flags, H_neg_out_dev, g_neg_out_dev);
            GMMVI_STEIN_WC(20, 32); GMMVI_STEIN_WC(24, 32);
#undef GMMVI_STEIN_WC
            default: break;
        }
    }
    switch (dp) {
#define GMMVI_STEIN_CASE(DPV, NBV)                                                                                  \
    return launch_stein<DPV, NBV>(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev,            \
                                  mapping_dev, map_offset, flags, H_neg_out_dev, g_neg_out_dev)
        case 2: GMMVI_STEIN_CASE(2, 1);
        case 4: GMMVI_STEIN_CASE(4, 1);
        case 8: GMMVI_STEIN_CASE(8, 1);
        case 10: GMMVI_STEIN_CASE(10, 1);
        case 12: GMMVI_STEIN_CASE(12, 1);
        case 16: GMMVI_STEIN_CASE(16, 1);
        case 20: GMMVI_STEIN_CASE(20, 1);
        case 24: GMMVI_STEIN_CASE(24, 1);
        case 32: if (two) GMMVI_STEIN_CASE(32, 2); else GMMVI_STEIN_CASE(32, 1);
        case 40: GMMVI_STEIN_CASE(40, 2);
        case 50: GMMVI_STEIN_CASE(50, 2);
        case 64: GMMVI_STEIN_CASE(64, 2);
#undef GMMVI_STEIN_CASE
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "unsupported dimension for gmmvi_stein (D must be <= 63)");
    }
}
