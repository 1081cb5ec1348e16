// Importance-weighted Stein estimate of the per-component expected gradient / Hessian
// (gmmvi_modules/ng_estimator.py:204-263, :171-188 self-normalised, :154-169 plain importance weights).
//
// For component k:  A_k = sum_n [g_n; 1] (e_kn [y_kn; 1])^T,  e_kn = exp(ld[k,n] - bg[n] - m),  g_n = grad log p~ - grad log q,
// y_kn = Sigma_k^-1 (x_n - mu_k).  The (D+1)x(D+1) matrix A_k carries sum e g y^T, sum e g (last column) and sum e
// (corner), so one contraction over the samples yields the Hessian, the gradient and the normaliser.
//
// Mapping (DESIGN.md "stein"): grid = (256-sample tiles, component chunks); 4 waves per workgroup, 64 samples each.
//   * the x and g tiles of the workgroup are fetched ONCE with fully coalesced loads and staged through LDS (the
//     per-lane rows of a row-major [N, D] array are 4*D bytes apart: loading them lane-by-lane costs one cache line per
//     lane per element and was the dominant stall of the first version of this kernel);
//   * each lane keeps its sample x in VGPRs and each wave keeps the [g;1] MFMA A-fragments of its 64 samples in VGPRs;
//     both are reused for every component of the chunk;
//   * per component: y by the register-resident forward/backward substitution (component block through scalar
//     loads), e*[y;1] rows written to the wave's LDS tile (row stride 32*NB+1: conflict-free), contraction of the 64
//     samples with v_mfma_f32_32x32x2_f32, then the four waves are merged through LDS (own maximum per wave, fixed
//     summation order) and the (component, tile) partial goes to a slab.
// stein_finalize sums the slab in fixed order (bitwise reproducible), normalises, symmetrises and negates.
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

template <int DP, int NB>
__global__ __launch_bounds__(256) void stein_partial_kernel(int K, int D, int chunk, const float* __restrict__ packed,
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
    float* Gs = sm;                                   // [256][LDW]  rows [g;1;0...]
    float* Ys = sm + 256 * LDW;                       // 4 x [64][LDW] rows e*[y;1;0...]; also staging / merge scratch
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
    for (int c = D1; c < W; ++c) Yw[lane * LDW + c] = 0.f;
    __syncthreads();
    // A fragments of this wave's 64 samples: lane (col = l & 31, half = l >> 5), step s -> Gs[2s + half][col]
    const int col = lane & 31, half = lane >> 5;
    float af[NB][32];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int s = 0; s < 32; ++s) af[a][s] = Gs[(wave * 64 + 2 * s + half) * LDW + 32 * a + col];

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

        float z[DP], y[DP];
        forward_subst_s<DP>(P, x, z);
        backward_subst_s<DP>(P, z, y);
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < D) Yw[lane * LDW + i] = e * y[i];
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

__global__ __launch_bounds__(256) void stein_finalize_kernel(int D, int R, int N, int flags, const float* __restrict__ part,
                                                             const float* __restrict__ part_m, float* __restrict__ H_neg,
                                                             float* __restrict__ g_neg) {
    extern __shared__ float A[];       // (D+1)^2, then R scale factors
    const int k = blockIdx.x;
    const int D1 = D + 1;
    float* scale_r = A + D1 * D1;
    float M = -3.0e38f;
    for (int r = threadIdx.x & 63; r < R; r += 64) M = fmaxf(M, part_m[(size_t)k * R + r]);
    M = wave_max(M);
    for (int r = threadIdx.x; r < R; r += 256) scale_r[r] = __expf(part_m[(size_t)k * R + r] - M);
    __syncthreads();
    const float* pk = part + (size_t)k * R * (size_t)(D1 * D1);
    for (int e = threadIdx.x; e < D1 * D1; e += 256) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;                    // four chains in flight, fixed final order
        int r = 0;
        for (; r + 3 < R; r += 4) {
            v0 = fmaf(pk[(size_t)(r + 0) * (D1 * D1) + e], scale_r[r + 0], v0);
            v1 = fmaf(pk[(size_t)(r + 1) * (D1 * D1) + e], scale_r[r + 1], v1);
            v2 = fmaf(pk[(size_t)(r + 2) * (D1 * D1) + e], scale_r[r + 2], v2);
            v3 = fmaf(pk[(size_t)(r + 3) * (D1 * D1) + e], scale_r[r + 3], v3);
        }
        for (; r < R; ++r) v0 = fmaf(pk[(size_t)r * (D1 * D1) + e], scale_r[r], v0);
        A[e] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    const bool snis = (flags & GMMVI_SELF_NORMALIZED) != 0;
    // A[i][j] = sum e g_i y_j, A[i][D] = sum e g_i, A[D][D] = sum e.
    // plain importance weights: 1/N * sum exp(ld - bg) v   (ng_estimator.py:146-152), Hessian not symmetrised
    const float scale = snis ? 1.f / A[D * D1 + D] : __expf(M) / (float)N;
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int i = e / D, j = e % D;
        const float v = snis ? 0.5f * (A[i * D1 + j] + A[j * D1 + i]) : A[i * D1 + j];
        H_neg[(size_t)k * D * D + e] = -v * scale;
    }
    for (int i = threadIdx.x; i < D; i += 256) g_neg[(size_t)k * D + i] = -A[i * D1 + D] * scale;
}

// ---------------------------------------------------------------------------------------------------------------------
// Component-stationary variant (DP <= 24): one wave = one component whose (mu, 1/diag, L) live in VGPRs for the whole
// kernel; the wave streams over the sample tiles of its chunk.  The substitution then runs on register operands only
// (no scalar-load or LDS feed in the dependent chain) and the (D+1)^2 accumulator stays in the MFMA registers across
// all tiles (online rescaling with a wave-uniform running maximum).  A workgroup = 4 components sharing the
// double-buffered, coalesced x / [g;1] tiles of 64 samples.
// ---------------------------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void stein_cs_kernel(int K, int D, int tiles_per_chunk, const float* __restrict__ packed,
                                                       const float* __restrict__ X, const float* __restrict__ TG,
                                                       const float* __restrict__ QG, int N, const float* __restrict__ ld,
                                                       const float* __restrict__ bg, const int32_t* __restrict__ mapping,
                                                       int map_offset, int flags, float* __restrict__ part,
                                                       float* __restrict__ part_m) {
    using PK = Pack<DP>;
    constexpr int W = 32, LDW = 33, T = PK::T;
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D1 = D + 1;
    const int ldx = D | 1;
    const int k = blockIdx.y * 4 + wave;
    const bool has_comp = k < K;
    const int chunk = blockIdx.x, n_chunks = gridDim.x;
    const int tile_begin = chunk * tiles_per_chunk;
    const int n_tiles_total = (N + 63) / 64;
    const int tile_end = min(n_tiles_total, tile_begin + tiles_per_chunk);
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    // LDS: Xs[2][64][ldx], Gs[2][64][LDW], Yw[4][64][LDW], Pw[4][STRIDE]
    float* Xs = sm;
    float* Gs = Xs + 2 * 64 * ldx;
    float* Yw = Gs + 2 * 64 * LDW + wave * 64 * LDW;
    float* Pw = Gs + 2 * 64 * LDW + 4 * 64 * LDW + wave * PK::STRIDE;

    // ---- component block -> LDS (coalesced) -> registers ------------------------------------------------------------------
    float mu[DP], rd[DP], Lr[T > 0 ? T : 1];
    if (has_comp)
        for (int i = lane; i < PK::STRIDE; i += 64) Pw[i] = packed[(size_t)k * PK::STRIDE + i];
    for (int c = D1; c < W; ++c) Yw[lane * LDW + c] = 0.f;               // padded columns of the Y tile stay zero
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DP; ++i) { mu[i] = has_comp ? Pw[PK::MU + i] : 0.f; rd[i] = has_comp ? Pw[PK::RD + i] : 1.f; }
#pragma unroll
    for (int i = 0; i < T; ++i) Lr[i] = has_comp ? Pw[PK::LROW + i] : 0.f;

    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;
    float m_run = -3.0e38f;
    const int col = lane & 31, half = lane >> 5;

    // software pipeline: the global loads of tile t+1 (x, g and the importance-weight logits) are issued before tile t is
    // computed and land in LDS afterwards, so their latency hides behind the substitution + MFMA of tile t
    constexpr int QX = (64 * DP + 255) / 256;
    float xr[QX], gr[QX];
    float a_next = -3.0e38f;
    auto issue_loads = [&](int tile) {
        const int n0 = tile * 64;
        const int n_here = min(64, N - n0);
#pragma unroll
        for (int q = 0; q < QX; ++q) {
            const int e = tid + 256 * q;
            const bool ok = e < 64 * D && e / D < n_here;
            const size_t gi = (size_t)n0 * D + e;
            xr[q] = ok ? X[gi] : 0.f;
            gr[q] = ok ? TG[gi] - QG[gi] : 0.f;                          // g = grad log p~ - grad log q (:248)
        }
        const int n = n0 + lane;
        a_next = -3.0e38f;
        if (has_comp && n < N) {
            if (own_only) a_next = (mapping[n] + map_offset == k) ? 0.f : -3.0e38f;
            else a_next = ld[(size_t)k * N + n] - bg[n];
        }
    };
    auto commit_loads = [&](int tile, int buf) {
        const int n_here = min(64, N - tile * 64);
        float* xs = Xs + buf * 64 * ldx;
        float* gs = Gs + buf * 64 * LDW;
#pragma unroll
        for (int q = 0; q < QX; ++q) {
            const int e = tid + 256 * q;
            if (e < 64 * D) {
                xs[(e / D) * ldx + (e % D)] = xr[q];
                gs[(e / D) * LDW + (e % D)] = gr[q];
            }
        }
        if (tid < 64) gs[tid * LDW + D] = (tid < n_here) ? 1.f : 0.f;    // the appended 1 of [g;1]
    };
    // zero the padded columns of both g buffers once
    for (int e = tid; e < 2 * 64 * (W - D1); e += 256) {
        const int r = e / (W - D1), c = D1 + e % (W - D1);
        Gs[r * LDW + c] = 0.f;
    }
    if (tile_begin < tile_end) { issue_loads(tile_begin); commit_loads(tile_begin, 0); }
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int buf = (tile - tile_begin) & 1;
        const float a_log = a_next;
        const bool more = tile + 1 < tile_end;
        if (more) issue_loads(tile + 1);
        const int n = tile * 64 + lane;
        const bool valid = n < N;
        const float* xs = Xs + buf * 64 * ldx;
        const float* gs = Gs + buf * 64 * LDW;
        if (has_comp) {
            const float m_new = fmaxf(m_run, wave_max(a_log));
            const float rescale = __expf(m_run - m_new);
            m_run = m_new;
            const float e = (valid && a_log > -1.0e38f) ? __expf(a_log - m_new) : 0.f;
            acc *= rescale;
            // z = L^-1 (x - mu) in place, then y = L^-T z in place (register operands only)
            float v[DP];
#pragma unroll
            for (int i = 0; i < DP; ++i) v[i] = (i < D) ? xs[lane * ldx + i] - mu[i] : 0.f;
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                float t = v[i];
#pragma unroll
                for (int j = 0; j < i; ++j) t = fmaf(-Lr[PK::rowofs(i) + j], v[j], t);
                v[i] = t * rd[i];
            }
#pragma unroll
            for (int i = DP - 1; i >= 0; --i) {
                float t = v[i];
#pragma unroll
                for (int j = i + 1; j < DP; ++j) t = fmaf(-Lr[PK::rowofs(j) + i], v[j], t);
                v[i] = t * rd[i];
            }
#pragma unroll
            for (int i = 0; i < DP; ++i)
                if (i < D) Yw[lane * LDW + i] = e * v[i];
            Yw[lane * LDW + D] = e;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // all 64 operand reads in flight, then the 32 dependent MFMAs
            float af[32], bf[32];
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                af[s2] = gs[(2 * s2 + half) * LDW + col];
                bf[s2] = Yw[(2 * s2 + half) * LDW + col];
            }
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s2], bf[s2], acc, 0, 0, 0);
        }
        if (more) commit_loads(tile + 1, buf ^ 1);
        __syncthreads();       // tile t+1 visible to every wave; buffers of tile t free for the loads issued next round
    }
    if (!has_comp) return;
    // ---- (component, chunk) partial straight from the accumulator registers -----------------------------------------------
    float* out = part + ((size_t)k * n_chunks + chunk) * (size_t)(D1 * D1);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = (t & 3) + 8 * (t >> 2) + 4 * half;
        if (i < D1 && col < D1) out[i * D1 + col] = acc[t];
    }
    if (lane == 0) part_m[(size_t)k * n_chunks + chunk] = m_run;
}

template <int DP>
static int launch_stein_cs(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                           const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping,
                           int map_offset, int flags, float* H_neg, float* g_neg) {
    using PK = Pack<DP>;
    const int D1 = D + 1;
    const int groups = (K + 3) / 4;
    const int n_tiles = (N + 63) / 64;
    // one workgroup per CU (the register-resident component limits occupancy to one wave per SIMD): chunks ~ CUs / groups
    int n_chunks = ctx->num_cus / groups;
    if (n_chunks < 1) n_chunks = 1;
    if (n_chunks > n_tiles) n_chunks = n_tiles;
    const int tiles_per_chunk = (n_tiles + n_chunks - 1) / n_chunks;
    n_chunks = (n_tiles + tiles_per_chunk - 1) / tiles_per_chunk;
    const size_t part_floats = (size_t)K * n_chunks * D1 * D1;
    int rc = gmmvi_ws_reserve(ctx, (part_floats + (size_t)K * n_chunks) * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    float* part = (float*)ctx->ws;
    float* part_m = part + part_floats;
    const size_t shmem = ((size_t)2 * 64 * (D | 1) + 2 * 64 * 33 + 4 * 64 * 33 + 4 * PK::STRIDE) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_cs_kernel<DP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    {
        GMMVI_PROF(ctx, "stein_partial");
        hipLaunchKernelGGL((stein_cs_kernel<DP>), dim3(n_chunks, groups), dim3(256), shmem, ctx->stream, K, D,
                           tiles_per_chunk, packed, X, tgrad, qgrad, N, ld, bg, mapping, map_offset, flags, part, part_m);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    GMMVI_PROF(ctx, "stein_finalize");
    hipLaunchKernelGGL(stein_finalize_kernel, dim3(K), dim3(256), ((size_t)D1 * D1 + n_chunks) * sizeof(float),
                       ctx->stream, D, n_chunks, N, flags, part, part_m, H_neg, g_neg);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
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
    const size_t shmem = (size_t)(256 + 4 * 64) * LDW * sizeof(float);
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
    GMMVI_PROF(ctx, "stein_finalize");
    hipLaunchKernelGGL(stein_finalize_kernel, dim3(K), dim3(256), ((size_t)D1 * D1 + n_tiles) * sizeof(float), ctx->stream,
                       D, n_tiles, N, flags, part, part_m, H_neg, g_neg);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

extern "C" int gmmvi_stein(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N,
                           const float* ld_dev, const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev,
                           const int32_t* mapping_dev, int map_offset, int flags, float* H_neg_out_dev,
                           float* g_neg_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D < GMMVI_MAX_DIM && N >= 1);
    GMMVI_ARG_CHECK(ctx, packed_dev && X_dev && qgrad_dev && tgrad_dev && H_neg_out_dev && g_neg_out_dev);
    if (flags & GMMVI_OWN_SAMPLES_ONLY) GMMVI_ARG_CHECK(ctx, mapping_dev != nullptr);
    else GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev);
    const int dp = gmmvi_padded_dim(D);
    const bool two = (D + 1) > 32;
    // The component-stationary kernel is experimental (GMMVI_STEIN_CS=1, DP <= 24): measured slower than the tiled
    // kernel on MI355X -- at DP = 20 the register-resident L spills to AGPRs (109 vs 82 us at the north-star shape), at
    // DP = 10 both are bound by the 32x32 MFMA tile (226 vs 165 us at K = 200, N = 20000); profiles/r01_notes.md.
    static const bool force_cs = getenv("GMMVI_STEIN_CS") != nullptr;
    if (force_cs && dp <= 24) {
        switch (dp) {
#define GMMVI_STEIN_CS(DPV)                                                                                          \
    case DPV: return launch_stein_cs<DPV>(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev,      \
                                          mapping_dev, map_offset, flags, H_neg_out_dev, g_neg_out_dev)
            GMMVI_STEIN_CS(2); GMMVI_STEIN_CS(4); GMMVI_STEIN_CS(8); GMMVI_STEIN_CS(10); GMMVI_STEIN_CS(12);
            GMMVI_STEIN_CS(16); GMMVI_STEIN_CS(20); GMMVI_STEIN_CS(24);
#undef GMMVI_STEIN_CS
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
