// Importance-weighted Stein estimate of the per-component expected gradient / Hessian
// (gmmvi_modules/ng_estimator.py:204-263, :171-188 self-normalised, :154-169 plain importance weights).
//
// For component k:  A_k = sum_n e_kn [g_n; 1] [y_kn; 1]^T,  e_kn = exp(ld[k,n] - bg[n] - m),  g_n = grad log p~ - grad log q,
// y_kn = Sigma_k^-1 (x_n - mu_k).  The (D+1)x(D+1) matrix A_k carries sum e g y^T, sum e g (last column) and sum e
// (corner), so one contraction over the samples yields the Hessian, the gradient and the normaliser.
//
// Mapping (DESIGN.md "stein"): grid = (sample ranges, components), 4 waves per workgroup.  Each wave takes 64
// samples per step: one lane per sample computes y by the register-resident forward/backward substitution
// (component block through scalar loads), writes e*[g;1] and [y;1] rows to its private LDS tile (row stride
// 32*NB+1: conflict-free), then contracts the 64 samples with v_mfma_f32_32x32x2_f32 (A = G^T, B = Y read straight
// from the tile).  A running wave-uniform maximum keeps e <= 1 (online rescaling of the accumulators).  Waves are
// merged through LDS, ranges through a slab summed in fixed order by stein_finalize (bitwise reproducible).
#include "common.h"

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
__global__ __launch_bounds__(256) void stein_partial_kernel(int K, int D, const float* __restrict__ packed,
                                                            const float* __restrict__ X, int N, int range_size,
                                                            const float* __restrict__ ld, const float* __restrict__ qgrad,
                                                            const float* __restrict__ bg, const float* __restrict__ tgrad,
                                                            const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                            float* __restrict__ part, float* __restrict__ part_m) {
    using PK = Pack<DP>;
    constexpr int W = 32 * NB;         // padded width of [g;1] and [y;1]
    constexpr int LDW = W + 1;         // LDS row stride
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y;
    const int r = blockIdx.x;
    const int D1 = D + 1;
    float* Gt = sm + (size_t)wave * (2 * 64 * LDW);
    float* Yt = Gt + 64 * LDW;
    const float* __restrict__ P = packed + (size_t)k * PK::STRIDE;
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;

    // zero the padded columns once (columns 0..D are rewritten every step)
    for (int c = D1; c < W; ++c) { Gt[lane * LDW + c] = 0.f; Yt[lane * LDW + c] = 0.f; }

    f32x16 acc[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[a][b][t] = 0.f;
    float m_run = -3.0e38f;

    const int n_begin = r * range_size;
    const int n_end = min(N, n_begin + range_size);
    for (int base = n_begin + wave * 64; base < n_end; base += 256) {
        const int n = base + lane;
        const bool valid = n < n_end;
        float a_log = -3.0e38f;
        if (valid) {
            if (own_only) a_log = (mapping[n] + map_offset == k) ? 0.f : -3.0e38f;
            else a_log = ld[(size_t)k * N + n] - bg[n];
        }
        const float m_new = fmaxf(m_run, wave_max(a_log));
        const float rescale = __expf(m_run - m_new);
        m_run = m_new;
        const float e = (valid && a_log > -1.0e38f) ? __expf(a_log - m_new) : 0.f;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] *= rescale;

        float x[DP], z[DP], y[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) x[i] = (valid && i < D) ? X[(size_t)n * D + i] : 0.f;
        forward_subst_s<DP>(P, x, z);
        backward_subst_s<DP>(P, z, y);
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                const float g = valid ? (tgrad[(size_t)n * D + i] - qgrad[(size_t)n * D + i]) : 0.f;
                Gt[lane * LDW + i] = e * g;
                Yt[lane * LDW + i] = valid ? y[i] : 0.f;
            }
        }
        Gt[lane * LDW + D] = e;
        Yt[lane * LDW + D] = 1.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        const int col = lane & 31, half = lane >> 5;
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int row = 2 * s + half;
            float af[NB], bf[NB];
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                af[a] = Gt[row * LDW + 32 * a + col];
                bf[a] = Yt[row * LDW + 32 * a + col];
            }
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the 4 waves (each wave's scratch lives inside its own tile region) ----
    __shared__ float sm_m[4];
    if (lane == 0) sm_m[wave] = m_run;
    __syncthreads();
    const float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
    const float f = __expf(m_run - M);
    float* red = Gt;                                   // [W][W] floats, W*W <= 2*64*LDW
    {
        const int col = lane & 31, half = lane >> 5;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int i = 32 * a + (t & 3) + 8 * (t >> 2) + 4 * half;
                    const int j = 32 * b + col;
                    red[i * W + j] = acc[a][b][t] * f;
                }
    }
    __syncthreads();
    float* out = part + ((size_t)k * gridDim.x + r) * (size_t)(D1 * D1);
    for (int e = threadIdx.x; e < D1 * D1; e += 256) {
        const int i = e / D1, j = e % D1;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += sm[(size_t)w * (2 * 64 * LDW) + i * W + j];
        out[e] = v;
    }
    if (threadIdx.x == 0) part_m[(size_t)k * gridDim.x + r] = M;
}

__global__ __launch_bounds__(256) void stein_finalize_kernel(int D, int R, int N, int flags, const float* __restrict__ part,
                                                             const float* __restrict__ part_m, float* __restrict__ H_neg,
                                                             float* __restrict__ g_neg) {
    extern __shared__ float A[];       // (D+1)^2
    const int k = blockIdx.x;
    const int D1 = D + 1;
    float M = -3.0e38f;
    for (int r = 0; r < R; ++r) M = fmaxf(M, part_m[(size_t)k * R + r]);
    for (int e = threadIdx.x; e < D1 * D1; e += 256) {
        float v = 0.f;
        for (int r = 0; r < R; ++r)
            v = fmaf(part[((size_t)k * R + r) * (size_t)(D1 * D1) + e], __expf(part_m[(size_t)k * R + r] - M), v);
        A[e] = v;
    }
    __syncthreads();
    const bool snis = (flags & GMMVI_SELF_NORMALIZED) != 0;
    // plain importance weights: 1/N * sum exp(ld - bg) v   (ng_estimator.py:146-152), Hessian not symmetrised
    const float scale = snis ? 1.f / A[D * D1 + D] : __expf(M) / (float)N;
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int i = e / D, j = e % D;
        const float v = snis ? 0.5f * (A[i * D1 + j] + A[j * D1 + i]) : A[i * D1 + j];
        H_neg[(size_t)k * D * D + e] = -v * scale;
    }
    for (int i = threadIdx.x; i < D; i += 256) g_neg[(size_t)k * D + i] = -A[i * D1 + D] * scale;
}

template <int DP, int NB>
static int launch_stein(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                        const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping, int map_offset,
                        int flags, float* H_neg, float* g_neg) {
    constexpr int LDW = 32 * NB + 1;
    const int D1 = D + 1;
    // sample ranges: aim at ~8 workgroups per CU, at least 256 samples (one step of 4 waves) per range
    long target = (8L * ctx->num_cus + K - 1) / K;
    if (target < 1) target = 1;
    long steps_total = ((long)N + 255) / 256;
    long steps_per_range = (steps_total + target - 1) / target;
    if (steps_per_range < 1) steps_per_range = 1;
    const int range_size = (int)(steps_per_range * 256);
    const int R = (int)(((long)N + range_size - 1) / range_size);
    size_t part_floats = (size_t)K * R * D1 * D1;
    size_t need = (part_floats + (size_t)K * R) * sizeof(float);
    int rc = gmmvi_ws_reserve(ctx, need);
    if (rc != GMMVI_OK) return rc;
    float* part = (float*)ctx->ws;
    float* part_m = part + part_floats;
    size_t shmem = (size_t)4 * 2 * 64 * LDW * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_partial_kernel<DP, NB>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    {
    GMMVI_PROF(ctx, "stein_partial");
    hipLaunchKernelGGL((stein_partial_kernel<DP, NB>), dim3(R, K), dim3(256), shmem, ctx->stream, K, D, packed, X, N,
                       range_size, ld, qgrad, bg, tgrad, mapping, map_offset, flags, part, part_m);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    GMMVI_PROF(ctx, "stein_finalize");
    hipLaunchKernelGGL(stein_finalize_kernel, dim3(K), dim3(256), (size_t)D1 * D1 * sizeof(float), ctx->stream, D, R,
                       N, flags, part, part_m, H_neg, g_neg);
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
