// Component packing and the fused {sample tile x component} density / log-sum-exp / gradient kernel.
//
// Mapping (DESIGN.md "mixture_eval"): one lane owns one sample (x, z, y and the running gradient live in VGPRs),
// one wave walks a strided subset of the components, a workgroup = 64 samples x W waves.  The component
// parameters are wave-uniform, so they are fetched with scalar loads (s_load_dwordxN through the constant
// cache) and feed v_fma as SGPR operands: no LDS traffic and no per-lane loads in the inner loop.  The triangular
// solve is fully unrolled for the padded dimension DP; partial (max, sum, gradient) of the W waves are merged
// through LDS.
#include "common.h"
#include <cmath>

// ---------------------------------------------------------------------------------------------------------------
// pack: (means, chols) -> kernel-side blocks; optional explicit inverse (sample_db.py:121)
// ---------------------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(64) void pack_kernel(int family, float nu, int K, int D, const float* __restrict__ means,
                                                  const float* __restrict__ chols, float* __restrict__ packed,
                                                  float* __restrict__ inv_chols) {
    using P = Pack<DP>;
    const int k = blockIdx.x;
    const int t = threadIdx.x;
    const float* L = chols + (size_t)k * D * D;
    float* out = packed + (size_t)k * P::STRIDE;
    for (int i = t; i < DP; i += 64) {
        out[P::MU + i] = i < D ? means[(size_t)k * D + i] : 0.f;
        out[P::RD + i] = i < D ? 1.f / L[i * D + i] : 1.f;
    }
    for (int e = t; e < DP * DP; e += 64) {
        int i = e / DP, j = e % DP;
        if (j < i) {
            float v = (i < D) ? L[i * D + j] : 0.f;
            out[P::LROW + P::rowofs(i) + j] = v;
            out[P::LCOL + P::colofs(j) + (i - j - 1)] = v;      // entry (row i, col j) lives in column j
        }
    }
    if (t == 0) {
        float s = 0.f;
        for (int i = 0; i < D; ++i) s += logf(L[i * D + i]);
        float c;
        if (family == GMMVI_GAUSS)
            c = -s - 0.5f * D * 1.8378770664093453f;             // log(2 pi)
        else
            c = lgammaf(0.5f * (nu + D)) - lgammaf(0.5f * nu) - 0.5f * D * logf(nu * 3.14159265358979f) - s;
        out[P::CONST] = c;
        for (int i = P::CONST + 1; i < P::STRIDE; ++i) out[i] = 0.f;
    }
    if (inv_chols != nullptr && t < D) {
        // column t of L^-1 by forward substitution: L x = e_t
        float* inv = inv_chols + (size_t)k * D * D;
        float x[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            float s = (i == t) ? 1.f : 0.f;
            if (i < D) {
                for (int j = 0; j < i; ++j) s -= L[i * D + j] * x[j];
                x[i] = s / L[i * D + i];
            } else {
                x[i] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < D) inv[i * D + t] = (i >= t) ? x[i] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// batched Cholesky for model construction / add_component (full_cov_gmm.py:23,:64-68)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cholesky_kernel(int D, const float* __restrict__ covs, float* __restrict__ chols,
                                                      int32_t* __restrict__ ok) {
    extern __shared__ float sm[];
    const int k = blockIdx.x, t = threadIdx.x;
    const int ld = D + 1;
    for (int e = t; e < D * D; e += 64) sm[(e / D) * ld + (e % D)] = covs[(size_t)k * D * D + e];
    __syncthreads();
    bool good = true;
    for (int j = 0; j < D; ++j) {
        float s = 0.f;
        if (t >= j && t < D) {
            s = sm[t * ld + j];
            for (int c = 0; c < j; ++c) s -= sm[t * ld + c] * sm[j * ld + c];
        }
        float p = __shfl(s, j);
        if (!(p > 0.f)) { good = false; break; }
        float d = sqrtf(p);
        __syncthreads();
        if (t == j) sm[t * ld + j] = d;
        else if (t > j && t < D) sm[t * ld + j] = s / d;
        __syncthreads();
    }
    for (int e = t; e < D * D; e += 64) {
        int i = e / D, j = e % D;
        float v = (j <= i) ? sm[i * ld + j] : 0.f;
        chols[(size_t)k * D * D + e] = good ? v : __builtin_nanf("");
    }
    if (t == 0 && ok) ok[k] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// mixture_eval
// ---------------------------------------------------------------------------------------------------------------
template <int DP>
__device__ __forceinline__ void forward_subst(const float* __restrict__ P, const float (&x)[DP], float (&z)[DP], float& q) {
    using PK = Pack<DP>;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        float t = x[i] - P[PK::MU + i];
#pragma unroll
        for (int j = 0; j < i; ++j) t = fmaf(-P[PK::LROW + PK::rowofs(i) + j], z[j], t);
        z[i] = t * P[PK::RD + i];
        q = fmaf(z[i], z[i], q);
    }
}

// y = L^-T z  (Sigma^-1 (x - mu) when z = L^-1 (x - mu))
template <int DP>
__device__ __forceinline__ void backward_subst(const float* __restrict__ P, const float (&z)[DP], float (&y)[DP]) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = DP - 1; i >= 0; --i) {
        float t = z[i];
#pragma unroll
        for (int j = i + 1; j < DP; ++j) t = fmaf(-P[PK::LCOL + PK::colofs(i) + (j - i - 1)], y[j], t);
        y[i] = t * P[PK::RD + i];
    }
}

template <int DP, int FAMILY, bool GRAD>
__global__ __launch_bounds__(1024) void mixture_eval_kernel(float nu, int K, int D, const float* __restrict__ packed,
                                                            const float* __restrict__ logw, const float* __restrict__ X,
                                                            int N, float* __restrict__ ld_out, float* __restrict__ lp_out,
                                                            float* __restrict__ grad_out) {
    using PK = Pack<DP>;
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const bool valid = n < N;

    float x[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) x[i] = (valid && i < D) ? X[(size_t)n * D + i] : 0.f;

    float m = -3.0e38f, s = 0.f;
    float acc[GRAD ? DP : 1];
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = 0.f;
    }
    const float nud = nu + (float)D;

    for (int k = wave; k < K; k += nwaves) {
        const float* __restrict__ P = packed + (size_t)k * PK::STRIDE;
        float z[DP], q;
        forward_subst<DP>(P, x, z, q);
        float ld, coef;
        if (FAMILY == GMMVI_GAUSS) {
            ld = fmaf(-0.5f, q, P[PK::CONST]);
            coef = -1.f;
        } else {
            ld = P[PK::CONST] - 0.5f * nud * log1pf(q / nu);
            coef = -nud / (nu + q);
        }
        if (ld_out != nullptr && valid) ld_out[(size_t)k * N + n] = ld;
        const float a = ld + logw[k];
        const float mn = fmaxf(m, a);
        const float sc = __expf(m - mn);
        const float e = __expf(a - mn);
        s = fmaf(s, sc, e);
        m = mn;
        if (GRAD) {
            float y[DP];
            backward_subst<DP>(P, z, y);
            const float ec = e * coef;
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = fmaf(acc[i], sc, ec * y[i]);
        }
    }
    if (lp_out == nullptr && !GRAD) return;

    // merge the W waves' partials: sm_m[w][lane], sm_s[w][lane], sm_acc[w][i][lane]
    float* sm_m = sm;
    float* sm_s = sm + nwaves * 64;
    float* sm_acc = sm + 2 * nwaves * 64;
    sm_m[wave * 64 + lane] = m;
    sm_s[wave * 64 + lane] = s;
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < DP; ++i) sm_acc[(wave * DP + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    float M = -3.0e38f;
    for (int w = 0; w < nwaves; ++w) M = fmaxf(M, sm_m[w * 64 + lane]);
    float S = 0.f;
    for (int w = 0; w < nwaves; ++w) S += sm_s[w * 64 + lane] * __expf(sm_m[w * 64 + lane] - M);
    if (wave == 0 && valid && lp_out != nullptr) lp_out[n] = M + __logf(S);
    if (GRAD && grad_out != nullptr) {
        const float inv = 1.f / S;
        for (int i = wave; i < D; i += nwaves) {
            float g = 0.f;
            for (int w = 0; w < nwaves; ++w) g += sm_acc[(w * DP + i) * 64 + lane] * __expf(sm_m[w * 64 + lane] - M);
            if (valid) grad_out[(size_t)n * D + i] = g * inv;
        }
    }
}

template <int DP>
static int launch_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed,
                               const float* logw, const float* X, int N, float* ld, float* lp, float* grad) {
    const bool want_grad = grad != nullptr;
    // waves per workgroup: enough to spread K, bounded by the LDS needed for the merge (<= 64 KiB) and 1024 threads
    int per_wave_floats = 64 * ((want_grad ? DP : 0) + 2);
    int max_w = (64 * 1024) / (per_wave_floats * 4);
    int nw = K < 16 ? K : 16;
    if (nw > max_w) nw = max_w;
    if (nw < 1) nw = 1;
    // plenty of sample tiles: fewer waves per tile keeps more tiles resident per CU
    int tiles = (N + 63) / 64;
    while (nw > 4 && (long)tiles * nw > 8L * 4 * ctx->num_cus) nw >>= 1;
    size_t shmem = (size_t)nw * per_wave_floats * 4;
    dim3 grid(tiles), block(nw * 64);
    GMMVI_PROF(ctx, want_grad ? "mixture_eval_grad" : "mixture_eval");
#define GMMVI_LAUNCH_ME(FAM, G)                                                                              \
    hipLaunchKernelGGL((mixture_eval_kernel<DP, FAM, G>), grid, block, shmem, ctx->stream, nu, K, D, packed, \
                       logw, X, N, ld, lp, grad)
    if (family == GMMVI_GAUSS) {
        if (want_grad) GMMVI_LAUNCH_ME(GMMVI_GAUSS, true); else GMMVI_LAUNCH_ME(GMMVI_GAUSS, false);
    } else {
        if (want_grad) GMMVI_LAUNCH_ME(GMMVI_STUDENT_T, true); else GMMVI_LAUNCH_ME(GMMVI_STUDENT_T, false);
    }
#undef GMMVI_LAUNCH_ME
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

extern "C" {

int gmmvi_pack_components(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* means_dev,
                          const float* chols_dev, float* packed_dev, float* inv_chols_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_MAX_DIM);
    GMMVI_ARG_CHECK(ctx, family == GMMVI_GAUSS || family == GMMVI_STUDENT_T);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && packed_dev);
    int dp = gmmvi_padded_dim(D);
    GMMVI_PROF(ctx, "pack_components");
    GMMVI_DISPATCH_DP(dp, hipLaunchKernelGGL((pack_kernel<DP>), dim3(K), dim3(64), 0, ctx->stream, family, nu, K, D,
                                             means_dev, chols_dev, packed_dev, inv_chols_dev));
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_cholesky(gmmvi_ctx* ctx, int K, int D, const float* covs_dev, float* chols_dev, int32_t* ok_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_MAX_DIM);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, covs_dev && chols_dev);
    hipLaunchKernelGGL(cholesky_kernel, dim3(K), dim3(64), (size_t)D * (D + 1) * 4, ctx->stream, D, covs_dev,
                       chols_dev, ok_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed_dev,
                       const float* logw_dev, const float* X_dev, int N, float* ld_out_dev, float* lp_out_dev,
                       float* grad_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM && N >= 0);
    GMMVI_ARG_CHECK(ctx, family == GMMVI_GAUSS || family == GMMVI_STUDENT_T);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, packed_dev && logw_dev && X_dev);
    GMMVI_ARG_CHECK(ctx, ld_out_dev || lp_out_dev || grad_out_dev);
    int dp = gmmvi_padded_dim(D);
    GMMVI_DISPATCH_DP(dp, return launch_mixture_eval<DP>(ctx, family, nu, K, D, packed_dev, logw_dev, X_dev, N,
                                                         ld_out_dev, lp_out_dev, grad_out_dev));
    return GMMVI_OK;
}

}  // extern "C"
