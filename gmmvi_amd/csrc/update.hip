// Per-component natural-gradient updates (gmmvi_modules/ng_based_component_updater.py):
//   KL-constrained trust region  :431-524 (bracketing_search :335-429, kl :244-333)   -- SURVEY.md Appendix A.1
//   direct                       :97-141
//   iBLR                         :160-223
// One 64-lane workgroup (a single wavefront) per component; all DxD matrices live in LDS with row stride D+1,
// lane t owns row t (D <= 64).  Control flow is wave-uniform: every decision is taken on values that are
// bitwise identical in all lanes (butterfly reductions / broadcasts), so the bracketing search follows the
// reference's stop rules decision for decision.  A non-positive or non-finite Cholesky pivot takes the
// reference's "NaN in the factor" branch (:320-324, :493).
#include "common.h"
#include "blocked.h"
#include <cfloat>

namespace {

struct Lds {
    int D, ld;
    float *Linv, *Q, *R, *C, *Wk;      // DxD, stride ld
    float *mu, *q, *r, *qn, *mun, *tmp; // D
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ bool wave_any(bool p) { return __any(p) != 0; }

// In-place lower Cholesky of the lower triangle of A (lane t = row t).  Returns false on a non-positive/NaN pivot.
__device__ bool chol_lower(float* A, int D, int ld) {
    const int t = threadIdx.x;
    for (int j = 0; j < D; ++j) {
        float s = 0.f;
        if (t >= j && t < D) {
            s = A[t * ld + j];
            for (int c = 0; c < j; ++c) s = fmaf(-A[t * ld + c], A[j * ld + c], s);
        }
        const float p = __shfl(s, j);
        if (!(p > 0.f) || !(p < FLT_MAX)) return false;
        const float d = sqrtf(p);
        __syncthreads();
        if (t == j) A[t * ld + j] = d;
        else if (t > j && t < D) A[t * ld + j] = s / d;
        __syncthreads();
    }
    return true;
}

// Out = L^-1 for lower-triangular L (lane c = column c).  Out's upper triangle is zero.
__device__ void tri_inverse(const float* L, float* Out, int D, int ld) {
    const int c = threadIdx.x;
    if (c < D) {
        for (int i = 0; i < D; ++i) {
            float s = (i == c) ? 1.f : 0.f;
            for (int j = 0; j < i; ++j) s = fmaf(-L[i * ld + j], Out[j * ld + c], s);   // Out[j][c] = 0 for j < c
            Out[i * ld + c] = (i >= c) ? s / L[i * ld + i] : 0.f;
        }
    }
    __syncthreads();
}

// Out = A^T A for lower-triangular A (lane i = row i): Out[i][j] = sum_{k >= max(i,j)} A[k][i] A[k][j]
__device__ void ata_lower(const float* A, float* Out, int D, int ld) {
    const int i = threadIdx.x;
    if (i < D) {
        for (int j = 0; j < D; ++j) {
            float s = 0.f;
            for (int k = (i > j ? i : j); k < D; ++k) s = fmaf(A[k * ld + i], A[k * ld + j], s);
            Out[i * ld + j] = s;
        }
    }
    __syncthreads();
}

// y = A x (full matrix), lane i = row i
__device__ void matvec(const float* A, const float* x, float* y, int D, int ld) {
    const int i = threadIdx.x;
    if (i < D) {
        float s = 0.f;
        for (int j = 0; j < D; ++j) s = fmaf(A[i * ld + j], x[j], s);
        y[i] = s;
    }
    __syncthreads();
}

// Solve C C^T x = b for lower-triangular C: x overwrites b (column-oriented substitutions with broadcasts).
__device__ void cho_solve_vec(const float* C, float* b, int D, int ld) {
    const int t = threadIdx.x;
    float v = (t < D) ? b[t] : 0.f;
    for (int j = 0; j < D; ++j) {                    // forward: C z = b
        const float zj = __shfl(v, j) / C[j * ld + j];
        if (t == j) v = zj;
        else if (t > j && t < D) v = fmaf(-C[t * ld + j], zj, v);
    }
    for (int j = D - 1; j >= 0; --j) {               // backward: C^T x = z
        const float xj = __shfl(v, j) / C[j * ld + j];
        if (t == j) v = xj;
        else if (t < j) v = fmaf(-C[j * ld + t], xj, v);
    }
    __syncthreads();
    if (t < D) b[t] = v;
    __syncthreads();
}

struct KlResult { float kl; bool ok; };

// kl() of the reference (:244-333, full-covariance branch) for a linear-space eta.  Leaves chol(Q') in s.C and the
// new mean in s.mun.
__device__ KlResult kl_probe(const Lds& s, float eta, float kl_const) {
    const int t = threadIdx.x, D = s.D, ld = s.ld;
    if (t < D) {
        s.qn[t] = (eta * s.q[t] + s.r[t]) / eta;                                       // :302
        for (int j = 0; j <= t; ++j) s.C[t * ld + j] = (eta * s.Q[t * ld + j] + s.R[t * ld + j]) / eta;   // :303
    }
    __syncthreads();
    KlResult res;
    res.ok = chol_lower(s.C, D, ld);                                                   // :319
    if (!res.ok) { res.kl = FLT_MAX; return res; }                                     // :320-324
    if (t < D) s.mun[t] = s.qn[t];
    __syncthreads();
    cho_solve_vec(s.C, s.mun, D, ld);                                                  // :326
    const float logdiag = (t < D) ? __logf(s.C[t * ld + t]) : 0.f;
    const float new_logdet = -2.f * wave_sum(logdiag);                                 // :327
    // trace term ||C^-1 L^-T||_F^2 (:328-329): lane c solves C w = (L^-1)[c, :]^T
    float tr = 0.f;
    if (t < D) {
        for (int i = 0; i < D; ++i) {
            float w = (i <= t) ? s.Linv[t * ld + i] : 0.f;
            for (int j = 0; j < i; ++j) w = fmaf(-s.C[i * ld + j], s.Wk[j * ld + t], w);
            w /= s.C[i * ld + i];
            s.Wk[i * ld + t] = w;
            tr = fmaf(w, w, tr);
        }
    }
    tr = wave_sum(tr);
    float dterm = 0.f;
    if (t < D) {
        float a = 0.f;
        for (int j = 0; j <= t; ++j) a = fmaf(s.Linv[t * ld + j], s.mu[j] - s.mun[j], a);   // :330,:332
        dterm = a * a;
    }
    dterm = wave_sum(dterm);
    __syncthreads();
    res.kl = 0.5f * (kl_const - new_logdet + tr + dterm);                              // :331
    if (!(res.kl == res.kl)) { res.kl = FLT_MAX; res.ok = false; }                      // NaN guard: same reject branch
    return res;
}

__device__ void carve(Lds& s, float* sm, int D) {
    s.D = D; s.ld = D + 1;
    const int m = D * (D + 1);
    s.Linv = sm; s.Q = sm + m; s.R = sm + 2 * m; s.C = sm + 3 * m; s.Wk = sm + 4 * m;
    float* v = sm + 5 * m;
    s.mu = v; s.q = v + D; s.r = v + 2 * D; s.qn = v + 3 * D; s.mun = v + 4 * D; s.tmp = v + 5 * D;
}

__device__ size_t lds_floats(int D) { return (size_t)5 * D * (D + 1) + 6 * D; }

// Loads L, mu, R into LDS and derives Linv, Q = Linv^T Linv, q = Q mu (:457-459).  L is left in s.C.
__device__ void load_component(Lds& s, const float* L, const float* mu, const float* R) {
    const int t = threadIdx.x, D = s.D, ld = s.ld;
    for (int e = t; e < D * D; e += 64) {
        s.C[(e / D) * ld + (e % D)] = L[e];
        s.R[(e / D) * ld + (e % D)] = R[e];
    }
    if (t < D) s.mu[t] = mu[t];
    __syncthreads();
    tri_inverse(s.C, s.Linv, D, ld);
    ata_lower(s.Linv, s.Q, D, ld);
    matvec(s.Q, s.mu, s.q, D, ld);
}

// new covariance Sigma' = Cinv^T Cinv (:486) and its Cholesky factor (:492) from chol(Q') in s.C.
// Result in s.Q (lower triangle).  Returns false when the factorisation fails.
__device__ bool new_chol_from_precision_chol(Lds& s) {
    tri_inverse(s.C, s.Wk, s.D, s.ld);
    ata_lower(s.Wk, s.Q, s.D, s.ld);
    return chol_lower(s.Q, s.D, s.ld);
}

__device__ void store_component(const Lds& s, float* L_out, float* mu_out) {
    const int t = threadIdx.x, D = s.D, ld = s.ld;
    for (int e = t; e < D * D; e += 64) {
        const int i = e / D, j = e % D;
        L_out[e] = (j <= i) ? s.Q[i * ld + j] : 0.f;
    }
    if (t < D) mu_out[t] = s.mun[t];
}

__device__ void finish_bookkeeping(int k, bool success, float l2_init, float* l2, float* num_updates,
                                   int32_t* success_out) {
    if (threadIdx.x == 0) {
        const float old = l2[k];
        l2[k] = success ? fmaxf(0.5f * old, l2_init) : fminf(1e-6f, 10.f * old);        // :520-523 (min on failure)
        num_updates[k] += 1.f;                                                           // :519
        if (success_out) success_out[k] = success ? 1 : 0;
    }
}

__global__ __launch_bounds__(64) void update_kl_kernel(int D, float* __restrict__ means, float* __restrict__ chols,
                                                       const float* __restrict__ H_neg, const float* __restrict__ g_neg,
                                                       const float* __restrict__ stepsizes, float temperature,
                                                       float l2_init, float* __restrict__ last_eta, float* __restrict__ l2,
                                                       float* __restrict__ num_updates, int32_t* __restrict__ success_out,
                                                       float* __restrict__ kl_out, int32_t* __restrict__ nprobes_out) {
    extern __shared__ float sm[];
    Lds s;
    carve(s, sm, D);
    const int k = blockIdx.x, t = threadIdx.x, ld = s.ld;
    float* Lg = chols + (size_t)k * D * D;
    float* mug = means + (size_t)k * D;
    load_component(s, Lg, mug, H_neg + (size_t)k * D * D);
    // reward_lin = R mu - g_neg (:455); kl_const = 2 sum log diag L - D (:456,:460)
    matvec(s.R, s.mu, s.r, D, ld);
    if (t < D) s.r[t] -= g_neg[(size_t)k * D + t];
    const float old_logdet = 2.f * wave_sum(t < D ? __logf(Lg[t * D + t]) : 0.f);
    const float kl_const = old_logdet - (float)D;
    __syncthreads();

    const float eps = stepsizes[k];
    const float last = last_eta[k];
    float lb, ub;
    if (last < 0.f) { lb = -20.f; ub = 80.f; }                                           // :462-466
    else { lb = fmaxf(0.f, __logf(last) - 3.f); ub = __logf(last) + 3.f; }                // :467-471
    float eta = 0.5f * (ub + lb);
    bool ub_ok = false;
    int probes = 0;
    for (int it = 0; it < 1000; ++it) {                                                   // :399
        const float e_eta = expf(eta);
        const float diff = fminf(expf(ub) - e_eta, e_eta - expf(lb));                     // :401
        if (diff < 1e-1f) break;
        const KlResult kr = kl_probe(s, e_eta, kl_const);                                 // :407
        ++probes;
        if (fabsf(eps - kr.kl) < 1e-1f * eps) { lb = ub = eta; break; }                   // :410-413
        if (eps > kr.kl) { ub = eta; ub_ok = true; } else { lb = eta; }                   // :415-419
        eta = 0.5f * (ub + lb);
    }
    if (ub_ok) lb = ub;                                                                   // :423-424
    const float lo = expf(lb), hi = expf(ub);                                             // :427
    const float eta_star = fmaxf(lo, temperature);                                        // :476
    bool success = (lo == hi);                                                            // :478
    float kl_val = -1.f;
    if (success) {
        const KlResult kr = kl_probe(s, eta_star, kl_const);                              // :480-482
        kl_val = kr.kl;
        success = kr.ok && kr.kl < FLT_MAX;                                               // :488
        if (success) success = new_chol_from_precision_chol(s);                           // :486-494
        if (success) {
            // NaN guard on the outputs (the reference's is_nan(new_chol) test, :493)
            bool bad = false;
            if (t < D) {
                bad = !(s.mun[t] == s.mun[t]);
                for (int j = 0; j <= t; ++j) bad |= !(s.Q[t * ld + j] == s.Q[t * ld + j]);
            }
            success = !wave_any(bad);
        }
    }
    if (success) store_component(s, Lg, mug);                                             // :499-504 (else keep old)
    if (t == 0) {
        last_eta[k] = success ? eta_star : -1.f;                                          // :504,:511,:524
        if (kl_out) kl_out[k] = success ? kl_val : -1.f;
        if (nprobes_out) nprobes_out[k] = probes;
    }
    finish_bookkeeping(k, success, l2_init, l2, num_updates, success_out);
}

// mode 0: direct (:97-141); mode 1: iBLR (:160-223)
__global__ __launch_bounds__(64) void update_plain_kernel(int mode, int D, float* __restrict__ means, float* __restrict__ chols,
                                                          const float* __restrict__ H_neg, const float* __restrict__ g_neg,
                                                          const float* __restrict__ stepsizes, float l2_init,
                                                          float* __restrict__ l2, float* __restrict__ num_updates,
                                                          int32_t* __restrict__ success_out) {
    extern __shared__ float sm[];
    Lds s;
    carve(s, sm, D);
    const int k = blockIdx.x, t = threadIdx.x, ld = s.ld;
    float* Lg = chols + (size_t)k * D * D;
    float* mug = means + (size_t)k * D;
    load_component(s, Lg, mug, H_neg + (size_t)k * D * D);
    const float step = stepsizes[k];
    if (mode == 0) {
        // new_lin = q + step (R mu - g);  new_prec = Q + step R
        matvec(s.R, s.mu, s.r, D, ld);
        if (t < D) {
            s.mun[t] = s.q[t] + step * (s.r[t] - g_neg[(size_t)k * D + t]);
            for (int j = 0; j <= t; ++j) s.C[t * ld + j] = s.Q[t * ld + j] + step * s.R[t * ld + j];
        }
        __syncthreads();
    } else {
        // Sigma = L L^T into Wk; T = R Sigma into C... correction = step/2 R Sigma R (:176-177)
        // load L again (load_component left chol in C but tri_inverse/ata did not touch it)
        if (t < D) {
            for (int j = 0; j < D; ++j) {                      // Wk = Sigma (row t)
                float a = 0.f;
                const int m = t < j ? t : j;
                for (int c = 0; c <= m; ++c) a = fmaf(Lg[t * D + c], Lg[j * D + c], a);
                s.Wk[t * ld + j] = a;
            }
        }
        __syncthreads();
        // new mean (:184-192): unchanged on the very first update of this component
        if (t < D) s.tmp[t] = g_neg[(size_t)k * D + t];
        __syncthreads();
        matvec(s.Wk, s.tmp, s.r, D, ld);                        // Sigma g_neg
        if (t < D) s.mun[t] = (num_updates[k] == 0.f) ? s.mu[t] : s.mu[t] - step * s.r[t];   // delta_mean = -g_neg
        // C <- R Sigma (row t), then Q' = Q + step (R + step/2 (R Sigma) R)
        float row[GMMVI_MAX_DIM];
        if (t < D) {
            for (int j = 0; j < D; ++j) {
                float a = 0.f;
                for (int c = 0; c < D; ++c) a = fmaf(s.R[t * ld + c], s.Wk[c * ld + j], a);
                row[j] = a;
            }
        }
        __syncthreads();
        if (t < D) {
            for (int j = 0; j <= t; ++j) {
                float a = 0.f;
                for (int c = 0; c < D; ++c) a = fmaf(row[c], s.R[c * ld + j], a);
                s.C[t * ld + j] = s.Q[t * ld + j] + step * (s.R[t * ld + j] + 0.5f * step * a);
            }
        }
        __syncthreads();
    }
    bool success = chol_lower(s.C, D, ld);
    if (success && mode == 0) cho_solve_vec(s.C, s.mun, D, ld);          // new_mean = new_prec^-1 new_lin (:116)
    if (success) success = new_chol_from_precision_chol(s);              // inv + cholesky (:117-118 / :199-200)
    if (success) {
        bool bad = false;
        if (t < D) {
            bad = !(s.mun[t] == s.mun[t]);
            for (int j = 0; j <= t; ++j) bad |= !(s.Q[t * ld + j] == s.Q[t * ld + j]);
        }
        success = !wave_any(bad);
    }
    if (success) store_component(s, Lg, mug);
    finish_bookkeeping(k, success, l2_init, l2, num_updates, success_out);
}

size_t update_lds_bytes(int D) { return ((size_t)5 * D * (D + 1) + 6 * D) * sizeof(float); }

}  // namespace

extern "C" {

int gmmvi_update_components_kl_reference(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                               const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                               float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                               float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                               int32_t* n_probes_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM);
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && H_neg_dev && g_neg_dev && stepsizes_dev && last_eta_dev && l2_dev &&
                             num_received_updates_dev);
    size_t shmem = update_lds_bytes(D);
    if (shmem > 64 * 1024)
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)update_kl_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    GMMVI_PROF(ctx, "update_kl_reference");
    hipLaunchKernelGGL(update_kl_kernel, dim3(K), dim3(64), shmem, ctx->stream, D, means_dev, chols_dev, H_neg_dev,
                       g_neg_dev, stepsizes_dev, temperature, l2_init, last_eta_dev, l2_dev, num_received_updates_dev,
                       success_out_dev, kl_out_dev, n_probes_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

static int launch_plain(gmmvi_ctx* ctx, int mode, int K, int D, float* means_dev, float* chols_dev,
                        const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev, float l2_init,
                        float* l2_dev, float* num_received_updates_dev, int32_t* success_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED);
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && H_neg_dev && g_neg_dev && stepsizes_dev && l2_dev &&
                             num_received_updates_dev);
    if (gmmvi_is_blocked_dim(D))
        return gmmvi_blocked_update_plain(ctx, mode, K, D, means_dev, chols_dev, H_neg_dev, g_neg_dev, stepsizes_dev, l2_init,
                                          l2_dev, num_received_updates_dev, success_out_dev);
    size_t shmem = update_lds_bytes(D);
    if (shmem > 64 * 1024)
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)update_plain_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    GMMVI_PROF(ctx, "update_plain");
    hipLaunchKernelGGL(update_plain_kernel, dim3(K), dim3(64), shmem, ctx->stream, mode, D, means_dev, chols_dev,
                       H_neg_dev, g_neg_dev, stepsizes_dev, l2_init, l2_dev, num_received_updates_dev, success_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_update_components_direct(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                                   const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                   float l2_init, float* l2_dev, float* num_received_updates_dev,
                                   int32_t* success_out_dev) {
    return launch_plain(ctx, 0, K, D, means_dev, chols_dev, H_neg_dev, g_neg_dev, stepsizes_dev, l2_init, l2_dev,
                        num_received_updates_dev, success_out_dev);
}

int gmmvi_update_components_iblr(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                                 const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                 float l2_init, float* l2_dev, float* num_received_updates_dev,
                                 int32_t* success_out_dev) {
    return launch_plain(ctx, 1, K, D, means_dev, chols_dev, H_neg_dev, g_neg_dev, stepsizes_dev, l2_init, l2_dev,
                        num_received_updates_dev, success_out_dev);
}

}  // extern "C"
