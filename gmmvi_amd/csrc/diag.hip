// Diagonal-covariance GMMs (models/diagonal_gmm.py:6-59 and every `if diagonal_covs` branch of the hot path).
//
// Layout: chol[K, D] holds sigma (square roots of the covariance diagonal).  Densities, gradients, sampling, the
// background density and the Stein contraction run through the dense kernels on the embedded factor
// L = diag(sigma) (gmmvi_diag_embed: the off-diagonal zeros contribute exact zeros, so the results are the
// diagonal formulas' results); the diagonal of the dense Stein estimate IS the diagonal estimate
// (ng_estimator.py:178-181: h[i] = sum_n w g[n,i] y[n,i]), read out by gmmvi_diag_extract.
// The component updates are elementwise and have their own kernels here:
//   KL-constrained trust region  ng_based_component_updater.py:431-524 with kl() :304-318 (diagonal branch)
//   iBLR                         :160-223 (diagonal branch :170-174, :188-189, :195-197)
// One wavefront per component; lane t owns dimensions t, t + 64, ... (D <= 512: at most 8 per lane, in registers).
// All decisions are taken on wave-uniform values (DPP/shuffle all-reduce), so the bracketing search follows the
// reference's stop rules decision for decision.
#include "common.h"
#include <cfloat>

namespace {

constexpr int kMaxPerLane = GMMVI_MAX_DIM_BLOCKED / 64;

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void diag_embed_kernel(const float* __restrict__ diag, int D, size_t total, float* __restrict__ dense) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const size_t k = e / ((size_t)D * D);
    const int r = (int)(e % ((size_t)D * D));
    const int i = r / D, j = r % D;
    dense[e] = (i == j) ? diag[k * D + i] : 0.f;
}

__global__ void diag_extract_kernel(const float* __restrict__ dense, int D, size_t total, float* __restrict__ diag) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const size_t k = e / D;
    const int i = (int)(e % D);
    diag[e] = dense[k * (size_t)D * D + (size_t)i * D + i];
}

__global__ void reciprocal_kernel(const float* __restrict__ src, size_t n, float* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = 1.f / src[e];
}

struct DiagState {
    float mu[kMaxPerLane], prec[kMaxPerLane], lin[kMaxPerLane], rq[kMaxPerLane], rl[kMaxPerLane], icho[kMaxPerLane];
};

// kl() of ng_based_component_updater.py:299-318 at linear eta; `valid` masks the lane's elements beyond D.
// Returns the KL (NaN when a new precision is negative, as the reference's sqrt/log produce).
template <int R>
__device__ __forceinline__ float diag_kl(const DiagState& s, const bool (&valid)[kMaxPerLane], float eta, int D,
                                         float (&new_mean)[kMaxPerLane], float (&new_prec)[kMaxPerLane]) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (valid[r]) {
            const float nl = (eta * s.lin[r] + s.rl[r]) / eta;                     // :301
            const float np_ = (eta * s.prec[r] + s.rq[r]) / eta;                   // :302
            const float nm = 1.f / np_ * nl;                                       // :306
            const float diff = s.mu[r] - nm;                                       // :308
            a += logf(np_ / s.prec[r]) + s.prec[r] / np_;                          // :314-315
            const float w = s.icho[r] * diff;
            b += w * w;                                                            // :317
            new_mean[r] = nm;
            new_prec[r] = np_;
        }
    }
    a = wave_sum_all(a) - (float)D;
    b = wave_sum_all(b);
    const float inner = (a != a) ? a : fmaxf(0.f, a);                               // tf.maximum propagates NaN
    return 0.5f * (inner + b);
}

template <int R>
__global__ __launch_bounds__(64) void update_diag_kl_kernel(int D, float* __restrict__ means, float* __restrict__ chols,
                                                            const float* __restrict__ h_neg,
                                                            const float* __restrict__ g_neg,
                                                            const float* __restrict__ stepsizes, float temperature,
                                                            float l2_init, float* __restrict__ last_eta,
                                                            float* __restrict__ l2, float* __restrict__ num_updates,
                                                            int32_t* __restrict__ success_out,
                                                            float* __restrict__ kl_out, int32_t* __restrict__ nprobes_out) {
    const int k = blockIdx.x, t = threadIdx.x;
    const size_t base = (size_t)k * D;
    DiagState s;
    bool valid[kMaxPerLane];
    float sigma[kMaxPerLane];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = t + 64 * r;
        valid[r] = d < D;
        if (valid[r]) {
            sigma[r] = chols[base + d];
            s.mu[r] = means[base + d];
            s.rq[r] = h_neg[base + d];
            s.rl[r] = s.rq[r] * s.mu[r] - g_neg[base + d];                         // :448
            s.icho[r] = 1.f / sigma[r];                                            // :450
            s.prec[r] = s.icho[r] * s.icho[r];                                     // :451
            s.lin[r] = s.prec[r] * s.mu[r];                                        // :452
        }
    }
    const float eps = stepsizes[k];
    const float last = last_eta[k];
    float lb, ub;
    if (last < 0.f) { lb = -20.f; ub = 80.f; }                                      // :462-466
    else { lb = fmaxf(0.f, logf(last) - 3.f); ub = logf(last) + 3.f; }              // :467-471
    float eta = 0.5f * (ub + lb);
    bool ub_ok = false;
    int probes = 0;
    float nm[kMaxPerLane], np_[kMaxPerLane];
    for (int it = 0; it < 1000; ++it) {                                             // :399
        const float e_eta = expf(eta);
        const float diff = fminf(expf(ub) - e_eta, e_eta - expf(lb));               // :401
        if (diff < 1e-1f) break;
        const float kl = diag_kl<R>(s, valid, e_eta, D, nm, np_);                    // :407
        ++probes;
        if (fabsf(eps - kl) < 1e-1f * eps) { lb = ub = eta; break; }                // :410-413
        if (eps > kl) { ub = eta; ub_ok = true; } else { lb = eta; }                // :415-419 (NaN: lb = eta)
        eta = 0.5f * (ub + lb);
    }
    if (ub_ok) lb = ub;                                                             // :423-424
    const float lo = expf(lb), hi = expf(ub);                                       // :427
    const float eta_star = fmaxf(lo, temperature);                                  // :476
    bool success = (lo == hi);                                                      // :478
    float kl_val = -1.f;
    if (success) {
        kl_val = diag_kl<R>(s, valid, eta_star, D, nm, np_);                         // :480-482
        success = kl_val < FLT_MAX;                                                 // :488 (false for NaN)
        if (success) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (valid[r]) {
                    const float inv_chol_inv = 1.f / sqrtf(np_[r]);                 // :305,:307
                    chols[base + t + 64 * r] = sqrtf(inv_chol_inv * inv_chol_inv);  // :484, :490
                    means[base + t + 64 * r] = nm[r];
                }
            }
        }
    }
    if (t == 0) {
        last_eta[k] = success ? eta_star : -1.f;                                    // :504,:511,:524
        if (kl_out) kl_out[k] = success ? kl_val : -1.f;
        if (nprobes_out) nprobes_out[k] = probes;
        const float old = l2[k];
        l2[k] = success ? fmaxf(0.5f * old, l2_init) : fminf(1e-6f, 10.f * old);    // :520-523
        num_updates[k] += 1.f;                                                      // :519
        if (success_out) success_out[k] = success ? 1 : 0;
    }
}

// NgBasedComponentUpdaterIblr, diagonal branch (:160-223)
__global__ __launch_bounds__(64) void update_diag_iblr_kernel(int D, float* __restrict__ means, float* __restrict__ chols,
                                                              const float* __restrict__ h_neg,
                                                              const float* __restrict__ g_neg,
                                                              const float* __restrict__ stepsizes, float l2_init,
                                                              float* __restrict__ l2, float* __restrict__ num_updates,
                                                              int32_t* __restrict__ success_out) {
    const int k = blockIdx.x, t = threadIdx.x;
    const size_t base = (size_t)k * D;
    const float step = stepsizes[k];
    const bool first = num_updates[k] == 0.f;
    bool bad = false;
    float nm[kMaxPerLane], nc[kMaxPerLane];
#pragma unroll
    for (int r = 0; r < kMaxPerLane; ++r) {
        const int d = t + 64 * r;
        if (d < D) {
            const float sg = chols[base + d], mu = means[base + d], h = h_neg[base + d];
            const float corr = step / 2.f * h * sg * sg * h;                        // :171-172
            const float icho = 1.f / sg;
            const float prec = icho * icho;                                         // :173-174
            const float dprec = h + corr;                                           // :181
            const float dmean = -g_neg[base + d];                                   // :182
            nm[r] = first ? mu : mu + step * sg * sg * dmean;                       // :184-189
            const float nprec = prec + step * dprec;                                // :194
            nc[r] = sqrtf(1.f / nprec);                                             // :196-197
            bad |= !(nc[r] == nc[r]);                                               // :202
        }
    }
    const bool success = __any(bad) == 0;
    if (success) {
#pragma unroll
        for (int r = 0; r < kMaxPerLane; ++r) {
            const int d = t + 64 * r;
            if (d < D) { means[base + d] = nm[r]; chols[base + d] = nc[r]; }
        }
    }
    if (t == 0) {
        const float old = l2[k];
        l2[k] = success ? fmaxf(0.5f * old, l2_init) : fminf(1e-6f, 10.f * old);    // :217-220
        num_updates[k] += 1.f;                                                      // :223
        if (success_out) success_out[k] = success ? 1 : 0;
    }
}

}  // namespace

extern "C" {

int gmmvi_diag_embed(gmmvi_ctx* ctx, int K, int D, const float* diag_dev, float* dense_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, diag_dev && dense_out_dev);
    const size_t total = (size_t)K * D * D;
    GMMVI_PROF(ctx, "diag_embed");
    hipLaunchKernelGGL(diag_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, diag_dev, D,
                       total, dense_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_diag_extract(gmmvi_ctx* ctx, int K, int D, const float* dense_dev, float* diag_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dense_dev && diag_out_dev);
    const size_t total = (size_t)K * D;
    GMMVI_PROF(ctx, "diag_extract");
    hipLaunchKernelGGL(diag_extract_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dense_dev,
                       D, total, diag_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_reciprocal_f32(gmmvi_ctx* ctx, const float* src_dev, size_t n, float* dst_dev) {
    if (n == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, src_dev && dst_dev);
    GMMVI_PROF(ctx, "reciprocal");
    hipLaunchKernelGGL(reciprocal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, src_dev, n,
                       dst_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_update_components_diag_kl(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_diag_dev,
                                    const float* h_neg_diag_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                    float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                                    float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                                    int32_t* n_probes_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED);
    GMMVI_ARG_CHECK(ctx, means_dev && chols_diag_dev && h_neg_diag_dev && g_neg_dev && stepsizes_dev && last_eta_dev &&
                             l2_dev && num_received_updates_dev);
    GMMVI_PROF(ctx, "update_diag_kl");
    const int per_lane = (D + 63) / 64;
#define GMMVI_DIAG_LAUNCH(R)                                                                                         \
    hipLaunchKernelGGL(update_diag_kl_kernel<R>, dim3(K), dim3(64), 0, ctx->stream, D, means_dev, chols_diag_dev,      \
                       h_neg_diag_dev, g_neg_dev, stepsizes_dev, temperature, l2_init, last_eta_dev, l2_dev,          \
                       num_received_updates_dev, success_out_dev, kl_out_dev, n_probes_out_dev)
    if (per_lane == 1) GMMVI_DIAG_LAUNCH(1);
    else if (per_lane == 2) GMMVI_DIAG_LAUNCH(2);
    else if (per_lane <= 4) GMMVI_DIAG_LAUNCH(4);
    else GMMVI_DIAG_LAUNCH(kMaxPerLane);
#undef GMMVI_DIAG_LAUNCH
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_update_components_diag_iblr(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_diag_dev,
                                      const float* h_neg_diag_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                      float l2_init, float* l2_dev, float* num_received_updates_dev,
                                      int32_t* success_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED);
    GMMVI_ARG_CHECK(ctx, means_dev && chols_diag_dev && h_neg_diag_dev && g_neg_dev && stepsizes_dev && l2_dev &&
                             num_received_updates_dev);
    GMMVI_PROF(ctx, "update_diag_iblr");
    hipLaunchKernelGGL(update_diag_iblr_kernel, dim3(K), dim3(64), 0, ctx->stream, D, means_dev, chols_diag_dev,
                       h_neg_diag_dev, g_neg_dev, stepsizes_dev, l2_init, l2_dev, num_received_updates_dev,
                       success_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // extern "C"
