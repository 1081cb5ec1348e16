// Expected log-ratios / component rewards (gmmvi_modules/weight_updater.py:56-75), the categorical trust-region
// weight update (:164-279), the direct update (:123-141) and the improvement-based
// stepsize rules (component_stepsize_adaptation.py:165-188, weight_stepsize_adaptation.py:141-156).
#include "common.h"
#include "riders.h"
#include "stepsize_rules.h"
#include "wave_reduce.h"
#include "combine.h"
#include <cfloat>

namespace {

__device__ __forceinline__ float wsum(float v) { return gmmvi_wave_sum(v); }
__device__ __forceinline__ float wmax(float v) { return gmmvi_wave_max(v); }

// block-wide reductions for 256 threads (4 waves); result identical in every thread
__device__ float block_max(float v, float* red) {
    v = wmax(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ float block_sum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// the per-thread pass of elr_kernel over samples n = t, t + 1024, ...: online softmax sums of a_n = ld[k,n] - bg[n] with
// rho_n = tlp[n] - beta logq[n].  R: number of chunk partials in logq ([R][N], merged as combine.h does), 0 = logq is final,
// -1 = run-time count Rrt.  R is a template parameter so that the loads of a round are straight-line code: a run-time choice
// per element makes the compiler branch around every load and wait for each one separately.
template <int R>
__device__ __forceinline__ void elr_accumulate(int N, const float* __restrict__ row, const float* __restrict__ bg,
                                               const float* __restrict__ tlp, const float* __restrict__ logq, int Rrt, float beta,
                                               float& m, float& s, float& se, float& s2) {
    constexpr int B = 4;                               // samples per round, loads in flight together
    for (int n0 = threadIdx.x; n0 < N; n0 += 1024 * B) {
        float av[B], rv[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int n = min(n0 + 1024 * u, N - 1);
            av[u] = row[n] - bg[n];
            float lq;
            if constexpr (R == 0) lq = logq[n];
            else if constexpr (R > 0) lq = combine_log_values_n<R>(logq, N, n);
            else lq = combine_log_values(logq, Rrt, N, n);
            rv[u] = tlp[n] - beta * lq;
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (n0 + 1024 * u < N) {
                const float a = av[u], rho = rv[u];
                const float mn = fmaxf(m, a);
                const float sc = __expf(m - mn), e = __expf(a - mn);
                s = fmaf(s, sc, e);
                se = fmaf(se, sc, e * rho);
                s2 = fmaf(s2, sc * sc, e * e);
                m = mn;
            }
        }
    }
}

// One 1024-thread workgroup per component: E_k = sum_n softmax_n(ld[k,n] - bg[n]) * (tlp[n] - beta logq[n]) in a single
// pass (per-thread running maximum with rescaling), then a fixed-order tree over the 16 waves.
// logq_R > 0: logq holds the logq_R chunk partials [logq_R][N] of a component-split sweep, merged here as combine.h does.
__device__ __forceinline__ void elr_body(int N, const float* __restrict__ ld, const float* __restrict__ bg,
                                         const float* __restrict__ tlp, const float* __restrict__ logq, int logq_R,
                                         float beta, const float* __restrict__ logw, int self_normalized,
                                         float* __restrict__ E_out, float* __restrict__ reward_out,
                                         float* __restrict__ ess_out) {
    __shared__ float red[4][16];
    const int k = blockIdx.x;
    const float* row = ld + (size_t)k * N;
    float m = -3.0e38f, s = 0.f, se = 0.f, s2 = 0.f;
    switch (logq_R) {
        case 0: elr_accumulate<0>(N, row, bg, tlp, logq, 0, beta, m, s, se, s2); break;
        case 2: elr_accumulate<2>(N, row, bg, tlp, logq, 2, beta, m, s, se, s2); break;
        case 3: elr_accumulate<3>(N, row, bg, tlp, logq, 3, beta, m, s, se, s2); break;
        case 4: elr_accumulate<4>(N, row, bg, tlp, logq, 4, beta, m, s, se, s2); break;
        case 5: elr_accumulate<5>(N, row, bg, tlp, logq, 5, beta, m, s, se, s2); break;
        case 6: elr_accumulate<6>(N, row, bg, tlp, logq, 6, beta, m, s, se, s2); break;
        case 7: elr_accumulate<7>(N, row, bg, tlp, logq, 7, beta, m, s, se, s2); break;
        case 8: elr_accumulate<8>(N, row, bg, tlp, logq, 8, beta, m, s, se, s2); break;
        case 10: elr_accumulate<10>(N, row, bg, tlp, logq, 10, beta, m, s, se, s2); break;
        case 12: elr_accumulate<12>(N, row, bg, tlp, logq, 12, beta, m, s, se, s2); break;
        case 16: elr_accumulate<16>(N, row, bg, tlp, logq, 16, beta, m, s, se, s2); break;
        default: elr_accumulate<-1>(N, row, bg, tlp, logq, logq_R, beta, m, s, se, s2); break;
    }
    // wave level
    const float mw = wmax(m);
    const float f = __expf(m - mw);
    s = wsum(s * f); se = wsum(se * f); s2 = wsum(s2 * f * f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = mw; red[1][wave] = s; red[2][wave] = se; red[3][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = -3.0e38f;
        for (int w = 0; w < 16; ++w) M = fmaxf(M, red[0][w]);
        float S = 0.f, SE = 0.f, S2 = 0.f;
        for (int w = 0; w < 16; ++w) {
            const float g = __expf(red[0][w] - M);
            S = fmaf(red[1][w], g, S); SE = fmaf(red[2][w], g, SE); S2 = fmaf(red[3][w], g * g, S2);
        }
        // plain importance weights (:66-71): 1/N sum_n exp(ld - bg) rho_n
        const float E = self_normalized ? SE / S : SE * __expf(M) / (float)N;
        if (E_out) E_out[k] = E;
        if (reward_out) reward_out[k] = beta * logw[k] + E;                              // :73
        if (ess_out) ess_out[k] = (S * S) / S2;                                          // sample_selector.py:154-158
    }
}

__global__ __launch_bounds__(1024) void elr_kernel(int N, const float* __restrict__ ld, const float* __restrict__ bg,
                                                   const float* __restrict__ tlp, const float* __restrict__ logq, int logq_R,
                                                   float beta, const float* __restrict__ logw, int self_normalized,
                                                   float* __restrict__ E_out, float* __restrict__ reward_out,
                                                   float* __restrict__ ess_out) {
    elr_body(N, ld, bg, tlp, logq, logq_R, beta, logw, self_normalized, E_out, reward_out, ess_out);
}

// the same launch carrying riders (riders.h): the single-call iteration's draw of the NEXT iteration's samples runs on the
// CUs the K workgroups of this launch leave idle (K = 100: 156 of 256)
template <int DP>
__global__ __launch_bounds__(1024) void elr_riders_kernel(int N, const float* __restrict__ ld, const float* __restrict__ bg,
                                                          const float* __restrict__ tlp, const float* __restrict__ logq,
                                                          int logq_R, float beta, const float* __restrict__ logw,
                                                          int self_normalized, float* __restrict__ E_out,
                                                          float* __restrict__ reward_out, float* __restrict__ ess_out,
                                                          Riders riders) {
    extern __shared__ __align__(16) float sm_riders[];
    if (riders_carried<DP>(riders, sm_riders)) return;
    elr_body(N, ld, bg, tlp, logq, logq_R, beta, logw, self_normalized, E_out, reward_out, ess_out);
}

// log-sum-exp over K values held in LDS by one wavefront
__device__ float wave_lse(const float* v, int K) {
    const int t = threadIdx.x;
    float m = -3.0e38f;
    for (int i = t; i < K; i += 64) m = fmaxf(m, v[i]);
    m = wmax(m);
    float s = 0.f;
    for (int i = t; i < K; i += 64) s += __expf(v[i] - m);
    s = wsum(s);
    return m + __logf(s);
}

// kl() of TrustRegionBasedWeightUpdater (:164-191): writes new log weights into nl, returns KL(new || old).
__device__ float weights_kl(float eta, float beta, const float* lw, const float* E, float* nl, int K) {
    const int t = threadIdx.x;
    const float a = (eta + 1.f) / (beta + eta), b = 1.f / (beta + eta);
    for (int i = t; i < K; i += 64) nl[i] = a * lw[i] + b * E[i];                         // :184-185
    __syncthreads();
    float l = wave_lse(nl, K);
    for (int i = t; i < K; i += 64) nl[i] = fmaxf(nl[i] - l, -69.07f);                    // :186-187
    __syncthreads();
    l = wave_lse(nl, K);
    float kl = 0.f;
    for (int i = t; i < K; i += 64) {
        const float v = nl[i] - l;                                                        // :188
        nl[i] = v;
        kl = fmaf(__expf(v), v - lw[i], kl);                                              // :190
    }
    __syncthreads();
    return wsum(kl);
}

// weights_kl for K <= 64 NE with the lane's NE elements (i = lane, lane + 64, ...) in registers: the same operations in the
// same order as the LDS form (every lane reduces its strided elements first, then the wave), without the LDS round trips and
// barriers of up to 50 probes.  v[] returns the new log weights of the lane's elements.
template <int NE>
__device__ __forceinline__ float weights_kl_reg(float eta, float beta, const float (&lw)[NE], const float (&E)[NE],
                                               const bool (&h)[NE], float (&v)[NE]) {
    const float a = (eta + 1.f) / (beta + eta), b = 1.f / (beta + eta);
    float nv[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) nv[e] = a * lw[e] + b * E[e];                            // :184-185
    auto lse = [&](const float (&x)[NE]) {
        float m = -3.0e38f;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (h[e]) m = fmaxf(m, x[e]);
        m = wmax(m);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (h[e]) s += __expf(x[e] - m);
        s = wsum(s);
        return m + __logf(s);
    };
    float l = lse(nv);
#pragma unroll
    for (int e = 0; e < NE; ++e) nv[e] = fmaxf(nv[e] - l, -69.07f);                       // :186-187
    l = lse(nv);
    float kl = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        v[e] = nv[e] - l;                                                                 // :188
        if (h[e]) kl = fmaf(__expf(v[e]), v[e] - lw[e], kl);                              // :190
    }
    return wsum(kl);
}

// the bisection of the trust-region update (:232-260) with the lane's elements in registers
template <int NE>
__device__ __forceinline__ void weights_search_reg(int K, float beta, float bound, const float* lw_s, const float* E_s, float* nl,
                                                   float& kl, float& eta, bool& updated) {
    const int t = threadIdx.x;
    float lw[NE], E[NE], v[NE];
    bool h[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        h[e] = t + 64 * e < K;
        lw[e] = h[e] ? lw_s[t + 64 * e] : 0.f;
        E[e] = h[e] ? E_s[t + 64 * e] : 0.f;
        v[e] = lw[e];                                                                     // nl starts as the old weights
    }
    float lb = -45.f, ub = 45.f;                                                          // :276-277
    float log_eta = 0.5f * (ub + lb);
    bool ub_ok = false;
    for (int it = 0; it < 50; ++it) {                                                     // :232
        eta = expf(log_eta);
        if (fabsf(expf(ub) - expf(lb)) < 1e-1f) break;                                    // :234-236
        kl = weights_kl_reg<NE>(eta, beta, lw, E, h, v);                                  // :238
        if (fabsf(bound - kl) < 1e-1f * bound) { lb = ub; break; }                        // :240-243
        if (bound > kl) { ub = log_eta; ub_ok = true; } else { lb = log_eta; }            // :245-249
        log_eta = 0.5f * (ub + lb);
    }
    if (lb == ub) {
        // :252-253 keep the last evaluated weights
    } else if (ub_ok) {
        eta = expf(ub);
        kl = weights_kl_reg<NE>(eta, beta, lw, E, h, v);                                  // :256-258
    } else {
        updated = false;                                                                  // :260
        kl = -1.f; eta = -1.f;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if (h[e]) nl[t + 64 * e] = v[e];                                                  // hand the result to the LDS image
    __syncthreads();
}

// mode 0: trust region (:193-279); mode 1: direct (:123-141).  Single wavefront; lw/E/nl live in LDS.
__global__ __launch_bounds__(64) void update_weights_kernel(int mode, int K, float* __restrict__ logw,
                                                            const float* __restrict__ E_in, const float* __restrict__ stepsize,
                                                            float beta, float* __restrict__ kl_eta_out,
                                                            float* __restrict__ exp_out) {
    extern __shared__ float sm[];
    float* lw = sm; float* E = sm + K; float* nl = sm + 2 * K;
    const int t = threadIdx.x;
    for (int i = t; i < K; i += 64) { lw[i] = logw[i]; E[i] = E_in[i]; nl[i] = logw[i]; }
    __syncthreads();
    if (K <= 1) return;                                                                   // :136 / :275
    const float bound = stepsize[0];
    float kl = -1.f, eta = -1.f;
    bool updated = true;
    if (mode == 1) {
        for (int i = t; i < K; i += 64) nl[i] = lw[i] + bound / beta * E[i];              // :137
        __syncthreads();
        float l = wave_lse(nl, K);
        for (int i = t; i < K; i += 64) nl[i] = fmaxf(nl[i] - l, -69.07f);                // :138-139
        __syncthreads();
        l = wave_lse(nl, K);
        for (int i = t; i < K; i += 64) nl[i] -= l;                                       // :140
        __syncthreads();
    } else {
        if (K <= 128) {
            weights_search_reg<2>(K, beta, bound, lw, E, nl, kl, eta, updated);
        } else if (K <= 256) {
            weights_search_reg<4>(K, beta, bound, lw, E, nl, kl, eta, updated);
        } else if (K <= 512) {                                                              // (adaptive runs grow there: the LDS form
            weights_search_reg<8>(K, beta, bound, lw, E, nl, kl, eta, updated);             //  below took 45 us at K = 290)
        } else if (K <= 1024) {
            weights_search_reg<16>(K, beta, bound, lw, E, nl, kl, eta, updated);
        } else {
            float lb = -45.f, ub = 45.f;                                                  // :276-277
            float log_eta = 0.5f * (ub + lb);
            bool ub_ok = false;
            for (int it = 0; it < 50; ++it) {                                             // :232
                eta = expf(log_eta);
                if (fabsf(expf(ub) - expf(lb)) < 1e-1f) break;                            // :234-236
                kl = weights_kl(eta, beta, lw, E, nl, K);                                 // :238
                if (fabsf(bound - kl) < 1e-1f * bound) { lb = ub; break; }                // :240-243
                if (bound > kl) { ub = log_eta; ub_ok = true; } else { lb = log_eta; }    // :245-249
                log_eta = 0.5f * (ub + lb);
            }
            if (lb == ub) {
                // :252-253 keep the last evaluated weights
            } else if (ub_ok) {
                eta = expf(ub);
                kl = weights_kl(eta, beta, lw, E, nl, K);                                 // :256-258
            } else {
                updated = false;                                                          // :260
                kl = -1.f; eta = -1.f;
            }
        }
    }
    if (!updated) {
        for (int i = t; i < K; i += 64) nl[i] = lw[i];
        __syncthreads();
    }
    // GMM.replace_weights (models/gmm.py:181): renormalise
    const float l = wave_lse(nl, K);
    for (int i = t; i < K; i += 64) {
        const float v = nl[i] - l;
        logw[i] = v;
        if (exp_out) exp_out[i] = expf(v);                  // GmmWrapper.replace_weights: weight history column (gmm_wrapper.py:182)
    }
    if (t == 0 && kl_eta_out) { kl_eta_out[0] = kl; kl_eta_out[1] = eta; }
}

__global__ void component_stepsize_kernel(int K, float* __restrict__ stepsizes, const float* __restrict__ prev,
                                          const float* __restrict__ last, float mn, float mx, float inc, float dec) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    stepsizes[k] = component_stepsize_rule(stepsizes[k], prev[k], last[k], mn, mx, inc, dec);
}

// ELBO proxy sum_k w_k R_k - sum_k w_k log w_k accumulated in fp64 and rounded to fp32 before the comparison
// (DESIGN.md quirk Q-elbo: keeps the float32.min sentinel of a fresh reward history from flipping the first test).
__global__ __launch_bounds__(64) void weight_stepsize_kernel(int K, const float* __restrict__ logw,
                                                             const float* __restrict__ rewards_last, float* __restrict__ state,
                                                             float mn, float mx, float inc, float dec) {
    weight_stepsize_wave(K, logw, rewards_last, state, mn, mx, inc, dec, threadIdx.x);
}

}  // namespace

extern "C" {

int gmmvi_expected_log_ratios(gmmvi_ctx* ctx, int K, int N, const float* ld_dev, const float* bg_dev,
                              const float* tlp_dev, const float* logq_dev, float beta, const float* logw_dev,
                              int self_normalized, float* E_out_dev, float* reward_out_dev, float* ess_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && N >= 1);
    GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev && tlp_dev && logq_dev && logw_dev);
    // a deferred merge (single-call iteration): log values of the post-update sweep are merged while they are read; anything
    // else still pending is launched first
    int logq_R = 0;
    {
        const CombineJob& j = ctx->pending;
        if (j.R > 0 && j.lp_out == logq_dev && j.N == N && !j.grad_out && !j.lp2_out) {
            logq_R = j.R;
            logq_dev = j.lp_parts;
            ctx->pending = CombineJob();
        } else {
            int rc = gmmvi_flush_pending_combine(ctx);
            if (rc != GMMVI_OK) return rc;
        }
    }
    GMMVI_PROF(ctx, "expected_log_ratios");
    const Riders riders = gmmvi_take_pending_riders(ctx, K, 1024);
    if (riders.prep_blocks + riders.sample_blocks > 0) {
        const int dp = gmmvi_padded_dim(riders.sample_blocks > 0 ? riders.sample.D : 2);
        GMMVI_DISPATCH_DP(dp, hipLaunchKernelGGL((elr_riders_kernel<DP>), dim3(K + riders.prep_blocks + riders.sample_blocks),
                                                 dim3(1024), riders_lds_bytes(riders), ctx->stream, N, ld_dev, bg_dev, tlp_dev,
                                                 logq_dev, logq_R, beta, logw_dev, self_normalized, E_out_dev, reward_out_dev,
                                                 ess_out_dev, riders));
    } else {
        hipLaunchKernelGGL(elr_kernel, dim3(K), dim3(1024), 0, ctx->stream, N, ld_dev, bg_dev, tlp_dev, logq_dev, logq_R, beta,
                           logw_dev, self_normalized, E_out_dev, reward_out_dev, ess_out_dev);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // extern "C"

// update_weights with the weight-history column written by the same kernel (used by fused.hip); C++ linkage, not exported
int gmmvi_update_weights_internal(gmmvi_ctx* ctx, int mode, int K, float* logw_dev, const float* E_dev,
                                  const float* stepsize_dev, float beta, float* kl_eta_out_dev, float* exp_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && K <= 4096);
    GMMVI_ARG_CHECK(ctx, logw_dev && E_dev && stepsize_dev);
    GMMVI_PROF(ctx, "update_weights");
    hipLaunchKernelGGL(update_weights_kernel, dim3(1), dim3(64), (size_t)3 * K * sizeof(float), ctx->stream, mode, K,
                       logw_dev, E_dev, stepsize_dev, beta, kl_eta_out_dev, exp_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

extern "C" {

int gmmvi_update_weights_kl(gmmvi_ctx* ctx, int K, float* logw_dev, const float* E_dev, const float* stepsize_dev,
                            float beta, float* kl_eta_out_dev) {
    return gmmvi_update_weights_internal(ctx, 0, K, logw_dev, E_dev, stepsize_dev, beta, kl_eta_out_dev, nullptr);
}

int gmmvi_update_weights_direct(gmmvi_ctx* ctx, int K, float* logw_dev, const float* E_dev,
                                const float* stepsize_dev, float beta) {
    return gmmvi_update_weights_internal(ctx, 1, K, logw_dev, E_dev, stepsize_dev, beta, nullptr, nullptr);
}

int gmmvi_component_stepsize_improvement(gmmvi_ctx* ctx, int K, float* stepsizes_dev, const float* rewards_prev_dev,
                                         const float* rewards_last_dev, float min_stepsize, float max_stepsize,
                                         float inc_factor, float dec_factor) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && stepsizes_dev && rewards_prev_dev && rewards_last_dev);
    GMMVI_PROF(ctx, "component_stepsize");
    hipLaunchKernelGGL(component_stepsize_kernel, dim3((K + 127) / 128), dim3(128), 0, ctx->stream, K, stepsizes_dev,
                       rewards_prev_dev, rewards_last_dev, min_stepsize, max_stepsize, inc_factor, dec_factor);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_weight_stepsize_improvement(gmmvi_ctx* ctx, int K, const float* logw_dev, const float* rewards_last_dev,
                                      float* state_dev, float min_stepsize, float max_stepsize, float inc_factor,
                                      float dec_factor) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && logw_dev && rewards_last_dev && state_dev);
    GMMVI_PROF(ctx, "weight_stepsize");
    hipLaunchKernelGGL(weight_stepsize_kernel, dim3(1), dim3(64), 0, ctx->stream, K, logw_dev, rewards_last_dev,
                       state_dev, min_stepsize, max_stepsize, inc_factor, dec_factor);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // extern "C"
