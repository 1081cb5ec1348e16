// Planar n-link robot target and its analytic gradient (target_distributions/planar_robot.py:29-66; the reference
// differentiates it with GradientTape, sample_selector.py:74-77).  GMM and Student-t mixture targets go through
// gmmvi_mixture_eval (density.hip).
#include "common.h"
#include "combine.h"
#include "riders.h"

__global__ void planar_kernel(int D, const float* __restrict__ prior_std, int G, const float* __restrict__ goals,
                              float lik_std, const float* __restrict__ X, int N, float* __restrict__ lp,
                              float* __restrict__ grad, CombineJob carried, Riders riders) {
    if (combine_carried(carried) || riders_carried_prep(riders)) return;   // workgroups past the samples: the merge of the previous sweep, bookkeeping
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* th = X + (size_t)n * D;
    float sinc[GMMVI_MAX_DIM], cosc[GMMVI_MAX_DIM];
    float c = 0.f, px = 0.f, py = 0.f, prior = 0.f;
    for (int i = 0; i < D; ++i) {
        const float t = th[i];
        c += t;
        float s, co;
        sincosf(c, &s, &co);
        sinc[i] = s; cosc[i] = co;
        px += co; py += s;
        const float sd = prior_std[i];
        const float r = t / sd;
        prior += -0.5f * r * r - logf(sd);
    }
    prior -= 0.5f * D * 1.8378770664093453f;
    const float inv_var = 1.f / (lik_std * lik_std);
    float best = -3.0e38f, gx = 0.f, gy = 0.f;
    for (int g = 0; g < G; ++g) {
        const float dx = px - goals[2 * g], dy = py - goals[2 * g + 1];
        const float ll = -0.5f * (dx * dx + dy * dy) * inv_var - 2.f * logf(lik_std) - 1.8378770664093453f;
        if (ll > best) { best = ll; gx = dx; gy = dy; }          // first maximum wins (argmax)
    }
    if (lp) lp[n] = prior + best;
    if (grad) {
        // d px / d theta_j = -sum_{i>=j} sin c_i ; d py / d theta_j = sum_{i>=j} cos c_i
        float ssum = 0.f, csum = 0.f;
        for (int j = D - 1; j >= 0; --j) {
            ssum += sinc[j]; csum += cosc[j];
            const float sd = prior_std[j];
            grad[(size_t)n * D + j] = -th[j] / (sd * sd) - (gx * (-ssum) + gy * csum) * inv_var;
        }
    }
}

extern "C" int gmmvi_target_planar(gmmvi_ctx* ctx, int D, const float* prior_std_dev, int G, const float* goals_dev,
                                   float likelihood_std, const float* X_dev, int N, float* lp_out_dev,
                                   float* grad_out_dev) {
    GMMVI_ARG_CHECK(ctx, D >= 1 && D <= GMMVI_MAX_DIM && G >= 1 && N >= 0 && likelihood_std > 0.f);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, prior_std_dev && goals_dev && X_dev);
    GMMVI_PROF(ctx, "target_planar");
    const Riders riders = gmmvi_take_pending_riders(ctx, (N + 127) / 128, 128, true);
    const CombineJob carried = gmmvi_take_pending_combine(ctx, 128, (N + 127) / 128 + riders.prep_blocks, true);
    hipLaunchKernelGGL(planar_kernel, dim3((N + 127) / 128 + riders.prep_blocks + carried.blocks), dim3(128), 0, ctx->stream, D,
                       prior_std_dev, G, goals_dev, likelihood_std, X_dev, N, lp_out_dev, grad_out_dev, carried, riders);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}
