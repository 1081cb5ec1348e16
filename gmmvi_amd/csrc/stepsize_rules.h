// Device-side stepsize rules shared by the stand-alone kernels (weights.hip) and the single-call iteration (fused.hip),
// so that both paths execute the same arithmetic.
#pragma once
#include <hip/hip_runtime.h>

// ImprovementBasedComponentStepsizeAdaptation.update_stepsize (component_stepsize_adaptation.py:177-186)
__device__ __forceinline__ float component_stepsize_rule(float cur, float prev, float last, float mn, float mx, float inc,
                                                         float dec) {
    return (prev >= last) ? fmaxf(dec * cur, mn) : fminf(inc * cur, mx);
}

// ImprovementBasedWeightStepsizeAdaptation.update_stepsize (weight_stepsize_adaptation.py:141-156) by the 64 lanes of
// ONE wavefront; state = {stepsize, previous ELBO proxy}.  The proxy is accumulated in fp64 and rounded once (DESIGN.md
// section 6, Q-elbo).
__device__ __forceinline__ void weight_stepsize_wave(int K, const float* __restrict__ logw,
                                                     const float* __restrict__ rewards_last, float* __restrict__ state,
                                                     float mn, float mx, float inc, float dec, int lane) {
    double a = 0.0;
    for (int i = lane; i < K; i += 64) {
        const double w = exp((double)logw[i]);
        a += w * (double)rewards_last[i] - w * (double)logw[i];                           // :147
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) {
        // a proxy still made of the float32.min sentinels of a fresh reward history (every component new: the first
        // iteration) equals float32.min up to the rounding of sum_k w_k ~ 1: pinned to float32.min itself, so that the first
        // comparison is "not greater" whatever the precision the weights are kept in (DESIGN.md section 6, Q-elbo)
        const float elbo = (a <= -3.4028e38) ? -3.402823466e38f : (float)a;
        const float prev = state[1];
        state[0] = (elbo > prev) ? fminf(inc * state[0], mx) : fmaxf(dec * state[0], mn);  // :149-156
        state[1] = elbo;
    }
}
