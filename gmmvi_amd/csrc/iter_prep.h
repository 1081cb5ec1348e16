// Bookkeeping of the single-call iteration as device code shared by fused.hip (stand-alone launch) and sampling.hip (extra
// blocks of the sampling launch).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "stepsize_rules.h"

struct PrepArgs {
    // DB mapping: dst[i] = src[i] + base
    int32_t* map_dst; const int32_t* map_src; int32_t map_base; int n_map;
    // model snapshot into the sample DB (sample_db.py:113-124): up to three word-wise copies
    uint32_t* cdst[3]; const uint32_t* csrc[3]; unsigned long long cwords[3];
    // stepsize rules
    int K; int cs_mode; float* stepsizes; const float* reward_prev; const float* reward_last;
    float cs_min, cs_max, cs_inc, cs_dec;
    int ws_mode; const float* logw; float* wstate; float ws_min, ws_max, ws_inc, ws_dec;
    // component shards (fused.hip: sharded iteration): the weight-stepsize rule runs over ALL ws_K components with its own
    // reward column (0 / NULL: K and reward_last, the unsharded case)
    int ws_K; const float* ws_reward_last;
};

// block 0 of n_blocks: the O(K) stepsize rules (wave 0 also runs the weight-stepsize reduction); blocks >= 1: copies.
// Called by the first `nthreads` threads (a multiple of 64) of every participating workgroup.
__device__ __forceinline__ void iter_prep_block(const PrepArgs& a, int block, int n_blocks, int nthreads = 256) {
    if (block == 0) {
        if (a.cs_mode == 1)
            for (int k = threadIdx.x; k < a.K; k += nthreads)
                a.stepsizes[k] = component_stepsize_rule(a.stepsizes[k], a.reward_prev[k], a.reward_last[k], a.cs_min,
                                                         a.cs_max, a.cs_inc, a.cs_dec);
        if (a.ws_mode == 1 && threadIdx.x < 64)
            weight_stepsize_wave(a.ws_K > 0 ? a.ws_K : a.K, a.logw, a.ws_reward_last ? a.ws_reward_last : a.reward_last, a.wstate,
                                 a.ws_min, a.ws_max, a.ws_inc, a.ws_dec, threadIdx.x);
        if (n_blocks > 1) return;                      // a single block does the copies as well
    }
    const int copy_blocks = n_blocks > 1 ? n_blocks - 1 : 1;
    const unsigned long long tid = (unsigned long long)(n_blocks > 1 ? block - 1 : 0) * nthreads + threadIdx.x;
    const unsigned long long step = (unsigned long long)copy_blocks * nthreads;
    for (unsigned long long i = tid; i < (unsigned long long)a.n_map; i += step) a.map_dst[i] = a.map_src[i] + a.map_base;
    // eight loads in flight per thread: source and destination may alias as far as the compiler knows, so a plain copy loop is a
    // chain of load -> store round trips (16 workgroups copying 130 k words took 13 us that way)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t* __restrict__ src = a.csrc[c];
        uint32_t* __restrict__ dst = a.cdst[c];
        const unsigned long long n = a.cwords[c];
        for (unsigned long long i = tid; i < n; i += 8 * step) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + u * step < n ? src[i + u * step] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u * step < n) dst[i + u * step] = v[u];
        }
    }
}
