// Wavefront-wide (64 lanes) reductions on the DPP network instead of ds_bpermute butterflies: four DPP steps give every
// lane the total of its row of 16, v_readlane adds the four rows.  ~10x shorter than six dependent __shfl_xor hops, which
// matters for single-wavefront kernels whose critical path is a chain of such reductions.  Result is wave-uniform.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL>
__device__ __forceinline__ float gmmvi_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <class Op>
__device__ __forceinline__ float gmmvi_wave_reduce(float v, Op op) {
    v = op(v, gmmvi_dpp<0xB1>(v));     // quad_perm [1,0,3,2]
    v = op(v, gmmvi_dpp<0x4E>(v));     // quad_perm [2,3,0,1]
    v = op(v, gmmvi_dpp<0x141>(v));    // row_half_mirror
    v = op(v, gmmvi_dpp<0x140>(v));    // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return op(op(r0, r1), op(r2, r3));
}

__device__ __forceinline__ float gmmvi_wave_sum(float v) {
    return gmmvi_wave_reduce(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float gmmvi_wave_max(float v) {
    return gmmvi_wave_reduce(v, [](float a, float b) { return fmaxf(a, b); });
}
