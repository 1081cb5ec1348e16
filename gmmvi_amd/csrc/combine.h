// Merge of the per-chunk partials of a component-split mixture sweep (density.hip: gridDim.y > 1):
//   lp[n] = log sum_r exp(lp_r[n]);  grad[n, :] = sum_r exp(lp_r[n] - lp[n]) grad_r[n, :]   (fixed order over r).
// One element function, three users with identical arithmetic: the stand-alone launch (comm.hip, also the E2 exchange of the
// sharded path), extra workgroups riding in the NEXT launch of the single-call iteration (a launch that exists anyway and does
// not read the merged arrays: the target evaluation carries the merge of the model sweep), and the expected-log-ratio kernel,
// which merges the log values of the post-update sweep while it reads them.
#pragma once
#include "common.h"

// thread = one element of the [N, D] gradient (coalesced over r-major partial arrays); the thread of column 0 also writes lp,
// the thread of column 1 (column 0 without a gradient) the second set of log values
template <int R>
__device__ __forceinline__ float combine_log_values_n(const float* __restrict__ parts, int N, int n) {
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = parts[(size_t)r * N + n];          // all loads in flight
    float m = -3.0e38f;
#pragma unroll
    for (int r = 0; r < R; ++r) m = fmaxf(m, v[r]);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += __expf(v[r] - m);
    return m + __logf(s);
}

__device__ __forceinline__ float combine_log_values(const float* __restrict__ parts, int R, int N, int n) {
    switch (R) {                                       // the usual chunk counts, unrolled (R is uniform: a scalar branch)
        case 2: return combine_log_values_n<2>(parts, N, n);
        case 3: return combine_log_values_n<3>(parts, N, n);
        case 4: return combine_log_values_n<4>(parts, N, n);
        case 5: return combine_log_values_n<5>(parts, N, n);
        case 6: return combine_log_values_n<6>(parts, N, n);
        case 7: return combine_log_values_n<7>(parts, N, n);
        case 8: return combine_log_values_n<8>(parts, N, n);
        case 10: return combine_log_values_n<10>(parts, N, n);
        case 12: return combine_log_values_n<12>(parts, N, n);
        case 16: return combine_log_values_n<16>(parts, N, n);
        default: break;
    }
    float m = -3.0e38f;
    for (int r = 0; r < R; ++r) m = fmaxf(m, parts[(size_t)r * N + n]);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += __expf(parts[(size_t)r * N + n] - m);
    return m + __logf(s);
}

// R > 0: compile-time chunk count (all loads of an element straight-line, in flight together); R == 0: run-time j.R
template <int R>
__device__ __forceinline__ void combine_element_n(const CombineJob& j, long e) {
    const bool with_grad = j.grad_out != nullptr && j.grad_parts != nullptr;
    const int width = with_grad ? j.D : 1;
    if (e >= (long)j.N * width) return;
    const int n = (int)(e / width), i = (int)(e - (long)n * width);
    const int Rr = R > 0 ? R : j.R;
    // distance between the parts of consecutive chunks / ranks: N values (N * D gradient entries) unless the parts sit inside a
    // gathered exchange buffer (fused.hip: sharded iteration)
    const int ps = j.part_stride > 0 ? (int)j.part_stride : j.N;
    const size_t gs = j.part_stride > 0 ? (size_t)j.part_stride : (size_t)j.N * j.D;
    float lp;
    if constexpr (R > 0) lp = combine_log_values_n<R>(j.lp_parts, ps, n); else lp = combine_log_values(j.lp_parts, Rr, ps, n);
    if (j.lp_out && i == 0) j.lp_out[n] = lp;
    if (j.lp2_out && j.lp2_parts && i == (width > 1 ? 1 : 0)) {
        if constexpr (R > 0) j.lp2_out[n] = combine_log_values_n<R>(j.lp2_parts, ps, n);
        else j.lp2_out[n] = combine_log_values(j.lp2_parts, Rr, ps, n);
    }
    if (with_grad) {
        float g = 0.f;
        if constexpr (R > 0) {
            float lv[R], gv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                lv[r] = j.lp_parts[(size_t)r * ps + n];
                gv[r] = j.grad_parts[(size_t)r * gs + (size_t)n * j.D + i];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) g = fmaf(__expf(lv[r] - lp), gv[r], g);
        } else {
            for (int r = 0; r < Rr; ++r)
                g = fmaf(__expf(j.lp_parts[(size_t)r * ps + n] - lp), j.grad_parts[(size_t)r * gs + (size_t)n * j.D + i], g);
        }
        j.grad_out[(size_t)n * j.D + i] = g;
    }
}

__device__ __forceinline__ void combine_element(const CombineJob& j, long e) {
    switch (j.R) {                                     // uniform
        case 2: combine_element_n<2>(j, e); break;
        case 3: combine_element_n<3>(j, e); break;
        case 4: combine_element_n<4>(j, e); break;
        case 5: combine_element_n<5>(j, e); break;
        case 6: combine_element_n<6>(j, e); break;
        case 7: combine_element_n<7>(j, e); break;
        case 8: combine_element_n<8>(j, e); break;
        case 10: combine_element_n<10>(j, e); break;
        case 12: combine_element_n<12>(j, e); break;
        case 16: combine_element_n<16>(j, e); break;
        default: combine_element_n<0>(j, e); break;
    }
}

// the carried form: workgroups first_block .. first_block + blocks - 1 of the carrying launch (blockIdx.y == 0 only)
template <int R>
__device__ __forceinline__ void combine_carried_n(const CombineJob& j) {
    const bool with_grad = j.grad_out != nullptr && j.grad_parts != nullptr;
    const long elems = (long)j.N * (with_grad ? j.D : 1);
    const long stride = (long)j.blocks * blockDim.x;
    for (long e = (long)((int)blockIdx.x - j.first_block) * blockDim.x + threadIdx.x; e < elems; e += stride)
        combine_element_n<R>(j, e);
}

__device__ __forceinline__ bool combine_carried(const CombineJob& j) {
    if (j.blocks == 0 || (int)blockIdx.x < j.first_block) return false;
    if (blockIdx.y == 0 && blockIdx.z == 0) {
        switch (j.R) {
            case 2: combine_carried_n<2>(j); break;
            case 3: combine_carried_n<3>(j); break;
            case 4: combine_carried_n<4>(j); break;
            case 5: combine_carried_n<5>(j); break;
            case 6: combine_carried_n<6>(j); break;
            case 7: combine_carried_n<7>(j); break;
            case 8: combine_carried_n<8>(j); break;
            case 10: combine_carried_n<10>(j); break;
            case 12: combine_carried_n<12>(j); break;
            case 16: combine_carried_n<16>(j); break;
            default: combine_carried_n<0>(j); break;
        }
    }
    return true;
}
