// Work of the single-call iteration that rides as EXTRA WORKGROUPS in a density launch that exists anyway (a kernel boundary
// costs ~2.3 us plus ramp and drain on this stack: tools/probe/pk_issue.hip):
//   * the bookkeeping blocks (iter_prep.h: the two stepsize rules, the model snapshot into the sample database) ride in the
//     dual sweep when the sampling launch that used to carry them is gone;
//   * the NEXT iteration's draw rides in the post-update sweep: x = mu + L eps of the next iteration needs the updated
//     components only -- not the weights -- so it can be made as soon as the component update is done
//     (sample_selector.py:204-219; fused.hip / optimization/fused.py: `presample_next`, `presampled`).
// The merge of a previous sweep's chunk partials (combine.h) is the third kind of rider and keeps its own struct.
#pragma once
#include "common.h"
#include "iter_prep.h"
#include "sample_block.h"

// true: this workgroup was a rider (the caller returns); workgroup sizes 256 .. 1024
template <int DP>
__device__ __forceinline__ bool riders_carried(const Riders& r, float* sm) {
    if ((r.prep_blocks | r.sample_blocks) == 0 || (int)blockIdx.x < r.first_block) return false;
    if (blockIdx.y != 0 || blockIdx.z != 0) return true;
    const int i = (int)blockIdx.x - r.first_block;
    if (i < r.prep_blocks) {
        const int nthreads = blockDim.x < 256 ? (int)blockDim.x : 256;
        if ((int)threadIdx.x < nthreads) iter_prep_block(r.prep, i, r.prep_blocks, nthreads);
    } else if (i < r.prep_blocks + r.sample_blocks) {
        const int j = i - r.prep_blocks;
        const SampleJob& s = r.sample;
        sample_block<DP>(sm, j % s.K, j / s.K, s.D, s.means, s.chols, s.offsets, s.seed, s.first_index, 0u, nullptr, s.X, s.mapping,
                         s.mapping_base, s.uniform_count);
    }
    return true;
}

// a carrier without the template parameter and the LDS of the sampling rider: bookkeeping blocks only (workgroups >= 64 threads)
__device__ __forceinline__ bool riders_carried_prep(const Riders& r) {
    if (r.prep_blocks == 0 || (int)blockIdx.x < r.first_block) return false;
    if (blockIdx.y != 0 || blockIdx.z != 0) return true;
    const int i = (int)blockIdx.x - r.first_block;
    const int nthreads = blockDim.x < 256 ? (int)blockDim.x : 256;
    if (i < r.prep_blocks && (int)threadIdx.x < nthreads) iter_prep_block(r.prep, i, r.prep_blocks, nthreads);
    return true;
}
