// Maximum mean discrepancy statistics (experiments/evaluation/mmd.py:41-58): the two double sums
//   compute_ustat(sample)      = sum_i sum_j exp(-(s_i - s_j)^T Kmat (s_i - s_j))             :41-48
//   kernel_mix(sample)         = sum_i sum_j exp(-(g_i - s_j)^T Kmat (g_i - s_j))             :50-58
// with the diagonal bandwidth matrix Kmat = inv(alpha * sigma).  One launch evaluates sum_{i<Na} sum_{j<Nb} for two
// row sets A, B; the reference's Python loop over i with an [N, D] temporary per step becomes a tiled pair sweep:
// lane = one row of A (its coordinates staged through LDS in chunks of 32 dimensions), a tile of 32 rows of B is
// broadcast from LDS, squared distances accumulate in registers, exp() once per pair.  Partial sums per workgroup
// (fp64) are written to a scratch array and added in a fixed order by a second single-wave launch, so the result
// does not depend on the execution order.
#include "common.h"

namespace {

constexpr int kTileJ = 32;
constexpr int kChunkD = 32;
constexpr int kTilesPerBlock = 8;      // a workgroup sweeps 8 * 32 rows of B

__global__ __launch_bounds__(64) void mmd_pair_sum_kernel(const float* __restrict__ A, int Na, const float* __restrict__ B,
                                                          int Nb, int D, const float* __restrict__ inv_bw,
                                                          double* __restrict__ partial) {
    __shared__ float As[64][kChunkD + 1];
    __shared__ float Bs[kTileJ][kChunkD];
    __shared__ float Cs[kChunkD];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 64 + t;
    const int j_begin = blockIdx.y * kTilesPerBlock * kTileJ;
    double total = 0.0;
    for (int tile = 0; tile < kTilesPerBlock; ++tile) {
        const int j0 = j_begin + tile * kTileJ;
        if (j0 >= Nb) break;
        float acc[kTileJ];
#pragma unroll
        for (int jj = 0; jj < kTileJ; ++jj) acc[jj] = 0.f;
        for (int d0 = 0; d0 < D; d0 += kChunkD) {
            const int nd = min(kChunkD, D - d0);
            __syncthreads();
            // stage A rows (coalesced over the chunk's dimensions), the B tile and the bandwidths
            for (int e = t; e < 64 * kChunkD; e += 64) {
                const int r = e / kChunkD, c = e % kChunkD;
                const int row = blockIdx.x * 64 + r;
                As[r][c] = (row < Na && c < nd) ? A[(size_t)row * D + d0 + c] : 0.f;
            }
            for (int e = t; e < kTileJ * kChunkD; e += 64) {
                const int r = e / kChunkD, c = e % kChunkD;
                Bs[r][c] = (j0 + r < Nb && c < nd) ? B[(size_t)(j0 + r) * D + d0 + c] : 0.f;
            }
            if (t < kChunkD) Cs[t] = (t < nd) ? inv_bw[d0 + t] : 0.f;
            __syncthreads();
            float a[kChunkD];
#pragma unroll
            for (int c = 0; c < kChunkD; ++c) a[c] = As[t][c];
#pragma unroll
            for (int jj = 0; jj < kTileJ; ++jj) {
                float s = acc[jj];
#pragma unroll
                for (int c = 0; c < kChunkD; ++c) {
                    const float df = a[c] - Bs[jj][c];
                    s = fmaf(df * Cs[c], df, s);
                }
                acc[jj] = s;
            }
        }
        if (i < Na) {
            float tile_sum = 0.f;
#pragma unroll
            for (int jj = 0; jj < kTileJ; ++jj)
                if (j0 + jj < Nb) tile_sum += expf(-acc[jj]);
            total += (double)tile_sum;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (t == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total;
}

__global__ __launch_bounds__(64) void mmd_reduce_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += 64) s += partial[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) *out = s;
}

}  // namespace

extern "C" {

size_t gmmvi_mmd_scratch_doubles(int Na, int Nb) {
    const size_t gx = ((size_t)Na + 63) / 64;
    const size_t gy = ((size_t)Nb + kTilesPerBlock * kTileJ - 1) / (kTilesPerBlock * kTileJ);
    return gx * gy;
}

int gmmvi_mmd_pair_sum(gmmvi_ctx* ctx, const float* A_dev, int Na, const float* B_dev, int Nb, int D,
                       const float* inv_bandwidth_dev, double* scratch_dev, double* sum_out_dev) {
    GMMVI_ARG_CHECK(ctx, Na >= 1 && Nb >= 1 && D >= 1);
    GMMVI_ARG_CHECK(ctx, A_dev && B_dev && inv_bandwidth_dev && scratch_dev && sum_out_dev);
    const unsigned gx = (unsigned)((Na + 63) / 64);
    const unsigned gy = (unsigned)((Nb + kTilesPerBlock * kTileJ - 1) / (kTilesPerBlock * kTileJ));
    GMMVI_ARG_CHECK(ctx, gy <= 65535u);
    {
        GMMVI_PROF(ctx, "mmd_pair_sum");
        hipLaunchKernelGGL(mmd_pair_sum_kernel, dim3(gx, gy), dim3(64), 0, ctx->stream, A_dev, Na, B_dev, Nb, D,
                           inv_bandwidth_dev, scratch_dev);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    GMMVI_PROF(ctx, "mmd_reduce");
    hipLaunchKernelGGL(mmd_reduce_kernel, dim3(1), dim3(64), 0, ctx->stream, scratch_dev, (int)(gx * gy), sum_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // extern "C"
