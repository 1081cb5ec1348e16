// Component-ordered sampling x = mu_k + L_k eps (models/gmm.py:361-386, models/full_cov_gmm.py:36-39) and the
// raw Philox streams.  One lane per sample; (mu_k, L_k) is read per lane because neighbouring lanes of a wave may
// belong to different components (the reads hit the same cache lines; N*D^2 flops are negligible beside the
// density kernels).
#include "common.h"
#include "philox.h"

__global__ void sample_components_kernel(int K, int D, const float* __restrict__ means, const float* __restrict__ chols,
                                         const int32_t* __restrict__ offsets, int N, uint64_t seed,
                                         uint64_t first_index, uint32_t stream_id, const float* __restrict__ eps_in,
                                         float* __restrict__ X, int32_t* __restrict__ mapping) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    // component of sample n: largest k with offsets[k] <= n (binary search over the K+1 prefix sums)
    int lo = 0, hi = K;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offsets[mid] <= n) lo = mid; else hi = mid;
    }
    const int k = lo;
    if (mapping) mapping[n] = k;
    float eps[GMMVI_MAX_DIM];
    if (eps_in) {
        for (int i = 0; i < D; ++i) eps[i] = eps_in[(size_t)n * D + i];
    } else {
        const uint64_t idx = first_index + (uint64_t)n;
        for (int b = 0; b * 4 < D; ++b) {
            float nn[4];
            philox_normal4(seed, idx, (uint32_t)b, stream_id, nn);
            for (int j = 0; j < 4; ++j)
                if (4 * b + j < D) eps[4 * b + j] = nn[j];
        }
    }
    const float* L = chols + (size_t)k * D * D;
    const float* mu = means + (size_t)k * D;
    for (int i = 0; i < D; ++i) {
        float v = mu[i];
        for (int j = 0; j <= i; ++j) v = fmaf(L[i * D + j], eps[j], v);
        X[(size_t)n * D + i] = v;
    }
}

__global__ void philox_normals_kernel(uint64_t seed, uint64_t first_index, uint32_t stream_id, int N, int D,
                                      float* __restrict__ out) {
    const int nb = (D + 3) / 4;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * nb) return;
    const int n = (int)(t / nb), b = (int)(t % nb);
    float nn[4];
    philox_normal4(seed, first_index + (uint64_t)n, (uint32_t)b, stream_id, nn);
    for (int j = 0; j < 4; ++j)
        if (4 * b + j < D) out[(size_t)n * D + 4 * b + j] = nn[j];
}

__global__ void philox_uniforms_kernel(uint64_t seed, uint64_t first_index, uint32_t stream_id, int N,
                                       float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint64_t idx = first_index + (uint64_t)n;
    Philox4 p = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), 0u, stream_id, (uint32_t)seed,
                              (uint32_t)(seed >> 32));
    out[n] = philox_u01(p.w[0]);
}

extern "C" {

int gmmvi_sample_components(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                            const int32_t* offsets_dev, int N, uint64_t seed, uint64_t first_index, int stream_id,
                            const float* eps_dev, float* X_out_dev, int32_t* mapping_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM && N >= 0);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && offsets_dev && X_out_dev);
    GMMVI_PROF(ctx, "sample_components");
    hipLaunchKernelGGL(sample_components_kernel, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, K, D, means_dev,
                       chols_dev, offsets_dev, N, seed, first_index, (uint32_t)stream_id, eps_dev, X_out_dev,
                       mapping_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_philox_normals(gmmvi_ctx* ctx, uint64_t seed, uint64_t first_index, int stream_id, int N, int D,
                         float* eps_out_dev) {
    GMMVI_ARG_CHECK(ctx, N >= 0 && D >= 1);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, eps_out_dev != nullptr);
    long total = (long)N * ((D + 3) / 4);
    hipLaunchKernelGGL(philox_normals_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, seed,
                       first_index, (uint32_t)stream_id, N, D, eps_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_philox_uniforms(gmmvi_ctx* ctx, uint64_t seed, uint64_t first_index, int stream_id, int N,
                          float* u_out_dev) {
    GMMVI_ARG_CHECK(ctx, N >= 0);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, u_out_dev != nullptr);
    hipLaunchKernelGGL(philox_uniforms_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, seed, first_index,
                       (uint32_t)stream_id, N, u_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // extern "C"
