// Component-ordered sampling x = mu_k + L_k eps (models/gmm.py:361-386, models/full_cov_gmm.py:36-39) and the
// raw Philox streams.  One lane per sample; (mu_k, L_k) is read per lane because neighbouring lanes of a wave may
// belong to different components (the reads hit the same cache lines; N*D^2 flops are negligible beside the
// density kernels).
#include "common.h"
#include "philox.h"
#include "iter_prep.h"
#include "sample_block.h"
#include "riders.h"
#include "blocked.h"

// grid = (component, 256-sample chunk of that component): sample_block.h
template <int DP>
__global__ __launch_bounds__(256) void sample_components_kernel(int K, int D, const float* __restrict__ means,
                                                                const float* __restrict__ chols,
                                                                const int32_t* __restrict__ offsets, int N, uint64_t seed,
                                                                uint64_t first_index, uint32_t stream_id,
                                                                const float* __restrict__ eps_in, float* __restrict__ X,
                                                                int32_t* __restrict__ mapping, int32_t mapping_base,
                                                                int n_chunks, int uniform_count, PrepArgs prep) {
    extern __shared__ float sm[];
    if ((int)blockIdx.y >= n_chunks) {                 // bookkeeping blocks of the single-call iteration (iter_prep.h)
        iter_prep_block(prep, blockIdx.x, gridDim.x);
        return;
    }
    sample_block<DP>(sm, blockIdx.x, blockIdx.y, D, means, chols, offsets, seed, first_index, stream_id, eps_in, X, mapping,
                     mapping_base, uniform_count);
}

// riders nothing carried (riders.h): their own launch
template <int DP>
__global__ __launch_bounds__(256) void riders_kernel(Riders r) {
    extern __shared__ float sm[];
    riders_carried<DP>(r, sm);
}

Riders gmmvi_take_pending_riders(gmmvi_ctx* ctx, int first_block, int threads, bool prep_only) {
    Riders r = ctx->riders;
    r.first_block = first_block;
    if (threads < 64 || threads % 64) { r.prep_blocks = 0; r.sample_blocks = 0; return r; }
    ctx->riders.prep_blocks = 0;
    if (threads < 256 || prep_only) r.sample_blocks = 0;      // sample_block needs 256 threads: that rider stays pending
    else ctx->riders.sample_blocks = 0;
    return r;
}

int gmmvi_flush_pending_riders(gmmvi_ctx* ctx) {
    if ((ctx->riders.prep_blocks | ctx->riders.sample_blocks) == 0) return GMMVI_OK;
    const Riders r = gmmvi_take_pending_riders(ctx, 0, 256);
    const int D = r.sample_blocks > 0 ? r.sample.D : 2;
    const int dp = gmmvi_padded_dim(D);
    GMMVI_PROF(ctx, "riders");
    GMMVI_DISPATCH_DP(dp, hipLaunchKernelGGL((riders_kernel<DP>), dim3(r.prep_blocks + r.sample_blocks), dim3(256),
                                             riders_lds_bytes(r), ctx->stream, r));
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
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

// C++ linkage (common.h): as gmmvi_sample_components, with an upper bound on the samples of any one component known to
// the caller -- the launch then covers ceil(bound / 256) chunks per component instead of ceil(N / 256)
static int launch_sample(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                         const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed, uint64_t first_index,
                         int stream_id, const float* eps_dev, float* X_out_dev, int32_t* mapping_out_dev, int32_t mapping_base,
                         const PrepArgs* prep, int uniform_count = 0) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_BLOCKED_MAX_DIM && N >= 0 && max_per_component >= 0);
    if (N == 0 && !prep) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && offsets_dev && X_out_dev);
    if (gmmvi_is_blocked_dim(D)) {
        GMMVI_ARG_CHECK(ctx, prep == nullptr && mapping_base == 0);
        return gmmvi_blocked_sample(ctx, K, D, means_dev, chols_dev, offsets_dev, N, max_per_component, seed, first_index,
                                    stream_id, eps_dev, X_out_dev, mapping_out_dev);
    }
    GMMVI_PROF(ctx, "sample_components");
    const int bound = max_per_component < N ? max_per_component : N;
    const int chunks = (bound + 255) / 256 > 0 ? (bound + 255) / 256 : 1;
    const size_t shmem = ((size_t)D * D + D + 256 * (size_t)(D | 1)) * sizeof(float);
    const int dp = gmmvi_padded_dim(D);
    PrepArgs none{};
    GMMVI_DISPATCH_DP(dp, hipLaunchKernelGGL((sample_components_kernel<DP>), dim3(K, chunks + (prep ? 1 : 0)), dim3(256),
                                             shmem, ctx->stream, K, D, means_dev, chols_dev, offsets_dev, N, seed,
                                             first_index, (uint32_t)stream_id, eps_dev, X_out_dev, mapping_out_dev,
                                             mapping_base, chunks, uniform_count, prep ? *prep : none));
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// C++ linkage (common.h): as gmmvi_sample_components, with an upper bound on the samples of any one component known to
// the caller -- the launch then covers ceil(bound / 256) chunks per component instead of ceil(N / 256)
int gmmvi_sample_components_bounded(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                                    const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed,
                                    uint64_t first_index, int stream_id, const float* eps_dev, float* X_out_dev,
                                    int32_t* mapping_out_dev) {
    return launch_sample(ctx, K, D, means_dev, chols_dev, offsets_dev, N, max_per_component, seed, first_index, stream_id,
                         eps_dev, X_out_dev, mapping_out_dev, 0, nullptr);
}

// C++ linkage (common.h): sampling for the single-call iteration -- mapping written as component index + mapping_base,
// the bookkeeping of iter_prep.h in K extra blocks of the same launch
int gmmvi_sample_components_prep(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                                 const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed, uint64_t first_index,
                                 float* X_out_dev, int32_t* mapping_out_dev, int32_t mapping_base, const PrepArgs& prep) {
    // the single-call iteration draws the same number of samples from every component (offsets[k] = k * max_per_component)
    const int uniform = (long)K * max_per_component == N ? max_per_component : 0;
    return launch_sample(ctx, K, D, means_dev, chols_dev, offsets_dev, N, max_per_component, seed, first_index, 0, nullptr,
                         X_out_dev, mapping_out_dev, mapping_base, &prep, uniform);
}

extern "C" {

int gmmvi_sample_components(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                            const int32_t* offsets_dev, int N, uint64_t seed, uint64_t first_index, int stream_id,
                            const float* eps_dev, float* X_out_dev, int32_t* mapping_out_dev) {
    // the per-component counts live on the device: cover the worst case (all N samples in one component); empty
    // chunks exit immediately
    return gmmvi_sample_components_bounded(ctx, K, D, means_dev, chols_dev, offsets_dev, N, N, seed, first_index, stream_id,
                                           eps_dev, X_out_dev, mapping_out_dev);
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
