// Component-ordered sampling x = mu_k + L_k eps (models/gmm.py:361-386, models/full_cov_gmm.py:36-39) and the
// raw Philox streams.  One lane per sample; (mu_k, L_k) is read per lane because neighbouring lanes of a wave may
// belong to different components (the reads hit the same cache lines; N*D^2 flops are negligible beside the
// density kernels).
#include "common.h"
#include "philox.h"
#include "iter_prep.h"
#include "blocked.h"

// grid = (component, 256-sample chunk of that component); (mu_k, L_k) staged in LDS and read as broadcasts, eps and x in
// registers (DP = padded dimension, loops unrolled), the output tile leaves through LDS with coalesced stores.
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
    const int k = blockIdx.x;
    // equal counts known to the caller (single-call iteration): no dependent load in front of everything else
    const int begin = uniform_count > 0 ? k * uniform_count : offsets[k];
    const int end = uniform_count > 0 ? begin + uniform_count : offsets[k + 1];
    const int base = begin + blockIdx.y * 256;
    if (base >= end) return;
    const int n_here = min(256, end - base);
    float* Ls = sm;                      // [D][D]
    float* mus = sm + D * D;             // [D]
    float* tile = mus + D;               // [256][ldx]
    const int ldx = D | 1;
    const int t = threadIdx.x;
    // (mu, L) are fetched into registers first and reach LDS after the random numbers are made: the loads (L2 round trips:
    // the previous iteration's update kernel wrote them on other CUs) overlap the Philox rounds
    constexpr int NL = (DP * DP + 255) / 256;
    float lreg[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) lreg[u] = (t + 256 * u < D * D) ? chols[(size_t)k * D * D + t + 256 * u] : 0.f;
    const float mreg = t < D ? means[(size_t)k * D + t] : 0.f;
    if (eps_in) {
        for (int e = t; e < n_here * D; e += 256) tile[(e / D) * ldx + (e % D)] = eps_in[(size_t)base * D + e];
    }
    const bool valid = t < n_here;
    if (!eps_in) {
        // the standard normals of the tile, four per Philox block (counter = sample index, block): the (sample, block) items are
        // spread over ALL threads -- a thread that made all of its sample's blocks itself spent 4 us of a 9 us launch in the
        // Philox rounds and the Box-Muller transforms (D = 20: five blocks a sample, 100 samples on 256 threads)
        constexpr int NB4 = (DP + 3) / 4;
        for (int item = t; item < n_here * NB4; item += 256) {
            const int smp = item / NB4, b = item - smp * NB4;
            float nn[4];
            philox_normal4(seed, first_index + (uint64_t)(base + smp), (uint32_t)b, stream_id, nn);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * b + j < D) tile[smp * ldx + 4 * b + j] = nn[j];
        }
    }
#pragma unroll
    for (int u = 0; u < NL; ++u)
        if (t + 256 * u < D * D) Ls[t + 256 * u] = lreg[u];
    if (t < D) mus[t] = mreg;
    __syncthreads();
    if constexpr (DP >= 32) {
        // X = mu + eps L^T as a matrix-core product (v_mfma_f32_16x16x4_f32): a wave owns 64 samples (four 16-row tiles of
        // eps), A = eps[16 mt + i][4 s + kk], B = L^T: B[kk][j] = L[16 nt + j][4 s + kk], k-steps beyond the diagonal block of
        // the lower-triangular L skipped.  (One lane per sample with the row of L broadcast from LDS is a chain of D^2 / 2
        // dependent multiply-adds on 100 of the 256 threads: 16 of the 21 us of the launch at D = 50.)
        typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
        constexpr int NT = (DP + 15) / 16, KS = (DP + 3) / 4;
        const int wave = t >> 6, lane = t & 63, i16 = lane & 15, kk = lane >> 4;
        const int s0 = 64 * wave;                      // first sample of this wave
        if (s0 < n_here) {
            sc_f32x4 acc[4][NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = sc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < KS; ++s4) {
                const int kcol = 4 * s4 + kk;
                float a[4], b[NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int smp = s0 + 16 * mt + i16;
                    a[mt] = (smp < n_here && kcol < D) ? tile[smp * ldx + kcol] : 0.f;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int row = 16 * nt + i16;
                    b[nt] = (row < D && kcol <= row) ? Ls[row * D + kcol] : 0.f;      // lower triangle only (and kcol < D)
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (4 * s4 > 16 * nt + 15) continue;                                // this k-step lies above the diagonal block
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
            }
            // every eps value of the wave's rows has been read (by this wave only): the results overwrite them in place.
            // D[i][j]: lane l, register r -> sample 16 mt + 4 (l >> 4) + r, dimension 16 nt + (l & 15)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int smp = s0 + 16 * mt + 4 * kk + r, dim = 16 * nt + i16;
                        if (smp < n_here && dim < D) tile[smp * ldx + dim] = acc[mt][nt][r] + mus[dim];
                    }
        }
        if (valid && mapping) mapping[base + t] = k + mapping_base;
    } else {
    float eps[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) eps[i] = (valid && i < D) ? tile[t * ldx + i] : 0.f;
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                float v = mus[i];
#pragma unroll
                for (int j = 0; j <= i; ++j) v = fmaf(Ls[i * D + j], eps[j], v);
                tile[t * ldx + i] = v;
            }
        }
        if (mapping) mapping[base + t] = k + mapping_base;
    }
    }
    __syncthreads();
    for (int e = t; e < n_here * D; e += 256) X[(size_t)base * D + e] = tile[(e / D) * ldx + (e % D)];
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
