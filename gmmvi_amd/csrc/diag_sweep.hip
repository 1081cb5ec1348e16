// Dedicated sweeps for DIAGONAL-covariance mixtures (models/diagonal_gmm.py:40-53, models/gmm.py:183-216,274-300 on a
// DiagonalGMM; sample_db.py:164-228 with diagonal snapshots): log N(x; mu_k, diag sigma_k^2), the (dual) log-sum-exp over the
// components, the responsibility-weighted gradient, sampling x = mu + sigma * eps and the Stein estimate's diagonal -- all
// O(D) per (sample, component) pair.  Until round 3 these ran through the dense kernels on the embedded factor diag(sigma):
// D / 2 times the multiply-adds (D^2 of them at D > 50, where the embedded factor took the blocked matrix-core path).
//
// Component block (gmmvi_diag_pack): [mu (D) | 1 / sigma (D) | 1 / sigma^2 (D) | log-normaliser | zeros] floats, padded by one
// 32-float piece so that the kernels may fetch whole pieces of any of the three vectors without leaving the block.
// Mapping: lane = sample; the block is wave-uniform, read through a constant-address-space pointer in pieces of 32 dimensions
// (scalar registers feeding v_sub / v_mul / v_fma); a wave walks a strided subset of the components; the lane keeps its x
// row in registers when D <= 32 and re-reads 32-dimension pieces of it (L1 / L2 hits) otherwise.
#include "common.h"
#include "combine.h"
#include "philox.h"
#include <cmath>

namespace {

typedef float ds_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* ds_cptr;

constexpr int DS_CH = 32;                                  // dimensions per piece

__host__ __device__ inline int ds_stride(int D) { return ((3 * D + 1 + DS_CH + 3) / 4) * 4; }

__global__ __launch_bounds__(64) void diag_pack_kernel(int D, const float* __restrict__ means, const float* __restrict__ sigma,
                                                       float* __restrict__ packed) {
    const int k = blockIdx.x, t = threadIdx.x;
    float* out = packed + (size_t)k * ds_stride(D);
    float lsum = 0.f;
    for (int i = t; i < D; i += 64) {
        const float sg = sigma[(size_t)k * D + i];
        out[i] = means[(size_t)k * D + i];
        out[D + i] = 1.f / sg;
        out[2 * D + i] = 1.f / (sg * sg);
        lsum += logf(sg);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if (t == 0) {
        out[3 * D] = -lsum - 0.5f * D * 1.8378770664093453f;              // - sum log sigma - D/2 log(2 pi)
        for (int i = 3 * D + 1; i < ds_stride(D); ++i) out[i] = 0.f;
    }
}

// one 32-dimension piece of the lane's sample row (dimensions beyond D read as 0)
__device__ __forceinline__ void ds_load_x(const float* __restrict__ xrow, int d0, int D, bool vec4, float (&x)[DS_CH]) {
    if (vec4 && d0 + DS_CH <= D) {
#pragma unroll
        for (int q = 0; q < DS_CH / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(xrow + d0 + 4 * q);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < DS_CH; ++i) x[i] = d0 + i < D ? xrow[d0 + i] : 0.f;
    }
}

// ---- densities + (dual) log-sum-exp -------------------------------------------------------------------------------------
// grid (64-sample tiles, component chunks); W waves per workgroup, wave w takes the components k_lo + w, k_lo + w + W, ...
// ld_out[K, N] (may be NULL), per-chunk log values lp_out[chunk][N] / lp2_out[chunk][N].
__global__ __launch_bounds__(512) void diag_eval_kernel(int K_total, int D, const float* __restrict__ packed,
                                                        const float* __restrict__ logw, const float* __restrict__ logw2,
                                                        const float* __restrict__ X, int N, float* __restrict__ ld_out,
                                                        float* __restrict__ lp_out, float* __restrict__ lp2_out) {
    __shared__ float sm_m[8 * 64], sm_s[8 * 64], sm_m2[8 * 64], sm_s2[8 * 64];
    const int kchunk = (K_total + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kchunk;
    const int K = min(K_total, k_lo + kchunk);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const bool valid = n < N;
    const float* __restrict__ xrow = X + (size_t)min(n, N - 1) * D;
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int stride = ds_stride(D);
    const bool dual = logw2 != nullptr;
    const bool one_piece = D <= DS_CH;
    float xr[DS_CH];
    if (one_piece) ds_load_x(xrow, 0, D, vec4, xr);
    float m = -3.0e38f, s = 0.f, m2 = -3.0e38f, s2 = 0.f;
    for (int k = k_lo + wave; k < K; k += nwaves) {
        const ds_cptr blk = (ds_cptr)(uintptr_t)(packed + (size_t)k * stride);
        float q = 0.f;
        for (int d0 = 0; d0 < D; d0 += DS_CH) {
            float x[DS_CH];
            if (one_piece) {
#pragma unroll
                for (int i = 0; i < DS_CH; ++i) x[i] = xr[i];
            } else {
                ds_load_x(xrow, d0, D, vec4, x);
            }
            float q0 = 0.f, q1 = 0.f;                                      // two chains of multiply-adds
#pragma unroll
            for (int i = 0; i < DS_CH; i += 2) {
                // (whole pieces are fetched: beyond dimension D the block continues with other data -- masked by the selects)
                const float t0 = (x[i] - blk[d0 + i]) * blk[D + d0 + i];
                const float t1 = (x[i + 1] - blk[d0 + i + 1]) * blk[D + d0 + i + 1];
                q0 = d0 + i < D ? fmaf(t0, t0, q0) : q0;
                q1 = d0 + i + 1 < D ? fmaf(t1, t1, q1) : q1;
            }
            q += q0 + q1;
        }
        const float ld = fmaf(-0.5f, q, blk[3 * D]);
        if (ld_out != nullptr && valid) ld_out[(size_t)k * N + n] = ld;
        const float a = ld + ((ds_cptr)(uintptr_t)logw)[k];
        const float mn = fmaxf(m, a);
        s = fmaf(s, __expf(m - mn), __expf(a - mn));
        m = mn;
        if (dual) {
            const float a2 = ld + ((ds_cptr)(uintptr_t)logw2)[k];
            const float mn2 = fmaxf(m2, a2);
            s2 = fmaf(s2, __expf(m2 - mn2), __expf(a2 - mn2));
            m2 = mn2;
        }
    }
    if (lp_out == nullptr && lp2_out == nullptr) return;
    sm_m[wave * 64 + lane] = m; sm_s[wave * 64 + lane] = s;
    if (dual) { sm_m2[wave * 64 + lane] = m2; sm_s2[wave * 64 + lane] = s2; }
    __syncthreads();
    if (wave == 0 && valid) {
        float M = -3.0e38f, S = 0.f;
        for (int w = 0; w < nwaves; ++w) M = fmaxf(M, sm_m[w * 64 + lane]);
        for (int w = 0; w < nwaves; ++w) S += sm_s[w * 64 + lane] * __expf(sm_m[w * 64 + lane] - M);
        if (lp_out) lp_out[(size_t)blockIdx.y * N + n] = M + __logf(S);
        if (dual && lp2_out) {
            float M2 = -3.0e38f, S2 = 0.f;
            for (int w = 0; w < nwaves; ++w) M2 = fmaxf(M2, sm_m2[w * 64 + lane]);
            for (int w = 0; w < nwaves; ++w) S2 += sm_s2[w * 64 + lane] * __expf(sm_m2[w * 64 + lane] - M2);
            lp2_out[(size_t)blockIdx.y * N + n] = M2 + __logf(S2);
        }
    }
}

// ---- gradient: grad[n, d] = - sum_k exp(logw_k + ld[k, n] - lp[n]) (x_nd - mu_kd) / sigma_kd^2 ------------------------------
// grid (64-sample tiles, 32-dimension pieces); W waves per workgroup split the components, merged through LDS in wave order.
__global__ __launch_bounds__(512) void diag_grad_kernel(int K, int D, const float* __restrict__ packed,
                                                        const float* __restrict__ logw, const float* __restrict__ X, int N,
                                                        const float* __restrict__ ld, const float* __restrict__ lp,
                                                        float* __restrict__ grad_out) {
    extern __shared__ float sm_acc[];                      // [W][DS_CH][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int n0 = blockIdx.x * 64, n = n0 + lane;
    const int nr = min(n, N - 1);
    const int d0 = blockIdx.y * DS_CH;
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int stride = ds_stride(D);
    float x[DS_CH], acc[DS_CH];
    ds_load_x(X + (size_t)nr * D, d0, D, vec4, x);
#pragma unroll
    for (int i = 0; i < DS_CH; ++i) acc[i] = 0.f;
    const float lpn = lp[nr];
    for (int k = wave; k < K; k += nwaves) {
        const ds_cptr blk = (ds_cptr)(uintptr_t)(packed + (size_t)k * stride);
        const float r = __expf(((ds_cptr)(uintptr_t)logw)[k] + ld[(size_t)k * N + nr] - lpn);
#pragma unroll
        for (int i = 0; i < DS_CH; ++i) {
            const float u = (x[i] - blk[d0 + i]) * blk[2 * D + d0 + i];          // (dimensions beyond D: never stored)
            acc[i] = fmaf(r, u, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < DS_CH; ++i) sm_acc[(wave * DS_CH + i) * 64 + lane] = acc[i];
    __syncthreads();
    // thread (sample, dimension) pairs of the 64 x 32 tile, coalesced along the dimension
    for (int e = threadIdx.x; e < 64 * DS_CH; e += blockDim.x) {
        const int sidx = e / DS_CH, i = e % DS_CH;
        if (n0 + sidx < N && d0 + i < D) {
            float g = 0.f;
            for (int w = 0; w < nwaves; ++w) g += sm_acc[(w * DS_CH + i) * 64 + sidx];
            grad_out[(size_t)(n0 + sidx) * D + d0 + i] = -g;
        }
    }
}

// ---- sampling x = mu_k + sigma_k * eps (component order; same Philox counters as the dense kernel) --------------------------
__global__ __launch_bounds__(256) void diag_sample_kernel(int K, int D, const float* __restrict__ means,
                                                          const float* __restrict__ sigma, const int32_t* __restrict__ offsets,
                                                          int N, uint64_t seed, uint64_t first_index, uint32_t stream_id,
                                                          const float* __restrict__ eps_in, float* __restrict__ X,
                                                          int32_t* __restrict__ mapping) {
    const int nb = (D + 3) / 4;
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (long)N * nb) return;
    const int n = (int)(item / nb), b = (int)(item % nb);
    // component of sample n: the last k with offsets[k] <= n
    int lo = 0, hi = K;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= n) lo = mid; else hi = mid;
    }
    const int k = lo;
    float nn[4];
    if (eps_in) {
#pragma unroll
        for (int j = 0; j < 4; ++j) nn[j] = 4 * b + j < D ? eps_in[(size_t)n * D + 4 * b + j] : 0.f;
    } else {
        philox_normal4(seed, first_index + (uint64_t)n, (uint32_t)b, stream_id, nn);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = 4 * b + j;
        if (d < D) X[(size_t)n * D + d] = fmaf(sigma[(size_t)k * D + d], nn[j], means[(size_t)k * D + d]);
    }
    if (b == 0 && mapping) mapping[n] = k;
}

// ---- Stein estimate, diagonal branch (ng_estimator.py:159-162, :178-181) -----------------------------------------------------
//   g_k[d] = E_w[gr[n, d]],  h_k[d] = E_w[gr[n, d] (x[n, d] - mu_k[d]) / sigma_k[d]^2],   gr = grad log p~ - grad log q,
// E_w = self-normalised importance weights softmax_n(ld[k, n] - bg[n]) or the plain 1 / n sum_n exp(ld - bg); with
// GMMVI_OWN_SAMPLES_ONLY over the component's own samples (mapping) with weights exp(0).  Outputs are the NEGATED estimates.
// grid (group of 4 components, 16-dimension piece): the x / gradient rows of a piece are read once for four components;
// 256 threads stride over the samples; fixed-order reduction (wave shuffles, then the four waves in order).
constexpr int DST_CG = 4, DST_CH = 16;
__global__ __launch_bounds__(256) void diag_stein_kernel(int K, int D, const float* __restrict__ packed, const float* __restrict__ X,
                                                         int N, const float* __restrict__ ld, const float* __restrict__ qgrad,
                                                         const float* __restrict__ bg, const float* __restrict__ tgrad,
                                                         const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                         float* __restrict__ h_neg, float* __restrict__ g_neg) {
    __shared__ float red[DST_CG][256];
    __shared__ float part[DST_CG * (2 * DST_CH + 1)][4];
    const int k0 = blockIdx.x * DST_CG, d0 = blockIdx.y * DST_CH, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const bool own = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0, snis = (flags & GMMVI_SELF_NORMALIZED) != 0;
    const int stride = ds_stride(D);
    auto logweight = [&](int c, int n) -> float {
        const int k = min(k0 + c, K - 1);
        if (own) return (mapping[n] + map_offset == k) ? 0.f : -3.0e38f;
        return ld[(size_t)k * N + n] - bg[n];
    };
    // pass 1: per component the maximum of the log weights and the number of samples the expectation runs over
    float mx[DST_CG], cnt[DST_CG];
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) { mx[c] = -3.0e38f; cnt[c] = 0.f; }
    for (int n = t; n < N; n += 256) {
#pragma unroll
        for (int c = 0; c < DST_CG; ++c) {
            const float a = logweight(c, n);
            mx[c] = fmaxf(mx[c], a);
            cnt[c] += a > -3.0e38f ? 1.f : 0.f;
        }
    }
    float M[DST_CG], n_used[DST_CG];
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) red[c][t] = mx[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
#pragma unroll
            for (int c = 0; c < DST_CG; ++c) red[c][t] = fmaxf(red[c][t], red[c][t + o]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) M[c] = red[c][0];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) red[c][t] = cnt[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
#pragma unroll
            for (int c = 0; c < DST_CG; ++c) red[c][t] += red[c][t + o];
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) n_used[c] = red[c][0];
    // pass 2: weighted sums of this piece's dimensions
    float sw[DST_CG], sg[DST_CG][DST_CH], sh[DST_CG][DST_CH], mu[DST_CG][DST_CH];
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) {
        sw[c] = 0.f;
        const float* blk = packed + (size_t)min(k0 + c, K - 1) * stride;
#pragma unroll
        for (int i = 0; i < DST_CH; ++i) { sg[c][i] = 0.f; sh[c][i] = 0.f; mu[c][i] = blk[d0 + i]; }
    }
    for (int n = t; n < N; n += 256) {
        float e[DST_CG];
        bool any = false;
#pragma unroll
        for (int c = 0; c < DST_CG; ++c) {
            const float a = logweight(c, n);
            e[c] = a > -3.0e38f ? __expf(a - M[c]) : 0.f;
            any |= e[c] != 0.f;
            sw[c] += e[c];
        }
        if (!any) continue;
#pragma unroll
        for (int i = 0; i < DST_CH; ++i) {
            if (d0 + i < D) {
                const float gr = tgrad[(size_t)n * D + d0 + i] - qgrad[(size_t)n * D + d0 + i];
                const float xv = X[(size_t)n * D + d0 + i];
#pragma unroll
                for (int c = 0; c < DST_CG; ++c) {
                    const float eg = e[c] * gr;
                    sg[c][i] += eg;
                    sh[c][i] = fmaf(eg, xv - mu[c][i], sh[c][i]);
                }
            }
        }
    }
    auto wave_sum = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
#pragma unroll
    for (int c = 0; c < DST_CG; ++c) {
        const float a = wave_sum(sw[c]);
        if (lane == 0) part[c * (2 * DST_CH + 1) + 2 * DST_CH][wave] = a;
#pragma unroll
        for (int i = 0; i < DST_CH; ++i) {
            const float g1 = wave_sum(sg[c][i]), h1 = wave_sum(sh[c][i]);
            if (lane == 0) { part[c * (2 * DST_CH + 1) + i][wave] = g1; part[c * (2 * DST_CH + 1) + DST_CH + i][wave] = h1; }
        }
    }
    __syncthreads();
    if (t < DST_CG * DST_CH) {
        const int c = t / DST_CH, i = t % DST_CH;
        const int k = k0 + c;
        if (k < K && d0 + i < D) {
            auto tot = [&](int row) { return (part[row][0] + part[row][1]) + (part[row][2] + part[row][3]); };
            const float S = tot(c * (2 * DST_CH + 1) + 2 * DST_CH);
            const float G = tot(c * (2 * DST_CH + 1) + i), H = tot(c * (2 * DST_CH + 1) + DST_CH + i);
            // self-normalised: / sum of the weights; plain: 1 / n sum exp(lw) v = exp(M) / n sum exp(lw - M) v (ng_estimator.py:146-152)
            float scale;
            if (snis) scale = S > 0.f ? 1.f / S : 0.f;
            else scale = n_used[c] > 0.f ? __expf(M[c]) / n_used[c] : __int_as_float(0x7fc00000);
            const float isq = packed[(size_t)k * stride + 2 * D + d0 + i];
            g_neg[(size_t)k * D + d0 + i] = -G * scale;
            h_neg[(size_t)k * D + d0 + i] = -H * scale * isq;
        }
    }
}

}  // namespace

extern "C" {

size_t gmmvi_diag_packed_stride(int D) { return (size_t)ds_stride(D); }

int gmmvi_diag_pack(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* sigma_dev, float* packed_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, means_dev && sigma_dev && packed_dev);
    GMMVI_PROF(ctx, "diag_pack");
    hipLaunchKernelGGL(diag_pack_kernel, dim3(K), dim3(64), 0, ctx->stream, D, means_dev, sigma_dev, packed_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_diag_mixture_eval(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* logw_dev, const float* logw2_dev,
                            const float* X_dev, int N, float* ld_out_dev, float* lp_out_dev, float* grad_out_dev,
                            float* lp2_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED && N >= 0);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, packed_dev && logw_dev && X_dev);
    GMMVI_ARG_CHECK(ctx, ld_out_dev || lp_out_dev || grad_out_dev);
    GMMVI_ARG_CHECK(ctx, (logw2_dev == nullptr) == (lp2_out_dev == nullptr));
    GMMVI_ARG_CHECK(ctx, logw2_dev == nullptr || lp_out_dev != nullptr);
    const bool want_grad = grad_out_dev != nullptr;
    const int tiles = (N + 63) / 64;
    const int nw = K < 8 ? K : 8;
    // component chunks so that ~2 workgroups per CU are in flight; every wave keeps at least two components
    int ky = 1;
    if (2L * tiles < 3L * ctx->num_cus) ky = (int)((2L * ctx->num_cus + tiles / 2) / tiles);
    if (ky > K / (2 * nw)) ky = K / (2 * nw);
    if (ky < 1) ky = 1;
    if (ky > 16) ky = 16;
    const int kchunk = (K + ky - 1) / ky;
    ky = (K + kchunk - 1) / kchunk;
    // scratch: ld when the caller does not want it but the gradient does, lp likewise, the chunk partials
    const bool need_lp = lp_out_dev != nullptr || want_grad;
    const size_t f_ld = (!ld_out_dev && want_grad) ? (size_t)K * N : 0;
    const size_t f_lp = (!lp_out_dev && want_grad) ? (size_t)N : 0;
    const size_t f_parts = ky > 1 ? (size_t)ky * N * (logw2_dev ? 2 : 1) : 0;
    int rc = gmmvi_ws_reserve(ctx, (f_ld + f_lp + f_parts) * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    float* wsf = (float*)ctx->ws;
    float* ld = ld_out_dev ? ld_out_dev : (f_ld ? wsf : nullptr);
    float* lp = lp_out_dev ? lp_out_dev : (f_lp ? wsf + f_ld : nullptr);
    float* parts = wsf + f_ld + f_lp;
    float* lp_k = ky > 1 ? parts : lp;
    float* lp2_k = logw2_dev ? (ky > 1 ? parts + (size_t)ky * N : lp2_out_dev) : nullptr;
    {
        GMMVI_PROF_UNITS(ctx, "diag_sweep", (double)N * K);
        hipLaunchKernelGGL(diag_eval_kernel, dim3(tiles, ky), dim3(64 * nw), 0, ctx->stream, K, D, packed_dev, logw_dev, logw2_dev,
                           X_dev, N, ld, need_lp ? lp_k : nullptr, lp2_k);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    if (ky > 1 && (need_lp || logw2_dev)) {
        GMMVI_PROF(ctx, "mixture_combine");
        rc = gmmvi_combine_partials_internal(ctx, ky, N, D, lp_k, nullptr, lp, nullptr, lp2_k, lp2_out_dev);
        if (rc != GMMVI_OK) return rc;
    }
    if (want_grad) {
        GMMVI_PROF_UNITS(ctx, "diag_grad", (double)N * K);
        const int gw = K < 8 ? K : 8;
        hipLaunchKernelGGL(diag_grad_kernel, dim3(tiles, (D + DS_CH - 1) / DS_CH), dim3(64 * gw), (size_t)gw * DS_CH * 64 * sizeof(float),
                           ctx->stream, K, D, packed_dev, logw_dev, X_dev, N, ld, lp, grad_out_dev);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    return GMMVI_OK;
}

int gmmvi_diag_sample(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* sigma_dev, const int32_t* offsets_dev,
                      int N, uint64_t seed, uint64_t first_index, int stream_id, const float* eps_dev, float* X_out_dev,
                      int32_t* mapping_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED && N >= 0);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, means_dev && sigma_dev && offsets_dev && X_out_dev);
    GMMVI_PROF(ctx, "diag_sample");
    const long items = (long)N * ((D + 3) / 4);
    hipLaunchKernelGGL(diag_sample_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream, K, D, means_dev,
                       sigma_dev, offsets_dev, N, seed, first_index, (uint32_t)stream_id, eps_dev, X_out_dev, mapping_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_diag_stein(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N, const float* ld_dev,
                     const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev, const int32_t* mapping_dev,
                     int map_offset, int flags, float* h_neg_diag_out_dev, float* g_neg_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM_BLOCKED && N >= 1);
    GMMVI_ARG_CHECK(ctx, packed_dev && X_dev && qgrad_dev && tgrad_dev && h_neg_diag_out_dev && g_neg_out_dev);
    if (flags & GMMVI_OWN_SAMPLES_ONLY) GMMVI_ARG_CHECK(ctx, mapping_dev != nullptr);
    else GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev);
    GMMVI_PROF_UNITS(ctx, "diag_stein", (double)N * K);
    hipLaunchKernelGGL(diag_stein_kernel, dim3((K + DST_CG - 1) / DST_CG, (D + DST_CH - 1) / DST_CH), dim3(256), 0, ctx->stream, K,
                       D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev, mapping_dev, map_offset, flags,
                       h_neg_diag_out_dev, g_neg_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // extern "C"
