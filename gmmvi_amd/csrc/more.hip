// MORE natural-gradient estimate (reference: optimization/gmmvi_modules/ng_estimator.py:266-376 with the quadratic
// ridge regression of optimization/least_squares.py:34-76,103-191), gfx950.
//
// Per component o the reference fits  R~(x) = x^T R x + x^T r + r0  to the rewards log p~(x) - log q(x) by importance-
// weighted ridge regression on whitened samples z = L_o^-1 (x - mu_o) with features
//     phi(z) = [ z_i z_j (i <= j, row-major upper triangle),  z,  1 ]          F = D(D+1)/2 + D + 1
// i.e. it solves  (Phi^T W Phi + lambda I') beta = Phi^T W rew   (bias row of I' is zero), then un-whitens.
//
//   more_lse_kernel     per component: log sum_n exp(ld[o,n] - bg[n])  (self-normalised weights, :353-356)
//   more_gram_kernel    the (F+1)x(F+1) Gram matrix of the rows  sqrt(w_n) [phi(z_n); rew_n]  -- one f32-MFMA SYRK per
//                       component: A = Phi^T W Phi, b = Phi^T W rew and sum w rew^2 in one contraction.  Workgroup =
//                       (component, sample chunk); per 64-sample tile every wave substitutes z (SGPR-fed L), the four
//                       waves write the weighted feature rows into one LDS image [F+1][64] and then contract it with
//                       v_mfma_f32_32x32x2_f32, each wave owning up to PP lower-triangular 32x32 block pairs whose
//                       accumulators stay in registers over the whole chunk.  Operands are read as ds_read_b128 (four
//                       k-steps per read; row stride 68 words = conflict-free for the b128 lane groups).
//   more_solve_kernel   per component: sums the chunk partials in fixed order into a packed lower triangle in LDS, adds the
//                       ridge, Cholesky-factorises in place (1024 threads), solves, maps the coefficients back
//                       (least_squares.py:177-189) and emits  H = L^-T Q_w L^-1  and  g = Q mu - lin = -L^-T lin_w.
//
// Sizes: the packed F x F triangle must fit the 160 KB LDS: F + 1 <= 256  <=>  D <= 21.  The ridge system is solved by
// Cholesky (the reference calls tf.linalg.solve = pivoted LU); a non-positive pivot marks the component's estimate as
// NaN, which the component updaters treat as a rejected update.
#include "common.h"
#include "subst.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int PHI_LD = 68;       // LDS row stride (words) of the feature image: 64 samples + 4 -> b128 reads conflict-free

template <int DP>
__device__ __forceinline__ void more_forward_subst(const float* __restrict__ P, const float (&x)[DP], float (&z)[DP]) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        float t = x[i] - P[PK::MU + i];
#pragma unroll
        for (int j = 0; j < i; ++j) t = fmaf(-P[PK::LROW + PK::rowofs(i) + j], z[j], t);
        z[i] = t * P[PK::RD + i];
    }
}

// log-normaliser of the importance weights of every component over the samples it uses
__global__ __launch_bounds__(1024) void more_lse_kernel(int N, const float* __restrict__ ld, const float* __restrict__ bg,
                                                        const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                        float* __restrict__ lse) {
    __shared__ float s_m[16], s_s[16];
    const int k = blockIdx.x, tid = threadIdx.x;
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    float m = -3.0e38f, s = 0.f;
    for (int n = tid; n < N; n += 1024) {
        float a;
        if (own_only) { if (mapping[n] + map_offset != k) continue; a = 0.f; }      // ng_estimator.py:110-118: lw = 0
        else a = ld[(size_t)k * N + n] - bg[n];
        if (!(a > -3.0e38f)) continue;
        if (a > m) { s = s * __expf(m - a) + 1.f; m = a; } else s += __expf(a - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        const float M = fmaxf(m, m2);
        s = s * __expf(m - M) + s2 * __expf(m2 - M);
        m = M;
    }
    if ((tid & 63) == 0) { s_m[tid >> 6] = m; s_s[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = s_m[0];
        for (int w = 1; w < 16; ++w) M = fmaxf(M, s_m[w]);
        float S = 0.f;
        for (int w = 0; w < 16; ++w) S += s_s[w] * __expf(s_m[w] - M);
        lse[k] = (S > 0.f) ? M + __logf(S) : 0.f;
    }
}

template <int DP, int PP>
__global__ __launch_bounds__(256, 2) void more_gram_kernel(int D, int N, int tiles_per_chunk, int nb,
                                                           const float* __restrict__ packed, const float* __restrict__ X,
                                                           const float* __restrict__ ld, const float* __restrict__ bg,
                                                           const float* __restrict__ tlp, const float* __restrict__ logq,
                                                           const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                           const float* __restrict__ lse, float* __restrict__ slab) {
    using PK = Pack<DP>;
    extern __shared__ float phi[];                     // [32 nb][PHI_LD]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = blockIdx.y, chunk = blockIdx.x, n_chunks = gridDim.x;
    const int T2 = D * (D + 1) / 2;                    // quadratic features
    const int F = T2 + D + 1;                          // features; row F carries the reward
    const int n_pairs = nb * (nb + 1) / 2;
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    const bool self_norm = (flags & GMMVI_SELF_NORMALIZED) != 0;
    const float* __restrict__ P = packed + (size_t)k * PK::STRIDE;
    const float lse_k = self_norm ? lse[k] : 0.f;

    // block pairs (bi >= bc) of this wave: p = wave + 4 pp
    int row_of[PP], col_of[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int p = wave + 4 * pp;
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
        row_of[pp] = (p < n_pairs) ? bi : -1;
        col_of[pp] = p - bi * (bi + 1) / 2;
    }
    f32x16 acc[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[pp][t] = 0.f;

    for (int e = tid; e < 32 * nb * PHI_LD; e += 256) phi[e] = 0.f;       // rows > F stay zero for the whole kernel
    __syncthreads();

    const int col = lane & 31, half = lane >> 5;
    const int tile_begin = chunk * tiles_per_chunk;
    const int tile_end = min((N + 63) / 64, tile_begin + tiles_per_chunk);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        // ---- every wave whitens the same 64 samples (lane = sample); each writes a quarter of the feature rows ----------
        const int n = tile * 64 + lane;
        const bool valid = n < N;
        float x[DP], z[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) x[i] = (valid && i < D) ? X[(size_t)n * D + i] : P[PK::MU + i];
        more_forward_subst<DP>(P, x, z);
        float sw = 0.f, rew = 0.f;
        if (valid) {
            float a;
            if (own_only) a = (mapping[n] + map_offset == k) ? 0.f : -3.0e38f;
            else a = ld[(size_t)k * N + n] - bg[n];
            if (a > -3.0e38f) sw = __expf(0.5f * (a - lse_k));           // sqrt of the importance weight (:353-358)
            rew = tlp[n] - logq[n];                                      // ng_estimator.py:346
        }
        const bool live = sw > 0.f;
        int f = 0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            if (i < D) {
                const float szi = sw * z[i];
#pragma unroll
                for (int j = i; j < DP; ++j) {
                    if (j < D) {
                        if ((f & 3) == wave) phi[f * PHI_LD + lane] = live ? szi * z[j] : 0.f;   // least_squares.py:113-124
                        ++f;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < D && ((T2 + i) & 3) == wave) phi[(T2 + i) * PHI_LD + lane] = live ? sw * z[i] : 0.f;
        if (((F - 1) & 3) == wave) phi[(F - 1) * PHI_LD + lane] = live ? sw : 0.f;
        if ((F & 3) == wave) phi[F * PHI_LD + lane] = live ? sw * rew : 0.f;
        __syncthreads();
        // ---- contraction over the 64 samples --------------------------------------------------------------------------
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            if (row_of[pp] < 0) continue;
            const float* pa = phi + (32 * row_of[pp] + col) * PHI_LD + 4 * half;
            const float* pb = phi + (32 * col_of[pp] + col) * PHI_LD + 4 * half;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 a4 = *reinterpret_cast<const float4*>(pa + 8 * q);
                const float4 b4 = *reinterpret_cast<const float4*>(pb + 8 * q);
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[pp], 0, 0, 0);
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[pp], 0, 0, 0);
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[pp], 0, 0, 0);
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[pp], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- partial Gram blocks of this chunk: slab[k][chunk][pair][32][32] ------------------------------------------------
    float* out = slab + ((size_t)k * n_chunks + chunk) * (size_t)n_pairs * 1024;
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        if (row_of[pp] < 0) continue;
        float* o = out + (size_t)(wave + 4 * pp) * 1024;
#pragma unroll
        for (int t = 0; t < 16; ++t) o[((t & 3) + 8 * (t >> 2) + 4 * half) * 32 + col] = acc[pp][t];
    }
}

__device__ __forceinline__ int tri_ofs(int i) { return i * (i + 1) / 2; }

// One workgroup per component: assemble, factorise, solve, un-whiten.
__global__ __launch_bounds__(1024) void more_solve_kernel(int D, int nb, int n_chunks, const float* __restrict__ slab,
                                                          const float* __restrict__ chols, const float* __restrict__ l2,
                                                          float* __restrict__ H_neg, float* __restrict__ g_neg) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    const int T2 = D * (D + 1) / 2;
    const int F = T2 + D + 1;
    const int n_pairs = nb * (nb + 1) / 2;
    float* tri = sm;                                   // packed lower triangle of the F x F system
    float* rhs = tri + tri_ofs(F);                     // [F]
    float* colv = rhs + F;                             // [F] scaled pivot column
    float* Ls = colv + F;                              // [D][D] component Cholesky factor
    float* Qs = Ls + D * D;                            // [D][D+1]  Q_w | lin_w
    float* Xs = Qs + D * (D + 1);                      // [D][D+1]
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;

    // ---- assemble: fixed-order sum over the sample chunks ------------------------------------------------------------
    const float* base = slab + (size_t)k * n_chunks * (size_t)n_pairs * 1024;
    const int il = tid >> 5, cl = tid & 31;
    for (int bi = 0, p = 0; bi < nb; ++bi)
        for (int bc = 0; bc <= bi; ++bc, ++p) {
            const int i = 32 * bi + il, c = 32 * bc + cl;
            if (i > F || c > i || c >= F) continue;
            float s = 0.f;
            for (int ch = 0; ch < n_chunks; ++ch) s += base[((size_t)ch * n_pairs + p) * 1024 + il * 32 + cl];
            if (i < F) tri[tri_ofs(i) + c] = s;
            else rhs[c] = s;
        }
    for (int e = tid; e < D * D; e += 1024) Ls[e] = chols[(size_t)k * D * D + e];
    __syncthreads();
    const float ridge = l2[k];
    for (int i = tid; i < F - 1; i += 1024) tri[tri_ofs(i) + i] += ridge;          // least_squares.py:71-73 (bias unregularised)
    __syncthreads();

    // ---- in-place Cholesky, right-looking, thread (ti, tc) covers rows ti (mod 32), columns tc (mod 32) -----------------
    for (int j = 0; j < F; ++j) {
        const float d = tri[tri_ofs(j) + j];
        if (!(d > 0.f)) { if (tid == 0) s_fail = 1; break; }                      // uniform: every thread reads the same d
        const float inv = rsqrtf(d);
        __syncthreads();                                                           // everyone has read the pivot
        for (int i = j + tid; i < F; i += 1024) {
            const float v = (i == j) ? d * inv : tri[tri_ofs(i) + j] * inv;
            tri[tri_ofs(i) + j] = v;
            colv[i] = v;
        }
        __syncthreads();
        for (int i = j + 1 + il; i < F; i += 32) {
            const float li = colv[i];
            float* rowp = tri + tri_ofs(i);
            for (int c = j + 1 + cl; c <= i; c += 32) rowp[c] = fmaf(-li, colv[c], rowp[c]);
        }
        __syncthreads();
    }
    __syncthreads();
    const bool fail = s_fail != 0;
    if (!fail) {
        // forward: L y = b
        for (int j = 0; j < F; ++j) {
            const float yj = rhs[j] / tri[tri_ofs(j) + j];
            for (int i = j + 1 + tid; i < F; i += 1024) rhs[i] = fmaf(-tri[tri_ofs(i) + j], yj, rhs[i]);
            __syncthreads();
        }
        for (int i = tid; i < F; i += 1024) rhs[i] /= tri[tri_ofs(i) + i];
        __syncthreads();
        // backward: L^T beta = y
        for (int j = F - 1; j >= 0; --j) {
            const float bj = rhs[j] / tri[tri_ofs(j) + j];
            const float* rowp = tri + tri_ofs(j);
            for (int i = tid; i < j; i += 1024) rhs[i] = fmaf(-rowp[i], bj, rhs[i]);
            __syncthreads();
        }
        for (int i = tid; i < F; i += 1024) colv[i] = rhs[i] / tri[tri_ofs(i) + i];  // beta
        __syncthreads();
        // Q_w = -(Qt + Qt^T) with Qt the upper-triangular fill of the quadratic coefficients (least_squares.py:177-179)
        for (int e = tid; e < D * (D + 1); e += 1024) {
            const int i = e / (D + 1), j = e % (D + 1);
            float v;
            if (j == D) v = colv[T2 + i];                                            // lin_w
            else {
                const int a = min(i, j), b = max(i, j);
                const float q = colv[a * D - a * (a - 1) / 2 + (b - a)];
                v = (a == b) ? -2.f * q : -q;
            }
            Qs[e] = v;
        }
        __syncthreads();
        // X = L^-T [Q_w | lin_w]   (lane = column)
        if (tid <= D) {
            for (int i = D - 1; i >= 0; --i) {
                float t = Qs[i * (D + 1) + tid];
                for (int j = i + 1; j < D; ++j) t = fmaf(-Ls[j * D + i], Xs[j * (D + 1) + tid], t);
                Xs[i * (D + 1) + tid] = t / Ls[i * D + i];
            }
        }
        __syncthreads();
        // H = X L^-1: row r of H solves L^T h = (row r of X)^T   (least_squares.py:185); g = Q mu - lin = -L^-T lin_w (:186-188,
        // ng_estimator.py:371-373)
        if (tid < D) {
            float* hrow = Qs + tid * (D + 1);                                        // reuse: row tid only touched by this lane
            for (int i = D - 1; i >= 0; --i) {
                float t = Xs[tid * (D + 1) + i];
                for (int j = i + 1; j < D; ++j) t = fmaf(-Ls[j * D + i], hrow[j], t);
                hrow[i] = t / Ls[i * D + i];
            }
            for (int i = 0; i < D; ++i) H_neg[((size_t)k * D + tid) * D + i] = hrow[i];
            g_neg[(size_t)k * D + tid] = -Xs[tid * (D + 1) + D];
        }
    } else {
        const float nanv = __int_as_float(0x7fc00000);
        for (int e = tid; e < D * D; e += 1024) H_neg[(size_t)k * D * D + e] = nanv;
        for (int e = tid; e < D; e += 1024) g_neg[(size_t)k * D + e] = nanv;
    }
}

template <int DP, int PP>
int launch_more(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* chols, const float* X, int N,
                const float* ld, const float* logq, const float* bg, const float* tlp, const int32_t* mapping,
                int map_offset, int flags, const float* l2, float* H_neg, float* g_neg) {
    const int F = D * (D + 1) / 2 + D + 1;
    const int nb = (F + 1 + 31) / 32;
    const int n_pairs = nb * (nb + 1) / 2;
    if (n_pairs > 4 * PP) return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: internal tiling error");
    const int tiles = (N + 63) / 64;
    int n_chunks = (2 * ctx->num_cus + K - 1) / K;             // two workgroups per CU
    if (n_chunks > tiles) n_chunks = tiles;
    if (n_chunks < 1) n_chunks = 1;
    const int tiles_per_chunk = (tiles + n_chunks - 1) / n_chunks;
    n_chunks = (tiles + tiles_per_chunk - 1) / tiles_per_chunk;
    const size_t slab_floats = (size_t)K * n_chunks * n_pairs * 1024;
    int rc = gmmvi_ws_reserve(ctx, (slab_floats + (size_t)K) * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    float* slab = (float*)ctx->ws;
    float* lse = slab + slab_floats;
    if (flags & GMMVI_SELF_NORMALIZED) {
        GMMVI_PROF(ctx, "more_lse");
        hipLaunchKernelGGL(more_lse_kernel, dim3(K), dim3(1024), 0, ctx->stream, N, ld, bg, mapping, map_offset, flags, lse);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    const size_t gram_lds = (size_t)32 * nb * PHI_LD * sizeof(float);
    static size_t gram_attr = 0;
    if (gram_lds > gram_attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)more_gram_kernel<DP, PP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds));
        gram_attr = gram_lds;
    }
    {
        GMMVI_PROF(ctx, "more_gram");
        hipLaunchKernelGGL((more_gram_kernel<DP, PP>), dim3(n_chunks, K), dim3(256), gram_lds, ctx->stream, D, N,
                           tiles_per_chunk, nb, packed, X, ld, bg, tlp, logq, mapping, map_offset, flags, lse, slab);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    const size_t solve_lds = ((size_t)F * (F + 1) / 2 + 2 * (size_t)F + (size_t)D * D + 2 * (size_t)D * (D + 1)) * sizeof(float);
    static size_t solve_attr = 0;
    if (solve_lds > solve_attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)more_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)solve_lds));
        solve_attr = solve_lds;
    }
    GMMVI_PROF(ctx, "more_solve");
    hipLaunchKernelGGL(more_solve_kernel, dim3(K), dim3(1024), solve_lds, ctx->stream, D, nb, n_chunks, slab, chols, l2,
                       H_neg, g_neg);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

}  // namespace

extern "C" int gmmvi_more(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* chols_dev,
                          const float* X_dev, int N, const float* ld_dev, const float* logq_dev, const float* bg_dev,
                          const float* tlp_dev, const int32_t* mapping_dev, int map_offset, int flags,
                          const float* l2_dev, float* H_neg_out_dev, float* g_neg_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && N >= 1);
    if (D > GMMVI_MORE_MAX_DIM)
        return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: D must be <= 21 (the F x F ridge system is solved in LDS)");
    GMMVI_ARG_CHECK(ctx, packed_dev && chols_dev && X_dev && logq_dev && tlp_dev && l2_dev && H_neg_out_dev && g_neg_out_dev);
    if (flags & GMMVI_OWN_SAMPLES_ONLY) GMMVI_ARG_CHECK(ctx, mapping_dev != nullptr);
    else GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev);
#define GMMVI_MORE_CASE(DPV, PPV)                                                                                    \
    return launch_more<DPV, PPV>(ctx, K, D, packed_dev, chols_dev, X_dev, N, ld_dev, logq_dev, bg_dev, tlp_dev,      \
                                 mapping_dev, map_offset, flags, l2_dev, H_neg_out_dev, g_neg_out_dev)
    switch (gmmvi_padded_dim(D)) {
        case 2: GMMVI_MORE_CASE(2, 1);
        case 4: GMMVI_MORE_CASE(4, 1);
        case 8: GMMVI_MORE_CASE(8, 1);
        case 10: GMMVI_MORE_CASE(10, 2);
        case 12: GMMVI_MORE_CASE(12, 2);
        case 16: GMMVI_MORE_CASE(16, 4);
        case 20: GMMVI_MORE_CASE(20, 9);
        case 24: GMMVI_MORE_CASE(24, 9);
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: unsupported dimension");
    }
#undef GMMVI_MORE_CASE
}
