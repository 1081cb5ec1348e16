// Importance-weighted Stein estimate of the per-component expected gradient / Hessian
// (gmmvi_modules/ng_estimator.py:204-263, :171-188 self-normalised, :154-169 plain importance weights).
//
// For component k:  sum_n e_kn g_n y_kn^T,  e_kn = exp(ld[k,n] - bg[n] - m),  g_n = grad log p~ - grad log q,
// y_kn = Sigma_k^-1 (x_n - mu_k).  The estimate is linear in y, so Sigma_k^-1 is applied once per component AFTER the sum over
// the samples: stein_moment_kernel accumulates the raw moment matrix  A_k = sum_n e_kn [g_n; 1] [x_n - mu_k; 1]^T  on the
// matrix cores ((D+1)x(D+1): sum e g d^T, sum e g in the last column, sum e in the corner), stein_finalize sums the
// per-range partials in fixed order (bitwise reproducible), applies L^-T L^-1 from the right, normalises, symmetrises and
// negates.  No per-sample triangular substitution is left on this path (the reference forms y per sample, :165-166, :184).
#include "common.h"
#include "blocked.h"
#include "wave_reduce.h"
#include "stein_finalize.h"
#include "stein_tile.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_max(float v) { return gmmvi_wave_max(v); }

template <int DP>
__global__ __launch_bounds__(1024) void stein_finalize_kernel(int D, int R, int N, int flags, const float* __restrict__ part,
                                                             const float* __restrict__ part_m, float* __restrict__ H_neg,
                                                             float* __restrict__ g_neg, const float* __restrict__ packed) {
    extern __shared__ float fin_lds[];
    stein_finalize_component<DP>(fin_lds, blockIdx.x, D, R, N, flags, part, part_m, H_neg, g_neg, packed);
}

template <int DP>
static int launch_stein_finalize(gmmvi_ctx* ctx, int K, int D, int R, int N, int flags, const float* part,
                                 const float* part_m, float* H_neg, float* g_neg, const float* packed) {
    const size_t shmem = stein_finalize_lds_floats(DP, D, R) * sizeof(float);
    static size_t attr = 64 * 1024;
    if (shmem > attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_finalize_kernel<DP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr = shmem;
    }
    // the products of the explicit-inverse route use every thread; the substitution route only D of them
    const int threads = Pack<DP>::FRAGS ? 1024 : 256;
    GMMVI_PROF(ctx, "stein_finalize");
    hipLaunchKernelGGL(stein_finalize_kernel<DP>, dim3(K), dim3(threads), shmem, ctx->stream, D, R, N, flags, part, part_m, H_neg,
                       g_neg, packed);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// =====================================================================================================================
// Moment form (all D <= 63): the estimate is LINEAR in y = Sigma_k^-1 (x - mu_k), so nothing has to be whitened per sample:
//     A_k = sum_n e_kn [g_n; 1] [x_n - mu_k; 1]^T            (one dense contraction over the samples)
//     sum_n e g y^T = A_k[:D,:D] Sigma_k^-1 = A_k[:D,:D] L_k^-T L_k^-1   (two triangular solves per COMPONENT, stein_finalize)
// The contraction is a GEMM whose left operand [g; 1] is common to all components; the right operands of NB components are
// stacked along the N side of the matrix-core tile (NB (D+1) columns in NT 16-column tiles), so the padding of a (D+1)-wide
// tile is paid once per stack on the M side only (D = 20: 21 rows in 2 x 16, 3 x 21 = 63 of 64 columns).
// v_mfma_f32_16x16x4_f32: A lane l -> A[i = l & 15][k = l >> 4], B lane l -> B[k = l >> 4][j = l & 15],
// D lane l, reg r -> D[i = 4 (l >> 4) + r][j = l & 15]  (tools/probe/mfma_f32_16x16x4_layout.hip).
//
// Grid (stack of NB components, sample range); 4 waves per workgroup, all on the same stack, the range split evenly over
// them.  A wave works through its samples in chunks of 64 WITHOUT any workgroup barrier: lane = sample loads its x / g row,
// transposed into a wave-private feature-major LDS image T[feature][sample] (row stride 72 floats = 18 sixteen-byte slots: a
// ds_read_b128 is served in four 16-lane groups -- lanes {0-3, 12-15, 20-27}, ... -- and within a group the rows c16 in
// {0-3, 12-15} (k-slot q) and {4-11} (k-slot q + 1) of the [g; 1] operand fall on sixteen different slots: conflict-free.
// The STACKED [x - mu; 1] operand reads rows (16 nt + c16) mod (D + 1), which wrap around the D + 1 rows of a component:
// rows eight apart then meet on one slot (D = 20, nt = 1: 16 / 8, 17 / 9, 18 / 10, 20 / 4) -- two-way conflicts on part of
// its reads, 55 % of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT, with LDS ~6 % of the wave cycles.  A per-16-rows skew
// of the image was measured and made it worse (3.7e5 -> 5.0e5 conflict cycles per launch: it breaks the other column
// tiles); an image in stacked order would triple the x stores.  Left as it is.), the importance weights e = exp(ld - bg - M) with lane = sample (M: the maximum over the wave's
// samples, found by a first pass over the log weights alone), then per 16 samples: MT + 2 NT ds_read_b128 (four MFMA steps each),
// (x - mu) e on the vector unit, 4 MT NT MFMAs.  Which sample sits in which k-slot of which step is free (the contraction
// sums over all of them): the lane with k-slot q takes samples 16 u + 4 q + {0,1,2,3} for its four steps of block u, which
// are contiguous in the image.
// The four waves are merged through LDS in fixed order (common maximum); one partial per (component, range) goes to the
// slab that stein_finalize sums -- K x R x (D+1)^2 floats with R ~ CUs / stacks ranges instead of one per 256 samples.
// =====================================================================================================================
// A lane's row of a row-major [*, D] array into registers, VW floats per load (VW > 1: D == DP, rows VW*4-byte aligned); the
// loads of a lane walk the same cache lines, only the first goes past the L1.  No per-lane predication: rows beyond the
// range are CLAMPED to a valid row by the caller (finite data, weight 0).
template <int DP, int VW>
__device__ __forceinline__ void sm_load_row(const float* __restrict__ row, int D, float (&v)[DP]) {
#pragma unroll
    for (int f = 0; f < DP; f += VW) {
        if (VW > 1 || f < SteinTile<DP>::PREV + 1 || f < D) {      // only the last few columns of an inexact fit branch
            if constexpr (VW == 4) {
                const float4 t = *reinterpret_cast<const float4*>(row + f);
                v[f] = t.x; v[f + 1] = t.y; v[f + 2] = t.z; v[f + 3] = t.w;
            } else if constexpr (VW == 2) {
                const float2 t = *reinterpret_cast<const float2*>(row + f);
                v[f] = t.x; v[f + 1] = t.y;
            } else {
                v[f] = row[f];
            }
        } else {
            v[f] = 0.f;
        }
    }
}

// registers -> feature-major image T[f][lane]: consecutive lanes, consecutive addresses, immediate offsets
template <int DP, bool EXACT>
__device__ __forceinline__ void sm_store_rows(float* t, int D, const float (&v)[DP]) {
#pragma unroll
    for (int f = 0; f < DP; ++f)
        if (EXACT || f < SteinTile<DP>::PREV + 1 || f < D) t[f * SM_RS] = v[f];
}

// VW > 1: the fast instances, D == DP (compile time) and all weights from ld - bg; VW == 1: any D in the class, scalar row
// loads, own-samples weights on request
template <int DP, int VW>
__global__ __launch_bounds__(256, (SteinTile<DP>::MT <= 2 ? 2 : 1))
void stein_moment_kernel(int K, int D_rt, int N, int wave_range, int stacks, int R, const float* __restrict__ packed,
                         const float* __restrict__ X, const float* __restrict__ TG, const float* __restrict__ QG,
                         const float* __restrict__ ld, const float* __restrict__ bg, const int32_t* __restrict__ mapping,
                         int map_offset, int own_rt, float* __restrict__ part, float* __restrict__ part_m) {
    using ST = SteinTile<DP>;
    constexpr int MT = ST::MT, NT = ST::NT, NB = ST::NB;
    constexpr bool EXACT = VW > 1;
    const int D = EXACT ? DP : D_rt;
    const bool own = EXACT ? false : own_rt != 0;
    extern __shared__ float sm[];
    __shared__ float sm_m[4][NB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D1 = D + 1;
    // work item (stack, range), range-major.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
    // workgroup b takes item (b % 8) * ceil(items / 8) + b / 8: an XCD then works through CONTIGUOUS items, i.e. one or two
    // sample ranges for every stack, and only those rows of x / grad / ld pass through its L2 (placement is for speed only)
    const int per_xcd = (stacks * R + 7) / 8;
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (item >= stacks * R) return;
    const int range_id = item / stacks;
    const int k0 = (item - range_id * stacks) * NB;
    const int nb = min(NB, K - k0);
    // wave-private images: Tg rows 0..D-1 = g, row D = 1 (valid samples), row D+1 = 0; Tx rows 0..D-1 = x, row D = 1;
    // Te rows 0..NB-1 = importance weights, row NB = 0
    const int rows_g = D1 + 1, rows_x = D1;
    constexpr int rows_e = NB + 1;
    const int wave_floats = (rows_g + rows_x + rows_e) * SM_RS;
    const int wave_stride = max(wave_floats, 16 * MT * (16 * NT + 1));      // (the region also takes the wave's result tile)
    float* Tg = sm + (size_t)wave * wave_stride;
    float* Tx = Tg + rows_g * SM_RS;
    float* Te = Tx + rows_x * SM_RS;
    for (int e = lane; e < SM_RS; e += 64) { Tg[D1 * SM_RS + e] = 0.f; Te[NB * SM_RS + e] = 0.f; Tx[D * SM_RS + e] = 1.f; }

    // ---- per-lane operand addresses: A rows of the MT row tiles, (x row, e row, mu) of the NT column tiles ---------------
    const int q = lane >> 4, c16 = lane & 15;
    int a_off[MT], x_off[NT], e_off[NT], cidx[NT];
    float mu[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + c16;
        a_off[mt] = (row < D1 ? row : D1) * SM_RS + 4 * q;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = 16 * nt + c16;
        const int comp = c / D1, j = c - comp * D1;
        const bool valid = comp < nb;
        cidx[nt] = valid ? comp : NB;
        x_off[nt] = (valid ? j : 0) * SM_RS + 4 * q;
        e_off[nt] = (valid ? comp : NB) * SM_RS + 4 * q;
        mu[nt] = (valid && j < D) ? packed[(size_t)(k0 + comp) * Pack<DP>::STRIDE + j] : 0.f;
    }
    const int w_begin = min(N, (range_id * 4 + wave) * wave_range);
    const int w_end = min(N, w_begin + wave_range);

    // log importance weight of (component slot c, sample n0 + lane).  Samples / components beyond the range are CLAMPED to
    // valid ones and their weight forced to "-inf" by a per-lane upper limit (a min, not a branch: the loads stay
    // unconditional and in flight together)
    float c_lim[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) c_lim[c] = c < nb ? 3.0e38f : -3.0e38f;
    auto log_weights = [&](int n0, float (&a)[NB]) {
        const int n = min(n0 + lane, N - 1);
        const float lim = (n0 + lane < w_end) ? 3.0e38f : -3.0e38f;
        if (own) {
            const int mp = mapping[n] + map_offset;
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = fminf((mp == k0 + c) ? 0.f : -3.0e38f, fminf(lim, c_lim[c]));
        } else {
            const float bgv = bg[n];
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = fminf(ld[(size_t)min(k0 + c, K - 1) * N + n] - bgv, fminf(lim, c_lim[c]));
        }
    };

    float a[NB], xr[DP], tr[DP], qr[DP];
    auto fetch = [&](int n0) {
        const size_t row = (size_t)min(n0 + lane, N - 1) * D;
        log_weights(n0, a);
        sm_load_row<DP, VW>(X + row, D, xr);
        sm_load_row<DP, VW>(TG + row, D, tr);
        sm_load_row<DP, VW>(QG + row, D, qr);
    };
    if (MT >= 4 && w_begin < w_end) fetch(w_begin);            // many row tiles: the first chunk's rows travel while pass 1 runs

    // ---- pass 1: the maximum of the log weights over this wave's samples, per component: the weights of pass 2 are referred
    // to it from the start, the accumulators never have to be rescaled ------------------------------------------------
    float M[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) M[c] = -3.0e38f;
    for (int n0 = w_begin; n0 < w_end; n0 += 64) {
        float a1[NB];
        log_weights(n0, a1);
#pragma unroll
        for (int c = 0; c < NB; ++c) M[c] = fmaxf(M[c], a1[c]);
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) M[c] = wave_max(M[c]);

    // ---- pass 2: chunks of 64 samples, software-pipelined: the rows of chunk i + 1 are fetched into registers while the
    // matrix cores work on chunk i.  Rows beyond the range are clamped to real rows (finite data) and carry weight 0, so
    // nothing has to be zeroed.
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
    if (MT < 4 && w_begin < w_end) fetch(w_begin);
    for (int n0 = w_begin; n0 < w_end; n0 += 64) {
        const int n_here = min(64, w_end - n0);
        sm_store_rows<DP, EXACT>(Tx + lane, D, xr);
#pragma unroll
        for (int f = 0; f < DP; ++f) tr[f] -= qr[f];               // g = grad log p~ - grad log q (:248)
        sm_store_rows<DP, EXACT>(Tg + lane, D, tr);
#pragma unroll
        for (int c = 0; c < NB; ++c) Te[c * SM_RS + lane] = (a[c] > -1.0e38f) ? __expf(a[c] - M[c]) : 0.f;
        Tg[D * SM_RS + lane] = 1.f;
        WAVE_LDS_SYNC();
        fetch(n0 + 64);                                            // past the end: clamped rows, never used
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (16 * u < n_here) {
                f32x4 av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(Tg + a_off[mt] + 16 * u);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(Tx + x_off[nt] + 16 * u);
                    const f32x4 ev = *reinterpret_cast<const f32x4*>(Te + e_off[nt] + 16 * u);
                    bv[nt] = (xv - mu[nt]) * ev;
                }
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][sp], bv[nt][sp], acc[mt][nt], 0, 0, 0);
            }
        }
        WAVE_LDS_SYNC();
    }

    if constexpr (MT >= 4) {
        // ---- merge the four waves (common maximum per component, fixed order), one partial per (component, range): every wave
        // parks its tile in its OWN image region (nobody else reads or writes it: no barrier in front), one barrier, then the
        // store loop adds the four tiles, each scaled to the common maximum ---------------------------------------------------
        constexpr int CW = 16 * NT + 1;
        float* C = sm + (size_t)wave * wave_stride;        // [16 MT][CW]
    #pragma unroll
        for (int nt = 0; nt < NT; ++nt)
    #pragma unroll
            for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                for (int r = 0; r < 4; ++r) C[(16 * mt + 4 * q + r) * CW + 16 * nt + c16] = acc[mt][nt][r];
        if (lane == 0) {
    #pragma unroll
            for (int c = 0; c < NB; ++c) sm_m[wave][c] = M[c];
        }
        __syncthreads();
        const int DD = D1 * D1;
    #pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {                                   // uniform
                const float m0 = sm_m[0][c], m1 = sm_m[1][c], m2 = sm_m[2][c], m3 = sm_m[3][c];
                const float Mall = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
                const float f0 = __expf(m0 - Mall), f1 = __expf(m1 - Mall), f2 = __expf(m2 - Mall), f3 = __expf(m3 - Mall);
                float* dst = part + ((size_t)(k0 + c) * R + range_id) * DD;
                for (int el = tid; el < DD; el += 256) {
                    const int i = el / D1, j = el - i * D1;
                    const float* src = sm + i * CW + c * D1 + j;
                    // (explicit fused multiply-adds: every instance of the kernel rounds the same way)
                    dst[el] = fmaf(src[3 * wave_stride], f3, fmaf(src[2 * wave_stride], f2, fmaf(src[wave_stride], f1, src[0] * f0)));
                }
                if (tid == 0) part_m[(size_t)(k0 + c) * R + range_id] = Mall;
            }
        }
    } else {
        // (up to three row tiles the tiles are small and the four rounds below cost less than the scaled four-way sum: D = 20
        // 24.4 against 25.0 us, D = 40 72.6 against 75.4; D = 50: 89.5 against 80.3)
        // ---- merge the four waves (common maximum per component, fixed order), one partial per (component, range) ---------
        if (lane == 0) {
    #pragma unroll
            for (int c = 0; c < NB; ++c) sm_m[wave][c] = M[c];
        }
        __syncthreads();                                   // also: every wave is done with its images
        float Mall[NB], fsc[NB];
    #pragma unroll
        for (int c = 0; c < NB; ++c) {
            Mall[c] = fmaxf(fmaxf(sm_m[0][c], sm_m[1][c]), fmaxf(sm_m[2][c], sm_m[3][c]));
            fsc[c] = __expf(M[c] - Mall[c]);
        }
        constexpr int CW = 16 * NT + 1;
        float* C = sm;                                     // [16 MT][CW]
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
    #pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float fs = 1.f;
    #pragma unroll
                    for (int c = 0; c < NB; ++c) fs = (cidx[nt] == c) ? fsc[c] : fs;
    #pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* dst = C + (16 * mt + 4 * q + r) * CW + 16 * nt + c16;
                            const float v = acc[mt][nt][r] * fs;
                            *dst = (w == 0) ? v : *dst + v;
                        }
                }
            }
            __syncthreads();
        }
        const int DD = D1 * D1;
        for (int el = tid; el < nb * DD; el += 256) {
            const int comp = el / DD, rem = el - comp * DD;
            const int i = rem / D1, j = rem - i * D1;
            part[((size_t)(k0 + comp) * R + range_id) * DD + rem] = C[i * CW + comp * D1 + j];
        }
        if (tid < nb) part_m[(size_t)(k0 + tid) * R + range_id] = Mall[tid];
    }
}

template <int DP, int VW>
static int launch_stein_moment(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                               const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping,
                               int map_offset, int flags, float* H_neg, float* g_neg, SteinSlab* slab_out) {
    using ST = SteinTile<DP>;
    const int D1 = D + 1;
    const int stacks = (K + ST::NB - 1) / ST::NB;
    // sample ranges: the grid fits the chip in ONE round of resident workgroups (a few workgroups over that would double the
    // time), at least 64 samples per wave
    // workgroups per CU: two for one row tile (D <= 15: few MFMAs per sample, the loads and the vector work of a second
    // workgroup fill the gaps: C4 28.0 against 33.3 us), one from two row tiles (D = 20: 143.4 against 144.6 us per iteration
    // with half the partial slab -- f32 MFMA and vector work of co-resident waves do not overlap)
    const int wgs_per_cu = ST::MT == 1 ? 2 : 1;
    static const int env_wgs = getenv("GMMVI_STEIN_WGS_PER_CU") ? atoi(getenv("GMMVI_STEIN_WGS_PER_CU")) : 0;
    int R = ((env_wgs > 0 ? env_wgs : wgs_per_cu) * ctx->num_cus) / stacks;
    if (R > (N + 255) / 256) R = (N + 255) / 256;
    if (R < 1) R = 1;
    int wave_range = (((N + R - 1) / R + 3) / 4 + 3) / 4 * 4;        // multiple of 4 samples (one MFMA k-step)
    R = (N + 4 * wave_range - 1) / (4 * wave_range);
    const size_t part_floats = (size_t)K * R * D1 * D1;
    int rc = gmmvi_ws_reserve(ctx, (part_floats + (size_t)K * R) * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    float* part = (float*)ctx->ws;
    float* part_m = part + part_floats;
    const int per_xcd = (stacks * R + 7) / 8;
    {
        size_t floats = (size_t)(2 * D1 + ST::NB + 2) * SM_RS;                 // a wave's images ...
        if (floats < (size_t)16 * ST::MT * (16 * ST::NT + 1)) floats = (size_t)16 * ST::MT * (16 * ST::NT + 1);     // ... or its result tile
        const size_t shmem = 4 * floats * sizeof(float);
        static size_t attr = 64 * 1024;
        if (shmem > attr) {
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_moment_kernel<DP, VW>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr = shmem;
        }
        GMMVI_PROF(ctx, "stein_partial");
        hipLaunchKernelGGL((stein_moment_kernel<DP, VW>), dim3(8 * per_xcd), dim3(256), shmem, ctx->stream, K, D, N, wave_range,
                           stacks, R, packed, X, tgrad, qgrad, ld, bg, mapping, map_offset,
                           (flags & GMMVI_OWN_SAMPLES_ONLY) ? 1 : 0, part, part_m);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    if (slab_out != nullptr) {                         // the caller finishes the estimate (update kernel prologue, fused.hip)
        slab_out->part = part; slab_out->part_m = part_m; slab_out->R = R;
        return GMMVI_OK;
    }
    return launch_stein_finalize<DP>(ctx, K, D, R, N, flags, part, part_m, H_neg, g_neg, packed);
}

static int stein_moment(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                        const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping, int map_offset,
                        int flags, float* H_neg, float* g_neg, SteinSlab* slab_out) {
    const int dp = gmmvi_padded_dim(D);
    // fast instance: D is exactly the padded dimension, rows aligned to the widest load their length allows, weights from ld - bg
    const uintptr_t bases = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(tgrad) | reinterpret_cast<uintptr_t>(qgrad);
    const int vw_nat = (dp % 4 == 0) ? 4 : 2;                         // every padded dimension is even
    const bool fast = D == dp && !(flags & GMMVI_OWN_SAMPLES_ONLY) && bases % (4 * vw_nat) == 0;
#define GMMVI_SM(DPV)                                                                                                  \
    case DPV:                                                                                                          \
        if (fast)                                                                                                      \
            return launch_stein_moment<DPV, (DPV % 4 == 0 ? 4 : 2)>(ctx, K, D, packed, X, N, ld, qgrad, bg, tgrad, mapping, \
                                                                    map_offset, flags, H_neg, g_neg, slab_out);         \
        return launch_stein_moment<DPV, 1>(ctx, K, D, packed, X, N, ld, qgrad, bg, tgrad, mapping, map_offset, flags, H_neg, g_neg, slab_out)
    switch (dp) {
        GMMVI_SM(2); GMMVI_SM(4); GMMVI_SM(8); GMMVI_SM(10); GMMVI_SM(12); GMMVI_SM(16); GMMVI_SM(20); GMMVI_SM(24);
        GMMVI_SM(32); GMMVI_SM(40); GMMVI_SM(50); GMMVI_SM(64);
        default: break;
    }
#undef GMMVI_SM
    return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_stein: unsupported dimension (D must be <= 63)");
}

extern "C" int gmmvi_stein(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N,
                           const float* ld_dev, const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev,
                           const int32_t* mapping_dev, int map_offset, int flags, float* H_neg_out_dev,
                           float* g_neg_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && (D < GMMVI_MAX_DIM || gmmvi_is_blocked_dim(D)) && N >= 1);
    GMMVI_ARG_CHECK(ctx, packed_dev && X_dev && qgrad_dev && tgrad_dev && H_neg_out_dev && g_neg_out_dev);
    if (flags & GMMVI_OWN_SAMPLES_ONLY) GMMVI_ARG_CHECK(ctx, mapping_dev != nullptr);
    else GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev);
    if (gmmvi_is_blocked_dim(D))
        return gmmvi_blocked_stein(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev, mapping_dev,
                                   map_offset, flags, H_neg_out_dev, g_neg_out_dev);
    return stein_moment(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev, mapping_dev, map_offset, flags,
                        H_neg_out_dev, g_neg_out_dev, nullptr);
}

// C++ linkage (common.h): the moment matrices only -- the partial slab stays in the context's scratch and the caller finishes
// the estimate (gmmvi_update_components_kl_from_slab: as the prologue of the update kernel where that is instantiated)
int gmmvi_stein_partials(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N, const float* ld_dev,
                         const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev, int flags, SteinSlab* slab) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D < GMMVI_MAX_DIM && !gmmvi_is_blocked_dim(D) && N >= 1 && slab != nullptr);
    GMMVI_ARG_CHECK(ctx, packed_dev && X_dev && qgrad_dev && tgrad_dev && ld_dev && bg_dev && !(flags & GMMVI_OWN_SAMPLES_ONLY));
    return stein_moment(ctx, K, D, packed_dev, X_dev, N, ld_dev, qgrad_dev, bg_dev, tgrad_dev, nullptr, 0, flags, nullptr, nullptr,
                        slab);
}

int gmmvi_stein_finalize_slab(gmmvi_ctx* ctx, int K, int D, const SteinSlab& slab, int N, int flags, const float* packed_dev,
                              float* H_neg_out_dev, float* g_neg_out_dev) {
    switch (gmmvi_padded_dim(D)) {
#define GMMVI_FIN(DPV) case DPV: return launch_stein_finalize<DPV>(ctx, K, D, slab.R, N, flags, slab.part, slab.part_m,       \
                                                                  H_neg_out_dev, g_neg_out_dev, packed_dev)
        GMMVI_FIN(2); GMMVI_FIN(4); GMMVI_FIN(8); GMMVI_FIN(10); GMMVI_FIN(12); GMMVI_FIN(16); GMMVI_FIN(20); GMMVI_FIN(24);
        GMMVI_FIN(32); GMMVI_FIN(40); GMMVI_FIN(50); GMMVI_FIN(64);
#undef GMMVI_FIN
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "stein_finalize: unsupported dimension");
    }
}
