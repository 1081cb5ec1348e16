// Component packing and the fused {sample tile x component} density / log-sum-exp / gradient kernel.
//
// Mapping (DESIGN.md "mixture_eval"): one lane owns one sample (x, z, y and the running gradient live in VGPRs),
// one wave walks a strided subset of the components, a workgroup = 64 samples x W waves.  The component
// parameters are wave-uniform, so they are fetched with scalar loads (s_load_dwordxN through the constant
// cache) and feed v_fma as SGPR operands: no LDS traffic and no per-lane loads in the inner loop.  The triangular
// solve is fully unrolled for the padded dimension DP; partial (max, sum, gradient) of the W waves are merged
// through LDS.
#include "common.h"
#include "combine.h"
#include "riders.h"
#include "subst.h"
#include "subst_phased.h"
#include "subst_pk.h"
#include "blocked.h"
#include <cmath>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------------------
// pack: (means, chols) -> kernel-side blocks; optional explicit inverse (sample_db.py:121)
// ---------------------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(64) void pack_kernel(int family, float nu, int K, int D, const float* __restrict__ means,
                                                  const float* __restrict__ chols, float* __restrict__ packed,
                                                  float* __restrict__ inv_chols) {
    using P = Pack<DP>;
    const int k = blockIdx.x;
    const int t = threadIdx.x;
    const float* L = chols + (size_t)k * D * D;
    float* out = packed + (size_t)k * P::STRIDE;
    for (int i = t; i < DP; i += 64) {
        out[P::MU + i] = i < D ? means[(size_t)k * D + i] : 0.f;
        out[P::RD + i] = i < D ? 1.f / L[i * D + i] : 1.f;
    }
    for (int e = t; e < DP * DP; e += 64) {
        int i = e / DP, j = e % DP;
        if (j < i) {
            float v = (i < D) ? L[i * D + j] : 0.f;
            out[P::LROW + P::rowofs(i) + j] = v;
            out[P::LCOL + P::colofs(j) + (i - j - 1)] = v;      // entry (row i, col j) lives in column j
        }
    }
    if (t == 0) {
        float s = 0.f;
        for (int i = 0; i < D; ++i) s += logf(L[i * D + i]);
        float c;
        if (family == GMMVI_GAUSS)
            c = -s - 0.5f * D * 1.8378770664093453f;             // log(2 pi)
        else
            c = lgammaf(0.5f * (nu + D)) - lgammaf(0.5f * nu) - 0.5f * D * logf(nu * 3.14159265358979f) - s;
        out[P::CONST] = c;
        for (int i = P::CONST + 1; i < P::FWD; ++i) out[i] = 0.f;
        out[P::SWH + DP] = c;
    }
    gmmvi_write_sweep_stream(out, DP, D, L, D, means + (size_t)k * D, t, 64);
    // L^-1: lane t solves L x = e_t (column t) by forward substitution; the dense inverse is staged in LDS, from where the
    // matrix-core fragments of the block (common.h) and the optional explicit inverse (sample_db.py:121) are written
    __shared__ float Li[DP * DP];
    if (P::FRAGS || inv_chols != nullptr) {
        float x[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            float sacc = (i == t) ? 1.f : 0.f;
            if (i < D && t < D) {
                for (int j = 0; j < i; ++j) sacc -= L[i * D + j] * x[j];
                x[i] = sacc / L[i * D + i];
            } else {
                x[i] = 0.f;
            }
        }
        if (t < DP) {
#pragma unroll
            for (int i = 0; i < DP; ++i) Li[i * DP + t] = (i >= t) ? x[i] : 0.f;
        }
    }
    __syncthreads();
    if (P::FRAGS) gmmvi_write_inverse_fragments(out, DP, D, Li, DP, t, 64);
    if (inv_chols != nullptr) {
        float* inv = inv_chols + (size_t)k * D * D;
        for (int e = t; e < D * D; e += 64) inv[e] = Li[(e / D) * DP + (e % D)];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// batched Cholesky for model construction / add_component (full_cov_gmm.py:23,:64-68)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cholesky_kernel(int D, const float* __restrict__ covs, float* __restrict__ chols,
                                                      int32_t* __restrict__ ok) {
    extern __shared__ float sm[];
    const int k = blockIdx.x, t = threadIdx.x;
    const int ld = D + 1;
    for (int e = t; e < D * D; e += 64) sm[(e / D) * ld + (e % D)] = covs[(size_t)k * D * D + e];
    __syncthreads();
    bool good = true;
    for (int j = 0; j < D; ++j) {
        float s = 0.f;
        if (t >= j && t < D) {
            s = sm[t * ld + j];
            for (int c = 0; c < j; ++c) s -= sm[t * ld + c] * sm[j * ld + c];
        }
        float p = __shfl(s, j);
        if (!(p > 0.f)) { good = false; break; }
        float d = sqrtf(p);
        __syncthreads();
        if (t == j) sm[t * ld + j] = d;
        else if (t > j && t < D) sm[t * ld + j] = s / d;
        __syncthreads();
    }
    for (int e = t; e < D * D; e += 64) {
        int i = e / D, j = e % D;
        float v = (j <= i) ? sm[i * ld + j] : 0.f;
        chols[(size_t)k * D * D + e] = good ? v : __builtin_nanf("");
    }
    if (t == 0 && ok) ok[k] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// mixture_eval
// ---------------------------------------------------------------------------------------------------------------
// gridDim.y > 1: the components are split over blockIdx.y; lp_out / grad_out then receive per-chunk partials
// ([chunk][N], [chunk][N][D]) that combine_partials merges.
typedef float me_f32x4 __attribute__((ext_vector_type(4)));
// a workgroup barrier for data that travels through LDS only
#define ME_LDS_BARRIER()                                       \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)
// rows of the last 16-row tile of the padded dimension when there are at most four of them (handled on the vector unit by the
// matrix-core kernels), else 0
__host__ __device__ constexpr int me_rem_rows(int dp) { return (dp % 16 != 0 && dp % 16 <= 4 && dp > 16) ? dp % 16 : 0; }
// the value of lane 16 g + R of v in every lane of lane group g (DPP row broadcast)
template <int R>
__device__ __forceinline__ float me_row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + R, 0xf, 0xf, false));
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void me_static_rows(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        me_static_rows<N, I + 1>(f);
    }
}
#ifndef GMMVI_ME_THREADS
#define GMMVI_ME_THREADS 1024
#define GMMVI_ME_MINW 1
#endif
// DP >= 40: at most 8 waves per workgroup, so that the compiler may use 256 VGPRs (x, z, y and the gradient accumulators are
// 4 DP registers; under the 128-register cap of a 16-wave workgroup the DP = 50 gradient kernel spilled 196 of them to scratch)
#ifndef GMMVI_ME_WIDE_DP
#define GMMVI_ME_WIDE_DP 40
#endif
#ifdef GMMVI_ME_STAMPS
__device__ long long g_me_wg[2 * 8192];    // experiment builds: wall-clock (100 MHz) start / end of every workgroup of the last launch
__device__ unsigned long long g_me_hw[8192];   // and where it ran (HW_ID, XCC_ID)
extern "C" int gmmvi_debug_wg_times(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_wg), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
extern "C" int gmmvi_debug_wg_hw(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_hw), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
__device__ long long g_me_ph[1024 * 16 * 16];  // packed kernel: 16 wall-clock phase stamps of every wave of the first 1024 workgroups
extern "C" int gmmvi_debug_wg_phases(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_ph), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
template <int DP, int FAMILY, bool GRAD>
__global__ __launch_bounds__(DP >= GMMVI_ME_WIDE_DP ? 512 : GMMVI_ME_THREADS, GMMVI_ME_MINW) void mixture_eval_kernel(float nu, int K_total, int D, const float* __restrict__ packed,
                                                            const float* __restrict__ logw, const float* __restrict__ X,
                                                            int N, float* __restrict__ ld_out, float* __restrict__ lp_out,
                                                            float* __restrict__ grad_out, const float* __restrict__ logw2,
                                                            float* __restrict__ lp2_out, CombineJob carried, Riders riders) {
    using PK = Pack<DP>;
    extern __shared__ __align__(16) float sm[];
    if (combine_carried(carried) || riders_carried<DP>(riders, sm)) return;   // workgroups past the sample tiles: the merge of the previous sweep, riders
    const int kchunk = (K_total + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kchunk;
    const int K = min(K_total, k_lo + kchunk);
    if (gridDim.y > 1) {
        if (lp_out) lp_out += (size_t)blockIdx.y * N;
        if (lp2_out) lp2_out += (size_t)blockIdx.y * N;
        if (GRAD && grad_out) grad_out += (size_t)blockIdx.y * N * D;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int n0 = blockIdx.x * 64;
    const int n_here = min(64, N - n0);
    const int n = n0 + lane;
    const bool valid = lane < n_here;
    float* sm_merge = sm;                              // staging, then the merge area
#ifdef GMMVI_ME_STAMPS             // experiment builds (tools/bench_sweep.py): phase time stamps of one wave
    if (threadIdx.x == 0) {
        g_me_wg[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_me_hw[blockIdx.y * gridDim.x + blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
    }
    unsigned long long stamp[16];
    int nstamp = 0, nbackward = 0;
    const long long wc0 = wall_clock64();
#define ME_STAMP() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (nstamp < 16) stamp[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
    ME_STAMP();
#else
#define ME_STAMP()
#endif

    // ---- x tile: rows that are a whole number of 16- or 8-byte pieces are read straight into the lane's registers (lane = sample:
    // the loads of a row walk the same cache lines, the W waves of the workgroup hit in L1); other shapes are staged
    // through LDS with a coalesced load (rows of a row-major [N, D] array are 4D bytes apart) ---------------------------
    const int ldx = D | 1;
    float x[DP];
    if (DP % 4 == 0 && D == DP && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const float4* xrow = reinterpret_cast<const float4*>(X + (size_t)min(n, N - 1) * D);
#pragma unroll
        for (int q4 = 0; q4 < DP / 4; ++q4) {
            const float4 v4 = xrow[q4];
            x[4 * q4] = v4.x; x[4 * q4 + 1] = v4.y; x[4 * q4 + 2] = v4.z; x[4 * q4 + 3] = v4.w;
        }
        if (!valid) {
#pragma unroll
            for (int i = 0; i < DP; ++i) x[i] = 0.f;
        }
    } else if (DP % 2 == 0 && D == DP && (reinterpret_cast<uintptr_t>(X) & 7) == 0) {       // 8-byte pieces (D = 10)
        const float2* xrow = reinterpret_cast<const float2*>(X + (size_t)min(n, N - 1) * D);
#pragma unroll
        for (int q2 = 0; q2 < DP / 2; ++q2) {
            const float2 v2 = xrow[q2];
            x[2 * q2] = v2.x; x[2 * q2 + 1] = v2.y;
        }
        if (!valid) {
#pragma unroll
            for (int i = 0; i < DP; ++i) x[i] = 0.f;
        }
    } else {
        for (int e = threadIdx.x; e < n_here * D; e += blockDim.x) sm_merge[(e / D) * ldx + (e % D)] = X[(size_t)n0 * D + e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DP; ++i) x[i] = (valid && i < D) ? sm_merge[lane * ldx + i] : 0.f;
        __syncthreads();
    }

    float m = -3.0e38f, s = 0.f;
    float m2 = -3.0e38f, s2 = 0.f;                      // second mixture over the same components (logw2), optional
    const bool dual = logw2 != nullptr;
    float acc[GRAD ? DP : 1];
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = 0.f;
    }
    const float nud = nu + (float)D;
    ME_STAMP();

    for (int k = k_lo + wave; k < K; k += nwaves) {
        ME_STAMP();
        // the block through a constant-address-space pointer, its loads software-pipelined (subst_phased.h)
        const sp_block_ptr blk = sp_block(packed + (size_t)k * PK::STRIDE);
        const float lw = ((sp_const_f32)(uintptr_t)logw)[k];
        const float lw2 = dual ? ((sp_const_f32)(uintptr_t)logw2)[k] : 0.f;
        float z[DP], vrd[DP], q, cst;
        float pc[2][32];
        SpPass<DP>::forward(blk, x, z, vrd, q, cst, pc);
        // the rows again for the backward pass, through a pointer the compiler cannot identify with the first one: otherwise it
        // keeps the values both passes share alive in VGPR lanes instead of loading them again
        sp_block_ptr blkb = blk;
        asm volatile("" : "+s"(blkb));
        if constexpr (GRAD) SpPass<DP>::backward_prefetch(blkb, pc);
        float ld, coef;
        if (FAMILY == GMMVI_GAUSS) {
            ld = fmaf(-0.5f, q, cst);
            coef = -1.f;
        } else {
            ld = cst - 0.5f * nud * log1pf(q / nu);
            coef = -nud / (nu + q);
        }
        const float a = ld + lw;
        const float mn = fmaxf(m, a);
        const float sc = __expf(m - mn);
        const float e = __expf(a - mn);
        s = fmaf(s, sc, e);
        m = mn;
        if (dual) {
            const float a2 = ld + lw2;
            const float mn2 = fmaxf(m2, a2);
            s2 = fmaf(s2, __expf(m2 - mn2), __expf(a2 - mn2));
            m2 = mn2;
        }
        if constexpr (GRAD) {
            SpPass<DP>::backward(blkb, z, vrd, pc);
            const float ec = e * coef;
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = fmaf(acc[i], sc, ec * z[i]);
            sp_pin<DP>(acc);
        }
        // (the store comes last: a branch in the middle of the pass would split it into basic blocks, and the optimiser then
        // sinks the multiply-adds of the backward substitution behind all the loads of its pieces)
        if (ld_out != nullptr && valid) ld_out[(size_t)k * N + n] = ld;
    }
    ME_STAMP();
    if (lp_out == nullptr && !GRAD) return;

    // merge the W waves' partials.  Round 1: the running maxima meet in LDS, every wave rescales ITS sums to the common
    // maximum (one exp per lane instead of one per (wave, dimension) pair in the reduction); round 2: plain sums in wave
    // order.  sm_m[w][lane], sm_s[w][lane], sm_acc[w][quad][lane] (four dimensions per 16-byte LDS access).
    // The barriers wait for LDS traffic only (ME_LDS_BARRIER): __syncthreads() also drains the global stores of the log
    // densities, ~4 000 cycles at the first barrier (profiles/r03_notes.md).
    constexpr int NQ = (DP + 3) / 4;
    float* sm_m = sm_merge;
    float* sm_s = sm_merge + nwaves * 64;
    me_f32x4* sm_acc = reinterpret_cast<me_f32x4*>(sm_merge + 2 * nwaves * 64);
    float* outt = sm_merge + 2 * nwaves * 64 + (GRAD ? (size_t)nwaves * NQ * 256 : 0);       // [64][ldx] tile of the result
    float* sm_m2 = outt + (GRAD ? 64 * ldx : 0);
    float* sm_s2 = sm_m2 + nwaves * 64;
    sm_m[wave * 64 + lane] = m;
    if (dual) sm_m2[wave * 64 + lane] = m2;
    ME_LDS_BARRIER();
    ME_STAMP();
    float M = -3.0e38f, M2 = -3.0e38f;
    for (int w = 0; w < nwaves; ++w) M = fmaxf(M, sm_m[w * 64 + lane]);
    const float f = __expf(m - M);
    sm_s[wave * 64 + lane] = s * f;
    if (dual) {
        for (int w = 0; w < nwaves; ++w) M2 = fmaxf(M2, sm_m2[w * 64 + lane]);
        sm_s2[wave * 64 + lane] = s2 * __expf(m2 - M2);
    }
    if (GRAD) {
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            me_f32x4 v4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v4[c] = 4 * q4 + c < DP ? acc[4 * q4 + c] * f : 0.f;
            sm_acc[(wave * NQ + q4) * 64 + lane] = v4;
        }
    }
    ME_STAMP();
    ME_LDS_BARRIER();
    ME_STAMP();
    float S = 0.f;
    for (int w = 0; w < nwaves; ++w) S += sm_s[w * 64 + lane];
    if (wave == 0 && valid && lp_out != nullptr) lp_out[n] = M + __logf(S);
    if (dual && wave == (nwaves > 1 ? 1 : 0) && valid && lp2_out != nullptr) {
        float S2 = 0.f;
        for (int w = 0; w < nwaves; ++w) S2 += sm_s2[w * 64 + lane];
        lp2_out[n] = M2 + __logf(S2);
    }
    if (GRAD && grad_out != nullptr) {
        const float inv = 1.f / S;
        // wave w reduces the dimension quads w, w + W, ...; results go to the [64][ldx] tile and leave coalesced
        for (int q4 = wave; q4 < NQ; q4 += nwaves) {
            me_f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < nwaves; ++w) g4 += sm_acc[(w * NQ + q4) * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (4 * q4 + c < D) outt[lane * ldx + 4 * q4 + c] = g4[c] * inv;
        }
        ME_STAMP();
        ME_LDS_BARRIER();
        ME_STAMP();
        for (int e = threadIdx.x; e < n_here * D; e += blockDim.x)
            grad_out[(size_t)n0 * D + e] = outt[(e / D) * ldx + (e % D)];
    }
#ifdef GMMVI_ME_STAMPS
    ME_STAMP();
    if (threadIdx.x == 0) g_me_wg[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = wall_clock64();
    if (GRAD && logw2 && blockIdx.x == 60 && blockIdx.y == 1 && threadIdx.x == 64 * 3 && (g_me_wg[2 * 8192 - 1]++ & 15) == 12) {
        const long long wc1 = wall_clock64();
        printf("me stamps (shader cycles; kernel %lld x 10 ns by the 100 MHz clock):", wc1 - wc0);
        for (int i = 1; i < nstamp; ++i) printf(" %llu", stamp[i] - stamp[i - 1]);
        printf("  total %llu, backward passes %d\n", stamp[nstamp - 1] - stamp[0], nbackward);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// mixture_eval, two samples per lane (packed f32 arithmetic)
// ---------------------------------------------------------------------------------------------------------------
// The launch of the kernel above is bound by the issue rate of its ~640 vector instructions per (component, 64 samples) pass:
// a wave64 v_fma_f32 issues once per 4 cycles per SIMD on gfx950, v_pk_fma_f32 at the same rate with two multiply-adds per
// lane (tools/probe/pk_issue.hip).  Here lane l owns samples l and l + 64 of a 128-sample tile and all per-dimension state is
// kept in aligned register pairs (subst_pk.h): ~520 packed instructions per (component, 128 samples).
//  * x does not live in registers: the tile sits in LDS in pair order ([dimension pair][lane] -> x_A[2i], x_B[2i], x_A[2i+1],
//    x_B[2i+1]: one conflict-free ds_read_b128 per dimension pair and pass), which keeps the gradient instance under 128
//    registers = four waves per SIMD.
//  * log-sum-exp against the running maximum of the component log densities alone (the weights enter as factors exp(log w),
//    >= 1e-30 by the floor of the weight update): ONE exp per sample and pass serves the model mixture, the second mixture of
//    the dual sweep and the rescaling of the gradient sums (the one-sample kernel pays four).
//  * the waves of a workgroup are merged by a tree through four LDS slots (eight for more than 8 waves; a wave writes the
//    slot it has just read: no barrier between a stage's reads and the next stage's writes), 47 KB instead of the 94 KB a
//    flat merge of 128 samples would take, so that two 8-wave workgroups share a CU.
// Chunk partials (gridDim.y > 1) leave as in the kernel above: log values and gradients normalised per chunk.
template <int DP, int FAMILY, bool GRAD>
// (padded D <= 24: up to 16 waves at <= 128 registers; wider dimensions: 8 waves, 256 registers -- the gradient instance keeps
// 4 DP registers of per-dimension state)
__global__ __launch_bounds__(DP > 24 ? 512 : 1024) void mixture_eval_pk_kernel(float nu, int K_total, int D, const float* __restrict__ packed,
                                                               const float* __restrict__ logw, const float* __restrict__ X, int N,
                                                               float* __restrict__ ld_out, float* __restrict__ lp_out,
                                                               float* __restrict__ grad_out, const float* __restrict__ logw2,
                                                               float* __restrict__ lp2_out, CombineJob carried, Riders riders) {
    using PK = Pack<DP>;
    static_assert(DP % 2 == 0, "padded dimensions are even");
    constexpr int NP2 = DP / 2;
    extern __shared__ __align__(16) float sm[];
    if (combine_carried(carried) || riders_carried<DP>(riders, sm)) return;   // workgroups past the sample tiles: the merge of the previous sweep, riders
    const int kchunk = (K_total + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kchunk;
    const int K = min(K_total, k_lo + kchunk);
    if (gridDim.y > 1) {
        if (lp_out) lp_out += (size_t)blockIdx.y * N;
        if (lp2_out) lp2_out += (size_t)blockIdx.y * N;
        if (GRAD && grad_out) grad_out += (size_t)blockIdx.y * N * D;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int n0 = blockIdx.x * 128;
    const int nA = n0 + lane, nB = n0 + 64 + lane;
    const bool validA = nA < N, validB = nB < N;
#ifdef GMMVI_ME_STAMPS
    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && wg_lin < 8192) {
        g_me_wg[2 * wg_lin] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_me_hw[wg_lin] = ((unsigned long long)xcc << 32) | hw;
    }
    int pk_ns = 0;
#define PK_STAMP() do { if (lane == 0 && wg_lin < 1024 && pk_ns < 16) g_me_ph[(wg_lin * 16 + wave) * 16 + pk_ns] = wall_clock64(); ++pk_ns; } while (0)
    PK_STAMP();                                        // 0: start
#else
#define PK_STAMP()
#endif

    // ---- x tile -> LDS in pair order (coalesced read of the tile's rows; padded dimensions and samples past N are zero) ----
    me_f32x4* xs4 = reinterpret_cast<me_f32x4*>(sm);                     // [NP2][64]
    {
        const int n_here = min(128, N - n0);
        for (int e = threadIdx.x; e < 128 * DP; e += blockDim.x) {
            const int sl = e / DP, i = e - sl * DP;
            const float v = (sl < n_here && i < D) ? X[(size_t)(n0 + sl) * D + i] : 0.f;
            sm[(((i >> 1) * 64) + (sl & 63)) * 4 + ((i & 1) << 1) + (sl >> 6)] = v;
        }
    }
    float* sm_merge = sm + NP2 * 64 * 4;
    ME_LDS_BARRIER();
    PK_STAMP();                                        // 1: x tile staged

    pk_f32x2 mld = pk_splat(-3.0e38f), s = pk_splat(0.f), s2 = pk_splat(0.f);
    const bool dual = logw2 != nullptr;
    pk_f32x2 acc[GRAD ? DP : 1];
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = pk_splat(0.f);
    }
    const float nud = nu + (float)D;
    // the wave that takes the odd components of a chunk rotates with the workgroup, so that the workgroups sharing a CU do
    // not put their longest waves on the same SIMD
    const int rot = (blockIdx.x + blockIdx.y) % nwaves;
    const int wslot = wave >= rot ? wave - rot : wave - rot + nwaves;

    for (int k = k_lo + wslot; k < K; k += nwaves) {
        const sp_block_ptr blk = sp_block(packed + (size_t)k * PK::STRIDE);
        const float w = __expf(((sp_const_f32)(uintptr_t)logw)[k]);
        const float w2 = dual ? __expf(((sp_const_f32)(uintptr_t)logw2)[k]) : 0.f;
        pk_f32x2 z[DP], q;
        float cst;
        float pc[2][32];
#pragma unroll
        for (int i2 = 0; i2 < NP2; ++i2) {
            const me_f32x4 v4 = xs4[i2 * 64 + lane];
            z[2 * i2] = pk_f32x2{v4.x, v4.y};
            z[2 * i2 + 1] = pk_f32x2{v4.z, v4.w};
        }
        PkPass<DP>::forward(blk, z, q, cst, pc);
        sp_block_ptr blkb = blk;                       // (an opaque copy: see the one-sample kernel)
        asm volatile("" : "+s"(blkb));
        if constexpr (GRAD) PkPass<DP>::backward_prefetch(blkb, pc);
        pk_f32x2 ld, coef;
        if (FAMILY == GMMVI_GAUSS) {
            ld = pk_fma(pk_splat(-0.5f), q, pk_splat(cst));
            coef = pk_splat(-1.f);
        } else {
            ld = pk_f32x2{cst - 0.5f * nud * log1pf(q.x / nu), cst - 0.5f * nud * log1pf(q.y / nu)};
            coef = pk_f32x2{-nud / (nu + q.x), -nud / (nu + q.y)};
        }
        // running maximum of the log densities; exactly one of (rescale factor, new term) is exp(-|d|), the other is 1
        const pk_f32x2 d = ld - mld;
        const float tA = __expf(-fabsf(d.x)), tB = __expf(-fabsf(d.y));
        const pk_f32x2 sc = pk_f32x2{d.x > 0.f ? tA : 1.f, d.y > 0.f ? tB : 1.f};
        const pk_f32x2 e = pk_f32x2{d.x > 0.f ? 1.f : tA, d.y > 0.f ? 1.f : tB};
        mld = pk_f32x2{fmaxf(mld.x, ld.x), fmaxf(mld.y, ld.y)};
        const pk_f32x2 we = e * pk_splat(w);
        s = pk_fma(s, sc, we);
        if (dual) s2 = pk_fma(s2, sc, e * pk_splat(w2));
        if constexpr (GRAD) {
            PkPass<DP>::backward(blkb, z, pc);
            const pk_f32x2 ec = we * coef;
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = pk_fma(acc[i], sc, ec * z[i]);
            pk_pin<DP>(acc);
        }
        if (ld_out != nullptr) {
            if (validA) ld_out[(size_t)k * N + nA] = ld.x;
            if (validB) ld_out[(size_t)k * N + nB] = ld.y;
        }
        PK_STAMP();                                    // 2 ..: after each component pass
    }
#ifdef GMMVI_ME_STAMPS
    pk_ns = 8;
#endif
    PK_STAMP();                                        // 8: loop left
    if (lp_out == nullptr && !GRAD) return;

    // ---- merge of the waves: common maximum, then a tree of sums through four slots ---------------------------------------
    constexpr int NV = (GRAD ? DP : 0) + 2;            // register pairs a wave hands over: s, s2, gradient sums
    pk_f32x2* sm_max = reinterpret_cast<pk_f32x2*>(sm_merge);            // [nwaves][64]
    pk_f32x2* slots = sm_max + nwaves * 64;                              // [4 or 8][NV][64]
    const int half0 = nwaves > 8 ? 8 : 4;                                // the first stage of the tree
    sm_max[wave * 64 + lane] = mld;
    ME_LDS_BARRIER();
    PK_STAMP();                                        // 9: maxima exchanged (includes the wait for the slowest wave)
    pk_f32x2 M = mld;
    for (int wv = 0; wv < nwaves; ++wv) {
        const pk_f32x2 o = sm_max[wv * 64 + lane];
        M = pk_f32x2{fmaxf(M.x, o.x), fmaxf(M.y, o.y)};
    }
    {
        const pk_f32x2 f = pk_f32x2{__expf(mld.x - M.x), __expf(mld.y - M.y)};
        s = s * f;
        s2 = s2 * f;
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = acc[i] * f;
        }
    }
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
        if (half < nwaves) {                           // (uniform: waves >= 2 * half hold nothing any more)
            const bool writer = wave >= half && wave < 2 * half;
            const bool reader = wave < half && wave + half < nwaves;
            // slot index = the slot the writer itself read one stage earlier (first stage: its partner's index)
            const int slot_w = half == half0 ? wave - half : wave;
            const int slot_r = half == half0 ? wave : wave + half;
            if (writer) {
                pk_f32x2* dst = slots + (size_t)slot_w * NV * 64 + lane;
                dst[0] = s;
                dst[64] = s2;
                if (GRAD) {
#pragma unroll
                    for (int i = 0; i < DP; ++i) dst[(2 + i) * 64] = acc[i];
                }
            }
            ME_LDS_BARRIER();
            if (reader) {
                const pk_f32x2* src = slots + (size_t)slot_r * NV * 64 + lane;
                s = s + src[0];
                s2 = s2 + src[64];
                if (GRAD) {
#pragma unroll
                    for (int i = 0; i < DP; ++i) acc[i] = acc[i] + src[(2 + i) * 64];
                }
            }
        }
    }
    PK_STAMP();                                        // 10: tree merge done
    if (wave != 0) return;
    if (lp_out != nullptr) {
        if (validA) lp_out[nA] = M.x + __logf(s.x);
        if (validB) lp_out[nB] = M.y + __logf(s.y);
    }
    if (dual && lp2_out != nullptr) {
        if (validA) lp2_out[nA] = M.x + __logf(s2.x);
        if (validB) lp2_out[nB] = M.y + __logf(s2.y);
    }
    if (GRAD && grad_out != nullptr) {
        const pk_f32x2 inv = pk_f32x2{1.f / s.x, 1.f / s.y};
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = acc[i] * inv;
        if (D == DP && DP % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15) == 0) {
            // a lane's row is a whole number of 16-byte pieces: neighbouring lanes' rows are neighbours in memory
#pragma unroll
            for (int q4 = 0; q4 < DP / 4; ++q4) {
                if (validA) reinterpret_cast<float4*>(grad_out + (size_t)nA * D)[q4] =
                    make_float4(acc[4 * q4].x, acc[4 * q4 + 1].x, acc[4 * q4 + 2].x, acc[4 * q4 + 3].x);
                if (validB) reinterpret_cast<float4*>(grad_out + (size_t)nB * D)[q4] =
                    make_float4(acc[4 * q4].y, acc[4 * q4 + 1].y, acc[4 * q4 + 2].y, acc[4 * q4 + 3].y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                if (i < D) {
                    if (validA) grad_out[(size_t)nA * D + i] = acc[i].x;
                    if (validB) grad_out[(size_t)nB * D + i] = acc[i].y;
                }
            }
        }
    }
    PK_STAMP();                                        // 11: results stored (wave 0)
#ifdef GMMVI_ME_STAMPS
    if (threadIdx.x == 0 && wg_lin < 8192) g_me_wg[2 * wg_lin + 1] = wall_clock64();
#endif
#undef PK_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// mixture_eval on the matrix cores
// ---------------------------------------------------------------------------------------------------------------
// z = L^-1 (x - mu) and y = L^-T z as dense contractions with the explicit inverse (operand fragments in the packed block,
// common.h): v_mfma_f32_16x16x4_f32 with A = a 16-row tile of L^-1 (coalesced 256-byte fragment loads, nothing on the scalar
// path), B = 16 samples x 4 dimensions of (x - mu).  A wave owns NTS sub-tiles of 16 samples (x fragments in registers for
// the whole launch) and walks a strided subset of the components; lane (g = l >> 4, n = l & 15) holds B[k = g][sample n] and
// receives rows 4 g + r of every 16-row tile of the result for sample n.  |z|^2 is completed across the four lane groups by
// two cross-lane adds; for the gradient z goes through a wave-private LDS image ([sample][dimension], permuted so that a
// lane's k-steps are contiguous) to become the B operand of the transposed contraction.  Structural zeros of the triangle
// are skipped per 16 x 4 fragment.  The log-sum-exp over the components, the K split over blockIdx.y and the merge of the
// workgroup's waves are those of the scalar-fed kernel.

#define ME_WAVE_LDS_SYNC()                                     \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

template <int DP, int FAMILY, bool GRAD, int NTS>
__global__ __launch_bounds__(512) void mixture_eval_mfma_kernel(float nu, int K_total, int D, const float* __restrict__ packed,
                                                                const float* __restrict__ logw, const float* __restrict__ X,
                                                                int N, float* __restrict__ ld_out, float* __restrict__ lp_out,
                                                                float* __restrict__ grad_out, const float* __restrict__ logw2,
                                                                float* __restrict__ lp2_out, CombineJob carried, Riders riders) {
    using PK = Pack<DP>;
    constexpr int MT = PK::MT, KS = PK::KS;
    extern __shared__ __align__(16) float sm[];
    if (combine_carried(carried) || riders_carried<DP>(riders, sm)) return;   // workgroups past the sample tiles: the merge of the previous sweep, riders
    constexpr int TS = 16 * NTS;                       // samples per workgroup tile
    constexpr int KSP = ((KS + 3) / 4) * 4;            // k-steps per lane in the z image, padded to 16-byte reads
    constexpr int ZW = 4 * KSP + 4;                    // row stride of the z image
    const int kchunk = (K_total + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kchunk;
    const int K = min(K_total, k_lo + kchunk);
    if (gridDim.y > 1) {
        if (lp_out) lp_out += (size_t)blockIdx.y * N;
        if (lp2_out) lp2_out += (size_t)blockIdx.y * N;
        if (GRAD && grad_out) grad_out += (size_t)blockIdx.y * N * D;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int g = lane >> 4, n16 = lane & 15;
    const int n0 = blockIdx.x * TS;
    const int n_here = min(TS, N - n0);
    // LDS: [nwaves][NTS][16][ZW] wave-private z images (gradient only), then the staging / merge area
    float* Zw = sm + (size_t)wave * (GRAD ? NTS * 16 * ZW : 0);
    float* sm_merge = sm + (size_t)nwaves * (GRAD ? NTS * 16 * ZW : 0);

    // ---- x tile: coalesced load staged through LDS; lane (g, n) keeps x[n][4 s + g] of its NTS sub-tiles -----------------
    const int ldx = D | 1;
    for (int e = threadIdx.x; e < n_here * D; e += blockDim.x) sm_merge[(e / D) * ldx + (e % D)] = X[(size_t)n0 * D + e];
    __syncthreads();
    float xb[NTS][KS];
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            xb[t][s] = (16 * t + n16 < n_here && 4 * s + g < D) ? sm_merge[(16 * t + n16) * ldx + 4 * s + g] : 0.f;
    __syncthreads();

    float m[NTS], sv[NTS], m2[NTS], s2[NTS];
#pragma unroll
    for (int t = 0; t < NTS; ++t) { m[t] = -3.0e38f; sv[t] = 0.f; m2[t] = -3.0e38f; s2[t] = 0.f; }
    const bool dual = logw2 != nullptr;
    me_f32x4 acc[GRAD ? NTS : 1][GRAD ? MT : 1];
    if (GRAD) {
#pragma unroll
        for (int t = 0; t < NTS; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][mt][r] = 0.f;
    }
    const float nud = nu + (float)D;

    for (int k = k_lo + wave; k < K; k += nwaves) {
        const float* __restrict__ Pk = packed + (size_t)k * PK::STRIDE;
        float af[PK::NF], ab[GRAD ? PK::NB : 1], mus[KS];
#pragma unroll
        for (int f = 0; f < PK::NF; ++f) af[f] = Pk[PK::FWD + 64 * f + lane];
        if (GRAD) {
#pragma unroll
            for (int f = 0; f < PK::NB; ++f) ab[f] = Pk[PK::BWD + 64 * f + lane];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) mus[s] = (4 * s + g < DP) ? Pk[PK::MU + 4 * s + g] : 0.f;
        const float cst = Pk[PK::CONST];
        const float lw = logw[k];
        const float lw2 = dual ? logw2[k] : 0.f;
        me_f32x4 z[NTS][MT];
        float ldv[NTS], ev[NTS], scv[NTS], cfv[NTS];
#pragma unroll
        for (int t = 0; t < NTS; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[t][mt][r] = 0.f;
        // k-step outermost: consecutive MFMAs go to different accumulators (NTS x MT independent chains), none waits for the
        // 40-cycle dependent-issue latency of its predecessor.
        // REM rows in the last 16-row tile (D = 50: rows 48, 49 -- two useful rows for 13 of the 37 forward fragments): those
        // rows go to the vector unit instead: lane (g, n) multiplies its own k-slots of row R (the fragment value of lane
        // 16 g + r, fetched by a DPP row broadcast) and the four lane groups are summed; the result lands where the matrix
        // cores would have put it (register r of the lanes of group 0).
        constexpr int REM = me_rem_rows(DP);
        float zr[REM > 0 ? NTS : 1][REM > 0 ? REM : 1];
        if constexpr (REM > 0) {
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int r = 0; r < REM; ++r) zr[t][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float bs[NTS];
#pragma unroll
            for (int t = 0; t < NTS; ++t) {
                bs[t] = xb[t][s] - mus[s];
#pragma unroll
                for (int mt = 0; mt < (REM > 0 ? MT - 1 : MT); ++mt)
                    if (s < PK::nf(mt))
                        z[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[PK::fwd_index(mt, s)], bs[t], z[t][mt], 0, 0, 0);
            }
            if constexpr (REM > 0) {
                me_static_rows<REM>([&](auto R) {
                    constexpr int r = R;
                    const float a = me_row_bcast<r>(af[PK::fwd_index(MT - 1, s)]);      // Linv[16 (MT-1) + r][4 s + g]
#pragma unroll
                    for (int t = 0; t < NTS; ++t) zr[t][r] = fmaf(a, bs[t], zr[t][r]);
                });
            }
        }
        if constexpr (REM > 0) {
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int r = 0; r < REM; ++r) {
                    float v = zr[t][r];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    z[t][MT - 1][r] = g == 0 ? v : 0.f;
                }
        }
        if constexpr (!GRAD && NTS == 4) {
            // lane = sample (16 g + n of sub-tile g): after the cross-lane completion every lane group holds |z|^2 of all four
            // sub-tiles; each lane keeps its own sub-tile's and runs ONE log-sum-exp step -- the four groups used to repeat the
            // same four tails (the sweep without the gradient is bound by them, not by the matrix cores)
            float qs = 0.f;
#pragma unroll
            for (int t = 0; t < NTS; ++t) {
                float q = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) q = fmaf(z[t][mt][r], z[t][mt][r], q);
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                qs = (g == t) ? q : qs;
            }
            float ld;
            if (FAMILY == GMMVI_GAUSS) ld = fmaf(-0.5f, qs, cst);
            else ld = cst - 0.5f * nud * log1pf(qs / nu);
            const float a = ld + lw;
            const float mn = fmaxf(m[0], a);
            sv[0] = fmaf(sv[0], __expf(m[0] - mn), __expf(a - mn));
            m[0] = mn;
            if (dual) {
                const float a2 = ld + lw2;
                const float mn2 = fmaxf(m2[0], a2);
                s2[0] = fmaf(s2[0], __expf(m2[0] - mn2), __expf(a2 - mn2));
                m2[0] = mn2;
            }
            if (ld_out != nullptr && lane < n_here) ld_out[(size_t)k * N + n0 + lane] = ld;
        } else {
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) q = fmaf(z[t][mt][r], z[t][mt][r], q);
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            float ld, coef;
            if (FAMILY == GMMVI_GAUSS) {
                ld = fmaf(-0.5f, q, cst);
                coef = -1.f;
            } else {
                ld = cst - 0.5f * nud * log1pf(q / nu);
                coef = -nud / (nu + q);
            }
            ldv[t] = ld;
            const float a = ld + lw;
            const float mn = fmaxf(m[t], a);
            const float sc = __expf(m[t] - mn);
            const float e = __expf(a - mn);
            sv[t] = fmaf(sv[t], sc, e);
            m[t] = mn;
            ev[t] = e; scv[t] = sc; cfv[t] = coef;
            if (dual) {
                const float a2 = ld + lw2;
                const float mn2 = fmaxf(m2[t], a2);
                s2[t] = fmaf(s2[t], __expf(m2[t] - mn2), __expf(a2 - mn2));
                m2[t] = mn2;
            }
        }
        if (ld_out != nullptr) {
            // lane group g stores sub-tile g: one instruction writes 16 NTS consecutive floats of row k
            float v = ldv[0];
#pragma unroll
            for (int t = 1; t < NTS; ++t) v = (g == t) ? ldv[t] : v;
            if (g < NTS && 16 * g + n16 < n_here) ld_out[(size_t)k * N + n0 + 16 * g + n16] = v;
        }
        }
        if (GRAD) {
            // z -> wave-private image Zs[t][n][p(i)], p(i) = (i & 3) KSP + (i >> 2): lane (g, n) holds i = 16 mt + 4 g + r
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * mt + g < KSP) Zw[(t * 16 + n16) * ZW + r * KSP + 4 * mt + g] = z[t][mt][r];
            ME_WAVE_LDS_SYNC();
#pragma unroll
            for (int t = 0; t < NTS; ++t) {
                float bz[KSP];
#pragma unroll
                for (int q4 = 0; q4 < KSP / 4; ++q4) {
                    const me_f32x4 v4 = *reinterpret_cast<const me_f32x4*>(Zw + (t * 16 + n16) * ZW + g * KSP + 4 * q4);
                    bz[4 * q4] = v4[0]; bz[4 * q4 + 1] = v4[1]; bz[4 * q4 + 2] = v4[2]; bz[4 * q4 + 3] = v4[3];
                }
                const float ec = ev[t] * cfv[t];
                me_f32x4 y[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[mt][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)                           // k-step outermost: MT independent chains
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (s >= 4 * mt)
                            y[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[PK::bwd_index(mt, s)], bz[s], y[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][mt][r] = fmaf(acc[t][mt][r], scv[t], ec * y[mt][r]);
            }
            ME_WAVE_LDS_SYNC();
        }
    }
    if (lp_out == nullptr && !GRAD) return;

    // merge the W waves' partials: sm_m[w][sample], sm_s[w][sample], sm_acc[w][i][sample]  (sample = 16 t + n).
    // Round 1: the running maxima meet in LDS and every wave rescales ITS sums to the common maximum (one exp per lane and
    // sub-tile instead of one per (dimension, sample, wave) in the reduction); round 2: plain sums in wave order.  The
    // barriers wait for LDS traffic only (ME_LDS_BARRIER): __syncthreads() would also drain the stores of the log densities.
    float* sm_m = sm_merge;
    float* sm_s = sm_merge + nwaves * TS;
    float* sm_acc = sm_merge + 2 * nwaves * TS;
    float* sm_m2 = sm_merge + (size_t)nwaves * TS * ((GRAD ? DP : 0) + 2) + (GRAD ? TS * ldx : 0);
    float* sm_s2 = sm_m2 + nwaves * TS;
    constexpr bool LANE_SAMPLE = !GRAD && NTS == 4;    // lane = sample (16 g + n of sub-tile g) instead of one column per lane group
    if constexpr (LANE_SAMPLE) {
        sm_m[wave * TS + lane] = m[0];
        if (dual) sm_m2[wave * TS + lane] = m2[0];
    } else if (g == 0) {
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            sm_m[wave * TS + 16 * t + n16] = m[t];
            if (dual) sm_m2[wave * TS + 16 * t + n16] = m2[t];
        }
    }
    ME_LDS_BARRIER();
    if constexpr (LANE_SAMPLE) {
        float M = -3.0e38f;
        for (int w = 0; w < nwaves; ++w) M = fmaxf(M, sm_m[w * TS + lane]);
        sm_s[wave * TS + lane] = sv[0] * __expf(m[0] - M);
        if (dual) {
            float M2 = -3.0e38f;
            for (int w = 0; w < nwaves; ++w) M2 = fmaxf(M2, sm_m2[w * TS + lane]);
            sm_s2[wave * TS + lane] = s2[0] * __expf(m2[0] - M2);
        }
    } else {
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            const int sidx = 16 * t + n16;
            float M = -3.0e38f;
            for (int w = 0; w < nwaves; ++w) M = fmaxf(M, sm_m[w * TS + sidx]);
            const float f = __expf(m[t] - M);
            if (g == 0) sm_s[wave * TS + sidx] = sv[t] * f;
            if (dual && g == 1) {
                float M2 = -3.0e38f;
                for (int w = 0; w < nwaves; ++w) M2 = fmaxf(M2, sm_m2[w * TS + sidx]);
                sm_s2[wave * TS + sidx] = s2[t] * __expf(m2[t] - M2);
            }
            if (GRAD) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * g + r;
                        if (i < DP) sm_acc[(wave * DP + i) * TS + sidx] = acc[t][mt][r] * f;
                    }
            }
        }
    }
    ME_LDS_BARRIER();
    // sample si = threadIdx.x (one thread per sample for the log values); 1 / S goes back to LDS for the gradient
    const int si = threadIdx.x;
    if (si < TS) {
        float M = -3.0e38f, S = 0.f;
        for (int w = 0; w < nwaves; ++w) { M = fmaxf(M, sm_m[w * TS + si]); S += sm_s[w * TS + si]; }
        if (si < n_here && lp_out != nullptr) lp_out[n0 + si] = M + __logf(S);
        if (dual && si < n_here && lp2_out != nullptr) {
            float M2 = -3.0e38f, S2 = 0.f;
            for (int w = 0; w < nwaves; ++w) { M2 = fmaxf(M2, sm_m2[w * TS + si]); S2 += sm_s2[w * TS + si]; }
            lp2_out[n0 + si] = M2 + __logf(S2);
        }
        if (GRAD) sm_s[si] = 1.f / S;                  // (row 0 of sm_s is read by this thread only: no hazard)
    }
    if (GRAD && grad_out != nullptr) {
        ME_LDS_BARRIER();
        // thread (dimension i, sample s_i) pairs spread over the workgroup; results go to a [TS][ldx] tile and leave coalesced
        float* outt = sm_acc + (size_t)nwaves * DP * TS;
        for (int e = threadIdx.x; e < D * TS; e += blockDim.x) {
            const int i = e / TS, s_i = e - i * TS;
            float gsum = 0.f;
            for (int w = 0; w < nwaves; ++w) gsum += sm_acc[(w * DP + i) * TS + s_i];
            outt[s_i * ldx + i] = gsum * sm_s[s_i];
        }
        ME_LDS_BARRIER();
        for (int e = threadIdx.x; e < n_here * D; e += blockDim.x)
            grad_out[(size_t)n0 * D + e] = outt[(e / D) * ldx + (e % D)];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mixture_eval on the matrix cores, component block shared by the workgroup
// ---------------------------------------------------------------------------------------------------------------
// In the kernel above every wave of a workgroup walks its own components and fetches their operand fragments (17 KB a
// component at D = 50) from the L2 for 16 NTS samples: 0.9 GB per launch at the C3 shape, the kernel's real bound.  Here the
// EIGHT waves of a workgroup work on the SAME component at a time, each on its own 16 NTS samples: the block (fragments, mu,
// log-normaliser) is brought to LDS once per workgroup and component (double-buffered: the next block's global loads are in
// flight while the current one is multiplied; one barrier per component), every MFMA takes its A operand with one
// conflict-free ds_read_b32.  A wave owns its samples for all components of the chunk, so nothing is merged across waves:
// log values and gradients leave straight from the registers.  The component chunks over blockIdx.y and the merge of their
// partials are those of the other kernels.
template <int DP, int FAMILY, bool GRAD, int NTS>
__global__ __launch_bounds__(512) void mixture_eval_mfma_ws_kernel(float nu, int K_total, int D, const float* __restrict__ packed,
                                                                   const float* __restrict__ logw, const float* __restrict__ X,
                                                                   int N, float* __restrict__ ld_out, float* __restrict__ lp_out,
                                                                   float* __restrict__ grad_out, const float* __restrict__ logw2,
                                                                   float* __restrict__ lp2_out, CombineJob carried, Riders riders) {
    using PK = Pack<DP>;
    constexpr int MT = PK::MT, KS = PK::KS;
    constexpr int TSW = 16 * NTS;                      // samples per wave
    constexpr int TS = 8 * TSW;                        // samples per workgroup
    constexpr int KSP = ((KS + 3) / 4) * 4;
    constexpr int ZW = 4 * KSP + 4;
    constexpr int NFR = 64 * (PK::NF + (GRAD ? PK::NB : 0));           // fragment floats of a block that are used
    constexpr int BLK = NFR + ((DP + 1 + 3) / 4) * 4;                   // + mu[DP], log-normaliser
    constexpr int NPRE = (BLK + 511) / 512;
    extern __shared__ __align__(16) float sm[];
    if (combine_carried(carried) || riders_carried<DP>(riders, sm)) return;
    const int kchunk = (K_total + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kchunk;
    const int K = min(K_total, k_lo + kchunk);
    if (gridDim.y > 1) {
        if (lp_out) lp_out += (size_t)blockIdx.y * N;
        if (lp2_out) lp2_out += (size_t)blockIdx.y * N;
        if (GRAD && grad_out) grad_out += (size_t)blockIdx.y * N * D;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n16 = lane & 15;
    const int n0 = blockIdx.x * TS + wave * TSW;       // this wave's first sample
    float* blk0 = sm;                                  // two component blocks
    float* Zw = sm + 2 * BLK + (size_t)wave * (GRAD ? NTS * 16 * ZW : 0);

    // one element of component k's LDS block: [fragments | mu | log-normaliser]
    auto block_src = [&](int k, int e) -> float {
        const float* Pk = packed + (size_t)k * PK::STRIDE;
        return e < NFR ? Pk[PK::FWD + e] : (e < NFR + DP ? Pk[PK::MU + (e - NFR)] : (e == NFR + DP ? Pk[PK::CONST] : 0.f));
    };
    // ---- the wave's x: lane (g, n) keeps x[n][4 s + g] of its NTS sub-tiles (rows beyond N: clamped, masked at the stores) ---
    float xb[NTS][KS];
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
        const float* xrow = X + (size_t)min(n0 + 16 * t + n16, N - 1) * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[t][s] = (4 * s + g < D) ? xrow[4 * s + g] : 0.f;
    }
    float m[NTS], sv[NTS], m2[NTS], s2[NTS];
#pragma unroll
    for (int t = 0; t < NTS; ++t) { m[t] = -3.0e38f; sv[t] = 0.f; m2[t] = -3.0e38f; s2[t] = 0.f; }
    const bool dual = logw2 != nullptr;
    me_f32x4 acc[GRAD ? NTS : 1][GRAD ? MT : 1];
    if (GRAD) {
#pragma unroll
        for (int t = 0; t < NTS; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][mt][r] = 0.f;
    }
    const float nud = nu + (float)D;

    float pre[NPRE];
    if (k_lo < K) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u)
            if (tid + 512 * u < BLK) blk0[tid + 512 * u] = block_src(k_lo, tid + 512 * u);
    }
    __syncthreads();
    for (int k = k_lo; k < K; ++k) {
        const float* B = blk0 + ((k - k_lo) & 1) * BLK;
        float* Bn = blk0 + (((k - k_lo) & 1) ^ 1) * BLK;
        const bool more = k + 1 < K;
        if (more) {
#pragma unroll
            for (int u = 0; u < NPRE; ++u) pre[u] = (tid + 512 * u < BLK) ? block_src(k + 1, min(tid + 512 * u, BLK - 1)) : 0.f;
        }
        const float* Fw = B + lane;                    // forward fragment f at Fw[64 f]
        const float* Bw = B + 64 * PK::NF + lane;      // backward fragment f at Bw[64 f]
        float mus[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) mus[s] = (4 * s + g < DP) ? B[NFR + 4 * s + g] : 0.f;
        const float cst = B[NFR + DP];
        const float lw = logw[k];
        const float lw2 = dual ? logw2[k] : 0.f;
        me_f32x4 z[NTS][MT];
        float ldv[NTS], ev[NTS], scv[NTS], cfv[NTS];
#pragma unroll
        for (int t = 0; t < NTS; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[t][mt][r] = 0.f;
        // (the REM rows of the last 16-row tile on the vector unit, as in the per-wave kernel: lane (g, n) reads the fragment value
        // of lane 16 g + r from the LDS block)
        constexpr int REM = me_rem_rows(DP);
        float zr[REM > 0 ? NTS : 1][REM > 0 ? REM : 1];
        if constexpr (REM > 0) {
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int r = 0; r < REM; ++r) zr[t][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float bx[NTS];
#pragma unroll
            for (int t = 0; t < NTS; ++t) bx[t] = xb[t][s] - mus[s];
#pragma unroll
            for (int mt = 0; mt < (REM > 0 ? MT - 1 : MT); ++mt) {
                if (s < PK::nf(mt)) {
                    const float a = Fw[64 * PK::fwd_index(mt, s)];                      // one fragment read serves NTS MFMAs
#pragma unroll
                    for (int t = 0; t < NTS; ++t) z[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx[t], z[t][mt], 0, 0, 0);
                }
            }
            if constexpr (REM > 0) {
#pragma unroll
                for (int r = 0; r < REM; ++r) {
                    const float a = B[64 * PK::fwd_index(MT - 1, s) + 16 * g + r];       // Linv[16 (MT-1) + r][4 s + g]
#pragma unroll
                    for (int t = 0; t < NTS; ++t) zr[t][r] = fmaf(a, bx[t], zr[t][r]);
                }
            }
        }
        if constexpr (REM > 0) {
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int r = 0; r < REM; ++r) {
                    float v = zr[t][r];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    z[t][MT - 1][r] = g == 0 ? v : 0.f;
                }
        }
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) q = fmaf(z[t][mt][r], z[t][mt][r], q);
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            float ld, coef;
            if (FAMILY == GMMVI_GAUSS) {
                ld = fmaf(-0.5f, q, cst);
                coef = -1.f;
            } else {
                ld = cst - 0.5f * nud * log1pf(q / nu);
                coef = -nud / (nu + q);
            }
            ldv[t] = ld;
            const float a = ld + lw;
            const float mn = fmaxf(m[t], a);
            const float sc = __expf(m[t] - mn);
            const float e = __expf(a - mn);
            sv[t] = fmaf(sv[t], sc, e);
            m[t] = mn;
            ev[t] = e; scv[t] = sc; cfv[t] = coef;
            if (dual) {
                const float a2 = ld + lw2;
                const float mn2 = fmaxf(m2[t], a2);
                s2[t] = fmaf(s2[t], __expf(m2[t] - mn2), __expf(a2 - mn2));
                m2[t] = mn2;
            }
        }
        if (ld_out != nullptr) {
            float v = ldv[0];
#pragma unroll
            for (int t = 1; t < NTS; ++t) v = (g == t) ? ldv[t] : v;
            if (g < NTS && n0 + 16 * g + n16 < N) ld_out[(size_t)k * N + n0 + 16 * g + n16] = v;
        }
        if (GRAD) {
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * mt + g < KSP) Zw[(t * 16 + n16) * ZW + r * KSP + 4 * mt + g] = z[t][mt][r];
            ME_WAVE_LDS_SYNC();
            float bz[NTS][KSP];
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int q4 = 0; q4 < KSP / 4; ++q4) {
                    const me_f32x4 v4 = *reinterpret_cast<const me_f32x4*>(Zw + (t * 16 + n16) * ZW + g * KSP + 4 * q4);
                    bz[t][4 * q4] = v4[0]; bz[t][4 * q4 + 1] = v4[1]; bz[t][4 * q4 + 2] = v4[2]; bz[t][4 * q4 + 3] = v4[3];
                }
            me_f32x4 y[NTS][MT];
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[t][mt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (s >= 4 * mt) {
                        const float a = Bw[64 * PK::bwd_index(mt, s)];
#pragma unroll
                        for (int t = 0; t < NTS; ++t) y[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bz[t][s], y[t][mt], 0, 0, 0);
                    }
#pragma unroll
            for (int t = 0; t < NTS; ++t) {
                const float ec = ev[t] * cfv[t];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][mt][r] = fmaf(acc[t][mt][r], scv[t], ec * y[t][mt][r]);
            }
            ME_WAVE_LDS_SYNC();
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < NPRE; ++u)
                if (tid + 512 * u < BLK) Bn[tid + 512 * u] = pre[u];
        }
        ME_LDS_BARRIER();              // (LDS only: __syncthreads() would drain the log-density stores of every component)
    }
    // ---- the wave's results leave from its registers: log values by the lanes of group 0, gradient rows 4 g + r of tile mt ---
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            const int n = n0 + 16 * t + n16;
            if (n < N) {
                if (lp_out != nullptr) lp_out[n] = m[t] + __logf(sv[t]);
                if (dual && lp2_out != nullptr) lp2_out[n] = m2[t] + __logf(s2[t]);
            }
        }
    }
    if (GRAD && grad_out != nullptr) {
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            // every lane of a column (same n16) holds the same m / sv: the sums run over the components, not over lanes
            const int n = n0 + 16 * t + n16;
            const float inv = 1.f / sv[t];
            if (n < N) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * g + r;
                        if (i < D) grad_out[(size_t)n * D + i] = acc[t][mt][r] * inv;
                    }
            }
        }
    }
}

// profile name of a density launch: the caller's tag (fused.hip knows which sweep of the iteration it issues), else by shape:
// the dual sweep (model + background over the same components), a sweep with the gradient (target evaluation or model
// density + gradient), a log-value sweep (post-update densities)
static const char* sweep_prof_name(const gmmvi_ctx* ctx, bool want_grad, bool dual) {
    if (ctx->prof_tag) return ctx->prof_tag;
    return dual ? "sweep_dual" : (want_grad ? "sweep_grad" : "sweep_values");
}

template <int DP, int NTS>
static int launch_mixture_eval_mfma(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed,
                                    const float* logw, const float* X, int N, float* ld, float* lp, float* grad,
                                    const float* logw2, float* lp2) {
    using PK = Pack<DP>;
    constexpr int TS = 16 * NTS, KSP = ((PK::KS + 3) / 4) * 4, ZW = 4 * KSP + 4;
    const bool want_grad = grad != nullptr;
    const bool want_merge = want_grad || lp != nullptr;
    static const int env_nw = getenv("GMMVI_ME_NW") ? atoi(getenv("GMMVI_ME_NW")) : 0;
    static const int env_ky = getenv("GMMVI_ME_KY") ? atoi(getenv("GMMVI_ME_KY")) : 0;
    auto lds_floats = [&](int nw) {
        size_t merge = (size_t)nw * TS * ((want_grad ? DP : 0) + 2) + (want_grad ? TS * (size_t)(D | 1) : 0) +
                       (logw2 ? (size_t)nw * 2 * TS : 0);
        size_t stage = TS * (size_t)(D | 1);
        return (want_grad ? (size_t)nw * NTS * 16 * ZW : 0) + (merge > stage ? merge : stage);
    };
    const int tiles = (N + TS - 1) / TS;
    // geometry as for the scalar-fed kernel: ky chunks of components over blockIdx.y (partials merged by combine_partials) so
    // that ~2 workgroups of 8 waves per CU are in flight when there are few sample tiles
    int ky = 1, nw = K < 8 ? K : 8;
    if (env_ky > 0) ky = env_ky < K ? env_ky : K;
    else if (K >= 16 && 2L * tiles <= 3L * ctx->num_cus) {
        ky = (int)((2L * ctx->num_cus + tiles / 2) / tiles);
        if (ky > K / 8) ky = K / 8;
        if (ky < 1) ky = 1;
    }
    const int kchunk = (K + ky - 1) / ky;
    ky = (K + kchunk - 1) / kchunk;
    if (nw > kchunk) nw = kchunk;
    if (env_nw > 0) nw = env_nw < kchunk ? env_nw : kchunk;
    if (nw > 8) nw = 8;
    while (nw > 1 && lds_floats(nw) * 4 > 64 * 1024) --nw;
    size_t shmem = lds_floats(nw) * 4;
    float* lp_k = lp;
    float* grad_k = grad;
    float* lp2_k = lp2;
    // the merge of the chunk partials: its own launch, or (single-call iteration) left to the next launch
    const bool defer = ctx->defer_combine && ky > 1 && want_merge;
    if (defer) {
        int rc = gmmvi_flush_pending_combine(ctx);
        if (rc != GMMVI_OK) return rc;
    }
    if (ky > 1 && want_merge) {
        size_t need = ((size_t)ky * N * (logw2 ? 2 : 1) + (want_grad ? (size_t)ky * N * D : 0)) * sizeof(float);
        int rc = defer ? gmmvi_defer_reserve(ctx, need) : gmmvi_ws_reserve(ctx, need);
        if (rc != GMMVI_OK) return rc;
        lp_k = (float*)(defer ? ctx->defer_ws : ctx->ws);
        lp2_k = logw2 ? lp_k + (size_t)ky * N : nullptr;
        grad_k = want_grad ? lp_k + (size_t)ky * N * (logw2 ? 2 : 1) : nullptr;
    }
    // block order: the launch's own tiles, the riders (the stepsize block first: it is the longest of them), the carried merge
    const Riders riders = gmmvi_take_pending_riders(ctx, tiles, nw * 64);
    const CombineJob carried = gmmvi_take_pending_combine(ctx, nw * 64, tiles + riders.prep_blocks + riders.sample_blocks);
    if (riders_lds_bytes(riders) > shmem) shmem = riders_lds_bytes(riders);
    dim3 grid(tiles + carried.blocks + riders.prep_blocks + riders.sample_blocks, ky), block(nw * 64);
    {
        GMMVI_PROF_UNITS(ctx, sweep_prof_name(ctx, want_grad, logw2 != nullptr), (double)N * K);
#define GMMVI_LAUNCH_MM(FAM, G)                                                                                     \
    do {                                                                                                            \
        if (shmem > 64 * 1024)                                                                                      \
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)mixture_eval_mfma_kernel<DP, FAM, G, NTS>,        \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));      \
        hipLaunchKernelGGL((mixture_eval_mfma_kernel<DP, FAM, G, NTS>), grid, block, shmem, ctx->stream, nu, K, D,  \
                           packed, logw, X, N, ld, lp_k, grad_k, logw2, lp2_k, carried, riders);                             \
    } while (0)
        if (family == GMMVI_GAUSS) {
            if (want_grad) GMMVI_LAUNCH_MM(GMMVI_GAUSS, true); else GMMVI_LAUNCH_MM(GMMVI_GAUSS, false);
        } else {
            if (want_grad) GMMVI_LAUNCH_MM(GMMVI_STUDENT_T, true); else GMMVI_LAUNCH_MM(GMMVI_STUDENT_T, false);
        }
#undef GMMVI_LAUNCH_MM
    }
    GMMVI_LAUNCH_CHECK(ctx);
    if (defer) {
        CombineJob& j = ctx->pending;
        j.R = ky; j.N = N; j.D = D;
        j.lp_parts = lp_k; j.grad_parts = grad_k; j.lp2_parts = lp2_k;
        j.lp_out = lp; j.grad_out = grad; j.lp2_out = lp2_k ? lp2 : nullptr;
    } else if (ky > 1 && want_merge) {
        GMMVI_PROF(ctx, "mixture_combine");
        int rc = gmmvi_combine_partials_internal(ctx, ky, N, D, lp_k, grad_k, lp, grad, lp2_k, lp2);
        if (rc != GMMVI_OK) return rc;
    }
    return GMMVI_OK;
}

template <int DP, int NTS>
static int launch_mixture_eval_mfma_ws(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed,
                                       const float* logw, const float* X, int N, float* ld, float* lp, float* grad,
                                       const float* logw2, float* lp2) {
    using PK = Pack<DP>;
    constexpr int TS = 128 * NTS, KSP = ((PK::KS + 3) / 4) * 4, ZW = 4 * KSP + 4;
    const bool want_grad = grad != nullptr;
    const bool want_merge = want_grad || lp != nullptr;
    const int nfr = 64 * (PK::NF + (want_grad ? PK::NB : 0));
    const int blk = nfr + ((DP + 1 + 3) / 4) * 4;
    size_t shmem = ((size_t)2 * blk + (want_grad ? (size_t)8 * NTS * 16 * ZW : 0)) * sizeof(float);
    const int tiles = (N + TS - 1) / TS;
    // component chunks so that every CU gets a workgroup (two where the LDS allows), at least eight components each (the
    // block pipeline has a prologue and every chunk writes partial outputs)
    const int per_cu = shmem * 2 <= 150 * 1024 ? 2 : 1;
    int ky = (per_cu * ctx->num_cus + tiles / 2) / tiles;
    if (ky > K / 8) ky = K / 8;
    if (ky < 1) ky = 1;
    const int kchunk = (K + ky - 1) / ky;
    ky = (K + kchunk - 1) / kchunk;
    float* lp_k = lp;
    float* grad_k = grad;
    float* lp2_k = lp2;
    const bool defer = ctx->defer_combine && ky > 1 && want_merge;
    if (defer) {
        int rc = gmmvi_flush_pending_combine(ctx);
        if (rc != GMMVI_OK) return rc;
    }
    if (ky > 1 && want_merge) {
        size_t need = ((size_t)ky * N * (logw2 ? 2 : 1) + (want_grad ? (size_t)ky * N * D : 0)) * sizeof(float);
        int rc = defer ? gmmvi_defer_reserve(ctx, need) : gmmvi_ws_reserve(ctx, need);
        if (rc != GMMVI_OK) return rc;
        lp_k = (float*)(defer ? ctx->defer_ws : ctx->ws);
        lp2_k = logw2 ? lp_k + (size_t)ky * N : nullptr;
        grad_k = want_grad ? lp_k + (size_t)ky * N * (logw2 ? 2 : 1) : nullptr;
    }
    // block order: the launch's own tiles, the riders (the stepsize block first: it is the longest of them), the carried merge
    const Riders riders = gmmvi_take_pending_riders(ctx, tiles, 512);
    const CombineJob carried = gmmvi_take_pending_combine(ctx, 512, tiles + riders.prep_blocks + riders.sample_blocks);
    if (riders_lds_bytes(riders) > shmem) shmem = riders_lds_bytes(riders);
    dim3 grid(tiles + carried.blocks + riders.prep_blocks + riders.sample_blocks, ky), block(512);
    {
        GMMVI_PROF_UNITS(ctx, sweep_prof_name(ctx, want_grad, logw2 != nullptr), (double)N * K);
#define GMMVI_LAUNCH_WS(FAM, G)                                                                                     \
    do {                                                                                                            \
        if (shmem > 64 * 1024)                                                                                      \
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)mixture_eval_mfma_ws_kernel<DP, FAM, G, NTS>,     \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));      \
        hipLaunchKernelGGL((mixture_eval_mfma_ws_kernel<DP, FAM, G, NTS>), grid, block, shmem, ctx->stream, nu, K,  \
                           D, packed, logw, X, N, ld, lp_k, grad_k, logw2, lp2_k, carried, riders);                          \
    } while (0)
        if (family == GMMVI_GAUSS) {
            if (want_grad) GMMVI_LAUNCH_WS(GMMVI_GAUSS, true); else GMMVI_LAUNCH_WS(GMMVI_GAUSS, false);
        } else {
            if (want_grad) GMMVI_LAUNCH_WS(GMMVI_STUDENT_T, true); else GMMVI_LAUNCH_WS(GMMVI_STUDENT_T, false);
        }
#undef GMMVI_LAUNCH_WS
    }
    GMMVI_LAUNCH_CHECK(ctx);
    if (defer) {
        CombineJob& j = ctx->pending;
        j.R = ky; j.N = N; j.D = D;
        j.lp_parts = lp_k; j.grad_parts = grad_k; j.lp2_parts = lp2_k;
        j.lp_out = lp; j.grad_out = grad; j.lp2_out = lp2_k ? lp2 : nullptr;
    } else if (ky > 1 && want_merge) {
        GMMVI_PROF(ctx, "mixture_combine");
        int rc = gmmvi_combine_partials_internal(ctx, ky, N, D, lp_k, grad_k, lp, grad, lp2_k, lp2);
        if (rc != GMMVI_OK) return rc;
    }
    return GMMVI_OK;
}

// the packed two-samples-per-lane sweep (mixture_eval_pk_kernel): 128-sample tiles, at most 8 waves per workgroup, component
// chunks over blockIdx.y so that about two workgroups per CU are in flight
template <int DP>
static int launch_mixture_eval_pk(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed,
                                  const float* logw, const float* X, int N, float* ld, float* lp, float* grad,
                                  const float* logw2, float* lp2) {
    const bool want_grad = grad != nullptr;
    const bool want_merge = want_grad || lp != nullptr;
    static const int env_ky = getenv("GMMVI_ME_PK_KY") ? atoi(getenv("GMMVI_ME_PK_KY")) : 0;
    static const int env_nw = getenv("GMMVI_ME_PK_NW") ? atoi(getenv("GMMVI_ME_PK_NW")) : 0;
    const int tiles = (N + 127) / 128;
    // Geometry: all workgroups resident at once (one round), and four waves per SIMD on the loaded CUs -- either ONE 16-wave
    // workgroup per CU or TWO 8-wave workgroups (the 16-wave form takes 108 KB of LDS at D = 20).  Measured inside the iteration
    // (profiles/r04_notes.md): north-star shape (79 tiles, K = 100) dual sweep 23.3 us at 3 chunks x 16 waves, 24.7 at 6 x 8,
    // 28.0 at 3 x 8 (two waves per SIMD do not cover the first-touch latency of blocks the previous launch wrote on other
    // XCDs); fewer chunks also mean fewer partials to merge.  With 157 tiles (D = 10, K = 200, N = 20 000) a 16-wave
    // workgroup per tile fills 61 % of the CUs and two per tile need two rounds: 3 chunks x 8 waves then (27.5 against 36.4 us).
    const long cus = ctx->num_cus;
    auto chunks_for = [&](long slots) {
        long c = slots / tiles;
        if (c > (K + 3) / 4) c = (K + 3) / 4;          // at least four components per chunk
        return (int)(c < 1 ? 1 : c);
    };
    auto fill = [&](int c, long slots) {               // share of the slots busy over the rounds the launch takes
        const long wgs = (long)tiles * c, rounds = (wgs + slots - 1) / slots;
        return (double)wgs / (double)(rounds * slots);
    };
    const int ky16 = chunks_for(cus), ky8 = chunks_for(2 * cus);
    bool wide = fill(ky16, cus) + 0.05 >= fill(ky8, 2 * cus);
    int ky = wide ? ky16 : ky8;
    // wider dimensions: 8 waves per workgroup; with the gradient ~220 registers and ~136 KB of LDS at D = 50 (one workgroup per
    // CU), without it 120 registers and 34 KB (two)
    if (DP > 24) ky = want_grad ? ky16 : ky8;
    if (env_ky > 0) ky = env_ky;
    if (ky > K) ky = K;
    if (ky < 1) ky = 1;
    const int kchunk = (K + ky - 1) / ky;
    ky = (K + kchunk - 1) / kchunk;
    int nw = wide && DP <= 24 ? 16 : 8;
    if (env_nw > 0) nw = env_nw;
    if (DP > 24 && nw > 8) nw = 8;
    if (nw > kchunk) nw = kchunk;
    if (nw > 16) nw = 16;
    const int nv = (want_grad ? DP : 0) + 2;
    size_t shmem = (size_t)(DP / 2) * 64 * 16 + (size_t)nw * 64 * 8 + (want_merge ? (size_t)(nw > 8 ? 8 : 4) * nv * 64 * 8 : 0);
    float* lp_k = lp;
    float* grad_k = grad;
    float* lp2_k = lp2;
    const bool defer = ctx->defer_combine && ky > 1 && want_merge;
    if (defer) {
        int rc = gmmvi_flush_pending_combine(ctx);
        if (rc != GMMVI_OK) return rc;
    }
    if (ky > 1 && want_merge) {
        size_t need = ((size_t)ky * N * (logw2 ? 2 : 1) + (want_grad ? (size_t)ky * N * D : 0)) * sizeof(float);
        int rc = defer ? gmmvi_defer_reserve(ctx, need) : gmmvi_ws_reserve(ctx, need);
        if (rc != GMMVI_OK) return rc;
        lp_k = (float*)(defer ? ctx->defer_ws : ctx->ws);
        lp2_k = logw2 ? lp_k + (size_t)ky * N : nullptr;
        grad_k = want_grad ? lp_k + (size_t)ky * N * (logw2 ? 2 : 1) : nullptr;
    }
    // block order: the launch's own tiles, the riders (the stepsize block first: it is the longest of them), the carried merge
    const Riders riders = gmmvi_take_pending_riders(ctx, tiles, nw * 64);
    const CombineJob carried = gmmvi_take_pending_combine(ctx, nw * 64, tiles + riders.prep_blocks + riders.sample_blocks);
    if (riders_lds_bytes(riders) > shmem) shmem = riders_lds_bytes(riders);
    dim3 grid(tiles + carried.blocks + riders.prep_blocks + riders.sample_blocks, ky), block(nw * 64);
    {
        GMMVI_PROF_UNITS(ctx, sweep_prof_name(ctx, want_grad, logw2 != nullptr), (double)N * K);
#define GMMVI_LAUNCH_MEPK(FAM, G)                                                                                      \
    do {                                                                                                               \
        if (shmem > 64 * 1024)                                                                                         \
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)mixture_eval_pk_kernel<DP, FAM, G>,               \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));         \
        hipLaunchKernelGGL((mixture_eval_pk_kernel<DP, FAM, G>), grid, block, shmem, ctx->stream, nu, K, D, packed, \
                           logw, X, N, ld, lp_k, grad_k, logw2, lp2_k, carried, riders);                                         \
    } while (0)
        // (the gradient instances exist up to padded D = 40: at 50 they need 223 registers and lose to the matrix-core kernel)
        if constexpr (DP <= 40) {
            if (family == GMMVI_GAUSS) {
                if (want_grad) GMMVI_LAUNCH_MEPK(GMMVI_GAUSS, true); else GMMVI_LAUNCH_MEPK(GMMVI_GAUSS, false);
            } else {
                if (want_grad) GMMVI_LAUNCH_MEPK(GMMVI_STUDENT_T, true); else GMMVI_LAUNCH_MEPK(GMMVI_STUDENT_T, false);
            }
        } else {
            if (want_grad) return gmmvi_fail(ctx, GMMVI_ERR_ARG, "mixture_eval_pk: no gradient instance for this dimension");
            if (family == GMMVI_GAUSS) GMMVI_LAUNCH_MEPK(GMMVI_GAUSS, false); else GMMVI_LAUNCH_MEPK(GMMVI_STUDENT_T, false);
        }
#undef GMMVI_LAUNCH_MEPK
    }
    GMMVI_LAUNCH_CHECK(ctx);
    if (defer) {
        CombineJob& j = ctx->pending;
        j.R = ky; j.N = N; j.D = D;
        j.lp_parts = lp_k; j.grad_parts = grad_k; j.lp2_parts = lp2_k;
        j.lp_out = lp; j.grad_out = grad; j.lp2_out = lp2_k ? lp2 : nullptr;
    } else if (ky > 1 && want_merge) {
        GMMVI_PROF(ctx, "mixture_combine");
        int rc = gmmvi_combine_partials_internal(ctx, ky, N, D, lp_k, grad_k, lp, grad, lp2_k, lp2);
        if (rc != GMMVI_OK) return rc;
    }
    return GMMVI_OK;
}

template <int DP>
static int launch_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed,
                               const float* logw, const float* X, int N, float* ld, float* lp, float* grad,
                               const float* logw2 = nullptr, float* lp2 = nullptr) {
    using PK = Pack<DP>;
    {
        // enough (128-sample tile, component) passes to give every SIMD a few: the packed kernel; otherwise the one-sample
        // kernel, whose 64-sample tiles spread a small problem over more CUs (GMMVI_ME_PK: 0 never, 1 always).
        // Padded D >= 32 (GMMVI_ME_PK_WIDE, experiments): the triangular substitution on the packed vector unit does D^2 / 2
        // multiply-adds per pair and side where the matrix-core kernels multiply whole 16 x 4 fragments of the explicit inverse
        static const int env_pk = getenv("GMMVI_ME_PK") ? atoi(getenv("GMMVI_ME_PK")) : -1;
        static const long env_pk_min = getenv("GMMVI_ME_PK_MIN") ? atol(getenv("GMMVI_ME_PK_MIN")) : 4096;
        static const int env_pk_wide = getenv("GMMVI_ME_PK_WIDE") ? atoi(getenv("GMMVI_ME_PK_WIDE")) : 1;
        const long passes = (long)((N + 127) / 128) * K;
        // measured inside the iteration at K = 100, N = 10^4 (profiles/r04_notes.md): without the gradient the packed kernel wins
        // at D = 40 / 50 (post-update sweep 41.9 -> 33.0 / 53.1 -> 45.7 us; D = 32: 25.8 / 26.1), with the gradient at D = 32 / 40
        // (dual sweep 57.5 -> 52.3 / 79.9 -> 74.4) but not at D = 50 (99 -> 115: 223 registers = two waves per SIMD)
        const bool shape_ok = !PK::FRAGS || (env_pk_wide && (grad == nullptr || DP <= 40));
        if constexpr (DP <= 50) {
            if (shape_ok && env_pk != 0 && (env_pk == 1 || passes >= env_pk_min))
                return launch_mixture_eval_pk<DP>(ctx, family, nu, K, D, packed, logw, X, N, ld, lp, grad, logw2, lp2);
        }
    }
    if constexpr (PK::FRAGS) {
        // enough samples for the workgroup-shared form (eight waves on one component at a time): the block goes to LDS once per
        // workgroup instead of to every wave through the L2
        // (with the gradient: C3 dual sweep 150 -> 107 us; without it the sweep is bound by the log-sum-exp tails of the 16-sample
        // sub-tiles, not by the fragment traffic, and stays on the kernel above: 61 against 59 us)
        static const int env_ws = getenv("GMMVI_ME_WS") ? atoi(getenv("GMMVI_ME_WS")) : -1;      // experiments: 0 off, 1 on, 2 all sweeps
        const bool ws = env_ws >= 0 ? env_ws != 0 : true;
        if (ws && N >= 2048 && K >= 16) {
            if (grad != nullptr) return launch_mixture_eval_mfma_ws<DP, 2>(ctx, family, nu, K, D, packed, logw, X, N, ld, lp, grad, logw2, lp2);
            if (env_ws == 2) return launch_mixture_eval_mfma_ws<DP, 4>(ctx, family, nu, K, D, packed, logw, X, N, ld, lp, grad, logw2, lp2);
        }
        // sub-tiles of 16 samples per wave pass: four (every fragment fetch of a component serves 64 samples) unless the
        // gradient state of the wide dimensions would not fit the registers
        if constexpr (DP >= 40) {                      // (D = 40 with the gradient: 60 us at two sub-tiles, 82 us at four)
            if (grad != nullptr) return launch_mixture_eval_mfma<DP, 2>(ctx, family, nu, K, D, packed, logw, X, N, ld, lp, grad, logw2, lp2);
        }
        return launch_mixture_eval_mfma<DP, 4>(ctx, family, nu, K, D, packed, logw, X, N, ld, lp, grad, logw2, lp2);
    } else {
    const bool want_grad = grad != nullptr;
    const bool want_merge = want_grad || lp != nullptr;
    static const int env_nw = getenv("GMMVI_ME_NW") ? atoi(getenv("GMMVI_ME_NW")) : 0;
    static const int env_ky = getenv("GMMVI_ME_KY") ? atoi(getenv("GMMVI_ME_KY")) : 0;
    auto lds_floats = [&](int nw) {
        size_t merge = (size_t)nw * 64 * ((want_grad ? (DP + 3) / 4 * 4 : 0) + 2) + (want_grad ? 64 * (size_t)(D | 1) : 0) +
                       (logw2 ? (size_t)nw * 128 : 0);
        size_t stage = 64 * (size_t)(D | 1);
        return merge > stage ? merge : stage;
    };
    const int tiles = (N + 63) / 64;
    // geometry: ky chunks of components over blockIdx.y (partials merged by combine_partials), nw waves per workgroup.
    // Measured on MI355X (tools/tune_mixture_eval.py; profiles/r01_notes.md): with few sample tiles (N = 10^4: 157) one
    // 16-wave workgroup per tile leaves 40 % of the CUs idle; 8 waves per workgroup and K split so that ~2 workgroups per CU
    // are in flight is 11-13 % faster including the (coalesced) combine launch.  Plenty of tiles: 16 waves, no split.
    const int nw_max = DP >= GMMVI_ME_WIDE_DP ? 8 : 16;
    int ky = 1, nw = K < nw_max ? K : nw_max;
    if (env_ky > 0) ky = env_ky < K ? env_ky : K;
    else if (K >= 16 && 2L * tiles <= 3L * ctx->num_cus) {
        ky = (int)((2L * ctx->num_cus + tiles / 2) / tiles);
        if (ky > K / 8) ky = K / 8;
        if (ky < 1) ky = 1;
        if (ky > 1) nw = 8;
    }
    const int kchunk = (K + ky - 1) / ky;
    ky = (K + kchunk - 1) / kchunk;
    if (nw > kchunk) nw = kchunk;
    if (env_nw > 0) nw = env_nw < kchunk ? env_nw : kchunk;
    if (nw > nw_max) nw = nw_max;
    while (nw > 1 && lds_floats(nw) * 4 > 96 * 1024) --nw;
    size_t shmem = lds_floats(nw) * 4;
    static const size_t env_shmem_min = getenv("GMMVI_ME_SHMEM_MIN") ? (size_t)atol(getenv("GMMVI_ME_SHMEM_MIN")) : 0;   // experiments: caps the workgroups per CU
    if (shmem < env_shmem_min) shmem = env_shmem_min;
    float* lp_k = lp;
    float* grad_k = grad;
    float* lp2_k = lp2;
    // the merge of the chunk partials: its own launch, or (single-call iteration) left to the next launch
    const bool defer = ctx->defer_combine && ky > 1 && want_merge;
    if (defer) {
        int rc = gmmvi_flush_pending_combine(ctx);
        if (rc != GMMVI_OK) return rc;
    }
    if (ky > 1 && want_merge) {
        size_t need = ((size_t)ky * N * (logw2 ? 2 : 1) + (want_grad ? (size_t)ky * N * D : 0)) * sizeof(float);
        int rc = defer ? gmmvi_defer_reserve(ctx, need) : gmmvi_ws_reserve(ctx, need);
        if (rc != GMMVI_OK) return rc;
        lp_k = (float*)(defer ? ctx->defer_ws : ctx->ws);
        lp2_k = logw2 ? lp_k + (size_t)ky * N : nullptr;
        grad_k = want_grad ? lp_k + (size_t)ky * N * (logw2 ? 2 : 1) : nullptr;
    }
    // block order: the launch's own tiles, the riders (the stepsize block first: it is the longest of them), the carried merge
    const Riders riders = gmmvi_take_pending_riders(ctx, tiles, nw * 64);
    const CombineJob carried = gmmvi_take_pending_combine(ctx, nw * 64, tiles + riders.prep_blocks + riders.sample_blocks);
    if (riders_lds_bytes(riders) > shmem) shmem = riders_lds_bytes(riders);
    dim3 grid(tiles + carried.blocks + riders.prep_blocks + riders.sample_blocks, ky), block(nw * 64);
    {
        GMMVI_PROF_UNITS(ctx, sweep_prof_name(ctx, want_grad, logw2 != nullptr), (double)N * K);
#define GMMVI_LAUNCH_ME(FAM, G)                                                                                     \
    do {                                                                                                            \
        if (shmem > 64 * 1024)                                                                                      \
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)mixture_eval_kernel<DP, FAM, G>,               \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));      \
        hipLaunchKernelGGL((mixture_eval_kernel<DP, FAM, G>), grid, block, shmem, ctx->stream, nu, K, D, packed, \
                           logw, X, N, ld, lp_k, grad_k, logw2, lp2_k, carried, riders);                                      \
    } while (0)
        if (family == GMMVI_GAUSS) {
            if (want_grad) GMMVI_LAUNCH_ME(GMMVI_GAUSS, true); else GMMVI_LAUNCH_ME(GMMVI_GAUSS, false);
        } else {
            if (want_grad) GMMVI_LAUNCH_ME(GMMVI_STUDENT_T, true); else GMMVI_LAUNCH_ME(GMMVI_STUDENT_T, false);
        }
#undef GMMVI_LAUNCH_ME
    }
    GMMVI_LAUNCH_CHECK(ctx);
    if (defer) {
        CombineJob& j = ctx->pending;
        j.R = ky; j.N = N; j.D = D;
        j.lp_parts = lp_k; j.grad_parts = grad_k; j.lp2_parts = lp2_k;
        j.lp_out = lp; j.grad_out = grad; j.lp2_out = lp2_k ? lp2 : nullptr;
    } else if (ky > 1 && want_merge) {
        GMMVI_PROF(ctx, "mixture_combine");
        int rc = gmmvi_combine_partials_internal(ctx, ky, N, D, lp_k, grad_k, lp, grad, lp2_k, lp2);
        if (rc != GMMVI_OK) return rc;
    }
    return GMMVI_OK;
    }
}

// C++ linkage (common.h): the register-path block layout (Pack<padded D>) whatever the blocked threshold says -- gmmvi_more
// reads that layout and re-packs the components of a blocked-path dimension (50 < D <= 63 by default) for its call
int gmmvi_pack_register_layout(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev, float* packed_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_MAX_DIM && means_dev && chols_dev && packed_dev);
    int dp = gmmvi_padded_dim(D);
    GMMVI_DISPATCH_DP(dp, hipLaunchKernelGGL((pack_kernel<DP>), dim3(K), dim3(64), 0, ctx->stream, (int)GMMVI_GAUSS, 0.f, K, D,
                                             means_dev, chols_dev, packed_dev, (float*)nullptr));
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

extern "C" {

int gmmvi_pack_components(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* means_dev,
                          const float* chols_dev, float* packed_dev, float* inv_chols_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_BLOCKED_MAX_DIM);
    GMMVI_ARG_CHECK(ctx, family == GMMVI_GAUSS || family == GMMVI_STUDENT_T);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && packed_dev);
    if (gmmvi_is_blocked_dim(D)) return gmmvi_blocked_pack(ctx, family, nu, K, D, means_dev, chols_dev, packed_dev, inv_chols_dev);
    int dp = gmmvi_padded_dim(D);
    GMMVI_PROF(ctx, "pack_components");
    GMMVI_DISPATCH_DP(dp, hipLaunchKernelGGL((pack_kernel<DP>), dim3(K), dim3(64), 0, ctx->stream, family, nu, K, D,
                                             means_dev, chols_dev, packed_dev, inv_chols_dev));
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_cholesky(gmmvi_ctx* ctx, int K, int D, const float* covs_dev, float* chols_dev, int32_t* ok_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 0 && D >= 1 && D <= GMMVI_BLOCKED_MAX_DIM);
    if (K == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, covs_dev && chols_dev);
    if (gmmvi_is_blocked_dim(D)) return gmmvi_blocked_cholesky(ctx, K, D, covs_dev, chols_dev, ok_dev);
    hipLaunchKernelGGL(cholesky_kernel, dim3(K), dim3(64), (size_t)D * (D + 1) * 4, ctx->stream, D, covs_dev,
                       chols_dev, ok_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed_dev,
                       const float* logw_dev, const float* X_dev, int N, float* ld_out_dev, float* lp_out_dev,
                       float* grad_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_BLOCKED_MAX_DIM && N >= 0);
    GMMVI_ARG_CHECK(ctx, family == GMMVI_GAUSS || family == GMMVI_STUDENT_T);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, packed_dev && logw_dev && X_dev);
    GMMVI_ARG_CHECK(ctx, ld_out_dev || lp_out_dev || grad_out_dev);
    if (gmmvi_is_blocked_dim(D))
        return gmmvi_blocked_mixture_eval(ctx, family, nu, K, D, packed_dev, logw_dev, nullptr, X_dev, N, ld_out_dev, lp_out_dev,
                                          grad_out_dev, nullptr);
    int dp = gmmvi_padded_dim(D);
    GMMVI_DISPATCH_DP(dp, return launch_mixture_eval<DP>(ctx, family, nu, K, D, packed_dev, logw_dev, X_dev, N,
                                                         ld_out_dev, lp_out_dev, grad_out_dev));
    return GMMVI_OK;
}

int gmmvi_mixture_eval_dual(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed_dev,
                            const float* logw_dev, const float* logw2_dev, const float* X_dev, int N, float* ld_out_dev,
                            float* lp_out_dev, float* grad_out_dev, float* lp2_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_BLOCKED_MAX_DIM && N >= 0);
    GMMVI_ARG_CHECK(ctx, family == GMMVI_GAUSS || family == GMMVI_STUDENT_T);
    if (N == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, packed_dev && logw_dev && logw2_dev && X_dev && lp_out_dev && lp2_out_dev);
    if (gmmvi_is_blocked_dim(D))
        return gmmvi_blocked_mixture_eval(ctx, family, nu, K, D, packed_dev, logw_dev, logw2_dev, X_dev, N, ld_out_dev,
                                          lp_out_dev, grad_out_dev, lp2_out_dev);
    int dp = gmmvi_padded_dim(D);
    GMMVI_DISPATCH_DP(dp, return launch_mixture_eval<DP>(ctx, family, nu, K, D, packed_dev, logw_dev, X_dev, N,
                                                         ld_out_dev, lp_out_dev, grad_out_dev, logw2_dev, lp2_out_dev));
    return GMMVI_OK;
}

}  // extern "C"
