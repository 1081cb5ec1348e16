// KL-constrained natural-gradient component update, production kernel
// (gmmvi_modules/ng_based_component_updater.py:431-524; bracketing_search :335-429; kl :244-333).
//
// Same decisions as the reference, different arithmetic route.  In the coordinates whitened by the old Cholesky factor
// L (Sigma = L L^T) the updated precision is Q' = L^-T (I + M/eta) L^-1 with M = L^T R L (R = expected_hessian_neg), and
//     KL(eta) = 1/2 [ logdet B - D + tr B^-1 + |B^-1 w|^2 / eta^2 ],   B = I + M/eta,  w = L^T (g_neg + (R_sym - R) mu)
// (derivation in DESIGN.md section 4).  M is reduced ONCE to tridiagonal form T = H^T M H (Householder), after which
// logdet, tr(B^-1) (two-sided pivot recurrences) and |B^-1 w| (Thomas solve) cost O(D) per eta.  The bisection of the
// reference is then evaluated speculatively: the 63 nodes of the next six levels of the bisection tree are probed in
// parallel, one lane per node, and the tree is walked with the reference's stop rules -- the visited etas, the
// comparisons and therefore the accepted eta are those of the sequential search.  The new factor follows from the
// UL factorisation B = U U^T:  L' = L U^-T is lower triangular with positive diagonal, i.e. chol(Sigma') itself.
// "Cholesky failed" (non-positive pivot / NaN) maps to KL = float32.max and to the reject branch (:320-324, :493).
// One wavefront per component, matrices in LDS (row stride D+1), lane = row (D <= 64).
#include "common.h"
#include "blocked.h"
#include "wave_reduce.h"
#include "stein_finalize.h"
#include <cfloat>
#include <type_traits>

#ifndef GMMVI_UKL_UNROLL_MAX
#define GMMVI_UKL_UNROLL_MAX 50      // static dimensions up to here get fully unrolled Householder / UL loops
#endif

namespace {

// compile-time loop: f(std::integral_constant<int, B>) ... f(std::integral_constant<int, E - 1>)
template <int B, int E, class F>
__device__ __forceinline__ void ukl_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        ukl_static_for<B + 1, E>(f);
    }
}

// hand-over through LDS inside the one wavefront that runs the chain phases: with several wavefronts in the workgroup only
// that wave's LDS queue has to drain (the others must not be waited for); a single-wave workgroup keeps the plain barrier
// (measured at D = 20: 37.9 us against 42.3 us with the wave-local form)
#define UKL_WSYNC()                                                \
    do {                                                           \
        if constexpr (NW == 1) {                                   \
            __syncthreads();                                       \
        } else {                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
            __builtin_amdgcn_wave_barrier();                       \
        }                                                          \
    } while (0)

typedef float ukl_v2 __attribute__((ext_vector_type(2)));

// out(i, j) = sum_c fa(c, i) fb(c, j), i, j < D, by 2 x 5 register blocks (one per thread: D <= 50 gives <= 250 blocks): seven
// LDS reads per ten multiply-adds instead of twenty
template <int D, class FA, class FB, class FO>
__device__ __forceinline__ void ukl_block_product(int t, FA fa, FB fb, FO out) {
    constexpr int NG = (D + 4) / 5, NP = (D + 1) / 2;
    if (t >= NP * NG) return;
    const int i0 = 2 * (t / NG), j0 = 5 * (t % NG);
    const int i1 = min(i0 + 1, D - 1);
    int jj[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) jj[u] = min(j0 + u, D - 1);
    float acc[2][5];
#pragma unroll
    for (int u = 0; u < 5; ++u) { acc[0][u] = 0.f; acc[1][u] = 0.f; }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float a0 = fa(c, i0), a1 = fa(c, i1);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const float b = fb(c, jj[u]);
            acc[0][u] = fmaf(a0, b, acc[0][u]);
            acc[1][u] = fmaf(a1, b, acc[1][u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        if (j0 + u < D) {
            out(i0, j0 + u, acc[0][u]);
            if (i0 + 1 < D) out(i0 + 1, j0 + u, acc[1][u]);
        }
    }
}

__device__ __forceinline__ float wsum(float v) { return gmmvi_wave_sum(v); }

struct Ws {
    int D, ld;
    float *L, *M, *Mc;                 // D x D (stride ld): old factor; M (tridiagonalised in place); copy of M / outputs
    float *mu, *w, *wt, *td, *te, *v, *q, *y, *z;   // D each
    float* pr;                          // per-lane probe scratch [2 D][64]
};

__device__ void carve(Ws& s, float* sm, int D) {
    s.D = D; s.ld = D + 1;
    const int m = D * (D + 1);
    s.L = sm; s.M = sm + m; s.Mc = sm + 2 * m;
    float* v = sm + 3 * m;
    s.mu = v; s.w = v + D; s.wt = v + 2 * D; s.td = v + 3 * D; s.te = v + 4 * D; s.v = v + 5 * D; s.q = v + 6 * D;
    s.y = v + 7 * D; s.z = v + 8 * D;
    s.pr = v + 9 * D;
}

size_t lds_bytes(int D) { return ((size_t)3 * D * (D + 1) + 9 * D + (size_t)2 * D * 64) * sizeof(float); }

// KL(eta) from the tridiagonal form; every lane may evaluate a different eta (scratch column = lane).  The two recurrences
// are chains of D dependent steps with a reciprocal each: v_rcp_f32 (1 ulp) instead of the IEEE division sequence halves the
// search phase (D = 20: 5.1 -> 3.8 us); accept / reject decisions and probe counts of the parity tests are unchanged.
template <int DC>
__device__ __forceinline__ float kl_tridiag(const Ws& s, float eta) {
    const int D = DC > 0 ? DC : s.D, lane = threadIdx.x;
    const float inv = 1.f / eta;
    float* dcol = s.pr + lane;                 // d_i at [i][lane], c_i at [D + i][lane]
    float dprev = 1.f, cprev = 0.f, logdet = 0.f;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float a = fmaf(s.td[i], inv, 1.f);
        float d = a, c = s.wt[i];
        if (i > 0) {
            const float b = s.te[i - 1] * inv;
            const float r = b * __builtin_amdgcn_rcpf(dprev);
            d = fmaf(-b, r, a);
            c = fmaf(-r, cprev, c);
        }
        ok = ok && (d > 0.f);
        logdet += __logf(d);
        dcol[i * 64] = d;
        dcol[(D + i) * 64] = c;
        dprev = d; cprev = c;
    }
    float dnext = 1.f, ynext = 0.f, tr = 0.f, yy = 0.f;
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
        const float a = fmaf(s.td[i], inv, 1.f);
        const float d = dcol[i * 64], c = dcol[(D + i) * 64];
        const float rd = __builtin_amdgcn_rcpf(d);
        float delta = a, y = c * rd;
        if (i < D - 1) {
            const float b = s.te[i] * inv;
            delta = fmaf(-b, b * __builtin_amdgcn_rcpf(dnext), a);
            y = (c - b * ynext) * rd;
        }
        tr += __builtin_amdgcn_rcpf(d + delta - a);
        yy = fmaf(y, y, yy);
        dnext = delta; ynext = y;
    }
    float kl = 0.5f * (logdet - (float)D + tr + yy * inv * inv);
    if (!ok || !(kl == kl)) kl = FLT_MAX;
    return kl;
}

// DC > 0: dimension known at compile time (inner loops unrolled, LDS reads issued in batches); DC == 0: generic.
template <int DC, int NW>
__global__ __launch_bounds__(64 * NW) void update_kl_fast_kernel(int Drt, float* __restrict__ means, float* __restrict__ chols,
                                                            float* H_neg, float* g_neg, SteinSlab slab, int slab_N, int slab_flags,
                                                            const float* __restrict__ packed_old,
                                                            const float* __restrict__ stepsizes, float temperature,
                                                            float l2_init, float* __restrict__ last_eta, float* __restrict__ l2,
                                                            float* __restrict__ num_updates, int32_t* __restrict__ success_out,
                                                            float* __restrict__ kl_out, int32_t* __restrict__ nprobes_out,
                                                            float* __restrict__ packed_out, int DPk) {
    extern __shared__ float sm[];
    const int D = DC > 0 ? DC : Drt;
    Ws s;
    carve(s, sm, D);
    const int k = blockIdx.x, t = threadIdx.x, ld = D + 1;
    constexpr int NTH = 64 * NW;
#ifdef GMMVI_UKL_STAMPS            // experiment builds (tools/ukl_probe.py): phase time stamps of component 0 into kl_out[1..6]
    long long stamp[12];
    stamp[0] = wall_clock64();
#define UKL_STAMP(i) stamp[i] = wall_clock64()
#else
#define UKL_STAMP(i)
#endif
    // single-call iteration: the Stein estimate arrives as its partial slab (stein_finalize.h).
    //  * self-normalised weights (H symmetric): the whitened matrix is formed DIRECTLY from the moment matrix C = sum e g (x-mu)^T,
    //    M = L^T (-sym(C Sigma^-1) / sum e) L = -sym(L^T C L^-T) / sum e, w = -L^T (sum e g) / sum e:
    //    one product and one triangular substitution instead of two substitutions (Sigma^-1), a round trip of H through global
    //    memory and two products.  H_neg / g_neg are not written on this route.
    //  * plain importance weights (H not symmetric, its lower triangle is mirrored): the estimate is finished as the
    //    stand-alone kernel does it, H_neg / g_neg go through global memory.
    bool direct = false;
    if constexpr (DC > 0) {
        if (slab.part != nullptr) {
            direct = (slab_flags & GMMVI_SELF_NORMALIZED) != 0 && (slab_flags & GMMVI_EXPLICIT_ESTIMATE) == 0 &&
                     (DC + 1) * (DC + 1) + slab.R + 4 + (Pack<DC>::FRAGS ? DC * DC : 0) <= 2 * DC * 64;
            if constexpr (DC <= 24) {                  // (wider instances are only launched with a slab on the direct route)
                if (!direct) {
                    stein_finalize_component<DC>(sm, k, D, slab.R, slab_N, slab_flags, slab.part, slab.part_m, H_neg, g_neg, packed_old);
                    __threadfence_block();
                    __syncthreads();
                }
            }
        }
    }
    float* Lg = chols + (size_t)k * D * D;
    float* mug = means + (size_t)k * D;
    const float* Rg = H_neg + (size_t)k * D * D;

    if (direct) {
        const int D1 = D + 1;
        float* A = s.pr;                               // the probe scratch is idle until the search
        const float Mx = stein_slab_sum(A, A + D1 * D1, k, D, slab.R, slab.part, slab.part_m);
        UKL_STAMP(7);
        const float scale = stein_moment_scale(A[D * D1 + D], Mx, slab_N, slab_flags);
        for (int e = t; e < D * D; e += NTH) {
            const int i = e / D, j = e % D;
            s.L[i * ld + j] = (j <= i) ? Lg[e] : 0.f;
        }
        if (t < D) {
            s.mu[t] = mug[t];
            s.y[t] = -A[t * D1 + D] * scale;           // g~ = g_neg (H is symmetric: no correction)
            s.v[t] = 1.f / Lg[t * D + t];
        }
        __syncthreads();
        // P = L^T C (into Mc): P[i][j] = sum_{c >= i} L[c][i] C[c][j]
        if constexpr (DC >= 32) {
            static_assert(DC < 32 || ((DC + 4) / 5) * ((DC + 1) / 2) <= 64 * NW, "one register block per thread");
            ukl_block_product<(DC >= 32 ? DC : 2)>(t, [&](int c, int i) { return s.L[c * ld + i]; },
                                                   [&](int c, int j) { return A[c * D1 + j]; },
                                                   [&](int i, int j, float v) { s.Mc[i * ld + j] = v; });
        } else {
        for (int o = t; o < D * D; o += NTH) {
            const int i = o / D, j = o % D;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) a = fmaf(s.L[c * ld + i], A[c * D1 + j], a);          // L[c][i] = 0 for c < i
            s.Mc[i * ld + j] = a;
        }
        }
        __syncthreads();
        UKL_STAMP(8);
        if constexpr (Pack<(DC > 0 ? DC : 2)>::FRAGS) {
            // blocks that carry L^-1 (operand fragments, common.h): S = P L^-T as a triangular product on all threads
            using PK = Pack<(DC > 0 ? DC : 2)>;
            constexpr int DPc = DC > 0 ? DC : 2;
            float* Li = A + ((D1 * D1 + slab.R + 3) & ~3);                  // dense L^-1 [DP][DP] behind the moment matrix
            const float* Pk = packed_old + (size_t)k * PK::STRIDE;
            for (int e = t; e < DPc * DPc; e += NTH) Li[e] = 0.f;
            __syncthreads();
            for (int e = t; e < 64 * PK::NF; e += NTH) {
                const int f = e >> 6, l = e & 63;
                int mt = 0, rem = f;
                for (;; ++mt) { const int nfm = PK::nf(mt); if (rem < nfm) break; rem -= nfm; }
                const int row = 16 * mt + (l & 15), col = 4 * rem + (l >> 4);
                if (row < DPc && col < DPc) Li[row * DPc + col] = Pk[PK::FWD + e];
            }
            __syncthreads();
            // (L^-1 is stored dense with its zeros: the sum runs over all m, straight-line)
            ukl_block_product<DPc>(t, [&](int m2, int r) { return s.Mc[r * ld + m2]; },
                                   [&](int m2, int j) { return Li[j * DPc + m2]; },
                                   [&](int r, int j, float v) { s.M[r * ld + j] = -scale * v; });
        } else if (t < D) {
            // S L^T = P, row r by lane r (forward over the columns); M = -scale S, symmetrised below
            float srow[DC > 0 ? DC : 1];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float a = s.Mc[t * ld + j];
#pragma unroll
                for (int m2 = 0; m2 < j; ++m2) a = fmaf(-srow[m2], s.L[j * ld + m2], a);
                srow[j] = a * s.v[j];
            }
#pragma unroll
            for (int j = 0; j < D; ++j) s.M[t * ld + j] = -scale * srow[j];
        }
        if (t < D) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) a = fmaf(s.L[c * ld + t], s.y[c], a);
            s.w[t] = a;
            s.wt[t] = a;
        }
        __syncthreads();
    } else {
    // ---- load: L, R_sym (lower triangle mirrored, as tf.linalg.cholesky reads only the lower part) into M -----------
    for (int e = t; e < D * D; e += NTH) {
        const int i = e / D, j = e % D;
        s.L[i * ld + j] = (j <= i) ? Lg[e] : 0.f;
        s.M[i * ld + j] = (j <= i) ? Rg[e] : Rg[j * D + i];
    }
    if (t < D) {
        s.mu[t] = mug[t];
        // g~ = g_neg + (R_sym - R) mu   (zero correction for a symmetric R)
        float gt = g_neg[(size_t)k * D + t];
        for (int j = t + 1; j < D; ++j) gt = fmaf(Rg[j * D + t] - Rg[t * D + j], mug[j], gt);
        s.y[t] = gt;
    }
    __syncthreads();
    // ---- T1 = R_sym L (into Mc), M = L^T T1, w = L^T g~ : the D^2 outputs are spread over all lanes (2 x 5 register blocks
    // for the wide instances) ---------------------------------------------------------------------------------------------
    if constexpr (DC >= 32) {
        ukl_block_product<(DC >= 32 ? DC : 2)>(t, [&](int c, int i) { return s.M[i * ld + c]; },
                                               [&](int c, int j) { return s.L[c * ld + j]; },
                                               [&](int i, int j, float v) { s.Mc[i * ld + j] = v; });
        __syncthreads();
        ukl_block_product<(DC >= 32 ? DC : 2)>(t, [&](int c, int i) { return s.L[c * ld + i]; },
                                               [&](int c, int j) { return s.Mc[c * ld + j]; },
                                               [&](int i, int j, float v) { s.M[i * ld + j] = v; });
    } else {
    for (int o = t; o < D * D; o += NTH) {
        const int i = o / D, j = o % D;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) a = fmaf(s.M[i * ld + c], s.L[c * ld + j], a);         // L[c][j] = 0 for c < j
        s.Mc[i * ld + j] = a;
    }
    __syncthreads();
    for (int o = t; o < D * D; o += NTH) {
        const int i = o / D, j = o % D;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) a = fmaf(s.L[c * ld + i], s.Mc[c * ld + j], a);        // L[c][i] = 0 for c < i
        s.M[i * ld + j] = a;
    }
    }
    if (t < D) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) a = fmaf(s.L[c * ld + t], s.y[c], a);
        s.w[t] = a;
        s.wt[t] = a;
    }
    __syncthreads();
    }
    for (int o = t; o < D * D; o += NTH) {          // exact symmetry for the reflectors; keep a copy for the final step
        const int i = o / D, j = o % D;
        if (j < i) {
            const float m = 0.5f * (s.M[i * ld + j] + s.M[j * ld + i]);
            s.M[i * ld + j] = m; s.M[j * ld + i] = m;
            s.Mc[i * ld + j] = m; s.Mc[j * ld + i] = m;
        } else if (j == i) {
            s.Mc[i * ld + i] = s.M[i * ld + i];
        }
    }
    __syncthreads();

    // ---- from here to the accepted factor one wavefront works alone (lane = row / tree node; the phases are chains of D
    // dependent steps): its LDS hand-overs only need the wave's own queue drained, the other waves wait at the barrier below
    bool success_w = false;
    float eta_w = 0.f, kl_w = -1.f;
    int probes_w = 0;
    if (NW == 1 || __builtin_amdgcn_readfirstlane(t >> 6) == 0) {          // wave-uniform: the region keeps its scalar branches
    UKL_STAMP(1);                                      // load + products done
    // ---- Householder tridiagonalisation of M, reflectors applied to wt.  Every lane forms the column norm and the two dot
    // products itself from broadcast LDS vectors: no cross-lane reduction chains on the critical path.  With a static D the
    // lane keeps its row of M (and a replica of wt) in registers, so a step costs ~20 LDS operations instead of ~200 --
    // a single wavefront pays the full issue latency of every one of them. --------------------------------------------------
    if (DC > 0) {
        constexpr int DR = DC > 0 ? DC : 1;
        constexpr int D4 = (DR + 3) / 4;
        constexpr int DH = 2 * D4;                  // pairs (padded to whole float4 reads)
        // the lane's row of M, the replica of wt and the broadcast vectors are kept as PAIRS: the dot products and the rank-2
        // update run on v_pk_fma_f32 / v_pk_mul_f32 (two elements per instruction) -- the phase is a single wavefront issuing
        // one vector instruction every four cycles, so its time is its instruction count.  Elements beyond D and the dead part
        // j <= c of the reflector are exact zeros.
        ukl_v2 mrow[DH], wtr[DH];
#pragma unroll
        for (int m2 = 0; m2 < DH; ++m2) {
            mrow[m2].x = (t < DR && 2 * m2 < DR) ? s.M[t * ld + 2 * m2] : 0.f;
            mrow[m2].y = (t < DR && 2 * m2 + 1 < DR) ? s.M[t * ld + 2 * m2 + 1] : 0.f;
            wtr[m2].x = (2 * m2 < DR) ? s.wt[2 * m2] : 0.f;
            wtr[m2].y = (2 * m2 + 1 < DR) ? s.wt[2 * m2 + 1] : 0.f;
        }
        // column c of M (element (j, c) lives in lane j) and p reach the other lanes through v_readlane: no LDS round trip in
        // the two hand-overs of a step (the D-step chain is latency bound)
        auto rl = [](float v, int lane_c) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_c)); };
        // one instantiation per column (ukl_static_for): c is a compile-time constant in every step -- `#pragma unroll` gives
        // up on this body from D ~ 40, and with a run-time c every select below stays a select
        ukl_static_for<0, (DR > 2 ? DR - 2 : 0)>([&](auto c_const) {
            constexpr int c = decltype(c_const)::value;
            constexpr int M0 = (c + 1) / 2;          // first pair with a live element (j > c)
            const float mc = (c & 1) ? mrow[c / 2].y : mrow[c / 2].x;
            ukl_v2 x[DH];
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) {
                x[m2].x = (2 * m2 < DR && 2 * m2 > c) ? rl(mc, 2 * m2 < DR ? 2 * m2 : 0) : 0.f;
                x[m2].y = (2 * m2 + 1 < DR && 2 * m2 + 1 > c) ? rl(mc, 2 * m2 + 1 < DR ? 2 * m2 + 1 : 0) : 0.f;
            }
            const float xt = mc;                     // this lane's own element of the column (its v_t below)
            const float x1 = ((c + 1) & 1) ? x[(c + 1) / 2].y : x[(c + 1) / 2].x;
            // dead part j <= c + 1 zeroed (compile-time), then the tail norm on pairs
            ukl_v2 t2 = {0.f, 0.f};
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) {
                ukl_v2 z = x[m2];
                if (2 * m2 <= c + 1) z.x = 0.f;
                if (2 * m2 + 1 <= c + 1) z.y = 0.f;
                t2 = z * z + t2;
            }
            const float tail = t2.x + t2.y;
            if (!(tail > 0.f)) {                    // column already tridiagonal (also covers NaN: handled later)
                if (t == 0) s.te[c] = x1;
                UKL_WSYNC();
                return;                             // uniform: every lane computed the same tail
            }
            const float nrm = __builtin_amdgcn_sqrtf(tail + x1 * x1);       // v_sqrt_f32 / v_rcp_f32 (1 ulp): the reflector only
                                                                            // has to be orthogonal to working precision
            const float alpha = (x1 > 0.f) ? -nrm : nrm;
            const float beta = __builtin_amdgcn_rcpf(nrm * nrm - alpha * x1);      // 2 / |v|^2, |v|^2 = 2 (alpha^2 - alpha x1)
            // v_j: 0 for j <= c, x1 - alpha for j = c + 1, x_j below
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) {
                if (2 * m2 <= c) x[m2].x = 0.f; else if (2 * m2 == c + 1) x[m2].x = x1 - alpha;
                if (2 * m2 + 1 <= c) x[m2].y = 0.f; else if (2 * m2 + 1 == c + 1) x[m2].y = x1 - alpha;
            }
            const bool act = (t > c) && (t < DR);
            ukl_v2 p2 = {0.f, 0.f};
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) p2 = mrow[m2] * x[m2] + p2;
            float p = p2.x + p2.y;
            p = act ? p * beta : 0.f;
            const float vv = act ? (t == c + 1 ? x1 - alpha : xt) : 0.f;
            ukl_v2 pj[DH];
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) {
                pj[m2].x = (2 * m2 < DR && 2 * m2 > c) ? rl(p, 2 * m2 < DR ? 2 * m2 : 0) : 0.f;
                pj[m2].y = (2 * m2 + 1 < DR && 2 * m2 + 1 > c) ? rl(p, 2 * m2 + 1 < DR ? 2 * m2 + 1 : 0) : 0.f;
            }
            ukl_v2 kk2 = {0.f, 0.f}, wd2 = {0.f, 0.f};
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) { kk2 = x[m2] * pj[m2] + kk2; wd2 = x[m2] * wtr[m2] + wd2; }
            const float kk = (kk2.x + kk2.y) * (0.5f * beta);
            const float wdot = (wd2.x + wd2.y) * beta;
            const float qq = p - kk * vv;
            const ukl_v2 vv2 = {vv, vv}, qq2 = {qq, qq}, kkn = {-kk, -kk}, wdn = {-wdot, -wdot};
#pragma unroll
            for (int m2 = M0; m2 < DH; ++m2) {
                const ukl_v2 u = kkn * x[m2] + pj[m2];              // p_j - kk v_j
                mrow[m2] = mrow[m2] - (vv2 * u + qq2 * x[m2]);
                wtr[m2] = wdn * x[m2] + wtr[m2];                    // replica of wt in every lane
            }
            if (t == 0) s.te[c] = alpha;
        });
        UKL_WSYNC();
        if (t < DR) {
            float dd = 0.f, wme = 0.f;
#pragma unroll
            for (int j = 0; j < DR; ++j) {
                const float mj = (j & 1) ? mrow[j / 2].y : mrow[j / 2].x;
                const float wj = (j & 1) ? wtr[j / 2].y : wtr[j / 2].x;
                dd = (j == t) ? mj : dd; wme = (j == t) ? wj : wme;
            }
            constexpr int JS = DR >= 2 ? DR - 2 : 0;
            const float sub = (JS & 1) ? mrow[JS / 2].y : mrow[JS / 2].x;
            s.td[t] = dd;
            s.wt[t] = wme;
            if (t == DR - 1 && DR >= 2) s.te[DR - 2] = sub;
        }
        UKL_WSYNC();
    } else {
    for (int c = 0; c + 2 < D; ++c) {
            const float x1 = s.M[(c + 1) * ld + c];
            float tail = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float xj = s.M[j * ld + c];
                tail = (j > c + 1) ? fmaf(xj, xj, tail) : tail;
            }
            if (!(tail > 0.f)) {                        // column already tridiagonal (also covers NaN: handled later)
                if (t == 0) s.te[c] = x1;
                continue;                               // uniform: every lane computed the same tail
            }
            const float nrm = sqrtf(tail + x1 * x1);
            const float alpha = (x1 > 0.f) ? -nrm : nrm;
            const float beta = 1.f / (nrm * nrm - alpha * x1);          // 2 / |v|^2, |v|^2 = 2 (alpha^2 - alpha x1)
            const bool act = (t > c) && (t < D);
            const float vv = act ? (t == c + 1 ? x1 - alpha : s.M[t * ld + c]) : 0.f;
            if (t < D) s.v[t] = vv;
            UKL_WSYNC();
            float p = 0.f;
            if (act) {
#pragma unroll
                for (int j = 0; j < D; ++j) p = fmaf(s.M[t * ld + j], s.v[j], p);              // v[j] = 0 for j <= c
                p *= beta;
            }
            if (t < D) s.q[t] = p;                                                             // p_j = 0 for j <= c
            UKL_WSYNC();
            float kk = 0.f, wdot = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float vj = s.v[j];
                kk = fmaf(vj, s.q[j], kk);
                wdot = fmaf(vj, s.wt[j], wdot);
            }
            kk *= 0.5f * beta;
            wdot *= beta;
            const float qq = p - kk * vv;
            UKL_WSYNC();                            // everyone has read wt / q before they change
            if (act) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float vj = s.v[j];
                    s.M[t * ld + j] -= vv * (s.q[j] - kk * vj) + qq * vj;                      // q[j] = v[j] = 0 for j <= c
                }
                s.wt[t] -= wdot * vv;
            }
            if (t == 0) s.te[c] = alpha;
            UKL_WSYNC();
        }
        if (t < D) s.td[t] = s.M[t * ld + t];
        if (t == 0 && D >= 2) s.te[D - 2] = s.M[(D - 1) * ld + (D - 2)];
        UKL_WSYNC();
    }

    UKL_STAMP(2);                                      // tridiagonal form done
    // ---- speculative bisection: 63 tree nodes (6 levels) per round, one lane per node ------------------------------------
    const float eps = stepsizes[k];
    const float last = last_eta[k];
    float lb, ub;
    if (last < 0.f) { lb = -20.f; ub = 80.f; }                                            // :462-466
    else { lb = fmaxf(0.f, __logf(last) - 3.f); ub = __logf(last) + 3.f; }                 // :467-471
    bool ub_ok = false, done = false;
    int probes = 0, iters = 0;
    float kl_ub = -1.f;                 // KL at the current upper bracket end (the probe that set it)
    while (!done && iters < 1000) {                                                        // :399
        // node of this lane: heap index t (1 = root); follow its bits from the root
        float nlb = lb, nub = ub;
        if (t >= 2) {
            const int depth = 31 - __clz(t);                   // number of steps from the root
            for (int b = depth - 1; b >= 0; --b) {
                const float mid = 0.5f * (nub + nlb);
                if ((t >> b) & 1) nlb = mid; else nub = mid;   // bit 1: "KL too large" branch (lb = eta)
            }
        }
        const float neta = 0.5f * (nub + nlb);
        const float e_eta = expf(neta);
        const float ndiff = fminf(expf(nub) - e_eta, e_eta - expf(nlb));                  // :401
        const float nkl = (t >= 1) ? kl_tridiag<DC>(s, e_eta) : 0.f;                           // :407
        int n = 1;
        for (int level = 0; level < 6 && !done && iters < 1000; ++level, ++iters) {
            const float diff = __shfl(ndiff, n), klv = __shfl(nkl, n), eta = __shfl(neta, n);
            if (diff < 1e-1f) { done = true; break; }                                      // :404-405
            ++probes;
            if (fabsf(eps - klv) < 1e-1f * eps) { lb = ub = eta; kl_ub = klv; done = true; break; }   // :410-413
            if (eps > klv) { ub = eta; ub_ok = true; kl_ub = klv; n = 2 * n; }             // :415-417
            else { lb = eta; n = 2 * n + 1; }                                              // :418-419
        }
    }
    UKL_STAMP(3);                                      // search done
    if (ub_ok) lb = ub;                                                                    // :423-424
    const float lo = expf(lb), hi = expf(ub);                                              // :426-427
    const float eta_star = fmaxf(lo, temperature);                                         // :476
    bool success = (lo == hi);                                                             // :478
    float kl_val = -1.f;
    const float inv = 1.f / eta_star;
    if (success) {
        // :480-482 evaluates KL at eta*; when eta* is the accepted bracket end (the usual case, eta* = lo >= temperature)
        // that is the value the search already holds for exactly this eta
        kl_val = (eta_star == lo && kl_ub >= 0.f) ? kl_ub : __shfl(kl_tridiag<DC>(s, eta_star), 0);
        success = kl_val < FLT_MAX;                                                        // :488
    }
    if (DC > 0) {
        // ---- static D: B = I + Mc/eta* = U U^T (U upper) factorised right-looking with the lane's row in registers; then
        // L' = L U^-T and z = U^-1 w by one back substitution per lane (lane t: row t of L, every lane also carries w), and
        // mu' = mu - L' z / eta*.  Lane t keeps row t of U in registers; the other lanes read it with v_readlane (both phases
        // are D-step dependency chains: an LDS hand-over per step was half their time). ----------------------------------------
        constexpr int DR = DC > 0 ? DC : 1;
        float xr[DR];                                              // B row, later the solution row (L' row t)
        float urow[DR];                                            // row t of U (entries j >= t)
#pragma unroll
        for (int c = 0; c < DR; ++c) urow[c] = 0.f;
        auto rlane = [](float v, int lane_c) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_c)); };
        float new_mu = 0.f;
        if (success) {
#pragma unroll
            for (int c = 0; c < DR; ++c)
                xr[c] = (t < DR && c >= t) ? ((c == t ? 1.f : 0.f) + s.Mc[t * ld + c] * inv) : 0.f;
            constexpr int UNR_J = DR <= GMMVI_UKL_UNROLL_MAX ? DR : 1;
#pragma unroll UNR_J
            for (int j = DR - 1; j >= 0; --j) {
                float bj = 0.f;                                    // xr[j] (j is wave-uniform; a constant when unrolled)
#pragma unroll
                for (int c = 0; c < DR; ++c) bj = (c == j) ? xr[c] : bj;
                const float p = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bj), j));
                if (!(p > 0.f) || !(p < FLT_MAX)) { success = false; break; }
                // v_sqrt_f32 / v_rcp_f32 (1 ulp each, like the rounding of the updates around them) keep the IEEE sequences out of
                // the D-step chain
                const float d = __builtin_amdgcn_sqrtf(p);
                const float u = (t == j) ? d : (t < j ? bj * __builtin_amdgcn_rcpf(d) : 0.f);
                urow[j] = u;                                       // row t of U stays in the lane's registers
                // column j of U reaches the other lanes through v_readlane (scalar operands of the update below): no LDS
                // round trip inside the D-step dependency chain
#pragma unroll
                for (int c = 0; c < DR; ++c) {
                    if (c < j) {
                        const float ucc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), c));
                        xr[c] = (c >= t) ? fmaf(-u, ucc, xr[c]) : xr[c];
                    }
                }
            }
            UKL_WSYNC();
        }
        if (success) {
            // back substitution U x = r for r = row t of L (x = row t of L') and, in every lane, r = w (z)
            float zr[DR];
#pragma unroll
            for (int c = 0; c < DR; ++c) { xr[c] = (t < DR) ? s.L[t * ld + c] : 0.f; zr[c] = s.w[c]; }
            // row i of U is read out of lane i's registers (v_readlane: scalar operands of the chains below)
#pragma unroll
            for (int i = DR - 1; i >= 0; --i) {
                float a = xr[i], b = zr[i];
#pragma unroll
                for (int j = i + 1; j < DR; ++j) {
                    const float uij = rlane(urow[j], i);
                    a = fmaf(-uij, xr[j], a);
                    b = fmaf(-uij, zr[j], b);
                }
                const float rd = __builtin_amdgcn_rcpf(rlane(urow[i], i));
                xr[i] = a * rd;
                zr[i] = b * rd;
            }
            float acc = 0.f;
            bool bad = false;
#pragma unroll
            for (int c = 0; c < DR; ++c) {
                acc = fmaf(xr[c], zr[c], acc);
                bad |= !(xr[c] == xr[c]);
                if (c == t) bad |= !(xr[c] > 0.f);
            }
            new_mu = s.mu[t < DR ? t : 0] - acc * inv;
            bad |= !(new_mu == new_mu);
            success = (__any(bad && t < DR) == 0);                                         // :493 is_nan(new_chol)
            UKL_WSYNC();
            if (success && t < DR) {
#pragma unroll
                for (int c = 0; c < DR; ++c) {
                    s.M[t * ld + c] = xr[c];
                    Lg[t * DR + c] = (c <= t) ? xr[c] : 0.f;
                }
                mug[t] = new_mu;
            }
        }
    } else {
    if (success) {
        // ---- B = I + Mc/eta*  ->  UL factor U (upper, B = U U^T) built in M -------------------------------------------------
        for (int o = t; o < D * D; o += 64) {
            const int i = o / D, j = o % D;
            if (j >= i) s.M[i * ld + j] = (j == i ? 1.f : 0.f) + s.Mc[i * ld + j] * inv;
        }
        UKL_WSYNC();
        for (int j = D - 1; j >= 0 && success; --j) {
            float a = 0.f;
            if (t <= j) {
                a = s.M[t * ld + j];
                for (int c = j + 1; c < D; ++c) a = fmaf(-s.M[t * ld + c], s.M[j * ld + c], a);
            }
            const float p = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), j));   // j is wave-uniform
            if (!(p > 0.f) || !(p < FLT_MAX)) { success = false; break; }
            const float d = sqrtf(p);
            if (t == j) s.M[t * ld + j] = d;                  // column j only: nobody reads it in this step
            else if (t < j) s.M[t * ld + j] = a / d;
            UKL_WSYNC();
        }
    }
    if (success) {
        // ---- Uinv (upper) into Mc: lane c solves U x = e_c from the bottom, its column in registers when D is static ------
        if (DC > 0) {
            constexpr int DR = DC > 0 ? DC : 1;
            float xr[DR];
#pragma unroll
            for (int i = DR - 1; i >= 0; --i) {
                float a = (i == t) ? 1.f : 0.f;
#pragma unroll
                for (int j = i + 1; j < DR; ++j) a = fmaf(-s.M[i * ld + j], xr[j], a);    // x_j = 0 for j > t
                xr[i] = (i <= t) ? a / s.M[i * ld + i] : 0.f;
            }
            if (t < D) {
#pragma unroll
                for (int i = 0; i < DR; ++i) s.Mc[i * ld + t] = xr[i];
            }
        } else if (t < D) {
            for (int i = D - 1; i >= 0; --i) {
                float a = (i == t) ? 1.f : 0.f;
                for (int j = i + 1; j < D; ++j) a = fmaf(-s.M[i * ld + j], s.Mc[j * ld + t], a);   // Mc[j][t] = 0, j > t
                s.Mc[i * ld + t] = (i <= t) ? a / s.M[i * ld + i] : 0.f;
            }
        }
        UKL_WSYNC();
        // z = Uinv w ; y = Uinv^T z = B^-1 w
        if (t < D) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) a = fmaf(s.Mc[t * ld + c], s.w[c], a);            // Uinv[t][c] = 0 for c < t
            s.z[t] = a;
        }
        UKL_WSYNC();
        if (t < D) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) a = fmaf(s.Mc[c * ld + t], s.z[c], a);            // Uinv[c][t] = 0 for c > t
            s.y[t] = a;
        }
        UKL_WSYNC();
        // new mean mu' = mu - L y / eta* ; new factor L' = L Uinv^T, its D(D+1)/2 entries spread over the 64 lanes,
        // written to M (U no longer needed)
        float new_mu = 0.f;
        bool bad = false;
        if (t < D) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) a = fmaf(s.L[t * ld + c], s.y[c], a);             // L[t][c] = 0 for c > t
            new_mu = s.mu[t] - a * inv;
            bad = !(new_mu == new_mu);
        }
        UKL_WSYNC();
        for (int o = t; o < D * D; o += 64) {
            const int i = o / D, j = o % D;
            if (j <= i) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) a = fmaf(s.L[i * ld + c], s.Mc[j * ld + c], a);
                s.M[i * ld + j] = a;
                bad |= !(a == a);
                if (i == j) bad |= !(a > 0.f);
            }
        }
        success = (__any(bad) == 0);                                                       // :493 is_nan(new_chol)
        UKL_WSYNC();
        if (success) {
            for (int e = t; e < D * D; e += 64) {
                const int i = e / D, j = e % D;
                Lg[e] = (j <= i) ? s.M[i * ld + j] : 0.f;
            }
            if (t < D) mug[t] = new_mu;
        }
    }
    }
    UKL_STAMP(4);                                      // new factor done
    success_w = success; eta_w = eta_star; kl_w = kl_val; probes_w = probes;
    }
    __shared__ int sh_success;
    if (t == 0) sh_success = success_w ? 1 : 0;
    __syncthreads();
    const bool success = sh_success != 0;
    if (packed_out != nullptr) {
        // packed parameter block of the (new or kept) component for the density kernels: layout of common.h Pack<DP>
        __syncthreads();
        const float* Lf = success ? s.M : s.L;                      // final factor (lower triangle), stride ld
        const float mu_f = (t < D) ? (success ? mug[t] : s.mu[t]) : 0.f;
        const int T = DPk * (DPk - 1) / 2;
        const PackDims pd = gmmvi_pack_dims(DPk);
        const int stride = pd.stride;
        float* out = packed_out + (size_t)k * stride;
        for (int i = t; i < DPk; i += NTH) {
            out[i] = (i < D) ? mu_f : 0.f;
            out[DPk + i] = (i < D) ? 1.f / Lf[i * ld + i] : 1.f;
        }
        for (int e = t; e < DPk * DPk; e += NTH) {
            const int i = e / DPk, j = e % DPk;
            if (j < i) {
                const float v = (i < D) ? Lf[i * ld + j] : 0.f;
                out[2 * DPk + i * (i - 1) / 2 + j] = v;
                out[2 * DPk + T + j * (DPk - 1) - j * (j - 1) / 2 + (i - j - 1)] = v;
            }
        }
        const float lsum = wsum((t < D) ? __logf(Lf[t * ld + t]) : 0.f);
        if (t == 0) {
            out[2 * DPk + 2 * T] = -lsum - 0.5f * D * 1.8378770664093453f;
            for (int i = 2 * DPk + 2 * T + 1; i < pd.fwd; ++i) out[i] = 0.f;
            if (pd.swh >= 0) out[pd.swh + DPk] = -lsum - 0.5f * D * 1.8378770664093453f;
        }
        gmmvi_write_sweep_stream(out, DPk, D, Lf, ld, success ? (const float*)mug : (const float*)s.mu, t, NTH);
        UKL_STAMP(5);                                  // packed block (without fragments) done
        // L^-1 of the final factor for the matrix-core fragments of the block: lane t solves L x = e_t (column t) from the
        // LDS image of L (every lane reads the same element: broadcast), the dense inverse goes through Mc
        __syncthreads();
        constexpr int LA = DC > 0 ? ((DC + 3) / 4) * 4 : 4;          // static instances: L again with 16-byte aligned rows
        float* La = s.pr;                                           // (the probe scratch is idle; broadcast ds_read_b128)
        if constexpr (DC > 0) {
            if (pd.nf_total > 0) {
                for (int e = t; e < DC * LA; e += NTH) {
                    const int i = e / LA, j = e - i * LA;
                    La[e] = (j <= i) ? Lf[i * ld + j] : 0.f;
                }
            }
            __syncthreads();
        }
        if constexpr (DC >= 32 && NW >= 2) {
            // 2 x 2 blocks, L = [[L11, 0], [L21, L22]] with H = 4 (D / 8) leading rows: X11 = L11^-1 on wave 0 and X22 = L22^-1 on
            // wave 1 at the same time (lane = column; chains of H (H - 1) / 2 steps instead of D (D - 1) / 2), then
            // X21 = -X22 (L21 X11) as two small products on all threads
            constexpr int H = 4 * (DC / 8), H2 = DC - H;
            float* Tm = La + DC * LA;                                   // [H2][H] scratch behind the aligned image
            const int wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;
            if (pd.nf_total > 0) {
                for (int e = t; e < DC * DC; e += NTH) s.Mc[(e / DC) * ld + (e % DC)] = 0.f;      // upper right block stays zero
                __syncthreads();
                if (wv == 0 && ln < H) {
                    float x[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        float lr[LA];
#pragma unroll
                        for (int q4 = 0; q4 <= i / 4; ++q4) {
                            const float4 v4 = reinterpret_cast<const float4*>(La + i * LA)[q4];
                            lr[4 * q4] = v4.x; lr[4 * q4 + 1] = v4.y; lr[4 * q4 + 2] = v4.z; lr[4 * q4 + 3] = v4.w;
                        }
                        float a = (i == ln) ? 1.f : 0.f;
#pragma unroll
                        for (int j = 0; j < i; ++j) a = fmaf(-lr[j], x[j], a);
                        x[i] = (i >= ln) ? a / lr[i] : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < H; ++i) s.Mc[i * ld + ln] = x[i];
                } else if (wv == 1 && ln < H2) {
                    float x[H2];
#pragma unroll
                    for (int i = 0; i < H2; ++i) {
                        float lr[LA];
#pragma unroll
                        for (int q4 = H / 4; q4 <= (H + i) / 4; ++q4) {                   // H is a multiple of 4: aligned reads
                            const float4 v4 = reinterpret_cast<const float4*>(La + (H + i) * LA)[q4];
                            lr[4 * q4] = v4.x; lr[4 * q4 + 1] = v4.y; lr[4 * q4 + 2] = v4.z; lr[4 * q4 + 3] = v4.w;
                        }
                        float a = (i == ln) ? 1.f : 0.f;
#pragma unroll
                        for (int j = 0; j < i; ++j) a = fmaf(-lr[H + j], x[j], a);
                        x[i] = (i >= ln) ? a / lr[H + i] : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < H2; ++i) s.Mc[(H + i) * ld + H + ln] = x[i];
                }
                __syncthreads();
                // T = L21 X11  ([H2][H]; X11[m][j] = 0 for m < j)
                for (int o = t; o < H2 * H; o += NTH) {
                    const int i = o / H, j = o - i * H;
                    float a = 0.f;
#pragma unroll
                    for (int m2 = 0; m2 < H; ++m2) a = fmaf(La[(H + i) * LA + m2], s.Mc[m2 * ld + j], a);
                    Tm[o] = a;
                }
                __syncthreads();
                // X21 = -X22 T  (X22[i][m] = 0 for m > i)
                for (int o = t; o < H2 * H; o += NTH) {
                    const int i = o / H, j = o - i * H;
                    float a = 0.f;
#pragma unroll
                    for (int m2 = 0; m2 < H2; ++m2) a = fmaf(s.Mc[(H + i) * ld + H + m2], Tm[m2 * H + j], a);
                    s.Mc[(H + i) * ld + j] = -a;
                }
            }
        } else
        if (pd.nf_total > 0 && t < D) {
            if constexpr (DC > 0) {
                float x[DC > 0 ? DC : 1];
#pragma unroll
                for (int i = 0; i < DC; ++i) {
                    float lr[LA];
#pragma unroll
                    for (int q4 = 0; q4 <= i / 4; ++q4) {
                        const float4 v4 = reinterpret_cast<const float4*>(La + i * LA)[q4];
                        lr[4 * q4] = v4.x; lr[4 * q4 + 1] = v4.y; lr[4 * q4 + 2] = v4.z; lr[4 * q4 + 3] = v4.w;
                    }
                    float a = (i == t) ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < i; ++j) a = fmaf(-lr[j], x[j], a);                // x_j = 0 for j < t
                    x[i] = (i >= t) ? a / lr[i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < DC; ++i) s.Mc[i * ld + t] = x[i];
            } else {
                for (int i = 0; i < D; ++i) {
                    float a = (i == t) ? 1.f : 0.f;
                    for (int j = t; j < i; ++j) a = fmaf(-Lf[i * ld + j], s.Mc[j * ld + t], a);
                    s.Mc[i * ld + t] = (i >= t) ? a / Lf[i * ld + i] : 0.f;
                }
            }
        }
        __syncthreads();
        if (pd.nf_total > 0) gmmvi_write_inverse_fragments(out, DPk, D, s.Mc, ld, t, NTH);
    }
#ifdef GMMVI_UKL_STAMPS
    UKL_STAMP(6);
    if (k == 0 && t == 0 && kl_out)
        for (int i = 1; i <= 6; ++i) kl_out[i] = (float)(stamp[i] - stamp[0]);
    if (k == 0 && t == 0 && !kl_out && slab.part)       // single-call iteration: no info array, print
        printf("update_kl stamps (10 ns): slabsum %lld P %lld rest-of-front %lld | front %lld householder %lld search %lld factor %lld pack %lld end %lld\n",
               stamp[7] - stamp[0], stamp[8] - stamp[7], stamp[1] - stamp[8], stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4],
               stamp[6] - stamp[5]);
    kl_out = nullptr;
#endif
    if (t == 0) {
        last_eta[k] = success ? eta_w : -1.f;                                              // :504,:511,:524
        if (kl_out) kl_out[k] = success ? kl_w : -1.f;
        if (nprobes_out) nprobes_out[k] = probes_w;
        const float old = l2[k];
        l2[k] = success ? fmaxf(0.5f * old, l2_init) : fminf(1e-6f, 10.f * old);           // :520-523 (min on failure)
        num_updates[k] += 1.f;                                                             // :519
        if (success_out) success_out[k] = success ? 1 : 0;
    }
}

}  // namespace

static int update_kl_launch(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev, float* H_neg_dev, float* g_neg_dev,
                            const SteinSlab& slab, int slab_N, int slab_flags, const float* packed_old_dev,
                            const float* stepsizes_dev, float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                            float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                            int32_t* n_probes_out_dev, float* packed_out_dev) {
    size_t shmem = lds_bytes(D);
    if (slab.part != nullptr) {
        const size_t fin = stein_finalize_lds_floats(gmmvi_padded_dim(D), D, slab.R) * sizeof(float);
        if (fin > shmem) shmem = fin;
    }
    GMMVI_PROF(ctx, "update_kl");
#define GMMVI_UKL(DCV, NWV)                                                                                        \
    do {                                                                                                           \
        if (shmem > 64 * 1024)                                                                                     \
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)update_kl_fast_kernel<DCV, NWV>,                 \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));     \
        hipLaunchKernelGGL((update_kl_fast_kernel<DCV, NWV>), dim3(K), dim3(64 * NWV), shmem, ctx->stream, D,      \
                           means_dev, chols_dev, H_neg_dev, g_neg_dev, slab, slab_N, slab_flags, packed_old_dev,   \
                           stepsizes_dev, temperature, l2_init, last_eta_dev, l2_dev, num_received_updates_dev,    \
                           success_out_dev, kl_out_dev, n_probes_out_dev, packed_out_dev, gmmvi_padded_dim(D));    \
    } while (0)
    // dimensions of the BASELINE configurations get unrolled instances; from D = 32 four wavefronts share the D^3 products,
    // the load / pack phases and the fragment writes (the chains in between stay on one)
    const bool pro = slab.part != nullptr;             // with the Stein prologue: four wavefronts (its slab sum is parallel work)
    switch (D) {
        case 4: if (pro) GMMVI_UKL(4, 4); else GMMVI_UKL(4, 1); break;
        case 10: if (pro) GMMVI_UKL(10, 4); else GMMVI_UKL(10, 1); break;
        case 20: if (pro) GMMVI_UKL(20, 4); else GMMVI_UKL(20, 1); break;
        case 32: GMMVI_UKL(32, 4); break;
        case 40: GMMVI_UKL(40, 4); break;
        case 50: GMMVI_UKL(50, 4); break;
        default:
            if (D > 24) GMMVI_UKL(0, 4); else GMMVI_UKL(0, 1);
            break;
    }
#undef GMMVI_UKL
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

extern "C" int gmmvi_update_components_kl(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                                          const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                          float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                                          float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                                          int32_t* n_probes_out_dev, float* packed_out_dev) {
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D <= GMMVI_BLOCKED_MAX_DIM);
    GMMVI_ARG_CHECK(ctx, means_dev && chols_dev && H_neg_dev && g_neg_dev && stepsizes_dev && last_eta_dev && l2_dev &&
                             num_received_updates_dev);
    if (gmmvi_is_blocked_dim(D))
        return gmmvi_blocked_update_kl(ctx, K, D, means_dev, chols_dev, H_neg_dev, g_neg_dev, stepsizes_dev, temperature, l2_init,
                                       last_eta_dev, l2_dev, num_received_updates_dev, success_out_dev, kl_out_dev,
                                       n_probes_out_dev, packed_out_dev);
    // (the estimate is only read: the kernel takes non-const pointers because its single-call form writes them first)
    return update_kl_launch(ctx, K, D, means_dev, chols_dev, const_cast<float*>(H_neg_dev), const_cast<float*>(g_neg_dev),
                            SteinSlab{nullptr, nullptr, 0}, 0, 0, nullptr, stepsizes_dev, temperature, l2_init, last_eta_dev, l2_dev,
                            num_received_updates_dev, success_out_dev, kl_out_dev, n_probes_out_dev, packed_out_dev);
}

// C++ linkage (common.h), single-call iteration: the Stein estimate is finished from its partial slab and the components are
// updated -- in ONE launch where the update kernel carries the prologue (the unrolled single-wave instances D = 4 / 10 / 20),
// by the stand-alone finalize launch followed by the update otherwise.  H_neg / g_neg receive the estimate on the explicit routes
// (plain importance weights, GMMVI_EXPLICIT_ESTIMATE, the finalize launch) ONLY: on the default direct route (self-normalised
// weights, D = 4 / 10 / 20 / 32 / 40 / 50) the kernel whitens the moment sums itself and never writes them.
int gmmvi_update_components_kl_from_slab(gmmvi_ctx* ctx, int K, int D, const SteinSlab& slab, int N, int stein_flags,
                                         const float* packed_old_dev, float* H_neg_dev, float* g_neg_dev, float* means_dev,
                                         float* chols_dev, const float* stepsizes_dev, float temperature, float l2_init,
                                         float* last_eta_dev, float* l2_dev, float* num_received_updates_dev,
                                         int32_t* success_out_dev, float* packed_out_dev) {
    // one launch: the instances that carry the finalize prologue (D = 4 / 10 / 20), and every unrolled instance on the direct
    // route (self-normalised weights, estimate not requested explicitly, the moment matrix fits the probe scratch)
    const bool direct = (stein_flags & GMMVI_SELF_NORMALIZED) != 0 && (stein_flags & GMMVI_EXPLICIT_ESTIMATE) == 0 &&
                        (size_t)(D + 1) * (D + 1) + slab.R + 4 + (D >= GMMVI_MFMA_DENSITY_FROM_DP ? (size_t)D * D : 0) <= (size_t)2 * D * 64;
    const bool fused = !gmmvi_is_blocked_dim(D) &&
                       ((D == 4 || D == 10 || D == 20) || ((D == 32 || D == 40 || D == 50) && direct));
    if (!fused) {
        int rc = gmmvi_stein_finalize_slab(ctx, K, D, slab, N, stein_flags, packed_old_dev, H_neg_dev, g_neg_dev);
        if (rc != GMMVI_OK) return rc;
        return gmmvi_update_components_kl(ctx, K, D, means_dev, chols_dev, H_neg_dev, g_neg_dev, stepsizes_dev, temperature,
                                          l2_init, last_eta_dev, l2_dev, num_received_updates_dev, success_out_dev, nullptr,
                                          nullptr, packed_out_dev);
    }
    return update_kl_launch(ctx, K, D, means_dev, chols_dev, H_neg_dev, g_neg_dev, slab, N, stein_flags, packed_old_dev,
                            stepsizes_dev, temperature, l2_init, last_eta_dev, l2_dev, num_received_updates_dev, success_out_dev,
                            nullptr, nullptr, packed_out_dev);
}
