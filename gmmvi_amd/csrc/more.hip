// MORE natural-gradient estimate (reference: optimization/gmmvi_modules/ng_estimator.py:266-376 with the quadratic
// ridge regression of optimization/least_squares.py:34-76,103-191), gfx950.
//
// Per component o the reference fits  R~(x) = x^T R x + x^T r + r0  to the rewards log p~(x) - log q(x) by importance-
// weighted ridge regression on whitened samples z = L_o^-1 (x - mu_o) with features
//     phi(z) = [ z_i z_j (i <= j, row-major upper triangle),  z,  1 ]          F = D(D+1)/2 + D + 1
// i.e. it solves  (Phi^T W Phi + lambda I') beta = Phi^T W rew   (bias row of I' is zero), then un-whitens.
//
//   more_lse_kernel     per component: log sum_n exp(ld[o,n] - bg[n])  (self-normalised weights, :353-356)
//   more_gram_kernel    the (F+1)x(F+1) Gram matrix of the rows  sqrt(w_n) [phi(z_n); rew_n]  -- one fp64-MFMA SYRK per
//                       component: A = Phi^T W Phi, b = Phi^T W rew and sum w rew^2 in one contraction.  Workgroup =
//                       (component, sample chunk), 8 waves; per 64-sample tile every wave substitutes z (SGPR-fed L), the
//                       waves write the weighted fp32 feature rows into one LDS image [F+1][64] and contract it with
//                       v_mfma_f64_16x16x4_f64 (rows widened exactly to fp64), each wave owning up to PP lower-triangular
//                       16x16 tile pairs whose accumulators stay in registers over the whole chunk.  Operands are read as
//                       ds_read_b128 (four k-steps per read).
//   more_solve_kernel   per component, fp64: sums the chunk partials in fixed order into a register-resident lower
//                       triangle, adds the ridge, Cholesky-factorises (1024 threads, one barrier per column), back-
//                       substitutes, maps the coefficients back (least_squares.py:177-189) and emits
//                       H = L^-T Q_w L^-1  and  g = Q mu - lin = -L^-T lin_w.
//
// Why fp64: the reference's ridge (1e-12) is far below fp32 resolution.  With fewer effective samples than features
// (the usual state early in a run: 100 samples per component, F = 231 at D = 20) the fp32 normal equations are
// numerically singular -- an fp32 LU/Cholesky returns noise-dominated coefficients (measured 10x the fp64 answer) or a
// negative pivot -- while the Gram matrix of the fp32 feature rows accumulated in fp64 keeps its exact rank and the ridge
// solution is the one the fp64 oracle computes.  MI355X runs fp64 MFMA at half the f32 rate, so this costs ~2x on the
// contraction, not 10x+.
//
// Sizes: the (F+1)x(F+1) triangle is held by one 1024-thread workgroup (36 doubles per thread): F + 1 <= 256  <=>
// D <= 21.  The ridge system is solved by Cholesky (the reference calls tf.linalg.solve = pivoted LU); a non-positive
// pivot marks the component's estimate as NaN, which the component updaters treat as a rejected update.
#include "common.h"
#include "subst.h"
#include "blocked.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

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

// D[i][j] of v_mfma_f64_16x16x4_f64 on gfx950: lane l, register r  ->  i = 4 r + l / 16, j = l % 16 (probed on the
// hardware: tools/probe/mfma_f64_layout.hip); operands A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16].
template <int DP, int PP>
__global__ __launch_bounds__(512) void more_gram_kernel(int D, int N, int tiles_per_chunk, int nb,
                                                        const float* __restrict__ packed, const float* __restrict__ X,
                                                        const float* __restrict__ ld, const float* __restrict__ bg,
                                                        const float* __restrict__ tlp, const float* __restrict__ logq,
                                                        const int32_t* __restrict__ mapping, int map_offset, int flags,
                                                        const float* __restrict__ lse, double* __restrict__ slab) {
    using PK = Pack<DP>;
    extern __shared__ float phi[];                     // [16 nb][PHI_LD] feature image, then 8 whitened tiles, then tab
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
    const int ZS_TILE = (D + 3) * 64;                  // per tile: rows 0..D-1 z, row D ones, D+1 reward, D+2 sqrt(weight)
    float* zs = phi + 16 * nb * PHI_LD;                // 8 tiles, whitened one per wave
    int* tab = reinterpret_cast<int*>(zs + 8 * ZS_TILE);   // feature f -> (row ia) | (row ib) << 16 of a zs tile

    // 16x16 tile pairs (bi >= bc) of this wave: p = wave + 8 pp; unused slots read tile 0 and are never stored
    int row_of[PP], col_of[PP];
    bool used[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int p = wave + 8 * pp;
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
        used[pp] = p < n_pairs;
        row_of[pp] = used[pp] ? bi : 0;
        col_of[pp] = used[pp] ? p - bi * (bi + 1) / 2 : 0;
    }
    f64x4 acc[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) acc[pp] = f64x4{0.0, 0.0, 0.0, 0.0};

    for (int e = tid; e < 16 * nb * PHI_LD; e += 512) phi[e] = 0.f;       // rows > F stay zero for the whole kernel
    for (int f = tid; f <= F; f += 512) {                                 // least_squares.py:113-124 feature order
        int ia, ib;
        if (f < T2) {
            int i = 0, rem = f;
            while (rem >= D - i) { rem -= D - i; ++i; }
            ia = i; ib = i + rem;
        } else if (f < T2 + D) { ia = f - T2; ib = D; }
        else if (f == F - 1) { ia = D; ib = D; }
        else { ia = D + 1; ib = D; }
        tab[f] = ia | (ib << 16);
    }
    __syncthreads();

    const int r16 = lane & 15, kg = lane >> 4;
    const int tile_begin = chunk * tiles_per_chunk;
    const int tile_end = min((N + 63) / 64, tile_begin + tiles_per_chunk);
    for (int t0 = tile_begin; t0 < tile_end; t0 += 8) {
        // ---- wave w whitens the 64 samples of tile t0 + w (lane = sample) -----------------------------------------------
        if (t0 + wave < tile_end) {
            float* zw = zs + wave * ZS_TILE;
            const int n = (t0 + wave) * 64 + lane;
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
                if (a > -3.0e38f) sw = __expf(0.5f * (a - lse_k));       // sqrt of the importance weight (:353-358)
                rew = tlp[n] - logq[n];                                  // ng_estimator.py:346
            }
            const bool live = sw > 0.f;
#pragma unroll
            for (int i = 0; i < DP; ++i)
                if (i < D) zw[i * 64 + lane] = live ? z[i] : 0.f;
            zw[D * 64 + lane] = 1.f;
            zw[(D + 1) * 64 + lane] = live ? rew : 0.f;
            zw[(D + 2) * 64 + lane] = live ? sw : 0.f;
        }
        __syncthreads();
      for (int u = 0; u < 8 && t0 + u < tile_end; ++u) {
        // ---- all waves: weighted feature rows  phi[f][n] = sw_n * zs[ia][n] * zs[ib][n] ------------------------------------
        {
            const float* zu = zs + u * ZS_TILE;
            const int n = tid & 63;
            const float swn = zu[(D + 2) * 64 + n];
#pragma unroll 4
            for (int f = tid >> 6; f <= F; f += 8) {
                const int t = tab[f];
                phi[f * PHI_LD + n] = (swn * zu[(t & 0xffff) * 64 + n]) * zu[(t >> 16) * 64 + n];
            }
        }
        __syncthreads();
        // ---- contraction over the 64 samples: fp32 rows widened (exactly) to fp64, fp64 accumulation --------------------
#pragma unroll
        for (int pp = 0; pp < PP; pp += 2) {
            constexpr int NQ = 2;
            float4 av[NQ][4], bv[NQ][4];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int pu = (pp + u < PP) ? pp + u : pp;
                const float* pa = phi + (16 * row_of[pu] + r16) * PHI_LD + 4 * kg;
                const float* pb = phi + (16 * col_of[pu] + r16) * PHI_LD + 4 * kg;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    av[u][q] = *reinterpret_cast<const float4*>(pa + 16 * q);
                    bv[u][q] = *reinterpret_cast<const float4*>(pb + 16 * q);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        if (pp + u < PP) {
                            const float a = t == 0 ? av[u][q].x : (t == 1 ? av[u][q].y : (t == 2 ? av[u][q].z : av[u][q].w));
                            const float b = t == 0 ? bv[u][q].x : (t == 1 ? bv[u][q].y : (t == 2 ? bv[u][q].z : bv[u][q].w));
                            acc[pp + u] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, (double)b, acc[pp + u], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
      }
    }
    // ---- partial Gram tiles of this chunk: slab[k][chunk][pair][j][i]  (column-major inside a tile) -------------------
    double* out = slab + ((size_t)k * n_chunks + chunk) * (size_t)n_pairs * 256;
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        if (!used[pp]) continue;
        double* o = out + (size_t)(wave + 8 * pp) * 256 + r16 * 16 + kg;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * r] = acc[pp][r];
    }
}

__device__ __forceinline__ double half_wave_sum(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}

// One workgroup per component: assemble the (F+1)x(F+1) Gram matrix, factorise, solve, un-whiten -- in fp64.
// The lower triangle lives in REGISTERS: thread (ti = tid % 32, tc = tid / 32) owns the elements (32 a + ti, 32 b + tc),
// 0 <= b <= a < 8 (36 doubles).  Right-looking Cholesky, one barrier per column: the owners of column j publish it
// (unscaled) and 1/pivot through a double-buffered LDS vector, every thread applies the rank-1 update to its elements.
// Row F of the matrix is the right-hand side, so the factorisation leaves y = L^-1 b in it; the back substitution walks
// the columns with a half-wave reduction (the owners of one column are 32 consecutive lanes).
__global__ __launch_bounds__(1024) void more_solve_kernel(int D, int nb, int n_chunks, const double* __restrict__ slab,
                                                          const float* __restrict__ chols, const float* __restrict__ l2,
                                                          float* __restrict__ H_neg, float* __restrict__ g_neg) {
    __shared__ double cb[2][256];       // published column (unscaled), double-buffered
    __shared__ double ys[256], dg[256], beta[256];   // y = L^-1 b, reciprocal diagonal, solution
    __shared__ double rblk[32], dblk[32 * 33];       // back substitution: block right-hand side, diagonal block
    extern __shared__ double smd[];     // Ls[D][D], Qs[D][D+1], Xs[D][D+1]
    const int tid = threadIdx.x;
    const int ti = tid & 31, tc = tid >> 5;
    const int k = blockIdx.x;
    const int T2 = D * (D + 1) / 2;
    const int F = T2 + D + 1;
    const int n_pairs = nb * (nb + 1) / 2;
    double* Ls = smd;
    double* Qs = Ls + D * D;
    double* Xs = Qs + D * (D + 1);

    // ---- assemble: fixed-order sum over the sample chunks ------------------------------------------------------------
    const double* base = slab + (size_t)k * n_chunks * (size_t)n_pairs * 256;
    const double ridge = (double)l2[k];
    double L[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int i = 32 * a + ti, c = 32 * b + tc;
            L[a][b] = (i == c && i < F - 1) ? ridge : 0.0;                  // least_squares.py:71-73 (bias unregularised)
        }
    // element (i, c) sits in tile (i / 16, c / 16) at [c % 16][i % 16]; with i = 32 a + ti, c = 32 b + tc that is tile
    // (2 a + ti / 16, 2 b + tc / 16): a per-thread base offset plus compile-time multiples
    const int th = ti >> 4, tch = tc >> 4;
    const int in_tile = (tc & 15) * 16 + (ti & 15);
    for (int ch = 0; ch < n_chunks; ++ch) {                                 // fixed order; independent loads in flight
        const double* src = base + (size_t)ch * n_pairs * 256 + in_tile;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const int i = 32 * a + ti, c = 32 * b + tc;
                const int I = 2 * a + th, J = 2 * b + tch;
                if (i <= F && c <= i && c < F) L[a][b] += src[(I * (I + 1) / 2 + J) * 256];
            }
    }
    for (int e = tid; e < D * D; e += 1024) Ls[e] = (double)chols[(size_t)k * D * D + e];

    // ---- Cholesky of the leading F x F block (row F rides along) -----------------------------------------------------
    // Per column j: [barrier] every thread reads the published column u = A[:, j] (unscaled) and applies
    // A[i][c] -= u_i u_c / u_j to its elements; the owners of column j+1 do their column first and publish it at once
    // (look-ahead), so the next barrier does not wait for a serial publish.  Elements above the diagonal of the diagonal
    // blocks are never read; they hold finite Schur-complement mirrors, so no predicate is needed for them.
    bool fail = false;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        if (32 * jb >= F || fail) break;
        if (tc == 0) {
            double* col0 = cb[0];                                            // 32 jb is even
#pragma unroll
            for (int a = jb; a < 8; ++a) col0[32 * a + ti] = L[a][jb];
        }
        for (int jj = 0; jj < 32; ++jj) {
            const int j = 32 * jb + jj;
            if (j >= F) break;
            const double* col = cb[j & 1];
            __syncthreads();
            const double d = col[j];
            if (!(d > 0.0)) { fail = true; break; }                          // uniform: every thread reads the same pivot
            double inv = __builtin_amdgcn_rcp(d);
            inv = fma(fma(-d, inv, 1.0), inv, inv);
            inv = fma(fma(-d, inv, 1.0), inv, inv);
            double ui[8];
#pragma unroll
            for (int a = jb; a < 8; ++a) ui[a] = col[32 * a + ti];
            const double uc0 = col[32 * jb + tc] * inv;
            const bool look = jj < 31 && j + 1 < F;
            if (look && tc == jj + 1) {
                double* nxt = cb[(j + 1) & 1];
#pragma unroll
                for (int a = jb; a < 8; ++a) {
                    L[a][jb] = fma(-ui[a], uc0, L[a][jb]);
                    nxt[32 * a + ti] = L[a][jb];
                }
            }
            const double ucj = (tc > jj + (look ? 1 : 0)) ? uc0 : 0.0;         // columns <= j are final, j+1 is done
#pragma unroll
            for (int a = jb; a < 8; ++a) L[a][jb] = fma(-ui[a], ucj, L[a][jb]);
#pragma unroll
            for (int b = jb + 1; b < 8; ++b) {
                const double ucb = col[32 * b + tc] * inv;
#pragma unroll
                for (int a = b; a < 8; ++a) L[a][b] = fma(-ui[a], ucb, L[a][b]);
            }
            if (tc == jj) {                                                  // finalise column j: divide by sqrt(pivot)
                const double rs = 1.0 / sqrt(d);
#pragma unroll
                for (int a = jb; a < 8; ++a) L[a][jb] *= rs;
            }
        }
    }
    __syncthreads();
    if (!fail) {
        // y = row F and the reciprocal diagonal -> LDS
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (ti == tc) dg[32 * a + ti] = 1.0 / L[a][a];
            if (32 * a + ti == F) {
#pragma unroll
                for (int b = 0; b <= a; ++b) ys[32 * b + tc] = L[a][b];
            }
        }
        __syncthreads();
        // back substitution L^T beta = y, one 32-column block per round: every half-wave reduces its column's product
        // with the already known beta, the diagonal block goes to LDS and wave 0 finishes the 32 unknowns without barriers
#pragma unroll
        for (int jb = 7; jb >= 0; --jb) {
            if (32 * jb >= F) continue;
            double part = 0.0;
#pragma unroll
            for (int a = jb + 1; a < 8; ++a) {
                const int i = 32 * a + ti;
                if (i < F) part = fma(L[a][jb], beta[i], part);
            }
            part = half_wave_sum(part);
            if (ti == 0) rblk[tc] = ys[32 * jb + tc] - part;
            dblk[ti * 33 + tc] = L[jb][jb];
            __syncthreads();
            if (tid < 64) {
                const int c = tid & 31;
                double rc = rblk[c];
                for (int jj = 31; jj >= 0; --jj) {
                    const int j = 32 * jb + jj;
                    if (j >= F) continue;
                    const double bj = __shfl(rc, jj) * dg[j];
                    if (tid == jj) beta[j] = bj;
                    if (c < jj) rc = fma(-dblk[jj * 33 + c], bj, rc);
                }
            }
            __syncthreads();
        }
        // Q_w = -(Qt + Qt^T) with Qt the upper-triangular fill of the quadratic coefficients (least_squares.py:177-179)
        for (int e = tid; e < D * (D + 1); e += 1024) {
            const int i = e / (D + 1), j = e % (D + 1);
            double v;
            if (j == D) v = beta[T2 + i];                                            // lin_w
            else {
                const int a = min(i, j), b = max(i, j);
                const double q = beta[a * D - a * (a - 1) / 2 + (b - a)];
                v = (a == b) ? -2.0 * q : -q;
            }
            Qs[e] = v;
        }
        __syncthreads();
        // X = L_o^-T [Q_w | lin_w]   (lane = column)
        if (tid <= D) {
            for (int i = D - 1; i >= 0; --i) {
                double t = Qs[i * (D + 1) + tid];
                for (int j = i + 1; j < D; ++j) t = fma(-Ls[j * D + i], Xs[j * (D + 1) + tid], t);
                Xs[i * (D + 1) + tid] = t / Ls[i * D + i];
            }
        }
        __syncthreads();
        // H = X L_o^-1: row r of H solves L_o^T h = (row r of X)^T (least_squares.py:185); g = Q mu - lin = -L_o^-T lin_w
        // (:186-188, ng_estimator.py:371-373)
        if (tid < D) {
            double* hrow = Qs + tid * (D + 1);                                       // row tid is only touched by this lane
            for (int i = D - 1; i >= 0; --i) {
                double t = Xs[tid * (D + 1) + i];
                for (int j = i + 1; j < D; ++j) t = fma(-Ls[j * D + i], hrow[j], t);
                hrow[i] = t / Ls[i * D + i];
            }
            for (int i = 0; i < D; ++i) H_neg[((size_t)k * D + tid) * D + i] = (float)hrow[i];
            g_neg[(size_t)k * D + tid] = (float)(-Xs[tid * (D + 1) + D]);
        }
    } else {
        const float nanv = __int_as_float(0x7fc00000);
        for (int e = tid; e < D * D; e += 1024) H_neg[(size_t)k * D * D + e] = nanv;
        for (int e = tid; e < D; e += 1024) g_neg[(size_t)k * D + e] = nanv;
    }
}

// =====================================================================================================================
// Large systems (D > 21: F + 1 > 256, up to 2 081 at D = 63).  The Gram matrix no longer fits one workgroup's registers:
//   more_gram_big_kernel    workgroup = (128 x 128 block of the lower triangle of G, component); it runs over ALL samples
//                           (fixed order, no partial slabs), builds only the <= 256 feature rows its block needs in the LDS
//                           image and contracts them with v_mfma_f64_16x16x4_f64, 8 tile pairs per wave; G (fp64, row stride
//                           LDG = 128 ceil((F+1)/128)) is written straight into global memory.
//   more_solve_big_kernel   one 1024-thread workgroup per component, fp64, G in global memory (L2-resident): ridge, blocked
//                           right-looking Cholesky (32-column panels: diagonal block by one wave in LDS, panel rows by one
//                           thread each, trailing update on 128 x 128 tiles staged in LDS with 4 x 4 outputs per thread), the
//                           right-hand side rides along as row F; blocked back substitution; un-whitening as the small kernel.
// =====================================================================================================================
template <int DP>
__global__ __launch_bounds__(512) void more_gram_big_kernel(int D, int N, int nblk, int LDG, const float* __restrict__ packed,
                                                            const float* __restrict__ X, const float* __restrict__ ld,
                                                            const float* __restrict__ bg, const float* __restrict__ tlp,
                                                            const float* __restrict__ logq, const int32_t* __restrict__ mapping,
                                                            int map_offset, int flags, const float* __restrict__ lse,
                                                            double* __restrict__ G) {
    using PK = Pack<DP>;
    extern __shared__ float phi[];                     // [256][PHI_LD] feature rows of the two blocks, then 4 whitened tiles, tab
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = blockIdx.y;
    int BI = 0;
    while ((BI + 1) * (BI + 2) / 2 <= (int)blockIdx.x) ++BI;
    const int BC = (int)blockIdx.x - BI * (BI + 1) / 2;
    const bool diag = BI == BC;
    const int T2 = D * (D + 1) / 2;
    const int F = T2 + D + 1;                          // features; row F carries the reward
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    const bool self_norm = (flags & GMMVI_SELF_NORMALIZED) != 0;
    const float* __restrict__ P = packed + (size_t)k * PK::STRIDE;
    const float lse_k = self_norm ? lse[k] : 0.f;
    const int ZS_TILE = (D + 3) * 64;                  // per tile: rows 0..D-1 z, row D ones, D+1 reward, D+2 sqrt(weight)
    float* zs = phi + 256 * PHI_LD;                    // 4 tiles
    int* tab = reinterpret_cast<int*>(zs + 4 * ZS_TILE);   // local row (0..255) -> (row ia) | (row ib) << 16 of a zs tile, or -1
    const int n_rows = diag ? 128 : 256;
    for (int r = tid; r < 256; r += 512) {
        const int f = 128 * (r < 128 ? BI : BC) + (r & 127);
        int code = -1;
        if (r < n_rows && f <= F) {                                     // least_squares.py:113-124 feature order
            int ia, ib;
            if (f < T2) {
                int i = 0, rem = f;
                while (rem >= D - i) { rem -= D - i; ++i; }
                ia = i; ib = i + rem;
            } else if (f < T2 + D) { ia = f - T2; ib = D; }
            else if (f == F - 1) { ia = D; ib = D; }
            else { ia = D + 1; ib = D; }
            code = ia | (ib << 16);
        }
        tab[r] = code;
    }
    for (int e = tid; e < 256 * PHI_LD; e += 512) phi[e] = 0.f;         // rows without a feature stay zero
    __syncthreads();

    // wave w owns the tile pairs q = w + 8 pp: row tile q / 8 of block BI, column tile q % 8 of block BC
    f64x4 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) acc[pp] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int r16 = lane & 15, kg = lane >> 4;
    const int col_base = diag ? 0 : 128;
    const int n_tiles = (N + 63) / 64;
    for (int t0 = 0; t0 < n_tiles; t0 += 4) {
        if (wave < 4 && t0 + wave < n_tiles) {                          // waves 0-3 whiten one tile each (lane = sample)
            float* zw = zs + wave * ZS_TILE;
            const int n = (t0 + wave) * 64 + lane;
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
                if (a > -3.0e38f) sw = __expf(0.5f * (a - lse_k));       // sqrt of the importance weight (:353-358)
                rew = tlp[n] - logq[n];                                  // ng_estimator.py:346
            }
            const bool live = sw > 0.f;
#pragma unroll
            for (int i = 0; i < DP; ++i)
                if (i < D) zw[i * 64 + lane] = live ? z[i] : 0.f;
            zw[D * 64 + lane] = 1.f;
            zw[(D + 1) * 64 + lane] = live ? rew : 0.f;
            zw[(D + 2) * 64 + lane] = live ? sw : 0.f;
        }
        __syncthreads();
        for (int u = 0; u < 4 && t0 + u < n_tiles; ++u) {
            {
                const float* zu = zs + u * ZS_TILE;
                const int n = tid & 63;
                const float swn = zu[(D + 2) * 64 + n];
                for (int r = tid >> 6; r < n_rows; r += 8) {
                    const int t = tab[r];
                    if (t >= 0) phi[r * PHI_LD + n] = (swn * zu[(t & 0xffff) * 64 + n]) * zu[(t >> 16) * 64 + n];
                }
            }
            __syncthreads();
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int q = wave + 8 * pp;
                const float* pa = phi + (16 * (q >> 3) + r16) * PHI_LD + 4 * kg;
                const float* pb = phi + (col_base + 16 * (q & 7) + r16) * PHI_LD + 4 * kg;
                float4 av[4], bv[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    av[qq] = *reinterpret_cast<const float4*>(pa + 16 * qq);
                    bv[qq] = *reinterpret_cast<const float4*>(pb + 16 * qq);
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    acc[pp] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[qq].x, (double)bv[qq].x, acc[pp], 0, 0, 0);
                    acc[pp] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[qq].y, (double)bv[qq].y, acc[pp], 0, 0, 0);
                    acc[pp] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[qq].z, (double)bv[qq].z, acc[pp], 0, 0, 0);
                    acc[pp] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[qq].w, (double)bv[qq].w, acc[pp], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    // D[i][j] of the fp64 MFMA: lane l, register r -> i = 4 r + l / 16, j = l % 16
    double* Gk = G + (size_t)k * LDG * LDG;
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
        const int q = wave + 8 * pp;
        const int ti = q >> 3, tj = q & 7;
        if (diag && tj > ti) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = 128 * BI + 16 * ti + 4 * r + kg, gj = 128 * BC + 16 * tj + r16;
            Gk[(size_t)gi * LDG + gj] = acc[pp][r];
        }
    }
}

// One workgroup per component; see the banner above.  G: [LDG][LDG] fp64, lower triangle + row F valid on entry.
__global__ __launch_bounds__(1024) void more_solve_big_kernel(int D, int LDG, double* __restrict__ Gall,
                                                              const float* __restrict__ chols, const float* __restrict__ l2,
                                                              float* __restrict__ H_neg, float* __restrict__ g_neg,
                                                              double* __restrict__ beta_all) {
    extern __shared__ double smd[];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    const int T2 = D * (D + 1) / 2;
    const int F = T2 + D + 1;
    double* G = Gall + (size_t)k * LDG * LDG;
    double* beta = beta_all + (size_t)k * LDG;
    double* Pi = smd;                                  // [128][33] panel rows of the row tile
    double* Pc = smd + 128 * 33;                       // [128][33] panel rows of the column tile
    double* Db = smd + 2 * 128 * 33;                   // [32][33] diagonal block
    double* red = Db + 32 * 33;                        // [32][33] reduction scratch
    const double ridge = (double)l2[k];
    for (int i = tid; i < F - 1; i += 1024) G[(size_t)i * LDG + i] += ridge;      // least_squares.py:71-73 (bias unregularised)
    if (tid == 0) s_fail = 0;
    __syncthreads();

    // ---- blocked right-looking Cholesky of the leading F x F block; row F (the right-hand side) rides along ------------------
    for (int jb = 0; jb < F; jb += 32) {
        const int nbc = min(32, F - jb);
        // diagonal block -> LDS, factorised by the first wave (lane = row)
        for (int e = tid; e < 32 * 32; e += 1024) {
            const int i = e >> 5, c = e & 31;
            Db[i * 33 + c] = (i < nbc && c <= i) ? G[(size_t)(jb + i) * LDG + jb + c] : (i == c ? 1.0 : 0.0);
        }
        __syncthreads();
        if (tid < 64) {
            const int i = tid & 31;
            for (int j = 0; j < nbc; ++j) {
                const double d = Db[j * 33 + j];
                if (!(d > 0.0)) { if (tid == 0) s_fail = 1; break; }            // uniform within the wave
                const double rs = 1.0 / sqrt(d);
                double lij = 0.0;
                if (tid < 32 && i >= j) { lij = Db[i * 33 + j] * rs; Db[i * 33 + j] = lij; }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (tid < 32 && i > j)
                    for (int c = j + 1; c <= i; ++c) Db[i * 33 + c] -= lij * Db[c * 33 + j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (s_fail) break;
        for (int e = tid; e < 32 * 32; e += 1024) {
            const int i = e >> 5, c = e & 31;
            if (i < nbc && c <= i) G[(size_t)(jb + i) * LDG + jb + c] = Db[i * 33 + c];
        }
        // panel rows i > jb + nbc - 1 (up to the right-hand side row F): x = a L_jj^-T, one thread per row
        for (int i = jb + nbc + tid; i <= F; i += 1024) {
            double x[32];
            double* row = G + (size_t)i * LDG + jb;
#pragma unroll
            for (int c = 0; c < 32; ++c) x[c] = (c < nbc) ? row[c] : 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                double t = x[c];
#pragma unroll
                for (int u = 0; u < c; ++u) t = fma(-x[u], Db[c * 33 + u], t);
                x[c] = t / Db[c * 33 + c];
            }
#pragma unroll
            for (int c = 0; c < 32; ++c)
                if (c < nbc) row[c] = x[c];
        }
        __threadfence_block();
        __syncthreads();
        // trailing update G[i][c] -= sum_t P[i][t] P[c][t] for jb + nbc <= c <= i <= F, c < F: 128 x 128 tiles, 4 x 4 per thread
        const int r0 = jb + nbc;
        const int nt = (F + 1 - r0 + 127) / 128;
        const int ty = tid >> 5, tx = tid & 31;                         // rows 4 ty .. 4 ty + 3, columns tx, tx + 32, tx + 64, tx + 96
        for (int ib = 0; ib < nt; ++ib) {
            for (int cbk = 0; cbk <= ib; ++cbk) {
                for (int e = tid; e < 128 * 32; e += 1024) {
                    const int rr = e >> 5, t = e & 31;
                    const int gi = r0 + 128 * ib + rr, gc = r0 + 128 * cbk + rr;
                    Pi[rr * 33 + t] = (gi <= F && t < nbc) ? G[(size_t)gi * LDG + jb + t] : 0.0;
                    Pc[rr * 33 + t] = (gc < F && t < nbc) ? G[(size_t)gc * LDG + jb + t] : 0.0;
                }
                __syncthreads();
                double a[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) a[u][v] = 0.0;
                for (int t = 0; t < 32; ++t) {
                    double pi[4], pc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { pi[u] = Pi[(4 * ty + u) * 33 + t]; pc[u] = Pc[(tx + 32 * u) * 33 + t]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) a[u][v] = fma(pi[u], pc[v], a[u][v]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int gi = r0 + 128 * ib + 4 * ty + u, gc = r0 + 128 * cbk + tx + 32 * v;
                        if (gi <= F && gc < F && gc <= gi) G[(size_t)gi * LDG + gc] -= a[u][v];
                    }
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    const bool fail = s_fail != 0;
    __syncthreads();
    if (!fail) {
        // ---- back substitution L^T beta = y (y = row F), 32 columns per round -----------------------------------------------
        for (int jb = ((F - 1) / 32) * 32; jb >= 0; jb -= 32) {
            const int nbc = min(32, F - jb);
            // part[c] = sum_{i >= jb + nbc} L[i][jb + c] beta[i]: thread (c = tid % 32, rows tid / 32, + 32, ...)
            {
                const int c = tid & 31, rr = tid >> 5;
                double part = 0.0;
                if (c < nbc)
                    for (int i = jb + nbc + rr; i < F; i += 32) part = fma(G[(size_t)i * LDG + jb + c], beta[i], part);
                red[rr * 33 + c] = part;
            }
            for (int e = tid; e < 32 * 32; e += 1024) {
                const int i = e >> 5, c = e & 31;
                Db[i * 33 + c] = (i < nbc && c <= i) ? G[(size_t)(jb + i) * LDG + jb + c] : (i == c ? 1.0 : 0.0);
            }
            __syncthreads();
            if (tid < 32) {
                double rc = 0.0;
                if (tid < nbc) {
                    double p = 0.0;
                    for (int rr = 0; rr < 32; ++rr) p += red[rr * 33 + tid];
                    rc = G[(size_t)F * LDG + jb + tid] - p;
                }
                for (int jj = nbc - 1; jj >= 0; --jj) {
                    const double bj = __shfl(rc, jj, 32) / Db[jj * 33 + jj];
                    if (tid == jj) beta[jb + jj] = bj;
                    if (tid < jj) rc = fma(-Db[jj * 33 + tid], bj, rc);
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        // ---- un-whitening (least_squares.py:177-189, ng_estimator.py:371-373), as the small kernel ---------------------------
        double* Ls = smd;
        double* Qs = Ls + D * D;
        double* Xs = Qs + D * (D + 1);
        for (int e = tid; e < D * D; e += 1024) Ls[e] = (double)chols[(size_t)k * D * D + e];
        for (int e = tid; e < D * (D + 1); e += 1024) {
            const int i = e / (D + 1), j = e % (D + 1);
            double v;
            if (j == D) v = beta[T2 + i];                                            // lin_w
            else {
                const int a = min(i, j), b = max(i, j);
                const double q = beta[a * D - a * (a - 1) / 2 + (b - a)];
                v = (a == b) ? -2.0 * q : -q;
            }
            Qs[e] = v;
        }
        __syncthreads();
        if (tid <= D) {                                                             // X = L_o^-T [Q_w | lin_w]   (lane = column)
            for (int i = D - 1; i >= 0; --i) {
                double t = Qs[i * (D + 1) + tid];
                for (int j = i + 1; j < D; ++j) t = fma(-Ls[j * D + i], Xs[j * (D + 1) + tid], t);
                Xs[i * (D + 1) + tid] = t / Ls[i * D + i];
            }
        }
        __syncthreads();
        if (tid < D) {                                                              // H = X L_o^-1, g = -L_o^-T lin_w
            double* hrow = Qs + tid * (D + 1);
            for (int i = D - 1; i >= 0; --i) {
                double t = Xs[tid * (D + 1) + i];
                for (int j = i + 1; j < D; ++j) t = fma(-Ls[j * D + i], hrow[j], t);
                hrow[i] = t / Ls[i * D + i];
            }
            for (int i = 0; i < D; ++i) H_neg[((size_t)k * D + tid) * D + i] = (float)hrow[i];
            g_neg[(size_t)k * D + tid] = (float)(-Xs[tid * (D + 1) + D]);
        }
    } else {
        const float nanv = __int_as_float(0x7fc00000);
        for (int e = tid; e < D * D; e += 1024) H_neg[(size_t)k * D * D + e] = nanv;
        for (int e = tid; e < D; e += 1024) g_neg[(size_t)k * D + e] = nanv;
    }
}

template <int DP>
int launch_more_big(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* chols, const float* X, int N,
                    const float* ld, const float* logq, const float* bg, const float* tlp, const int32_t* mapping,
                    int map_offset, int flags, const float* l2, float* H_neg, float* g_neg) {
    const int F = D * (D + 1) / 2 + D + 1;
    const int nblk = (F + 1 + 127) / 128;
    const int LDG = 128 * nblk;
    const size_t g_doubles = (size_t)K * LDG * LDG, b_doubles = (size_t)K * LDG;
    int rc = gmmvi_ws_reserve(ctx, (g_doubles + b_doubles) * sizeof(double) + (size_t)K * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    double* G = (double*)ctx->ws;
    double* beta = G + g_doubles;
    float* lse = (float*)(beta + b_doubles);
    if (flags & GMMVI_SELF_NORMALIZED) {
        GMMVI_PROF(ctx, "more_lse");
        hipLaunchKernelGGL(more_lse_kernel, dim3(K), dim3(1024), 0, ctx->stream, N, ld, bg, mapping, map_offset, flags, lse);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    const size_t gram_lds = ((size_t)256 * PHI_LD + (size_t)4 * (D + 3) * 64 + 256) * sizeof(float);
    static size_t gram_attr = 0;
    if (gram_lds > gram_attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)more_gram_big_kernel<DP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds));
        gram_attr = gram_lds;
    }
    {
        GMMVI_PROF(ctx, "more_gram");
        hipLaunchKernelGGL((more_gram_big_kernel<DP>), dim3(nblk * (nblk + 1) / 2, K), dim3(512), gram_lds, ctx->stream, D, N,
                           nblk, LDG, packed, X, ld, bg, tlp, logq, mapping, map_offset, flags, lse, G);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    size_t solve_doubles = (size_t)2 * 128 * 33 + 2 * 32 * 33;
    const size_t unwhiten = (size_t)D * D + 2 * (size_t)D * (D + 1);
    if (unwhiten > solve_doubles) solve_doubles = unwhiten;
    const size_t solve_lds = solve_doubles * sizeof(double);
    static size_t solve_attr = 0;
    if (solve_lds > solve_attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)more_solve_big_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds));
        solve_attr = solve_lds;
    }
    GMMVI_PROF(ctx, "more_solve");
    hipLaunchKernelGGL(more_solve_big_kernel, dim3(K), dim3(1024), solve_lds, ctx->stream, D, LDG, G, chols, l2, H_neg, g_neg,
                       beta);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// workgroups per launch that fill the chip in whole rounds: the count c in [lo, hi] maximising c K / (ceil(c K / cus) cus)
static int more_pick_chunks(int K, int cus, int tiles) {
    int best = 1; double best_eff = 0.0;
    const int hi = tiles < 4 ? 1 : tiles / 4;                 // at least four 64-sample tiles per workgroup
    for (int c = 1; c <= hi && (long)c * K <= 6L * cus; ++c) {
        const long wgs = (long)c * K;
        const double eff = (double)wgs / (double)(((wgs + cus - 1) / cus) * cus);
        const bool enough = wgs >= cus || c == hi;
        if ((enough ? eff : eff * 0.5) > best_eff + 1e-9) { best_eff = enough ? eff : eff * 0.5; best = c; }
    }
    return best;
}

template <int DP, int PP>
int launch_more(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* chols, const float* X, int N,
                const float* ld, const float* logq, const float* bg, const float* tlp, const int32_t* mapping,
                int map_offset, int flags, const float* l2, float* H_neg, float* g_neg) {
    const int F = D * (D + 1) / 2 + D + 1;
    const int nb = (F + 1 + 15) / 16;
    const int n_pairs = nb * (nb + 1) / 2;
    if (n_pairs > 8 * PP) return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: internal tiling error");
    const int tiles = (N + 63) / 64;
    int n_chunks = more_pick_chunks(K, ctx->num_cus, tiles);
    const int tiles_per_chunk = (tiles + n_chunks - 1) / n_chunks;
    n_chunks = (tiles + tiles_per_chunk - 1) / tiles_per_chunk;
    const size_t slab_doubles = (size_t)K * n_chunks * n_pairs * 256;
    int rc = gmmvi_ws_reserve(ctx, slab_doubles * sizeof(double) + (size_t)K * sizeof(float));
    if (rc != GMMVI_OK) return rc;
    double* slab = (double*)ctx->ws;
    float* lse = (float*)(slab + slab_doubles);
    if (flags & GMMVI_SELF_NORMALIZED) {
        GMMVI_PROF(ctx, "more_lse");
        hipLaunchKernelGGL(more_lse_kernel, dim3(K), dim3(1024), 0, ctx->stream, N, ld, bg, mapping, map_offset, flags, lse);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    const size_t gram_lds = ((size_t)16 * nb * PHI_LD + (size_t)8 * (D + 3) * 64 + (size_t)(F + 1)) * sizeof(float);
    static size_t gram_attr = 0;
    if (gram_lds > gram_attr) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)more_gram_kernel<DP, PP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds));
        gram_attr = gram_lds;
    }
    {
        GMMVI_PROF(ctx, "more_gram");
        hipLaunchKernelGGL((more_gram_kernel<DP, PP>), dim3(n_chunks, K), dim3(512), gram_lds, ctx->stream, D, N,
                           tiles_per_chunk, nb, packed, X, ld, bg, tlp, logq, mapping, map_offset, flags, lse, slab);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    const size_t solve_lds = ((size_t)D * D + 2 * (size_t)D * (D + 1)) * sizeof(double);
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
    GMMVI_ARG_CHECK(ctx, D < GMMVI_MAX_DIM);
    // the kernels read the register-path block layout (Pack<DP>); a dimension on the blocked path (50 < D <= 63 by default)
    // hands over [mu | log-normaliser | pad | dense L^-1] blocks of another stride: the components are re-packed for this call
    // (the means are the first D floats of a blocked block); beyond D = 63 there is no MORE route (the Stein kernels need D + 1
    // <= 64 lanes, the whitening its row in registers)
    float* repacked = nullptr;
    if (gmmvi_is_blocked_dim(D)) {
        if (D >= GMMVI_MAX_DIM)
            return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: no MORE route beyond D = 63");
        const size_t rstride = gmmvi_packed_stride_dp(gmmvi_padded_dim(D));
        float* means_tmp = nullptr;
        GMMVI_HIP_CHECK(ctx, hipMalloc(&repacked, ((size_t)K * rstride + (size_t)K * D) * sizeof(float)));
        means_tmp = repacked + (size_t)K * rstride;
        hipError_t e = hipMemcpy2DAsync(means_tmp, (size_t)D * sizeof(float), packed_dev, gmmvi_blocked_stride(D) * sizeof(float),
                                        (size_t)D * sizeof(float), (size_t)K, hipMemcpyDeviceToDevice, ctx->stream);
        int rc = e == hipSuccess ? gmmvi_pack_register_layout(ctx, K, D, means_tmp, chols_dev, repacked)
                                 : gmmvi_fail(ctx, GMMVI_ERR_HIP, std::string("gmmvi_more: ") + hipGetErrorString(e));
        if (rc != GMMVI_OK) { (void)hipFree(repacked); return rc; }
        packed_dev = repacked;
    }
    // (the temporary blocks live until the launches below have run: freed after a stream synchronisation)
    struct Cleanup {
        gmmvi_ctx* c; float* p;
        ~Cleanup() { if (p) { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); } }
    } cleanup{ctx, repacked};
    GMMVI_ARG_CHECK(ctx, packed_dev && chols_dev && X_dev && logq_dev && tlp_dev && l2_dev && H_neg_out_dev && g_neg_out_dev);
    if (flags & GMMVI_OWN_SAMPLES_ONLY) GMMVI_ARG_CHECK(ctx, mapping_dev != nullptr);
    else GMMVI_ARG_CHECK(ctx, ld_dev && bg_dev);
#define GMMVI_MORE_CASE(DPV, PPV)                                                                                    \
    return launch_more<DPV, PPV>(ctx, K, D, packed_dev, chols_dev, X_dev, N, ld_dev, logq_dev, bg_dev, tlp_dev,      \
                                 mapping_dev, map_offset, flags, l2_dev, H_neg_out_dev, g_neg_out_dev)
    if (D > GMMVI_MORE_REGISTER_MAX_DIM) {            // the ridge system no longer fits one workgroup's registers: tiled route
#define GMMVI_MORE_BIG(DPV)                                                                                          \
    return launch_more_big<DPV>(ctx, K, D, packed_dev, chols_dev, X_dev, N, ld_dev, logq_dev, bg_dev, tlp_dev,       \
                                mapping_dev, map_offset, flags, l2_dev, H_neg_out_dev, g_neg_out_dev)
        switch (gmmvi_padded_dim(D)) {
            case 24: GMMVI_MORE_BIG(24);
            case 32: GMMVI_MORE_BIG(32);
            case 40: GMMVI_MORE_BIG(40);
            case 50: GMMVI_MORE_BIG(50);
            case 64: GMMVI_MORE_BIG(64);
            default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: unsupported dimension");
        }
#undef GMMVI_MORE_BIG
    }
    switch (gmmvi_padded_dim(D)) {
        case 2: GMMVI_MORE_CASE(2, 1);
        case 4: GMMVI_MORE_CASE(4, 1);
        case 8: GMMVI_MORE_CASE(8, 1);
        case 10: GMMVI_MORE_CASE(10, 2);
        case 12: GMMVI_MORE_CASE(12, 3);
        case 16: GMMVI_MORE_CASE(16, 7);
        case 20: GMMVI_MORE_CASE(20, 15);
        case 24: GMMVI_MORE_CASE(24, 17);
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_more: unsupported dimension");
    }
#undef GMMVI_MORE_CASE
}
