// Blocked path for 50 < D <= 512 (blocked.h): densities, sampling, Stein estimate and the KL-constrained update with the
// O(D^2)-per-pair work expressed as batched f32 contractions on the matrix cores (bgemm: on the f32 instruction, or -- wide and
// long launches, the default -- on the bf16 instruction through 3-way split operands at f32 accuracy) and the
// O(D^3)-per-component factorisations as one-workgroup-per-component kernels on L2-resident matrices.
//
// Reference arithmetic: models/full_cov_gmm.py:56-62 (z = L^-1 (x - mu) -- here Z = (X - mu) L^-T with the explicit
// inverse the reference also keeps for its sample database, optimization/sample_db.py:121), models/gmm.py:183-216,274-300,
// :361-386, gmmvi_modules/ng_estimator.py:146-263, gmmvi_modules/ng_based_component_updater.py:244-524.
#include "blocked.h"
#include "bf16_split.h"
#include "wave_reduce.h"
#include <cfloat>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------------
// batched contraction  C[b] (+)= alpha * opA(A[b]) * opB(B[b]) (+ bias)   on v_mfma_f32_32x32x2_f32
// ---------------------------------------------------------------------------------------------------------------------------
// Workgroup = 128 x (32 NT) tile of C, 4 waves stacked along M: wave w owns rows 32 w .. 32 w + 31 and NT 32 x 32 MFMA
// tiles (16 NT accumulator registers); NT in {2..5} is chosen per launch so that the padded width wastes the least
// (D = 300: NT = 5, 320 columns).  The k dimension advances 16 at a time through two LDS images per operand, stored
// k-major ([k][row]; row strides of 132 / 32 NT + 4 words keep 16-byte alignment for ds_write_b128 and let the lanes of a
// ds_read_b32 walk consecutive rows).  Every thread moves 4 consecutive elements along the contiguous direction of its
// operand (one 16-byte load when the operand is aligned), global loads of step s+1 are in flight while step s is
// multiplied.  The f32 MFMA shares the vector ALU on gfx950 (it does not co-execute), so the staging code is kept to a
// few vector instructions per 16-byte load: uniform base pointers + per-thread offsets fixed for the whole launch.
// The A operand takes an element-wise prologue (subtract a vector along k, scale along k, scale along rows), which is how
// "x - mu", the importance weights and the responsibilities enter without being materialised.  A triangular opB skips
// the 32-column sub-tiles whose k range is structurally zero.
struct BG {
    const float* A; const float* B; float* C;
    int M, N, Kd;
    int lda, ldb, ldc;
    long long sA, sB, sC;            // batch strides (floats)
    int a_kmajor;                    // 0: A[m * lda + k]      1: A[k * lda + m]
    int b_kmajor;                    // 0: B[n * ldb + k]      1: B[k * ldb + n]
    const float* a_sub; long long s_asub;        // [Kd]  A(m, k) -= a_sub[k]
    const float* a_kscale; long long s_aks;      // [Kd]  A(m, k) *= a_kscale[k]
    const float* a_rscale; long long s_ars;      // [M]   A(m, k) *= a_rscale[m]
    const float* a_rsub; long long s_arsub;      // [M]   with a_kscale (k-major A only): A(m, k) = (A(m, k) - a_rsub[m]) * a_kscale[k]
    const float* c_bias; long long s_cb;         // [N]   C(m, n) += c_bias[n]
    const int32_t* row_off;          // optional [batches + 1]: batch b owns rows [row_off[b], row_off[b+1]) of A and C (M = bound)
    int inner;                       // > 0: workgroup z accumulates the batches [z * inner, min((z+1) * inner, inner_total)) into slab z of C
    int inner_total;
    int tri;                         // 1: opB(k, n) = 0 for k > n   2: opB(k, n) = 0 for k < n
    int accumulate;                  // C += result
    int ksplit;                      // > 1: grid.z = batches * ksplit, slab z of C receives the partial sum over its k range
    float alpha;
    // optional: per-row sums of squares of this workgroup's result columns, rowsq[blockIdx.x * rs_tile + blockIdx.z * rs_batch +
    // row] (one partial per column tile; the caller adds the tiles).  no_store: the result itself is not written.
    float* rowsq; long long rs_tile, rs_batch;
    int no_store;
    int debug;                       // experiments (GMMVI_BG_DEBUG): 1 no MFMAs, 2 no split / LDS writes, 4 no global loads
};

constexpr int BM = 128, BK = 16, LDA_S = BM + 4;

// 4 consecutive floats, the first nvalid of them in range (others 0); one 16-byte load when allowed
__device__ __forceinline__ float4 ld4(const float* p, int nvalid, bool vec) {
    if (vec && nvalid >= 4) return *reinterpret_cast<const float4*>(p);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) r.x = p[0];
    if (nvalid > 1) r.y = p[1];
    if (nvalid > 2) r.z = p[2];
    if (nvalid > 3) r.w = p[3];
    return r;
}
__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- split-operand route (SPLIT = 1): the same f32 contraction on the bf16 matrix cores ------------------------------
// Every f32 operand element is written as x = x1 + x2 + x3 with three bf16 values (round-to-nearest splits: |x2| <= 2^-9 |x|,
// |x3| <= 2^-18 |x|, residual <= 2^-27 |x|); a product x y is the sum of the six partial products x_i y_j with i + j <= 4
// (each EXACT in f32: 8 x 8 significant bits), accumulated in f32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  The
// dropped terms (x2 y3, x3 y2, x3 y3 and the residuals) are below 2^-25 |x y|: less than the rounding of one f32 multiply-add,
// so the result carries the error of an f32 contraction (tests/test_hip_blocked.py measures both routes against fp64).
// Six bf16 MFMAs (6 x 32 cycles per 32 x 32 x 16) replace eight f32 MFMAs (8 x 64 cycles): 2.7 x the matrix-core rate.
// LDS images per operand and plane, 16 k per step: a k-contiguous operand is stored in fragment order ([k / 8][row][k % 8]:
// lane l of a ds_read_b128 reads 16 consecutive bytes after lane l - 1); a k-major operand is stored [k][row] (row stride
// 2 * rowsp bytes = 64 or 192 mod 256: the four k rows of a transposed read fall into disjoint banks) and read with
// ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of a 4 x 16 block -- the MFMA operand order.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// (bf16x2_t, f32x2_t, cvt_pk_bf16, split_pair: bf16_split.h)
constexpr int split_rowsp(int rows) {                  // row stride (elements) of a k-major bf16 image
    int r = rows;
    while (r % 128 != 32 && r % 128 != 96) r += 4;
    return r;
}
__device__ __forceinline__ i32x4 lds_tr_frag(const unsigned char* base, int byte_off, int row_stride_bytes) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + byte_off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + byte_off + 4 * row_stride_bytes));
    i32x4 r;
    r[0] = __builtin_bit_cast(int2, lo).x; r[1] = __builtin_bit_cast(int2, lo).y;
    r[2] = __builtin_bit_cast(int2, hi).x; r[3] = __builtin_bit_cast(int2, hi).y;
    return r;
}

// NT: 32-column MFMA tiles per wave; AK / BKM: operand is k-major in memory; PRO: prologue on A (0 none, 1 subtract a
// vector along k, 2 scale rows, 3 scale along k, 4 subtract a vector along the rows, then scale along k: k-major A only).  Every step takes one of two routes, chosen uniformly: the fast route
// (aligned operands, a full 16-wide k step, every 16-byte piece entirely valid or entirely void -- decided once per thread)
// issues all loads back to back without a branch; the edge route handles ragged ends element by element.
template <int NT, int AK, int BKM, int PRO>
__global__ __launch_bounds__(256, PRO == 2 ? 2 : 3) void bgemm_kernel(BG g) {
    constexpr int BN = 32 * NT, LDB_S = BN + 4;
    constexpr int NVB = (BN * BK / 4 + 255) / 256;          // 16-byte pieces of the B tile per thread
    __shared__ __align__(16) float As[2][BK * LDA_S];
    __shared__ __align__(16) float Bs[2][BK * LDB_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nsplit = g.ksplit > 1 ? g.ksplit : 1;
    const int bz = blockIdx.z / nsplit, kz = blockIdx.z - bz * nsplit;
    int Mb = g.M;
    long long rowbase = 0;
    if (g.row_off) { rowbase = g.row_off[bz]; Mb = g.row_off[bz + 1] - (int)rowbase; }
    if (m0 >= Mb) return;
    int kb = 0, ke = g.Kd;
    if (g.tri == 1) ke = min(g.Kd, n0 + BN);
    else if (g.tri == 2) kb = min(g.Kd, n0) & ~3;
    if (nsplit > 1) {
        const int chunk = ((g.Kd + nsplit - 1) / nsplit + BK - 1) / BK * BK;
        kb = min(g.Kd, kz * chunk);
        ke = min(g.Kd, kb + chunk);
    }
    const int spb = (ke - kb + BK - 1) / BK;
    const int nb = g.inner > 0 ? min(g.inner, g.inner_total - bz * g.inner) : 1;
    const int total = nb * spb;
    const float* A0 = g.A + (AK ? rowbase : rowbase * g.lda);
    const float* pro = PRO == 1 ? g.a_sub : (PRO == 2 ? g.a_rscale : (PRO >= 3 ? g.a_kscale : nullptr));
    const long long s_pro = PRO == 1 ? g.s_asub : (PRO == 2 ? g.s_ars : (PRO >= 3 ? g.s_aks : 0));
    const bool vecR = PRO != 4 || (al16(g.a_rsub) && (g.s_arsub & 3) == 0);
    const bool vecA = al16(A0) && ((g.lda | g.sA) & 3) == 0;
    const bool vecB = al16(g.B) && ((g.ldb | g.sB) & 3) == 0;
    const bool vecP = PRO == 0 || (al16(pro) && (s_pro & 3) == 0);
    // fast route: no 16-byte piece may straddle the end of its operand
    const bool fast_ok = vecA && vecB && (PRO != 1 || vecP) && vecR && (!AK || (Mb & 3) == 0 || m0 + BM <= Mb) &&
                         (!BKM || (g.N & 3) == 0 || n0 + BN <= g.N);

    // per-thread pieces: A tile = 512 pieces (2 per thread), B tile = 4 BN pieces
    // k-contiguous operand: piece = (row r, 4 consecutive k); k-major operand: piece = (k row, 4 consecutive rows)
    int a_r[2], a_k[2], a_off[2];
    bool a_ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + 256 * u;
        if (AK) { a_k[u] = idx >> 5; a_r[u] = 4 * (idx & 31); } else { a_r[u] = idx >> 2; a_k[u] = 4 * (idx & 3); }
        a_ok[u] = m0 + a_r[u] < Mb;
        a_off[u] = a_ok[u] ? (AK ? a_k[u] * g.lda + m0 + a_r[u] : (m0 + a_r[u]) * g.lda + a_k[u]) : 0;
    }
    int b_c[NVB], b_k[NVB], b_off[NVB];
    bool b_ok[NVB];
#pragma unroll
    for (int u = 0; u < NVB; ++u) {
        const int idx = tid + 256 * u;
        if (BKM) { b_k[u] = idx / (BN / 4); b_c[u] = 4 * (idx % (BN / 4)); } else { b_c[u] = idx >> 2; b_k[u] = 4 * (idx & 3); }
        if (idx >= BN * BK / 4) { b_c[u] = BN; b_k[u] = 0; }            // no piece
        b_ok[u] = b_c[u] < BN && n0 + b_c[u] < g.N;
        b_off[u] = b_ok[u] ? (BKM ? b_k[u] * g.ldb + n0 + b_c[u] : (n0 + b_c[u]) * g.ldb + b_k[u]) : 0;
    }

    float4 ra[2], rb[NVB], pv[2], pw[2];
    bool pending = false;                   // fast route: prologue / masking of the staged pieces still to be applied
    auto gload = [&](int step) {
        pending = false;
        const int bi = step / spb;
        const long long b = (long long)bz * (g.inner > 0 ? g.inner : 1) + bi;
        const int k0 = kb + (step - bi * spb) * BK;
        const float* Ab = A0 + b * g.sA;
        const float* Bb = g.B + b * g.sB;
        const float* pb = PRO ? pro + b * s_pro : nullptr;
        const float* pr = PRO == 4 ? g.a_rsub + b * g.s_arsub : nullptr;
        if (fast_ok && k0 + BK <= ke) {
            const float* Ak = Ab + (AK ? (long long)k0 * g.lda : (long long)k0);
            const float* Bk = Bb + (BKM ? (long long)k0 * g.ldb : (long long)k0);
            // loads only: the prologue and the masking wait for the data, so they run in sstore, after this step's MFMAs
#pragma unroll
            for (int u = 0; u < 2; ++u) ra[u] = *reinterpret_cast<const float4*>(Ak + a_off[u]);
#pragma unroll
            for (int u = 0; u < NVB; ++u) rb[u] = *reinterpret_cast<const float4*>(Bk + b_off[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (PRO == 1) {
                    if (AK) { const float s = pb[k0 + a_k[u]]; pv[u] = make_float4(s, s, s, s); }
                    else pv[u] = *reinterpret_cast<const float4*>(pb + k0 + a_k[u]);
                } else if (PRO == 2) {
                    // rows that do not exist read element 0 (masked in sstore): no branch around the load
                    if (AK) pv[u] = *reinterpret_cast<const float4*>(pb + (a_ok[u] ? m0 + a_r[u] : 0));
                    else { const float s = pb[a_ok[u] ? m0 + a_r[u] : 0]; pv[u] = make_float4(s, s, s, s); }
                } else if (PRO == 3) {
                    if (AK) { const float s = pb[k0 + a_k[u]]; pv[u] = make_float4(s, s, s, s); }
                    else pv[u] = ld4(pb + k0 + a_k[u], 4, vecP);
                } else if (PRO == 4) {
                    const float s = pb[k0 + a_k[u]];
                    pv[u] = make_float4(s, s, s, s);
                    pw[u] = *reinterpret_cast<const float4*>(pr + (a_ok[u] ? m0 + a_r[u] : 0));
                }
            }
            pending = true;
            return;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int gm = m0 + a_r[u], gk = k0 + a_k[u];
            float4 v;
            if (AK) {                                           // 4 consecutive rows of one k
                const int nv = (gk < ke) ? Mb - gm : 0;
                v = ld4(Ab + (long long)gk * g.lda + gm, nv, vecA);
                if (nv > 0) {
                    if (PRO == 1) { const float s = pb[gk]; v.x -= s; v.y -= s; v.z -= s; v.w -= s; }
                    if (PRO == 3) { const float s = pb[gk]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
                    if (PRO == 2) { const float4 s = ld4(pb + gm, nv, vecP); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
                    if (PRO == 4) {
                        const float4 r4 = ld4(pr + gm, nv, vecR);
                        const float s = pb[gk];
                        v.x = (v.x - r4.x) * s; v.y = (v.y - r4.y) * s; v.z = (v.z - r4.z) * s; v.w = (v.w - r4.w) * s;
                    }
                    if (nv < 4) { if (nv < 2) v.y = 0.f; if (nv < 3) v.z = 0.f; v.w = 0.f; }
                }
            } else {                                            // 4 consecutive k of one row
                const int nv = (gm < Mb) ? ke - gk : 0;
                v = ld4(Ab + (long long)gm * g.lda + gk, nv, vecA);
                if (nv > 0) {
                    if (PRO == 1) { const float4 s = ld4(pb + gk, nv, vecP); v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w; }
                    if (PRO == 3) { const float4 s = ld4(pb + gk, nv, vecP); v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w; }
                    if (PRO == 2) { const float s = pb[gm]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
                    if (nv < 4) { if (nv < 2) v.y = 0.f; if (nv < 3) v.z = 0.f; v.w = 0.f; }
                }
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NVB; ++u) {
            const int gn = n0 + b_c[u], gk = k0 + b_k[u];
            if (BKM) {
                const int nv = (gk < ke && b_c[u] < BN) ? g.N - gn : 0;
                rb[u] = ld4(Bb + (long long)gk * g.ldb + gn, nv, vecB);
            } else {
                const int nv = (gn < g.N && b_c[u] < BN) ? ke - gk : 0;
                rb[u] = ld4(Bb + (long long)gn * g.ldb + gk, nv, vecB);
            }
        }
    };
    auto sstore = [&](int buf) {
        if (pending) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4 v = ra[u];
                if (PRO == 1) { v.x -= pv[u].x; v.y -= pv[u].y; v.z -= pv[u].z; v.w -= pv[u].w; }
                if (PRO == 4) { v.x -= pw[u].x; v.y -= pw[u].y; v.z -= pw[u].z; v.w -= pw[u].w; }
                if (PRO >= 2) { v.x *= pv[u].x; v.y *= pv[u].y; v.z *= pv[u].z; v.w *= pv[u].w; }
                ra[u] = a_ok[u] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NVB; ++u) rb[u] = b_ok[u] ? rb[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (AK) {
                *reinterpret_cast<float4*>(&As[buf][a_k[u] * LDA_S + a_r[u]]) = ra[u];
            } else {
                float* d = &As[buf][a_k[u] * LDA_S + a_r[u]];
                d[0] = ra[u].x; d[LDA_S] = ra[u].y; d[2 * LDA_S] = ra[u].z; d[3 * LDA_S] = ra[u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < NVB; ++u) {
            if (b_c[u] >= BN) continue;
            if (BKM) {
                *reinterpret_cast<float4*>(&Bs[buf][b_k[u] * LDB_S + b_c[u]]) = rb[u];
            } else {
                float* d = &Bs[buf][b_k[u] * LDB_S + b_c[u]];
                d[0] = rb[u].x; d[LDB_S] = rb[u].y; d[2 * LDB_S] = rb[u].z; d[3 * LDB_S] = rb[u].w;
            }
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (total > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int step = 0; step < total; ++step) {
        if (step + 1 < total) gload(step + 1);
        const int k0 = kb + (step % spb) * BK;
        const float* a_ = As[buf] + half * LDA_S + wave * 32 + col;
        const float* b_ = Bs[buf] + half * LDB_S + col;
        float av[BK / 2];
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) av[k2] = a_[2 * k2 * LDA_S];
        // sub-tile t (columns n0 + 32 t ..) is skipped for this k range when opB vanishes there or the columns do not exist.
        // (One uniform branch per sub-tile: straight-line variants per live range were measured -- the accumulator copies
        // between the variants double the register count and halve the occupancy, 1.6x slower.)
        // the B operands of sub-tile t + 1 are read (unconditionally) before the MFMAs of sub-tile t are issued, so the LDS
        // latency of a block hides behind the previous block's eight MFMAs.  The MFMAs are issued through inline asm with
        // the accumulators pinned to AGPRs: with the builtin the compiler kept acc[] in VGPRs across the (uniform) branches
        // and copied 16 registers into and out of one AGPR tuple around every block (400 v_accvgpr moves per k step).
        float bv[2][BK / 2];
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) bv[0][k2] = b_[2 * k2 * LDB_S];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) {
#pragma unroll
                for (int k2 = 0; k2 < BK / 2; ++k2) bv[(t + 1) & 1][k2] = b_[2 * k2 * LDB_S + 32 * (t + 1)];
            }
            const int c0 = n0 + 32 * t;
            if (c0 < g.N && !(g.tri == 1 && k0 >= c0 + 32) && !(g.tri == 2 && k0 + BK <= c0)) {
#pragma unroll
                for (int k2 = 0; k2 < BK / 2; ++k2)
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(av[k2]), "v"(bv[t & 1][k2]));
            }
        }
        if (step + 1 < total) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // the last MFMA (16 passes) must have retired before its accumulators are read (inline asm: no automatic hazard nops)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (g.rowsq != nullptr) {
        // lane (col, half) holds rows (r & 3) + 8 (r >> 2) + 4 half of its column in every sub-tile: square-sum over the
        // sub-tiles, then over the 32 lanes of the half
        float rs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (n0 + 32 * t + col < g.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = g.alpha * acc[t][r]; rs[r] = fmaf(v, v, rs[r]); }
            }
        }
        float* rq = g.rowsq + (long long)blockIdx.x * g.rs_tile + (long long)blockIdx.z * g.rs_batch + rowbase;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = rs[r];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            const int i = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (col == 0 && i < Mb) rq[i] = v;
        }
    }
    if (g.no_store) return;
    float* Cb = g.C + (long long)blockIdx.z * g.sC + rowbase * g.ldc;
    const float* bias = g.c_bias ? g.c_bias + (long long)bz * g.s_cb : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = n0 + 32 * t + col;
        if (j >= g.N) continue;
        const float bj = bias ? bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < Mb) {
                float* p = Cb + (long long)i * g.ldc + j;
                float v = fmaf(g.alpha, acc[t][r], bj);
                if (g.accumulate) v += *p;
                *p = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the same contraction, split-operand route: role-specialised workgroup on v_mfma_f32_32x32x16_bf16
// ---------------------------------------------------------------------------------------------------------------------------
// A wave issues one instruction every four cycles whatever its kind, an MFMA of this shape occupies the matrix core for 32:
// the kernel is paced by the matrix cores only if fewer than ~7 other instructions are issued per MFMA, address arithmetic
// and loop control included (first version of this route, every wave staging and multiplying: 8.9 vector + 7.4 scalar
// instructions per MFMA, matrix cores 19 % busy).  Hence:
//  * the B operand (component data / shared rows: reused by every row tile) is split ONCE per launch by bsplit_image_kernel
//    into the exact byte order of the LDS stage (per 16-k step: [plane][k half][column][8 k] bf16, transposed, masked for
//    triangular operands, zero-padded): staging B is a 16-byte copy;
//  * 512 threads: waves 0..3 MULTIPLY (wave w owns rows 32 w .. of the 128 x 32 NT tile; NT up to 10, so D <= 320 is ONE
//    column tile and (x - mu) is read and split once per component): LDS fragment reads and MFMAs only, the first fragments
//    of the next step are read before the last MFMAs of this one are issued; waves 4..7 STAGE: global loads (three register
//    stages: the loads of step s + 3 are issued when step s has been handed to LDS), the element-wise prologue and the
//    3-way split of A, the LDS writes;
//  * LDS is a ring of three stages; one LDS-only barrier per step (s_waitcnt lgkmcnt(0) + s_barrier: the stagers' global
//    loads stay in flight across it): during step t the multipliers read stage t % 3 (and the head of stage (t + 1) % 3,
//    complete since the previous barrier) while the stagers fill stage (t + 2) % 3.
#define BG_LDS_BARRIER()                                       \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

template <int N> using bg_ic = std::integral_constant<int, N>;

// bytes of the B image of one (batch, 16-k step): three planes x two k halves x ncolp columns x 8 bf16
__host__ __device__ inline size_t bimg_step_bytes(int ncolp) { return (size_t)96 * ncolp; }

// B image: thread = (column n, k half h) of one (step, batch): eight k values -> three 16-byte fragments
__global__ __launch_bounds__(256) void bsplit_image_kernel(BG g, int ncolp, int nsteps, unsigned char* img) {
    const int step = blockIdx.x, b = blockIdx.y;
    const float* Bb = g.B + (long long)b * g.sB;
    unsigned char* out = img + ((size_t)b * nsteps + step) * bimg_step_bytes(ncolp);
    for (int e = threadIdx.x; e < 2 * ncolp; e += 256) {
        const int h = e / ncolp, n = e - h * ncolp;
        const int k0 = step * BK + 8 * h;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            bool live = n < g.N && k < g.Kd;
            if (g.tri == 1) live = live && k <= n;
            if (g.tri == 2) live = live && k >= n;
            v[j] = live ? (g.b_kmajor ? Bb[(long long)k * g.ldb + n] : Bb[(long long)n * g.ldb + k]) : 0.f;
        }
        uint32_t p[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint4*>(out + ((size_t)(pl * 2 + h) * ncolp + n) * 16) = make_uint4(p[pl][0], p[pl][1], p[pl][2], p[pl][3]);
    }
}

// experiment switches (GMMVI_BG_DEBUG) exist only in -DBG_STAMPS builds: a runtime branch around a load or a hand-over changes
// what the compiler can prove about the loads in flight
#ifdef BG_STAMPS
#define BG_DBG(bit) (g.debug & (bit))
#else
#define BG_DBG(bit) 0
#endif
#ifdef BG_STAMPS
#define BG_CLK() __builtin_readcyclecounter()
#define BG_BARRIER_TIMED(acc_)                                 \
    do {                                                       \
        const unsigned long long b0__ = BG_CLK();              \
        BG_LDS_BARRIER();                                      \
        acc_ += BG_CLK() - b0__;                               \
    } while (0)
#else
#define BG_BARRIER_TIMED(acc_) BG_LDS_BARRIER()
#endif

// bytes of the LDS ring of bgemm_ws_kernel<NT, AK, .>
template <int NT, int AK>
constexpr int bg_ws_ring_bytes() {
    constexpr int BN = 32 * NT;
    constexpr int APL = AK ? BK * split_rowsp(BM) * 2 : 2 * (BM * 16 + 128), BPL = 2 * (BN * 16 + 128);
    return 3 * (3 * (APL + BPL) + 4096);
}

// one tile (bx, by, bzz) of the launch: both roles leave this function after 1 + total3 barriers
template <int NT, int AK, int PRO>
__device__ __forceinline__ void bgemm_ws_tile(const BG& g, const unsigned char* __restrict__ bimg, int ncolp, int nsteps,
                                              unsigned char* ring, const int bx, const int by, const int bzz) {
    constexpr int BN = 32 * NT;
#ifdef BG_STAMPS
    const unsigned long long st0 = BG_CLK();
    unsigned long long bwait = 0;
#endif
    constexpr int RSA = split_rowsp(BM);
    // fragment-order plane of a k-contiguous A: two k halves of [row][8 k] (+128 bytes between them: the two halves a
    // ds_write_b64 touches fall into different banks); B stage = the image slice, k halves BHS apart
    constexpr int AHS = BM * 16 + 128, BHS = BN * 16 + 128;
    constexpr int APL = AK ? BK * RSA * 2 : 2 * AHS, BPL = 2 * BHS;
    // bytes of one ring stage: three planes of each operand (+4 KB: where the staging threads put the chunks that do not
    // exist, so that their stores need no branch)
    constexpr int STG = 3 * (APL + BPL) + 4096;
    constexpr int NCB = (6 * BN + 255) / 256;               // 16-byte chunks of the B stage per staging thread
    static_assert(3 * STG == bg_ws_ring_bytes<NT, AK>(), "ring size");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = by * BM, n0 = bx * BN;
    const int nsplit = g.ksplit > 1 ? g.ksplit : 1;
    const int bz = bzz / nsplit, kz = bzz - bz * nsplit;
    int Mb = g.M;
    long long rowbase = 0;
    if (g.row_off) { rowbase = g.row_off[bz]; Mb = g.row_off[bz + 1] - (int)rowbase; }
    if (m0 >= Mb) return;
    int kb = 0, ke = g.Kd;                                  // kb: a multiple of 16 (the image is cut in 16-k steps from k = 0)
    if (g.tri == 1) ke = min(g.Kd, n0 + BN);
    else if (g.tri == 2) kb = min(g.Kd, n0) & ~15;
    if (nsplit > 1) {
        const int chunk = ((g.Kd + nsplit - 1) / nsplit + BK - 1) / BK * BK;
        kb = min(g.Kd, kz * chunk);
        ke = min(g.Kd, kb + chunk);
    }
    const int spb = (ke - kb + BK - 1) / BK;
    const int nb = g.inner > 0 ? min(g.inner, g.inner_total - bz * g.inner) : 1;
    const int total = nb * spb;
    const int total3 = (total + 2) / 3 * 3;            // barriers after the first one, both roles

    if (wave >= 4) {
        // ------------------------------------------------ staging waves ------------------------------------------------
        // (priority: the multiplying waves are the older ones and win the arbitration for the vector issue port otherwise)
        if (!BG_DBG(256)) __builtin_amdgcn_s_setprio(3);
        const int pt = tid - 256;
        const float* A0 = g.A + (AK ? rowbase : rowbase * g.lda);
        const float* pro = PRO == 1 ? g.a_sub : (PRO == 2 ? g.a_rscale : (PRO >= 3 ? g.a_kscale : nullptr));
        const long long s_pro = PRO == 1 ? g.s_asub : (PRO == 2 ? g.s_ars : (PRO >= 3 ? g.s_aks : 0));
        const bool vecR = PRO != 4 || (al16(g.a_rsub) && (g.s_arsub & 3) == 0);
        const bool vecA = al16(A0) && ((g.lda | g.sA) & 3) == 0;
        const bool vecP = PRO == 0 || (al16(pro) && (s_pro & 3) == 0);
        // A pieces as in bgemm_kernel; a_at: where a piece goes inside a plane (units of 8 bytes)
        int a_r[2], a_k[2], a_at[2];
        bool a_ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = pt + 256 * u;
            if (AK) { a_k[u] = idx >> 5; a_r[u] = 4 * (idx & 31); } else { a_r[u] = idx >> 2; a_k[u] = 4 * (idx & 3); }
            a_ok[u] = m0 + a_r[u] < Mb;
            a_at[u] = AK ? (a_k[u] * RSA + a_r[u]) >> 2 : ((a_k[u] >> 3) * AHS + a_r[u] * 16 + ((a_k[u] >> 2) & 1) * 8) >> 3;
        }
        // B chunks: chunk q = (plane-and-half ph = q / BN, column c = q % BN): image offset and LDS offset, fixed for the launch
        unsigned b_go[NCB], b_lo[NCB];
        bool b_has[NCB];
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            const int q = pt + 256 * u;
            const int ph = q / BN, c = q - ph * BN;
            b_has[u] = q < 6 * BN;
            b_go[u] = b_has[u] ? ((unsigned)ph * ncolp + n0 + c) * 16u : 0u;
            b_lo[u] = b_has[u] ? (unsigned)(3 * APL + (ph >> 1) * BPL + (ph & 1) * BHS + c * 16) : (unsigned)(STG - 4096 + pt * 16);
        }
        const size_t img_step = bimg_step_bytes(ncolp);
        auto split_store_a = [&](unsigned char* sa, const float4* va) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint32_t lo[3], hi[3];
                split_pair(va[u].x, va[u].y, lo[0], lo[1], lo[2]);
                split_pair(va[u].z, va[u].w, hi[0], hi[1], hi[2]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) reinterpret_cast<uint2*>(sa + pl * APL)[a_at[u]] = make_uint2(lo[pl], hi[pl]);
            }
        };
        // Fast route (decided once per workgroup): aligned A, every 16-byte piece entirely inside or entirely outside the
        // operand (k range and row range), byte offsets that fit 32 bits.  One straight-line load routine, no merge of
        // register values with another route: the loads of three steps stay in flight.
        const long long extA = AK ? (long long)g.Kd * g.lda : (long long)Mb * g.lda;
        const bool fast = vecA && vecP && vecR && (!AK || (Mb & 3) == 0 || m0 + BM <= Mb) && (AK || ((kb | ke) & 3) == 0) &&
                          extA < (1ll << 29);
        if (fast) {
            // byte offsets of the pieces from the operand's place at (batch, k = kb); pieces outside the row range read piece 0
            unsigned a_bo[2], a_b0[2], p_bo[2], r_bo[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a_bo[u] = a_ok[u] ? 4u * (unsigned)(AK ? a_k[u] * g.lda + m0 + a_r[u] : (m0 + a_r[u]) * g.lda + a_k[u]) : 0u;
                a_b0[u] = a_ok[u] ? 4u * (unsigned)(AK ? m0 + a_r[u] : (m0 + a_r[u]) * g.lda) : 0u;    // the same piece at k = kb
                // prologue vector: along k (PRO 1, 3, 4) or along the rows (PRO 2); PRO 4: second vector along the rows
                p_bo[u] = PRO == 2 ? 4u * (unsigned)(a_ok[u] ? m0 + a_r[u] : 0) : 4u * (unsigned)a_k[u];
                r_bo[u] = 4u * (unsigned)(a_ok[u] ? m0 + a_r[u] : 0);
            }
            const unsigned a_kstep = 4u * (unsigned)(AK ? g.lda : 1);
            float4 ra[3][2], pv[3][2], pw[3][2];
            i32x4 rb[3][NCB];
            int klim[3] = {0, 0, 0};                 // ke - k0 of the step a register stage holds
            int ld_bi = 0, ld_ks = 0, ld_n = 0;      // (inner batch, k step, index) of the next step to load: steps are loaded in order
            const char* run_A = nullptr; const unsigned char* run_B = nullptr; const unsigned char* run_B0 = nullptr;
            const char* run_pb = nullptr; const char* run_pr = nullptr;
            auto set_batch = [&](long long b) {
                run_A = reinterpret_cast<const char*>(A0 + b * g.sA + (AK ? (long long)kb * g.lda : (long long)kb));
                run_B0 = bimg + ((size_t)(g.sB ? b : 0) * nsteps + (kb >> 4)) * img_step;
                run_B = run_B0;
                run_pb = PRO ? reinterpret_cast<const char*>(pro + b * s_pro + (PRO == 2 ? 0 : kb)) : nullptr;
                run_pr = PRO == 4 ? reinterpret_cast<const char*>(g.a_rsub + b * g.s_arsub) : nullptr;
            };
            set_batch((long long)bz * (g.inner > 0 ? g.inner : 1));
            auto gload = [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                // (behind the last step the last step is loaded again and never handed over: every path through the loop
                // issues the same number of loads, which lets the compiler wait with vmcnt(two stages) instead of vmcnt(0))
                if (BG_DBG(4)) return;
                // running pointers: the places of this step's operands are those of the previous step + one k step; only a new
                // inner batch computes them from scratch (scalar work only, the same loads follow on both paths)
                const int k0 = kb + ld_ks * BK;
                const unsigned kofs = BG_DBG(128) ? 0u : (unsigned)(ld_ks * BK);
                klim[R] = ke - k0;
                const char* Ab = run_A;
                const unsigned char* Bi = BG_DBG(64) ? run_B0 : run_B;
                const char* pb = run_pb;
                const char* pr = run_pr;
                if (++ld_n < total) {
                    run_B += img_step;
                    if (++ld_ks == spb) {
                        ld_ks = 0;
                        ++ld_bi;
                        set_batch((long long)bz * (g.inner > 0 ? g.inner : 1) + ld_bi);
                    }
                }
                // a piece whose k lies behind ke reads the piece of its row at k = kb (inside the operand whatever the length of
                // the tile's k range: kb < ke), masked at the hand-over
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned at = a_k[u] < klim[R] ? a_bo[u] + kofs * a_kstep : a_b0[u];
                    ra[R][u] = *reinterpret_cast<const float4*>(Ab + at);
                }
#pragma unroll
                for (int u = 0; u < NCB; ++u) rb[R][u] = *reinterpret_cast<const i32x4*>(Bi + b_go[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (PRO == 2) {
                        if (AK) pv[R][u] = *reinterpret_cast<const float4*>(pb + p_bo[u]);
                        else { const float sc = *reinterpret_cast<const float*>(pb + p_bo[u]); pv[R][u] = make_float4(sc, sc, sc, sc); }
                    } else if (PRO != 0) {
                        const unsigned ko = a_k[u] < klim[R] ? kofs : 0u;
                        if (AK) { const float sc = *reinterpret_cast<const float*>(pb + (p_bo[u] + 4u * ko)); pv[R][u] = make_float4(sc, sc, sc, sc); }
                        else pv[R][u] = *reinterpret_cast<const float4*>(pb + (p_bo[u] + 4u * ko));
                        if (PRO == 4) pw[R][u] = *reinterpret_cast<const float4*>(pr + r_bo[u]);
                    }
                }
            };
            auto sstore = [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                unsigned char* stage = ring + R * STG;       // step s lives in register stage and ring stage s % 3
                float4 va[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float4 v = ra[R][u];
                    if (PRO == 1) { v.x -= pv[R][u].x; v.y -= pv[R][u].y; v.z -= pv[R][u].z; v.w -= pv[R][u].w; }
                    if (PRO == 4) { v.x -= pw[R][u].x; v.y -= pw[R][u].y; v.z -= pw[R][u].z; v.w -= pw[R][u].w; }
                    if (PRO >= 2) { v.x *= pv[R][u].x; v.y *= pv[R][u].y; v.z *= pv[R][u].z; v.w *= pv[R][u].w; }
                    va[u] = (a_ok[u] && a_k[u] < klim[R]) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (!BG_DBG(2)) split_store_a(stage, va);
#pragma unroll
                for (int u = 0; u < NCB; ++u)
                    if (!BG_DBG(8)) *reinterpret_cast<i32x4*>(ring + R * STG + b_lo[u]) = rb[R][u];
            };
            if (total > 0) {
                // (scheduling barriers: the loads must be ISSUED in step order -- the waits count loads issued later)
                gload(bg_ic<0>());
                __builtin_amdgcn_sched_barrier(0);
                gload(bg_ic<1>());
                __builtin_amdgcn_sched_barrier(0);
                gload(bg_ic<2>());
                __builtin_amdgcn_sched_barrier(0);
                sstore(bg_ic<0>());
                gload(bg_ic<0>());
                __builtin_amdgcn_sched_barrier(0);
                if (1 < total) sstore(bg_ic<1>());
                gload(bg_ic<1>());
                __builtin_amdgcn_sched_barrier(0);
            }
            BG_LDS_BARRIER();
            // during step t: fill ring stage (t + 2) % 3 with step t + 2, then load step t + 5 into the freed registers.  The loop
            // runs over total3 steps (total rounded up to a multiple of 3; the extra steps only meet the barrier) and has no
            // early exit: every path to a hand-over has issued the same loads in the same order
#ifdef BG_STAMPS
            const unsigned long long pst1 = BG_CLK();
#endif
#ifdef BG_STAMPS
            unsigned long long tss = 0, tgl = 0, c0_, c1_;
#define BG_T(acc_, stmt)  do { c0_ = BG_CLK(); stmt; c1_ = BG_CLK(); acc_ += c1_ - c0_; } while (0)
#else
#define BG_T(acc_, stmt)  do { stmt; } while (0)
#endif
            for (int t = 0; t < total3; t += 3) {
                BG_T(tss, if (t + 2 < total) sstore(bg_ic<2>()));
                BG_T(tgl, gload(bg_ic<2>()));
                BG_BARRIER_TIMED(bwait);
                BG_T(tss, if (t + 3 < total) sstore(bg_ic<0>()));
                BG_T(tgl, gload(bg_ic<0>()));
                BG_BARRIER_TIMED(bwait);
                BG_T(tss, if (t + 4 < total) sstore(bg_ic<1>()));
                BG_T(tgl, gload(bg_ic<1>()));
                BG_BARRIER_TIMED(bwait);
            }
#ifdef BG_STAMPS
            if ((g.debug & 32) && pt == 0 && blockIdx.x == 7 && bzz == 1) {
                const unsigned long long pst2 = BG_CLK();
                printf("stager  by=%d total=%d: start %llu prime %llu  loop %llu  per step: all %llu  hand-over %llu  loads %llu  barrier %llu\n", by, total,
                       st0 % 100000000ull, pst1 - st0, pst2 - pst1, (pst2 - pst1) / (unsigned long long)total3, tss / (unsigned long long)total3,
                       tgl / (unsigned long long)total3, bwait / (unsigned long long)total3);
            }
#endif
            return;
        }
        // General route (misaligned or ragged A): element-wise loads, one step at a time, same ring protocol
        auto stage_step = [&](int step) {
            const int bi = step / spb;
            const long long b = (long long)bz * (g.inner > 0 ? g.inner : 1) + bi;
            const int k0 = kb + (step - bi * spb) * BK;
            const float* Ab = A0 + b * g.sA;
            const unsigned char* Bi = bimg + ((size_t)(g.sB ? b : 0) * nsteps + (k0 >> 4)) * img_step;
            const float* pb = PRO ? pro + b * s_pro : nullptr;
            const float* pr = PRO == 4 ? g.a_rsub + b * g.s_arsub : nullptr;
            unsigned char* stage = ring + (step % 3) * STG;
            float4 va[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gm = m0 + a_r[u], gk = k0 + a_k[u];
                float4 v;
                if (AK) {                                           // 4 consecutive rows of one k
                    const int nv = (gk < ke) ? Mb - gm : 0;
                    v = ld4(Ab + (long long)gk * g.lda + gm, nv, vecA);
                    if (nv > 0) {
                        if (PRO == 1) { const float sc = pb[gk]; v.x -= sc; v.y -= sc; v.z -= sc; v.w -= sc; }
                        if (PRO == 3) { const float sc = pb[gk]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                        if (PRO == 2) { const float4 sc = ld4(pb + gm, nv, vecP); v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
                        if (PRO == 4) {
                            const float4 r4 = ld4(pr + gm, nv, vecR);
                            const float sc = pb[gk];
                            v.x = (v.x - r4.x) * sc; v.y = (v.y - r4.y) * sc; v.z = (v.z - r4.z) * sc; v.w = (v.w - r4.w) * sc;
                        }
                        if (nv < 4) { if (nv < 2) v.y = 0.f; if (nv < 3) v.z = 0.f; v.w = 0.f; }
                    }
                } else {                                            // 4 consecutive k of one row
                    const int nv = (gm < Mb) ? ke - gk : 0;
                    v = ld4(Ab + (long long)gm * g.lda + gk, nv, vecA);
                    if (nv > 0) {
                        if (PRO == 1) { const float4 sc = ld4(pb + gk, nv, vecP); v.x -= sc.x; v.y -= sc.y; v.z -= sc.z; v.w -= sc.w; }
                        if (PRO == 3) { const float4 sc = ld4(pb + gk, nv, vecP); v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
                        if (PRO == 2) { const float sc = pb[gm]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                        if (nv < 4) { if (nv < 2) v.y = 0.f; if (nv < 3) v.z = 0.f; v.w = 0.f; }
                    }
                }
                va[u] = v;
            }
            split_store_a(stage, va);
#pragma unroll
            for (int u = 0; u < NCB; ++u)
                if (b_has[u]) *reinterpret_cast<i32x4*>(stage + b_lo[u]) = *reinterpret_cast<const i32x4*>(Bi + b_go[u]);
        };
        if (0 < total) stage_step(0);
        if (1 < total) stage_step(1);
        BG_LDS_BARRIER();
        for (int t = 0; t < total3; ++t) {
            if (t + 2 < total) stage_step(t + 2);
            BG_LDS_BARRIER();
        }
        return;
    }

    // ---------------------------------------------------- multiplying waves ----------------------------------------------------
    const int col = lane & 31, half = lane >> 5;
    // lane's place in a transposed read (k-major A): group lane / 16 covers operand rows 16 (group & 1) .., k 8 (group >> 1) ..;
    // lane 4 q + p of the group addresses k row q, rows 4 p .. 4 p + 3
    const int tr_k = 8 * (lane >> 5) + ((lane >> 2) & 3), tr_r = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int a_rd = AK ? (tr_k * RSA + wave * 32 + tr_r) * 2 : half * AHS + (wave * 32 + col) * 16;
    const int b_rd = 3 * APL + half * BHS + col * 16;
    auto aload = [&](const unsigned char* stage, i32x4* dst) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (AK) dst[pl] = lds_tr_frag(stage + pl * APL, a_rd, RSA * 2);
            else dst[pl] = *reinterpret_cast<const i32x4*>(stage + pl * APL + a_rd);
        }
    };
    auto bload = [&](const unsigned char* stage, int t, i32x4* dst) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const i32x4*>(stage + pl * BPL + b_rd + 512 * t);
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // sub-tile t (columns n0 + 32 t ..) takes part in a step unless opB vanishes there (triangular operands) or the columns do
    // not exist: t_lo <= t < t_hi, from the step's k0
    const int t_end = min(NT, (g.N - n0 + 31) >> 5);
    BG_LDS_BARRIER();
#ifdef BG_STAMPS
    const unsigned long long st1 = BG_CLK();
#endif
    i32x4 af[3], bf[2][3], afn[3], bfn[3];
    if (total > 0) { aload(ring, af); bload(ring, 0, bf[0]); }
    int slot = 0, ks = 0;
    for (int step = 0; step < total; ++step) {
        const int k0 = kb + ks * BK;
        if (++ks == spb) ks = 0;
        const int t_lo = g.tri == 1 ? max(0, (k0 - n0) >> 5) : 0;                       // live iff k0 < n0 + 32 t + 32
        const int t_hi = g.tri == 2 ? min(t_end, (k0 + BK - n0 + 31) >> 5) : t_end;     // live iff k0 + 16 > n0 + 32 t
        const unsigned char* stage = ring + slot * STG;
        slot = slot == 2 ? 0 : slot + 1;
        const unsigned char* next = ring + slot * STG;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // (fragments of a sub-tile that does not take part are not read: LDS bandwidth is what paces this role)
            if (t + 1 < NT) { if (t + 1 >= t_lo && t + 1 < t_hi) bload(stage, t + 1, bf[(t + 1) & 1]); }
            else if (step + 1 < total) { aload(next, afn); bload(next, 0, bfn); }
            if (t >= t_lo && t < t_hi && !BG_DBG(1)) {
                const i32x4* b = bf[t & 1];
                // smallest partial products first
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(af[2]), "v"(b[0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(af[0]), "v"(b[2]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(af[1]), "v"(b[1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(af[1]), "v"(b[0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(af[0]), "v"(b[1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(af[0]), "v"(b[0]));
            }
        }
        BG_BARRIER_TIMED(bwait);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { af[pl] = afn[pl]; bf[0][pl] = bfn[pl]; }
    }
    for (int step = total; step < total3; ++step) BG_LDS_BARRIER();
#ifdef BG_STAMPS
    const unsigned long long st2 = BG_CLK();
#endif
    // the last MFMA must have retired before its accumulators are read (inline asm: no automatic hazard nops)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (g.rowsq != nullptr) {
        // lane (col, half) holds rows (r & 3) + 8 (r >> 2) + 4 half of its column in every sub-tile: square-sum over the
        // sub-tiles, then over the 32 lanes of the half on the DPP network (four steps inside a row of 16, two v_readlane
        // per half); lane i of the wave collects the sum of row i: one 128-byte store
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float z = g.alpha * acc[t][r];
                if (n0 + 32 * t + col < g.N) v = fmaf(z, z, v);
            }
            v += gmmvi_dpp<0xB1>(v);
            v += gmmvi_dpp<0x4E>(v);
            v += gmmvi_dpp<0x141>(v);
            v += gmmvi_dpp<0x140>(v);
            const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
                             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
            const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
                             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
            const int i0 = (r & 3) + 8 * (r >> 2);
            mine = lane == i0 ? s0 : (lane == i0 + 4 ? s1 : mine);
        }
        float* rq = g.rowsq + (long long)bx * g.rs_tile + (long long)bzz * g.rs_batch + rowbase;
        const int i = m0 + wave * 32 + lane;
        if (lane < 32 && i < Mb) rq[i] = mine;
    }
    if (g.no_store) return;
    float* Cb = g.C + (long long)bzz * g.sC + rowbase * g.ldc;
    const float* bias = g.c_bias ? g.c_bias + (long long)bz * g.s_cb : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = n0 + 32 * t + col;
        if (j >= g.N) continue;
        const float bj = bias ? bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < Mb) {
                float* p = Cb + (long long)i * g.ldc + j;
                float v = fmaf(g.alpha, acc[t][r], bj);
                if (g.accumulate) v += *p;
                *p = v;
            }
        }
    }
#ifdef BG_STAMPS
    if ((g.debug & 32) && tid == 0 && blockIdx.x == 7 && bzz == 1) {
        const unsigned long long st3 = BG_CLK();
        printf("multipl by=%d total=%d: start %llu  +prime %llu  +loop %llu (barrier %llu)  +epilogue %llu\n", by, total, st0 % 100000000ull, st1 - st0,
               st2 - st1, bwait, st3 - st2);
    }
#endif
}

// Persistent launch: one workgroup per CU walks over the tiles (row tile fastest, then column tile, then batch / k-split slab).
// The stagers leave a tile before the multipliers (which still write their results) and start loading the next one: the
// dispatch gap between two workgroups of a 150 KB-LDS kernel and the first loads of a tile are no longer exposed
// (measured per tile of the whitening launch before: 49 k cycles in the step loop, 36 k around it).
template <int NT, int AK, int PRO>
__global__ __launch_bounds__(512, 1) void bgemm_ws_kernel(BG g, const unsigned char* __restrict__ bimg, int ncolp, int nsteps,
                                                          int ntx, int nty, int ntz) {
    __shared__ __align__(16) unsigned char ring[bg_ws_ring_bytes<NT, AK>()];
    const int G = gridDim.x;
    const int ntiles = ntx * nty * ntz;
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int by = tile % nty, rest = tile / nty;
        const int bx = rest % ntx, bzz = rest / ntx;
        bgemm_ws_tile<NT, AK, PRO>(g, bimg, ncolp, nsteps, ring, bx, by, bzz);
    }
}

BG bg_zero() {
    BG g;
    memset(&g, 0, sizeof(g));
    g.alpha = 1.f;
    return g;
}

// GMMVI_BLOCKED_F32=1 keeps every contraction on the f32 matrix-core instruction (v_mfma_f32_32x32x2_f32); the default is
// the split-operand route above
bool bgemm_split_enabled() {
    static const bool off = getenv("GMMVI_BLOCKED_F32") != nullptr && atoi(getenv("GMMVI_BLOCKED_F32")) != 0;
    return !off;
}

// Route of one launch.  The split route pays a pre-pass over B and runs one 512-thread workgroup per CU: measured at K = 64,
// N = 19 968 (tools/time_blocked.py, whitening / whitening + gradient / Stein, us): D = 64: 269 / 609 / 348 against the f32
// route's 188 / 446 / 442; D = 128: 415 / 1132 / 842 against 387 / 987 / 1266; D = 192: 644 / 1649 / 1057 against 950 / 2114 /
// 1731 -- it wins from 160 result columns on, and for long contractions (the Stein sums over the samples) at every width.
bool bgemm_use_split(int n, int kd) { return bgemm_split_enabled() && (n >= 160 || kd >= 512); }

// tile widths of the two routes: f32 route 32 x {3, 4, 5}, split route 32 x {3, 4, 5, 6, 8, 10}: the least padded total width,
// ties go to the wider tile.  -> NT, *tiles = number of column tiles
int bgemm_tile_width(int n, bool split, int* tiles) {
    static const int f32_nt[] = {5, 4, 3}, split_nt[] = {10, 8, 6, 5, 4, 3};
    const int* cand = split ? split_nt : f32_nt;
    const int nc = split ? 6 : 3;
    int nt = cand[0], best = 1 << 30;
    for (int i = 0; i < nc; ++i) {
        const int w = (n + 32 * cand[i] - 1) / (32 * cand[i]) * 32 * cand[i];
        if (w < best) { best = w; nt = cand[i]; }
    }
    *tiles = best / (32 * nt);
    return nt;
}

// number of column tiles bgemm_launch will use for a result of n columns
int bgemm_col_tiles(int n, int kd) {
    int tiles = 1;
    (void)bgemm_tile_width(n, bgemm_use_split(n, kd), &tiles);
    return tiles;
}

int bimg_reserve(gmmvi_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->bimg_bytes) return GMMVI_OK;
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->bimg) GMMVI_HIP_CHECK(ctx, hipFree(ctx->bimg));
    ctx->bimg = nullptr;
    ctx->bimg_bytes = 0;
    const size_t want = bytes + bytes / 4;
    GMMVI_HIP_CHECK(ctx, hipMalloc(&ctx->bimg, want));
    ctx->bimg_bytes = want;
    return GMMVI_OK;
}

template <int AK, int BKM, int PRO>
int bgemm_launch(gmmvi_ctx* ctx, const BG& g0, int batches_outer) {
    const bool split = bgemm_use_split(g0.N, g0.Kd);
    int tiles = 1;
    const int nt = bgemm_tile_width(g0.N, split, &tiles);
    dim3 grid(tiles, (g0.M + BM - 1) / BM, batches_outer * (g0.ksplit > 1 ? g0.ksplit : 1));
    if (!split) {
        switch (nt) {
            case 3: hipLaunchKernelGGL((bgemm_kernel<3, AK, BKM, PRO>), grid, dim3(256), 0, ctx->stream, g0); break;
            case 4: hipLaunchKernelGGL((bgemm_kernel<4, AK, BKM, PRO>), grid, dim3(256), 0, ctx->stream, g0); break;
            default: hipLaunchKernelGGL((bgemm_kernel<5, AK, BKM, PRO>), grid, dim3(256), 0, ctx->stream, g0); break;
        }
        GMMVI_LAUNCH_CHECK(ctx);
        return GMMVI_OK;
    }
    static const int dbg = getenv("GMMVI_BG_DEBUG") ? atoi(getenv("GMMVI_BG_DEBUG")) : 0;
    BG g = g0;
    g.debug = dbg;
    // the B image: every batch of B (one if B is shared: sB == 0), every 16-k step of [0, Kd)
    const int ncolp = tiles * nt * 32, nsteps = (g.Kd + BK - 1) / BK;
    const long long nbatch = g.sB ? (g.inner > 0 ? (long long)g.inner_total : (long long)batches_outer) : 1;
    const size_t bytes = (size_t)nbatch * nsteps * bimg_step_bytes(ncolp);
    int rc = bimg_reserve(ctx, bytes);
    if (rc != GMMVI_OK) return rc;
    unsigned char* img = static_cast<unsigned char*>(ctx->bimg);
    hipLaunchKernelGGL(bsplit_image_kernel, dim3(nsteps, (unsigned)nbatch), dim3(256), 0, ctx->stream, g, ncolp, nsteps, img);
    GMMVI_LAUNCH_CHECK(ctx);
    const int ntiles = (int)(grid.x * grid.y * grid.z);
    static const int env_grid = getenv("GMMVI_BG_GRID") ? atoi(getenv("GMMVI_BG_GRID")) : 0;      // experiments: fewer workgroups than CUs
    const int cap = env_grid > 0 ? env_grid : ctx->num_cus;
    const dim3 pgrid(ntiles < cap ? ntiles : cap);
#define BG_WS_LAUNCH(NT_) hipLaunchKernelGGL((bgemm_ws_kernel<NT_, AK, PRO>), pgrid, dim3(512), 0, ctx->stream, g, img, ncolp, nsteps, \
                                             (int)grid.x, (int)grid.y, (int)grid.z)
    switch (nt) {
        case 3: BG_WS_LAUNCH(3); break;
        case 4: BG_WS_LAUNCH(4); break;
        case 5: BG_WS_LAUNCH(5); break;
        case 6: BG_WS_LAUNCH(6); break;
        case 8: BG_WS_LAUNCH(8); break;
        default: BG_WS_LAUNCH(10); break;
    }
#undef BG_WS_LAUNCH
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// the operand layouts / prologues the callers below use
int bgemm(gmmvi_ctx* ctx, const BG& g, int batches_outer) {
    if (g.M <= 0 || g.N <= 0 || batches_outer <= 0) return GMMVI_OK;
    const int pro = g.a_sub ? 1 : (g.a_rscale ? 2 : (g.a_kscale ? (g.a_rsub ? 4 : 3) : 0));
    const int key = g.a_kmajor * 100 + g.b_kmajor * 10 + pro;
    switch (key) {
        case 1: return bgemm_launch<0, 0, 1>(ctx, g, batches_outer);      // whitening: (X - mu) L^-T
        case 0: return bgemm_launch<0, 0, 0>(ctx, g, batches_outer);      // sampling: eps L^T
        case 12: return bgemm_launch<0, 1, 2>(ctx, g, batches_outer);     // gradient: (r Z) L^-1
        case 10: return bgemm_launch<0, 1, 0>(ctx, g, batches_outer);     // A L^-1, R L
        case 114: return bgemm_launch<1, 1, 4>(ctx, g, batches_outer);    // Stein: (X1 - mu1)^T diag(e) G1
        case 100: return bgemm_launch<1, 0, 0>(ctx, g, batches_outer);    // C^T L^-T
        case 110: return bgemm_launch<1, 1, 0>(ctx, g, batches_outer);    // L^T (R L)
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "bgemm: operand layout not instantiated");
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {          // red: >= 16 floats of LDS; result block-uniform
    v = gmmvi_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
// two sums at the price of one (red: >= 32 floats)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = gmmvi_wave_sum(a);
    b = gmmvi_wave_sum(b);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[16 + w] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
    for (int i = 0; i < nw; ++i) { sa += red[i]; sb += red[16 + i]; }
    a = sa; b = sb;
}
// Variants without the leading barrier: the caller guarantees that a barrier separates this call from the previous readers
// of the same `red` words (the tridiagonalisation alternates between two disjoint ranges, with barriers of its own in between).
__device__ __forceinline__ float block_sum_nb(float v, float* red) {
    v = gmmvi_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ void block_sum2_nb(float& a, float& b, float* red) {
    a = gmmvi_wave_sum(a);
    b = gmmvi_wave_sum(b);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[16 + w] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
    for (int i = 0; i < nw; ++i) { sa += red[i]; sb += red[16 + i]; }
    a = sa; b = sb;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = gmmvi_wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = -3.0e38f;
    for (int i = 0; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}

// Forward substitution T x = r with a lower-triangular T, one thread per right-hand side, 32 rows at a time:
// thread t solves for rhs(i, t), i = 0..D-1, knowing that its solution vanishes for i < i0; X[i * ldx + t] receives it
// (coalesced over t).  Per block of 32 rows the rows of T are staged in LDS (Ts: 32 x (D + 4) floats) and the thread keeps
// its 32 partial sums in registers: every solved x is read back from memory once per block (not once per row), the
// multiplies run on broadcast 16-byte LDS reads.  TCM: T is given column-major (T[i][c] at Tm[c * D + i]).
constexpr int TRB = 32;
// row stride of the staged rows: a multiple of 4 (16-byte reads) whose quarter is odd (the matrix-core operand reads walk the
// rows: stride 4 * odd spreads 32 rows over 16 banks)
__host__ __device__ inline int blk_trsm_ld(int D) {
    const int ld = ((D + 3) / 4) * 4 + 4;
    return (ld / 4) % 2 == 0 ? ld + 4 : ld;
}
inline size_t blk_trsm_lds_floats(int D) { return (size_t)TRB * blk_trsm_ld(D) + (size_t)(D + 1) * (TRB + 1); }     // + sums of up to D + 1 right-hand sides

template <bool TCM, class RhsF>
__device__ void blk_trsm(int D, const float* __restrict__ Tm, int nrhs, RhsF rhs, int i0, float* X, int ldx, float* Ts) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, nwave = blockDim.x >> 6, col = lane & 31, half = lane >> 5;
    const int ld = blk_trsm_ld(D);
    float* Ss = Ts + (size_t)TRB * ld;                     // [rhs][33]: what the solved rows contribute to the current block
    for (int I = 0; I < D; I += TRB) {
        const int nbk = min(TRB, D - I), w = I + nbk;
        __syncthreads();
        if (TCM) {
            for (int e = t; e < w * TRB; e += blockDim.x) {
                const int c = e / TRB, r = e % TRB;
                if (r < nbk) Ts[r * ld + c] = (c <= I + r) ? Tm[(size_t)c * D + I + r] : 0.f;
            }
        } else {
            for (int c = t; c < w; c += blockDim.x) {          // row by row: no index divisions, every load independent
#pragma unroll 8
                for (int r = 0; r < nbk; ++r) Ts[r * ld + c] = Tm[(size_t)(I + r) * D + c];
            }
        }
        __syncthreads();
        // what the solved rows contribute to this block, S[r][t] = sum_{c < I} T[I + r][c] x_c(t), as matrix-core tiles (wave w
        // takes the 32-rhs tiles w, w + waves, ..: A operand from the staged rows, B operand = dword loads of X from the L2),
        // handed to the right-hand sides' threads through LDS (this sum as per-thread multiply-adds over
        // broadcast LDS reads was the bulk of the 0.96 M cycles of the solve at D = 300)
        if (I > 0) {
            const int ntile = (nrhs + TRB - 1) / TRB;
            for (int tile = wave; tile < ntile; tile += nwave) {
                const int t0 = TRB * tile;
                const float* pa = Ts + col * ld;                               // rows past nbk: unused results
                const float* pb = X + min(t0 + col, nrhs - 1);
                f32x16 sacc;
#pragma unroll
                for (int j = 0; j < 16; ++j) sacc[j] = 0.f;
                for (int c = 0; c < I; c += 16) {                              // I is a multiple of 32
                    float av[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        av[u] = pa[c + 2 * u + half];
                        bv[u] = pb[(size_t)(c + 2 * u + half) * ldx];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], sacc, 0, 0, 0);
                }
                if (t0 + col < nrhs) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) Ss[(size_t)(t0 + col) * (TRB + 1) + (j & 3) + 8 * (j >> 2) + 4 * half] = sacc[j];
                }
            }
            __syncthreads();
        }
        if (t >= nrhs) continue;
        float acc[TRB];
        if (I + TRB <= i0) {                                   // structurally zero rows
#pragma unroll
            for (int r = 0; r < TRB; ++r)
                if (r < nbk) X[(size_t)(I + r) * ldx + t] = 0.f;
            continue;
        }
#pragma unroll
        for (int r = 0; r < TRB; ++r) acc[r] = (r < nbk) ? rhs(I + r, t) - (I > 0 ? Ss[(size_t)t * (TRB + 1) + r] : 0.f) : 0.f;
#pragma unroll
        for (int r = 0; r < TRB; ++r) {
            if (r < nbk) {
                float sv = acc[r];
#pragma unroll
                for (int q4 = 0; q4 < r; q4 += 4) {                            // 16-byte broadcast reads of the pivot row
                    const float4 pr = *reinterpret_cast<const float4*>(Ts + r * ld + I + q4);
                    sv = fmaf(-pr.x, acc[q4], sv);
                    if (q4 + 1 < r) sv = fmaf(-pr.y, acc[q4 + 1], sv);
                    if (q4 + 2 < r) sv = fmaf(-pr.z, acc[q4 + 2], sv);
                    if (q4 + 3 < r) sv = fmaf(-pr.w, acc[q4 + 3], sv);
                }
                acc[r] = sv / Ts[r * ld + I + r];
                X[(size_t)(I + r) * ldx + t] = acc[r];
            }
        }
    }
    __syncthreads();
}

// Cholesky factorisation A = C C^T of the symmetric matrix a(i, j) (i >= j read), left-looking over blocks of 32 columns,
// one thread per row; the factor is built column-major (W[c * D + i] = C[i][c]: thread i walks coalesced rows of W).
// Per column block J the sums over the finished columns, S[i][r] = sum_{c < J} C[i][c] C[J + r][c] for the rows i >= J, are
// matrix-core tiles (v_mfma_f32_32x32x2_f32: wave w takes the 32-row tiles w, w + waves, ..; both operands are dword loads of
// W straight from the L2, sixteen k in flight) handed to the row threads through LDS (Sm: D x 33; this accumulation as
// per-thread multiply-adds over broadcast LDS reads was 0.6 M of the factorisation's 1.4 M cycles at D = 300); then every
// thread finishes its 32 entries, one wave factorises the 32 x 32 diagonal block in LDS (Dg), the other rows solve against
// it.  LDS: blk_chol_lds_floats(D).  Returns false (block-uniform) on a non-positive or non-finite pivot.
constexpr int DGS = TRB + 4;                            // row stride of the diagonal block in LDS: 16-byte aligned rows
inline size_t blk_chol_lds_floats(int D) { return (size_t)(((D * (TRB + 1) + 3) / 4) * 4) + TRB * DGS + 4; }

template <class ElemF>
__device__ bool blk_cholesky(int D, ElemF a, float* W, float* lds) {
    float* Sm = lds;                                   // [i][33], rows J <= i < D
    float* Dg = lds + (size_t)(((D * (TRB + 1) + 3) / 4) * 4);          // [32][DGS]
    int* fail = reinterpret_cast<int*>(Dg + TRB * DGS);
    const int i = threadIdx.x;
    const int lane = i & 63, wave = i >> 6, nwave = blockDim.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    if (i == 0) *fail = 0;
    for (int J = 0; J < D; J += TRB) {
        const int nbk = min(TRB, D - J);
        __syncthreads();
        // S tiles: result row m = r (A operand: C[J + r][c]), result column n = matrix row i (B operand: C[i][c]); the lane
        // holds column n = col and the rows m = (j & 3) + 8 (j >> 2) + 4 half of its 16 registers
        const int ntile = (D - J + TRB - 1) / TRB;
        for (int tile = wave; J > 0 && tile < ntile; tile += nwave) {
            const int i0 = J + TRB * tile;
            const float* pa = W + min(J + col, D - 1);             // rows past the matrix read its last row (never used)
            const float* pb = W + min(i0 + col, D - 1);
            f32x16 sacc;
#pragma unroll
            for (int j = 0; j < 16; ++j) sacc[j] = 0.f;
            for (int c = 0; c < J; c += 16) {                      // J is a multiple of 32
                float av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = pa[(size_t)(c + 2 * u + half) * D];
                    bv[u] = pb[(size_t)(c + 2 * u + half) * D];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], sacc, 0, 0, 0);
            }
            if (i0 + col < D) {
#pragma unroll
                for (int j = 0; j < 16; ++j) Sm[(size_t)(i0 + col) * (TRB + 1) + (j & 3) + 8 * (j >> 2) + 4 * half] = sacc[j];
            }
        }
        __syncthreads();
        float acc[TRB];
        const bool mine = i >= J && i < D;
        if (mine) {
#pragma unroll
            for (int r = 0; r < TRB; ++r)
                acc[r] = (r < nbk && J + r <= i) ? a(i, J + r) - (J > 0 ? Sm[(size_t)i * (TRB + 1) + r] : 0.f) : 0.f;
            if (i < J + nbk) {
#pragma unroll
                for (int r = 0; r < TRB; ++r) Dg[(i - J) * DGS + r] = acc[r];
            }
        }
        __syncthreads();
        if (i < 32) {
            // half of wave 0: 32 x 32 Cholesky, lane = row.  The lane keeps ITS row in registers; the pivot row r (complete
            // after step r - 1: every step publishes the new column) comes from LDS as 16-byte broadcast reads; the sums run
            // over q = 0, 1, .. as before (bit-identical), but with static indices: no dependent LDS read per term (that loop
            // was 57 k cycles a block: more than half of the factorisation at D = 300)
            const int l = i;
            float rw[TRB];
#pragma unroll
            for (int c = 0; c < TRB; ++c) rw[c] = Dg[l * DGS + c];
            bool bad = false;
#pragma unroll
            for (int r = 0; r < TRB; ++r) {
                if (r < nbk && !bad) {
                    float sv = rw[r];
#pragma unroll
                    for (int q4 = 0; q4 < r; q4 += 4) {
                        const float4 pr = *reinterpret_cast<const float4*>(Dg + r * DGS + q4);
                        sv = fmaf(-rw[q4], pr.x, sv);
                        if (q4 + 1 < r) sv = fmaf(-rw[q4 + 1], pr.y, sv);
                        if (q4 + 2 < r) sv = fmaf(-rw[q4 + 2], pr.z, sv);
                        if (q4 + 3 < r) sv = fmaf(-rw[q4 + 3], pr.w, sv);
                    }
                    const float pv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), r));
                    if (!(pv > 0.f) || !(pv < FLT_MAX)) {
                        bad = true;
                    } else {
                        const float d = sqrtf(pv);
                        rw[r] = (l == r) ? d : sv / d;
                        if (l >= r && l < nbk) Dg[l * DGS + r] = rw[r];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                }
            }
            if (bad && i == 0) *fail = 1;
        }
        __syncthreads();
        if (*fail) return false;
        if (mine) {
            if (i < J + nbk) {
                const int l = i - J;
#pragma unroll
                for (int r = 0; r < TRB; ++r)
                    if (r <= l) W[(size_t)(J + r) * D + i] = Dg[l * DGS + r];
            } else {
#pragma unroll
                for (int r = 0; r < TRB; ++r) {
                    if (r < nbk) {
                        float sv = acc[r];
#pragma unroll
                        for (int q4 = 0; q4 < r; q4 += 4) {                    // 16-byte broadcast reads of the pivot row
                            const float4 pr = *reinterpret_cast<const float4*>(Dg + r * DGS + q4);
                            sv = fmaf(-acc[q4], pr.x, sv);
                            if (q4 + 1 < r) sv = fmaf(-acc[q4 + 1], pr.y, sv);
                            if (q4 + 2 < r) sv = fmaf(-acc[q4 + 2], pr.z, sv);
                            if (q4 + 3 < r) sv = fmaf(-acc[q4 + 3], pr.w, sv);
                        }
                        acc[r] = sv / Dg[r * DGS + r];
                        W[(size_t)(J + r) * D + i] = acc[r];
                    }
                }
            }
        }
    }
    __syncthreads();
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// component blocks: [mu | const | pad | L^-1]
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(704) void blk_pack_kernel(int family, float nu, int D, const float* __restrict__ means,
                                                       const float* __restrict__ chols, float* __restrict__ packed,
                                                       size_t ps, int lo, float* __restrict__ inv_out) {
    extern __shared__ __align__(16) float dyn[];
    __shared__ float red[16];
    const int k = blockIdx.x, t = threadIdx.x;
    const float* L = chols + (size_t)k * D * D;
    float* out = packed + (size_t)k * ps;
    float* Linv = out + lo;
    if (t < D) out[t] = means[(size_t)k * D + t];
    const float s = block_sum(t < D ? logf(L[(size_t)t * D + t]) : 0.f, red);
    if (t == 0) {
        out[D] = (family == GMMVI_GAUSS)
                     ? -s - 0.5f * D * 1.8378770664093453f
                     : lgammaf(0.5f * (nu + D)) - lgammaf(0.5f * nu) - 0.5f * D * logf(nu * 3.14159265358979f) - s;
        for (int i = D + 1; i < lo; ++i) out[i] = 0.f;
    }
    blk_trsm<false>(D, L, D, [](int i, int tt) { return i == tt ? 1.f : 0.f; }, t, Linv, D, dyn);
    if (inv_out != nullptr) {
        float* o = inv_out + (size_t)k * D * D;
        for (int e = t; e < D * D; e += blockDim.x) o[e] = Linv[e];
    }
}

__global__ __launch_bounds__(704) void blk_cholesky_kernel(int D, const float* __restrict__ covs, float* __restrict__ W,
                                                           float* __restrict__ chols, int32_t* __restrict__ ok) {
    extern __shared__ __align__(16) float dyn[];
    const int k = blockIdx.x, t = threadIdx.x;
    const float* A = covs + (size_t)k * D * D;
    float* Wk = W + (size_t)k * D * D;
    const bool good = blk_cholesky(D, [A, D](int i, int j) { return A[(size_t)i * D + j]; }, Wk, dyn);
    __syncthreads();
    float* o = chols + (size_t)k * D * D;
    for (int e = t; e < D * D; e += blockDim.x) {
        const int i = e / D, j = e % D;
        o[e] = good ? (j <= i ? Wk[(size_t)j * D + i] : 0.f) : __builtin_nanf("");
    }
    if (t == 0 && ok) ok[k] = good ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// densities
// ---------------------------------------------------------------------------------------------------------------------------
// q[row] = sum over the column tiles of the |z|^2 partials the whitening launch left, and the component log-density
__global__ __launch_bounds__(256) void blk_qfinish_kernel(int family, float nu, int D, int N, long long rows, int tiles,
                                                          const float* __restrict__ qpart, const float* __restrict__ packed,
                                                          size_t ps, float* __restrict__ q, float* __restrict__ ld) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s = qpart[row];
    for (int t = 1; t < tiles; ++t) s += qpart[(long long)t * rows + row];
    const float c = packed[(size_t)(row / N) * ps + D];
    if (q) q[row] = s;
    if (ld) ld[row] = (family == GMMVI_GAUSS) ? fmaf(-0.5f, s, c) : c - 0.5f * (nu + D) * log1pf(s / nu);
}

__global__ __launch_bounds__(256) void blk_lse_kernel(int K, int N, const float* __restrict__ ld, const float* __restrict__ logw,
                                                      const float* __restrict__ logw2, float* __restrict__ lp,
                                                      float* __restrict__ lp2) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float m = -3.0e38f, s = 0.f, m2 = -3.0e38f, s2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const float v = ld[(size_t)k * N + n];
        const float a = v + logw[k];
        const float mn = fmaxf(m, a);
        s = fmaf(s, __expf(m - mn), __expf(a - mn));
        m = mn;
        if (logw2) {
            const float a2 = v + logw2[k];
            const float mn2 = fmaxf(m2, a2);
            s2 = fmaf(s2, __expf(m2 - mn2), __expf(a2 - mn2));
            m2 = mn2;
        }
    }
    if (lp) lp[n] = m + __logf(s);
    if (lp2 && logw2) lp2[n] = m2 + __logf(s2);
}

// responsibilities times the family's gradient coefficient: rw[kb][n] = exp(logw_k + ld[k][n] - lp[n]) * coef
__global__ __launch_bounds__(256) void blk_resp_kernel(int family, float nu, int D, int N, int kn, const float* __restrict__ ld,
                                                       const float* __restrict__ logw, const float* __restrict__ lp,
                                                       const float* __restrict__ q, float* __restrict__ rw) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)kn * N) return;
    const int kb = (int)(e / N), n = (int)(e % N);
    const float r = __expf(logw[kb] + ld[e] - lp[n]);
    const float coef = (family == GMMVI_GAUSS) ? -1.f : -(nu + (float)D) / (nu + q[e]);
    rw[e] = r * coef;
}

// dst[b][e] (+)= sum_s src[b * S + s][e]: partial results of a split contraction, summed in fixed order
__global__ __launch_bounds__(256) void blk_sum_slabs_kernel(int S, size_t slab, const float* __restrict__ src,
                                                            float* __restrict__ dst, int accumulate) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= slab) return;
    const float* p = src + (size_t)blockIdx.y * S * slab + e;
    float a = accumulate ? dst[(size_t)blockIdx.y * slab + e] : 0.f;
    for (int s = 0; s < S; ++s) a += p[(size_t)s * slab];
    dst[(size_t)blockIdx.y * slab + e] = a;
}

__global__ void blk_mapping_kernel(int K, const int32_t* __restrict__ offsets, int32_t* __restrict__ mapping) {
    const int k = blockIdx.x;
    const int begin = offsets[k], end = offsets[k + 1];
    for (int n = begin + blockIdx.y * blockDim.x + threadIdx.x; n < end; n += gridDim.y * blockDim.x) mapping[n] = k;
}

size_t z_budget_floats() {
    // scratch budget for the whitened samples of a component chunk (default 4 GiB); read per call so tests can shrink it
    const char* e = getenv("GMMVI_BLOCKED_ZBYTES");
    return e ? (size_t)atoll(e) / 4 : ((size_t)4 << 30) / 4;
}

// Z[kb][n][0:D] = (x_n - mu_k) L_k^-T for the components k0 .. k0 + kn - 1 (row stride ldz)
// qpart != nullptr: the launch also leaves |z|^2 partials, one per column tile: qpart[tile * kn * N + kb * N + n];
// store_z = false: Z itself is not written (a density pass that does not need the gradient)
int blk_forward(gmmvi_ctx* ctx, int D, const float* packed, int k0, int kn, const float* X, int N, float* Z, int ldz,
                float* qpart = nullptr, bool store_z = true) {
    const size_t ps = gmmvi_blocked_stride(D);
    BG g = bg_zero();
    g.A = X; g.lda = D; g.sA = 0; g.a_kmajor = 0;
    g.a_sub = packed + (size_t)k0 * ps; g.s_asub = (long long)ps;
    g.B = packed + (size_t)k0 * ps + gmmvi_blocked_linv_ofs(D); g.ldb = D; g.sB = (long long)ps; g.b_kmajor = 0;
    g.C = Z; g.ldc = ldz; g.sC = (long long)N * ldz;
    g.M = N; g.N = D; g.Kd = D; g.tri = 1;
    g.rowsq = qpart; g.rs_tile = (long long)kn * N; g.rs_batch = N; g.no_store = store_z ? 0 : 1;
    GMMVI_PROF_UNITS(ctx, "blocked_forward", (double)kn * N);
    return bgemm(ctx, g, kn);
}

}  // namespace

#define BLK_TRY(call) do { int rc__ = (call); if (rc__ != GMMVI_OK) return rc__; } while (0)

static int blk_threads(int D) { return ((D + 63) / 64) * 64; }
// factorisation kernels: one wavefront per 32-row / 32-rhs tile of the matrix-core sums when that fits (D <= 320: <= 11 waves),
// else one thread per row as everywhere
static int blk_threads_mm(int D) { return D <= 320 ? 64 * ((D + 31) / 32) : blk_threads(D); }

// the factorisation kernels stage up to ~70 KB of LDS at D = 512: raise the dynamic limit once
static int blk_lds_attr(gmmvi_ctx* ctx) {
    if (ctx->func_attr_done & 4u) return GMMVI_OK;          // per device: remembered per context
    const int lim = 156 * 1024;            // (the kernels also hold a few static words)
    GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blk_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blk_cholesky_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    ctx->func_attr_done |= 4u;
    return GMMVI_OK;
}

int gmmvi_blocked_pack(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* means, const float* chols,
                       float* packed, float* inv_chols) {
    GMMVI_PROF(ctx, "blocked_pack");
    BLK_TRY(blk_lds_attr(ctx));
    hipLaunchKernelGGL(blk_pack_kernel, dim3(K), dim3(blk_threads_mm(D)), blk_trsm_lds_floats(D) * sizeof(float), ctx->stream, family,
                       nu, D, means, chols, packed, gmmvi_blocked_stride(D), gmmvi_blocked_linv_ofs(D), inv_chols);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_blocked_cholesky(gmmvi_ctx* ctx, int K, int D, const float* covs, float* chols, int32_t* ok) {
    BLK_TRY(gmmvi_ws_reserve(ctx, (size_t)K * D * D * sizeof(float)));
    BLK_TRY(blk_lds_attr(ctx));
    hipLaunchKernelGGL(blk_cholesky_kernel, dim3(K), dim3(blk_threads_mm(D)), blk_chol_lds_floats(D) * sizeof(float), ctx->stream, D,
                       covs, (float*)ctx->ws, chols, ok);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_blocked_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed, const float* logw,
                               const float* logw2, const float* X, int N, float* ld_out, float* lp, float* grad, float* lp2) {
    const size_t ps = gmmvi_blocked_stride(D);
    const bool need_lse = lp || grad || lp2;
    const int LP = ((D + 1 + 3) / 4) * 4;                  // row stride of Z: the Stein estimate reuses it with [.; 1; 0..] appended
    const size_t zrow = (size_t)N * LP, gslab = (size_t)N * D;
    size_t kc = z_budget_floats() / zrow;
    const int Kc = (int)(kc < 1 ? 1 : (kc > (size_t)K ? (size_t)K : kc));
    const int nchunks = (K + Kc - 1) / Kc;
    const size_t f_z = (size_t)Kc * zrow, f_q = (size_t)Kc * N, f_rw = grad ? (size_t)Kc * N : 0;
    const size_t f_ld = (!ld_out && need_lse) ? (size_t)K * N : 0, f_lp = (grad && !lp) ? (size_t)N : 0;
    // gradient: the component loop runs inside the workgroups of one [N, D] tile grid; with few tiles it is split into S
    // component ranges (≈8 workgroups per CU), each writing a partial gradient
    int S = 1;
    if (grad) {
        const int tiles = ((N + 127) / 128) * ((D + 159) / 160);
        S = (8 * ctx->num_cus + tiles - 1) / tiles;
        if (S > Kc / 4) S = Kc / 4;
        if (S > 16) S = 16;
        if (S < 1) S = 1;
        if (bgemm_use_split(D, D)) {
            // split route: one persistent workgroup per CU walks over the tiles -- the S with the fewest k steps per CU (whole
            // rounds of tiles x components of a range; C5 shard: 156 row tiles x S = 8 ranges of 8 components = 5 rounds, 40
            // components per CU against 39 ideal; the rule above gave 7 ranges of 10: 50)
            const int tc = ((N + 127) / 128) * bgemm_col_tiles(D, D);
            const int spb = (D + BK - 1) / BK;
            long long best = -1;
            for (int c = 1; c <= 16 && c <= (Kc / 4 > 1 ? Kc / 4 : 1); ++c) {
                const int inner = (Kc + c - 1) / c;
                const long long rounds = ((long long)tc * ((Kc + inner - 1) / inner) + ctx->num_cus - 1) / ctx->num_cus;
                const long long cost = rounds * ((long long)inner * spb + 12);          // + the fixed cost of a tile, in steps
                if (best < 0 || cost < best) { best = cost; S = c; }
            }
        }
    }
    const size_t f_gp = S > 1 ? (size_t)S * gslab : 0;
    const int ctiles = bgemm_col_tiles(D, D);              // the whitening launch: D columns, contraction over D
    const size_t f_qp = (size_t)ctiles * Kc * N;              // |z|^2 partials of the whitening launch, one per column tile
    // Z is needed after the whitening launch only by the gradient pass and by the Stein hand-over (which requires the gradient)
    const bool store_z = grad != nullptr;
    BLK_TRY(gmmvi_ws_reserve(ctx, (f_z + f_q + f_rw + f_ld + f_lp + f_gp + f_qp) * sizeof(float)));
    float* Z = (float*)ctx->ws;
    float* q = Z + f_z;
    float* rw = q + f_q;
    float* ldp = ld_out ? ld_out : (f_ld ? rw + f_rw : nullptr);
    float* lpp = lp ? lp : (f_lp ? rw + f_rw + f_ld : nullptr);
    float* gpart = rw + f_rw + f_ld + f_lp;
    float* qpart = gpart + f_gp;
    // the row norms |z|^2 come out of the whitening launch itself (per column tile; Z is not read again)
    auto rowsq = [&](int k0, int kn, bool write_ld) {
        const long long rows = (long long)kn * N;
        GMMVI_PROF(ctx, "blocked_rowsq");
        hipLaunchKernelGGL(blk_qfinish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, family, nu, D, N,
                           rows, ctiles, qpart, packed + (size_t)k0 * ps, ps, q, (write_ld && ldp) ? ldp + (size_t)k0 * N : nullptr);
    };
    for (int c = 0; c < nchunks; ++c) {
        const int k0 = c * Kc, kn = (K - k0 < Kc) ? K - k0 : Kc;
        BLK_TRY(blk_forward(ctx, D, packed, k0, kn, X, N, Z, LP, qpart, store_z));
        rowsq(k0, kn, true);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    if (need_lse) {
        GMMVI_PROF(ctx, "blocked_lse");
        hipLaunchKernelGGL(blk_lse_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, K, N, ldp, logw, logw2, lpp, lp2);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    if (grad) {
        for (int c = 0; c < nchunks; ++c) {
            const int k0 = c * Kc, kn = (K - k0 < Kc) ? K - k0 : Kc;
            if (nchunks > 1) {
                BLK_TRY(blk_forward(ctx, D, packed, k0, kn, X, N, Z, LP, qpart, true));
                rowsq(k0, kn, false);
                GMMVI_LAUNCH_CHECK(ctx);
            }
            const long long tot = (long long)kn * N;
            hipLaunchKernelGGL(blk_resp_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, family, nu, D, N,
                               kn, ldp + (size_t)k0 * N, logw + k0, lpp, q, rw);
            GMMVI_LAUNCH_CHECK(ctx);
            BG g = bg_zero();
            g.A = Z; g.lda = LP; g.sA = (long long)zrow; g.a_kmajor = 0;
            g.a_rscale = rw; g.s_ars = N;
            g.B = packed + (size_t)k0 * ps + gmmvi_blocked_linv_ofs(D); g.ldb = D; g.sB = (long long)ps; g.b_kmajor = 1;
            g.ldc = D;
            g.M = N; g.N = D; g.Kd = D; g.tri = 2; g.inner_total = kn;
            GMMVI_PROF_UNITS(ctx, "blocked_grad", (double)kn * N);
            if (S > 1) {                       // component ranges over blockIdx.z, partial gradients summed in fixed order
                g.C = gpart; g.sC = (long long)gslab; g.inner = (kn + S - 1) / S;
                const int nz = (kn + g.inner - 1) / g.inner;
                BLK_TRY(bgemm(ctx, g, nz));
                hipLaunchKernelGGL(blk_sum_slabs_kernel, dim3((unsigned)((gslab + 255) / 256), 1), dim3(256), 0, ctx->stream, nz,
                                   gslab, gpart, grad, c > 0 ? 1 : 0);
                GMMVI_LAUNCH_CHECK(ctx);
            } else {
                g.C = grad; g.sC = 0; g.inner = kn; g.accumulate = c > 0;
                BLK_TRY(bgemm(ctx, g, 1));
            }
        }
    }
    return GMMVI_OK;
}

int gmmvi_blocked_sample(gmmvi_ctx* ctx, int K, int D, const float* means, const float* chols, const int32_t* offsets, int N,
                         int max_per_component, uint64_t seed, uint64_t first_index, int stream_id, const float* eps,
                         float* X_out, int32_t* mapping_out) {
    if (N == 0) return GMMVI_OK;
    if (eps == nullptr) {
        BLK_TRY(gmmvi_ws_reserve(ctx, (size_t)N * D * sizeof(float)));
        BLK_TRY(gmmvi_philox_normals(ctx, seed, first_index, stream_id, N, D, (float*)ctx->ws));
        eps = (const float*)ctx->ws;
    }
    const int bound = max_per_component < N ? max_per_component : N;
    BG g = bg_zero();
    g.A = eps; g.lda = D; g.sA = 0; g.a_kmajor = 0;
    g.B = chols; g.ldb = D; g.sB = (long long)D * D; g.b_kmajor = 0;       // opB(k = m, n = i) = L[i][m]: zero for m > i
    g.C = X_out; g.ldc = D; g.sC = 0;
    g.c_bias = means; g.s_cb = D;
    g.row_off = offsets;
    g.M = bound; g.N = D; g.Kd = D; g.tri = 1;
    {
        GMMVI_PROF(ctx, "blocked_sample");
        BLK_TRY(bgemm(ctx, g, K));
    }
    if (mapping_out) {
        int by = (bound + 255) / 256;
        if (by < 1) by = 1;
        if (by > 64) by = 64;
        hipLaunchKernelGGL(blk_mapping_kernel, dim3(K, by), dim3(256), 0, ctx->stream, K, offsets, mapping_out);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    return GMMVI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Stein estimate (gmmvi_modules/ng_estimator.py:146-263), moment form as in stein.hip: the estimate is linear in
// y = Sigma^-1 (x - mu), so C_k = sum_n e_kn [x_n - mu_k; 1] [g_n; 1]^T is ONE contraction over the samples per component with
// operands shared by all components (rows [x; 1] and [g; 1]; centring and importance weight enter as the A prologue) --
// nothing whitened is read or kept -- and Sigma_k^-1 = L^-T L^-1 is applied once per component by two triangular contractions:
// H = sym((C[:D,:D]^T L^-T) L^-1) / sum e.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// rows [g_n; 1; 0..] and [x_n; 1; 0..] with the row stride LP = D + 1 rounded up to a multiple of 4 (16-byte aligned rows)
__global__ __launch_bounds__(256) void blk_stein_rows_kernel(int N, int D, int LP, const float* __restrict__ X,
                                                             const float* __restrict__ tgrad, const float* __restrict__ qgrad,
                                                             float* __restrict__ X1, float* __restrict__ G1) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * LP) return;
    const int n = (int)(e / LP), i = (int)(e % LP);
    const float one = (i == D) ? 1.f : 0.f;
    G1[e] = (i < D) ? tgrad[(size_t)n * D + i] - qgrad[(size_t)n * D + i] : one;
    X1[e] = (i < D) ? X[(size_t)n * D + i] : one;
}

// rows [mu_k; 0; 0..] (the "1" column is not centred)
__global__ __launch_bounds__(256) void blk_stein_mu_kernel(int D, int LP, size_t ps, const float* __restrict__ packed,
                                                           float* __restrict__ mu1) {
    const int kb = blockIdx.x;
    for (int i = threadIdx.x; i < LP; i += 256) mu1[(size_t)kb * LP + i] = (i < D) ? packed[(size_t)kb * ps + i] : 0.f;
}

// importance weights of one component over all samples, referred to their maximum: e[kb][n] = exp(a_n - M), Mk[kb] = M
__global__ __launch_bounds__(1024) void blk_stein_weights_kernel(int N, int k0, const float* __restrict__ ld,
                                                                 const float* __restrict__ bg, const int32_t* __restrict__ mapping,
                                                                 int map_offset, int flags, float* __restrict__ e,
                                                                 float* __restrict__ Mk) {
    __shared__ float red[16];
    const int kb = blockIdx.x, k = k0 + kb;
    const bool own_only = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    float m = -3.0e38f;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const float a = own_only ? ((mapping[n] + map_offset == k) ? 0.f : -3.0e38f) : ld[(size_t)k * N + n] - bg[n];
        m = fmaxf(m, a);
    }
    const float M = block_max(m, red);
    for (int n = threadIdx.x; n < N; n += 1024) {
        const float a = own_only ? ((mapping[n] + map_offset == k) ? 0.f : -3.0e38f) : ld[(size_t)k * N + n] - bg[n];
        e[(size_t)kb * N + n] = (a > -1.0e38f) ? __expf(a - M) : 0.f;
    }
    if (threadIdx.x == 0) Mk[kb] = M;
}

// C[b][a] = sum e d_b g'_a (row D of C: sum e g'_a), T = sum e g y^T
__global__ __launch_bounds__(256) void blk_stein_finalize_kernel(int D, int N, int flags, const float* __restrict__ Craw,
                                                                 const float* __restrict__ T, const float* __restrict__ Mk,
                                                                 float* __restrict__ H_neg, float* __restrict__ g_neg) {
    const int kb = blockIdx.x, D1 = ((D + 1 + 3) / 4) * 4;          // row stride of the augmented matrix
    const float* C = Craw + (size_t)kb * D1 * D1;
    const float* Tk = T + (size_t)kb * D * D;
    const bool snis = (flags & GMMVI_SELF_NORMALIZED) != 0;
    // own samples only: the divisor is the number of own samples = sum e (weights exp(0), ng_estimator.py:110-118,146-152)
    const bool own = (flags & GMMVI_OWN_SAMPLES_ONLY) != 0;
    // (an empty own-sample set: zeros in the self-normalised branch, NaN in the plain one, as upstream's reductions give)
    const float se = C[(size_t)D * D1 + D];
    const float scale = snis ? (se > 0.f ? 1.f / se : 0.f) : (own ? 1.f / se : __expf(Mk[kb]) / (float)N);
    // (blockIdx.y slices the elements: one workgroup per component left three quarters of the chip idle, 126 us at the C5 shard)
    for (int e = blockIdx.y * 256 + threadIdx.x; e < D * D; e += gridDim.y * 256) {
        const int i = e / D, j = e % D;
        const float v = snis ? 0.5f * (Tk[e] + Tk[(size_t)j * D + i]) : Tk[e];
        H_neg[(size_t)kb * D * D + e] = -v * scale;
    }
    if (blockIdx.y == 0)
        for (int i = threadIdx.x; i < D; i += 256) g_neg[(size_t)kb * D + i] = -C[(size_t)D * D1 + i] * scale;
}

}  // namespace

int gmmvi_blocked_stein(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                        const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping, int map_offset,
                        int flags, float* H_neg, float* g_neg) {
    const int LP = ((D + 1 + 3) / 4) * 4;                  // augmented width D + 1, padded to 16-byte rows (pad columns zero)
    const size_t ps = gmmvi_blocked_stride(D);
    const size_t rows = (size_t)N * LP;
    // component chunks only bound the scratch of the weights and the per-component matrices (default budget 4 GiB)
    const size_t per_k = (size_t)N + 2 * (size_t)LP * LP + 2 * (size_t)D * D + LP + 4;
    size_t kc = z_budget_floats() / per_k;
    const int Kc = (int)(kc < 1 ? 1 : (kc > (size_t)K ? (size_t)K : kc));
    // the contraction over the samples is split into S ranges per component so that the launch has ~8 workgroups per CU
    // (a component alone has only ceil(LP/128) * ceil(LP/160) output tiles); the partial matrices are summed in fixed order
    const int tiles = ((LP + 127) / 128) * ((LP + 159) / 160);
    int S = (8 * ctx->num_cus + tiles * Kc - 1) / (tiles * Kc);
    if (S > N / 256) S = N / 256;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    if (bgemm_use_split(LP, N)) {
        // split route: one persistent workgroup per CU walks over the tiles -- the S that needs the fewest k steps per CU
        // (whole rounds of tiles x steps of a tile; C5 shard: 3 row tiles x 64 components x S = 4 -> exactly 3 tiles per CU)
        const int tc = ((LP + 127) / 128) * bgemm_col_tiles(LP, N) * Kc;
        long long best = -1;
        for (int c = 1; c <= 16 && c <= (N / 256 > 1 ? N / 256 : 1); ++c) {
            const long long rounds = ((long long)tc * c + ctx->num_cus - 1) / ctx->num_cus;
            const long long steps = ((N + c - 1) / c + BK - 1) / BK + 12;            // + the fixed cost of a tile, in steps
            if (best < 0 || rounds * steps < best) { best = rounds * steps; S = c; }
        }
    }
    const size_t f_x = rows, f_g = rows, f_e = (size_t)Kc * N, f_m = ((size_t)Kc + 3) / 4 * 4, f_mu = (size_t)Kc * LP;
    const size_t f_c = (size_t)Kc * LP * LP, f_w = (size_t)Kc * D * D, f_t = (size_t)Kc * D * D;
    const size_t f_p = S > 1 ? (size_t)Kc * S * LP * LP : 0;
    BLK_TRY(gmmvi_ws_reserve(ctx, (f_x + f_g + f_e + f_m + f_mu + f_c + f_w + f_t + f_p) * sizeof(float)));
    float* X1 = (float*)ctx->ws;
    float* G1 = X1 + f_x;
    float* e = G1 + f_g;
    float* Mk = e + f_e;
    float* mu1 = Mk + f_m;
    float* Craw = mu1 + f_mu;
    float* Wt = Craw + f_c;
    float* T = Wt + f_w;
    float* Cpart = T + f_t;
    hipLaunchKernelGGL(blk_stein_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, N, D, LP, X, tgrad,
                       qgrad, X1, G1);
    GMMVI_LAUNCH_CHECK(ctx);
    for (int k0 = 0; k0 < K; k0 += Kc) {
        const int kn = (K - k0 < Kc) ? K - k0 : Kc;
        hipLaunchKernelGGL(blk_stein_mu_kernel, dim3(kn), dim3(256), 0, ctx->stream, D, LP, ps, packed + (size_t)k0 * ps, mu1);
        hipLaunchKernelGGL(blk_stein_weights_kernel, dim3(kn), dim3(1024), 0, ctx->stream, N, k0, ld, bg, mapping, map_offset,
                           flags, e, Mk);
        GMMVI_LAUNCH_CHECK(ctx);
        {
            BG g = bg_zero();
            g.A = X1; g.lda = LP; g.sA = 0; g.a_kmajor = 1;             // opA(m = b, k = n) = (X1[n][b] - mu1[b]) e[n]
            g.a_kscale = e; g.s_aks = N;
            g.a_rsub = mu1; g.s_arsub = LP;
            g.B = G1; g.ldb = LP; g.sB = 0; g.b_kmajor = 1;             // opB(k = n, n = a) = G1[n][a]
            g.C = S > 1 ? Cpart : Craw; g.ldc = LP; g.sC = (long long)LP * LP;
            g.M = LP; g.N = LP; g.Kd = N; g.ksplit = S;
            GMMVI_PROF_UNITS(ctx, "blocked_stein_accumulate", (double)kn * N);
            BLK_TRY(bgemm(ctx, g, kn));
            if (S > 1) {
                const size_t slab = (size_t)LP * LP;
                hipLaunchKernelGGL(blk_sum_slabs_kernel, dim3((unsigned)((slab + 255) / 256), kn), dim3(256), 0, ctx->stream, S,
                                   slab, Cpart, Craw, 0);
                GMMVI_LAUNCH_CHECK(ctx);
            }
        }
        {
            // Wt[a][i] = sum_j C[j][a] Linv[i][j]  ( = (A L^-T)[a][i] with A = C^T )
            BG g = bg_zero();
            g.A = Craw; g.lda = LP; g.sA = (long long)LP * LP; g.a_kmajor = 1;          // opA(m = a, k = j) = C[j][a]
            g.B = packed + (size_t)k0 * ps + gmmvi_blocked_linv_ofs(D); g.ldb = D; g.sB = (long long)ps; g.b_kmajor = 0;   // opB(k = j, n = i) = Linv[i][j]
            g.C = Wt; g.ldc = D; g.sC = (long long)D * D;
            g.M = D; g.N = D; g.Kd = D; g.tri = 1;                                       // Linv[i][j] = 0 for j > i
            GMMVI_PROF(ctx, "blocked_stein_unwhiten");
            BLK_TRY(bgemm(ctx, g, kn));
        }
        {
            // T = Wt L^-1
            BG g = bg_zero();
            g.A = Wt; g.lda = D; g.sA = (long long)D * D; g.a_kmajor = 0;
            g.B = packed + (size_t)k0 * ps + gmmvi_blocked_linv_ofs(D); g.ldb = D; g.sB = (long long)ps; g.b_kmajor = 1;
            g.C = T; g.ldc = D; g.sC = (long long)D * D;
            g.M = D; g.N = D; g.Kd = D; g.tri = 2;
            GMMVI_PROF(ctx, "blocked_stein_unwhiten");
            BLK_TRY(bgemm(ctx, g, kn));
        }
        const int fin_slices = (D * D + 256 * 32 - 1) / (256 * 32);                 // ~32 elements per thread
        hipLaunchKernelGGL(blk_stein_finalize_kernel, dim3(kn, fin_slices), dim3(256), 0, ctx->stream, D, N, flags, Craw, T, Mk,
                           H_neg + (size_t)k0 * D * D, g_neg + (size_t)k0 * D);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    return GMMVI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// KL-constrained component update (ng_based_component_updater.py:431-524): the whitened / tridiagonal route of update_kl.hip
// with the D x D matrices in global memory (L2-resident), one workgroup per component and step.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// R_sym (lower triangle mirrored, as tf.linalg.cholesky reads it) and g~ = g_neg + (R_sym - R) mu
__global__ __launch_bounds__(256) void blk_upd_prep_kernel(int D, const float* __restrict__ H_neg, const float* __restrict__ g_neg,
                                                           const float* __restrict__ means, float* __restrict__ Rs,
                                                           float* __restrict__ gt) {
    const int k = blockIdx.x;
    const float* R = H_neg + (size_t)k * D * D;
    const float* mu = means + (size_t)k * D;
    float* o = Rs + (size_t)k * D * D;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < D * D; e += gridDim.y * 256) {
        const int i = e / D, j = e % D;
        o[e] = (j <= i) ? R[e] : R[(size_t)j * D + i];
    }
    for (int t = blockIdx.y * 256 + threadIdx.x; t < D; t += gridDim.y * 256) {
        // four independent partial sums: the loads of four terms are in flight together (one chain of D dependent
        // load + multiply-add steps made this loop 100 of the kernel's 112 us at D = 300)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int j = t + 1;
        for (; j + 3 < D; j += 4) {
            a0 = fmaf(R[(size_t)j * D + t] - R[(size_t)t * D + j], mu[j], a0);
            a1 = fmaf(R[(size_t)(j + 1) * D + t] - R[(size_t)t * D + j + 1], mu[j + 1], a1);
            a2 = fmaf(R[(size_t)(j + 2) * D + t] - R[(size_t)t * D + j + 2], mu[j + 2], a2);
            a3 = fmaf(R[(size_t)(j + 3) * D + t] - R[(size_t)t * D + j + 3], mu[j + 3], a3);
        }
        for (; j < D; ++j) a0 = fmaf(R[(size_t)j * D + t] - R[(size_t)t * D + j], mu[j], a0);
        gt[(size_t)k * D + t] = g_neg[(size_t)k * D + t] + ((a0 + a1) + (a2 + a3));
    }
}

// M <- (M + M^T)/2, copy to Mc; w = L^T g~ (also the vector the reflectors act on)
__global__ __launch_bounds__(256) void blk_upd_sym_kernel(int D, const float* __restrict__ chols, const float* __restrict__ gt,
                                                          float* __restrict__ M, float* __restrict__ Mc, float* __restrict__ w,
                                                          float* __restrict__ wt) {
    const int k = blockIdx.x;
    float* Mk = M + (size_t)k * D * D;
    float* Ck = Mc + (size_t)k * D * D;
    const float* L = chols + (size_t)k * D * D;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < D * D; e += gridDim.y * 256) {
        const int i = e / D, j = e % D;
        if (j < i) {
            const float m = 0.5f * (Mk[e] + Mk[(size_t)j * D + i]);
            Mk[e] = m; Mk[(size_t)j * D + i] = m;
            Ck[e] = m; Ck[(size_t)j * D + i] = m;
        } else if (j == i) {
            Ck[e] = Mk[e];
        }
    }
    for (int t = blockIdx.y * 256 + threadIdx.x; t < D; t += gridDim.y * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;              // four independent partial sums (as in blk_upd_prep_kernel)
        const float* gk = gt + (size_t)k * D;
        int c = t;
        for (; c + 3 < D; c += 4) {
            a0 = fmaf(L[(size_t)c * D + t], gk[c], a0);
            a1 = fmaf(L[(size_t)(c + 1) * D + t], gk[c + 1], a1);
            a2 = fmaf(L[(size_t)(c + 2) * D + t], gk[c + 2], a2);
            a3 = fmaf(L[(size_t)(c + 3) * D + t], gk[c + 3], a3);
        }
        for (; c < D; ++c) a0 = fmaf(L[(size_t)c * D + t], gk[c], a0);
        const float a = (a0 + a1) + (a2 + a3);
        w[(size_t)k * D + t] = a;
        wt[(size_t)k * D + t] = a;
    }
}

// Householder tridiagonalisation of the symmetric M (in place) with the reflectors applied to wt.  Thread (i, grp): column
// i of the trailing block, rows j = c+1+grp, +G, ... (coalesced over i, eight loads in flight); partial products meet in
// LDS.  ONE sweep over the trailing block per step: the rank-2 update of step c (M -= v q^T + q v^T, exactly symmetric)
// and the product M' v' for the reflector of step c+1 share it -- the next reflector only needs row c+1 of the updated
// matrix, which is formed first (one element per thread).
__global__ __launch_bounds__(1024) void blk_tridiag_kernel(int D, int DT, int G, float* __restrict__ Mall,
                                                           float* __restrict__ wt_all, float* __restrict__ td_all,
                                                           float* __restrict__ te_all) {
    extern __shared__ float sm[];
    float* va = sm;                       // reflector of the current step
    float* vb = sm + D;                   // reflector of the next step
    float* q = sm + 2 * D;
    float* xn = sm + 3 * D;               // row c+1 of the updated matrix
    float* part = sm + 4 * D;
    float* red = part + (size_t)G * DT;
    const int k = blockIdx.x, tid = threadIdx.x;
    const int i = tid % DT, grp = tid / DT;
    float* M = Mall + (size_t)k * D * D;
    float* te = te_all + (size_t)k * D;
    const bool g0 = grp == 0 && i < D;
    float wti = g0 ? wt_all[(size_t)k * D + i] : 0.f;

    // reflector for column c from its sub-diagonal part x (x_i = value of thread i, group 0): v_i, beta; alpha or the
    // untouched sub-diagonal entry goes to te[c].  All results block-uniform except vi.
    float beta = 0.f, vi = 0.f, p = 0.f;
    auto make_reflector = [&](int c, float xi, float x1, float* vdst) {
        const bool mine = g0 && i > c;
        const float tail = block_sum_nb((mine && i > c + 1) ? xi * xi : 0.f, red + 32);
        float v_here = 0.f;
        if (!(tail > 0.f)) {                       // column already tridiagonal (NaN lands here too: handled by the search)
            beta = 0.f;
            if (tid == 0) te[c] = x1;
        } else {
            const float nrm = sqrtf(tail + x1 * x1);
            const float alpha = (x1 > 0.f) ? -nrm : nrm;
            beta = 1.f / (nrm * nrm - alpha * x1);
            v_here = mine ? (i == c + 1 ? x1 - alpha : xi) : 0.f;
            if (tid == 0) te[c] = alpha;
        }
        if (g0) vdst[i] = v_here;
        return v_here;
    };

    if (D >= 3) {
        // step 0: reflector of column 0, p = beta M v by a plain sweep
        const float x1 = M[1];
        const float xi = (g0 && i > 0) ? M[i] : 0.f;
        vi = make_reflector(0, xi, x1, va);
        __syncthreads();
        float a0 = 0.f;
        if (i > 0 && i < D) {
            float av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int j = 1 + grp;
            for (; j + 7 * G < D; j += 8 * G) {
#pragma unroll
                for (int u = 0; u < 8; ++u) av[u] = fmaf(M[(size_t)(j + u * G) * D + i], va[j + u * G], av[u]);
            }
            for (; j < D; j += G) a0 = fmaf(M[(size_t)j * D + i], va[j], a0);
            a0 += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        }
        part[grp * DT + i] = a0;
        __syncthreads();
        p = 0.f;
        if (g0 && i > 0) {
            for (int gi = 0; gi < G; ++gi) p += part[gi * DT + i];
            p *= beta;
        }
    }
    float* v = va;
    float* v2 = vb;
    for (int c = 0; c + 2 < D; ++c) {
        float s_vp = vi * p, s_vw = vi * wti;
        block_sum2_nb(s_vp, s_vw, red);         // red[0..31]; the reflector norm uses red[32..47]
        const float kk = 0.5f * beta * s_vp;
        const float wdot = beta * s_vw;
        if (g0) q[i] = (i > c) ? p - kk * vi : 0.f;
        wti -= wdot * vi;
        __syncthreads();
        const bool more = c + 3 < D;                   // another reflector follows
        float beta_n = 0.f, vi_n = 0.f;
        if (more) {
            // row c+1 of the updated matrix: its entries right of the diagonal are the next column
            float xv = 0.f;
            if (g0 && i > c) {
                xv = M[(size_t)(c + 1) * D + i] - __fadd_rn(__fmul_rn(v[c + 1], q[i]), __fmul_rn(q[c + 1], v[i]));
                xn[i] = xv;
            }
            __syncthreads();
            const float x1n = xn[c + 2];
            const float beta_keep = beta;
            vi_n = make_reflector(c + 1, (g0 && i > c + 1) ? xv : 0.f, x1n, v2);
            beta_n = beta;
            beta = beta_keep;
            __syncthreads();
        }
        // one sweep: M[j][i] -= v_j q_i + q_j v_i, and the partial product with the next reflector
        float a0 = 0.f;
        if (i > c && i < D) {
            const float vi2 = v[i], qi2 = q[i];
            float av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int j = c + 1 + grp;
            for (; j + 7 * G < D; j += 8 * G) {
                float mv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) mv[u] = M[(size_t)(j + u * G) * D + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jj = j + u * G;
                    const float m = mv[u] - __fadd_rn(__fmul_rn(v[jj], qi2), __fmul_rn(q[jj], vi2));
                    M[(size_t)jj * D + i] = m;
                    if (more) av[u] = fmaf(m, v2[jj], av[u]);
                }
            }
            for (; j < D; j += G) {
                const float m = M[(size_t)j * D + i] - __fadd_rn(__fmul_rn(v[j], qi2), __fmul_rn(q[j], vi2));
                M[(size_t)j * D + i] = m;
                if (more) a0 = fmaf(m, v2[j], a0);
            }
            a0 += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        }
        part[grp * DT + i] = a0;
        __syncthreads();
        p = 0.f;
        if (more && g0 && i > c + 1) {
            for (int gi = 0; gi < G; ++gi) p += part[gi * DT + i];
            p *= beta_n;
        }
        beta = beta_n;
        vi = vi_n;
        float* tsw = v; v = v2; v2 = tsw;
    }
    __syncthreads();
    if (g0) {
        td_all[(size_t)k * D + i] = M[(size_t)i * D + i];
        wt_all[(size_t)k * D + i] = wti;
        if (i == D - 1 && D >= 2) te[D - 2] = M[(size_t)(D - 1) * D + D - 2];
    }
}

// The same reduction with the matrix in REGISTERS (D <= 320): NW = 8 wavefronts (two per SIMD: 256 registers a thread),
// wavefront tj owns rows j = tj + NW a (a < BR), lane l columns i = l + 64 b (b < BC), a thread keeps its BR x BC elements for
// the whole kernel (D = 300: 38 x 5 on 512 threads, 380 KB of the CU's 512 KB register file; with 16 wavefronts the 128
// registers a thread gets spill ~20 of them to scratch and the kernel is no faster than the global-memory one) -- a step no longer streams the trailing block through the L2 (the
// global-memory kernel above is bound by the ~8 loads a thread keeps in flight: 8 us per step at D = 300).  Per step a thread
// needs the reflector vectors at its rows (uniform in the wavefront: 16-byte broadcast reads from a copy stored in row order
// [tj][a]) and at its columns (one value per lane), 3 BR + 2 BC values for BR BC elements.  Rows / columns that are already
// reduced see v = q = 0 and stay bit-exact; whole 4-row blocks and 64-column blocks of them are skipped.  (The update is
// symmetric up to the rounding of the fused multiply-add the compiler contracts it to.)  Row c+1 of the
// matrix (the source of the next reflector) lives in one wavefront, one value per lane and column block: it goes to LDS.  The per-column state
// (p, v_i, the running w~) lives in thread i.  Four barriers per step: every thread recomputes q_{c+1} from the partial
// sums, so the next column can be formed without waiting for q.
__device__ __forceinline__ float blk_q_elem(float p, float kk, float v) { return p - kk * v; }

template <int NW, int BR, int BC>
__global__ __launch_bounds__(64 * NW) void blk_tridiag_reg_kernel(int D, const float* __restrict__ Mall, float* __restrict__ wt_all,
                                                               float* __restrict__ td_all, float* __restrict__ te_all) {
    constexpr int RBP = (BR + 3) / 4 * 4;          // row slots per wavefront in the row-order copies
    constexpr int NBK = RBP / 4;
    constexpr int DT = 64 * BC;
    constexpr int NWS = NW == 16 ? 4 : (NW == 8 ? 3 : 2);           // log2(NW)
    static_assert(NW == 16 || NW == 8 || NW == 4, "row classes: a power of two");
    static_assert(64 * NW >= DT, "one state thread per column");
    extern __shared__ __align__(16) float sm[];
    float* vNa = sm;                               // reflector (two buffers: current / next), natural order
    float* vNb = sm + DT;
    float* qN = sm + 2 * DT;
    float* xn = sm + 3 * DT;                       // row c+1
    float* vRa = sm + 4 * DT;                      // the same vectors in row order: element j at (j % NW) RBP + j / NW
    float* vRb = sm + 4 * DT + NW * RBP;
    float* qR = sm + 4 * DT + 2 * NW * RBP;
    float* part = sm + 4 * DT + 3 * NW * RBP;      // [NW][DT] partial products
    float* red = part + NW * DT;                   // [48]
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int tj = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* M = Mall + (size_t)k * D * D;
    float* te = te_all + (size_t)k * D;
    const int i = tid;                             // the column whose state this thread keeps
    const bool cs = tid < D;
    const int pr = (i & (NW - 1)) * RBP + (i >> NWS);
    float wti = cs ? wt_all[(size_t)k * D + i] : 0.f;
    for (int e = tid; e < 4 * DT + 3 * NW * RBP; e += 64 * NW) sm[e] = 0.f;

    float m[BR][BC];
#pragma unroll
    for (int a = 0; a < BR; ++a)
#pragma unroll
        for (int b = 0; b < BC; ++b) {
            const int j = tj + NW * a, ci = lane + 64 * b;
            m[a][b] = (j < D && ci < D) ? M[(size_t)j * D + ci] : 0.f;
        }
    __syncthreads();

    // reflector from the sub-diagonal part x of a column (x_i in thread i), tail = sum_{i > c+1} x_i^2: writes v (both orders),
    // alpha or the untouched sub-diagonal entry to te[c]; returns v_i, sets beta_out
    auto reflector = [&](int c, float xi, float x1, float tail, int buf, float& beta_out) {
        float v_here = 0.f;
        if (!(tail > 0.f)) {                       // column already tridiagonal (NaN lands here too: handled by the search)
            beta_out = 0.f;
            if (tid == 0) te[c] = x1;
        } else {
            const float nrm = sqrtf(tail + x1 * x1);
            const float alpha = (x1 > 0.f) ? -nrm : nrm;
            beta_out = 1.f / (nrm * nrm - alpha * x1);
            v_here = (cs && i > c) ? (i == c + 1 ? x1 - alpha : xi) : 0.f;
            if (tid == 0) te[c] = alpha;
        }
        if (cs) { (buf ? vNb : vNa)[i] = v_here; (buf ? vRb : vRa)[pr] = v_here; }
        return v_here;
    };
    auto column_sum = [&](int col) {               // fixed order over the NW row classes
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < NW; ++g) s += part[g * DT + col];
        return s;
    };

    float beta = 0.f, vi = 0.f, p = 0.f, pc1 = 0.f;
    if (D >= 3) {
        // step 0: reflector of column 0, p = beta M v
        const float x1 = M[1];
        const float xi = (cs && i > 0) ? M[i] : 0.f;
        float t2 = gmmvi_wave_sum((cs && i > 1) ? xi * xi : 0.f);
        if (lane == 0) red[32 + tj] = t2;
        __syncthreads();
        float tail = 0.f;
#pragma unroll
        for (int g = 0; g < NW; ++g) tail += red[32 + g];
        vi = reflector(0, xi, x1, tail, 0, beta);
        __syncthreads();
        {
            float acc[BC];
#pragma unroll
            for (int b = 0; b < BC; ++b) acc[b] = 0.f;
#pragma unroll
            for (int a = 0; a < BR; ++a) {
                const float vr = vRa[tj * RBP + a];
#pragma unroll
                for (int b = 0; b < BC; ++b) acc[b] = fmaf(m[a][b], vr, acc[b]);
            }
#pragma unroll
            for (int b = 0; b < BC; ++b) part[tj * DT + lane + 64 * b] = acc[b];
        }
        __syncthreads();
        p = (cs && i > 0) ? beta * column_sum(i) : 0.f;
        pc1 = beta * column_sum(1);
    }
    int cur = 0;
#ifdef GMMVI_TRI_STAMPS
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#define TRI_STAMP(x) do { const long long tn = wall_clock64(); tacc[x] += tn - tprev; tprev = tn; } while (0)
#else
#define TRI_STAMP(x)
#endif
    for (int c = 0; c + 2 < D; ++c) {
        const bool more = c + 3 < D;                   // another reflector follows
        const int a_lo = c >= tj ? ((c - tj) >> NWS) + 1 : 0;             // first live row slot of this wavefront (rows > c)
        const int b_lo = (c + 1) >> 6;                                 // first column block with a live column
        const float* vNc = cur ? vNb : vNa;
        const float* vRc = (cur ? vRb : vRa) + tj * RBP;
        // row c+1 (not yet updated) into LDS: it lives in ONE wavefront (tj = (c+1) % NW, slot (c+1) / NW), spread over the lanes
        if (more && tj == ((c + 1) & (NW - 1))) {
            // run-time (uniform) slot -> static register index: two levels of scalar branches
            const int asel = (c + 1) >> NWS, ahi = asel >> 3, alo = asel & 7;
#pragma unroll
            for (int hi = 0; hi < (BR + 7) / 8; ++hi) {
                if (hi == ahi) {
#pragma unroll
                    for (int lo = 0; lo < 8; ++lo) {
                        if (8 * hi + lo < BR && lo == alo) {
#pragma unroll
                            for (int b = 0; b < BC; ++b) xn[lane + 64 * b] = m[8 * hi + lo][b];
                        }
                    }
                }
            }
        }
        float s_vp = gmmvi_wave_sum(vi * p), s_vw = gmmvi_wave_sum(vi * wti);
        if (lane == 0) { red[tj] = s_vp; red[16 + tj] = s_vw; }
        TRI_STAMP(0);
        __syncthreads();                                                               // 1
        TRI_STAMP(1);
        s_vp = 0.f; s_vw = 0.f;
#pragma unroll
        for (int g = 0; g < NW; ++g) { s_vp += red[g]; s_vw += red[16 + g]; }
        const float kk = 0.5f * beta * s_vp;
        const float wdot = beta * s_vw;
        const float q_i = (cs && i > c) ? blk_q_elem(p, kk, vi) : 0.f;
        if (cs) { qN[i] = q_i; qR[pr] = q_i; }
        wti -= wdot * vi;
        float beta_n = 0.f, vi_n = 0.f, xv = 0.f;
        if (more) {
            // row c+1 of the updated matrix: its entries right of the diagonal are the next column
            const float vc1 = vNc[c + 1];
            const float qc1 = blk_q_elem(pc1, kk, vc1);
            if (cs && i > c) {
                xv = xn[i] - __fadd_rn(__fmul_rn(vc1, q_i), __fmul_rn(qc1, vi));
                xn[i] = xv;
            }
            const float t2 = gmmvi_wave_sum((cs && i > c + 2) ? xv * xv : 0.f);
            if (lane == 0) red[32 + tj] = t2;
        }
        __syncthreads();                                                               // 2
        TRI_STAMP(2);
        if (more) {
            float tail = 0.f;
#pragma unroll
            for (int g = 0; g < NW; ++g) tail += red[32 + g];
            vi_n = reflector(c + 1, (cs && i > c + 1) ? xv : 0.f, xn[c + 2], tail, cur ^ 1, beta_n);
        }
        __syncthreads();                                                               // 3
        TRI_STAMP(3);
        // one sweep: M[j][i] -= v_j q_i + q_j v_i, and the partial product with the next reflector.  Row blocks outermost: the
        // three 16-byte broadcast reads of a block serve all the thread's column blocks
        {
            const float* vRn = (cur ? vRa : vRb) + tj * RBP;
            const float* qRc = qR + tj * RBP;
            float vcb[BC], qcb[BC], acc[BC];
#pragma unroll
            for (int b = 0; b < BC; ++b) {
                vcb[b] = vNc[lane + 64 * b]; qcb[b] = qN[lane + 64 * b];
                acc[b] = 0.f;
            }
#pragma unroll
            for (int blk = 0; blk < NBK; ++blk) {
                if (4 * blk + 3 >= a_lo) {
                    const float4 v4 = *reinterpret_cast<const float4*>(vRc + 4 * blk);
                    const float4 q4 = *reinterpret_cast<const float4*>(qRc + 4 * blk);
                    const float4 w4 = *reinterpret_cast<const float4*>(vRn + 4 * blk);
                    const float vv[4] = {v4.x, v4.y, v4.z, v4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
                    const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int b = 0; b < BC; ++b) {
                        if (b >= b_lo) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (4 * blk + u < BR) {
                                    const float mn = m[4 * blk + u][b] - __fadd_rn(__fmul_rn(vv[u], qcb[b]), __fmul_rn(qq[u], vcb[b]));
                                    m[4 * blk + u][b] = mn;
                                    acc[b] = fmaf(mn, ww[u], acc[b]);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < BC; ++b)
                if (b >= b_lo) part[tj * DT + lane + 64 * b] = acc[b];
        }
        TRI_STAMP(4);
        __syncthreads();                                                               // 4
        p = (more && cs && i > c + 1) ? beta_n * column_sum(i) : 0.f;
        pc1 = more ? beta_n * column_sum(c + 2) : 0.f;
        beta = beta_n;
        vi = vi_n;
        cur ^= 1;
        TRI_STAMP(5);
    }
#ifdef GMMVI_TRI_STAMPS
    if (blockIdx.x == 0 && (tid == 0 || tid == 64 * NW - 1))
        printf("tridiag stamps (100 MHz ticks) tid %d: dump+sums %lld, barrier1 %lld, q/xv+barrier2 %lld, reflector+barrier3 %lld, sweep %lld, barrier4+colsum %lld\n",
               tid, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5]);
#endif
#pragma unroll
    for (int a = 0; a < BR; ++a)
#pragma unroll
        for (int b = 0; b < BC; ++b) {
            const int j = tj + NW * a, ci = lane + 64 * b;
            if (j == ci && ci < D) td_all[(size_t)k * D + ci] = m[a][b];
            if (D >= 2 && j == D - 1 && ci == D - 2) te[D - 2] = m[a][b];
        }
    if (cs) wt_all[(size_t)k * D + i] = wti;
}

// instance (rows per wavefront, column blocks) of the register-resident tridiagonalisation; 0 = D > 320, the global-memory
// kernel runs
static int blk_tridiag_instance(int D) {
    if (D > 320) return 0;
    const int br = (D + 7) / 8, bc = (D + 63) / 64;
    const int brs[] = {8, 16, 24, 32, 38, 40};
    for (int r : brs) if (br <= r) return r * 8 + bc;
    return 0;
}

// KL(eta) from the tridiagonal form (update_kl.hip kl_tridiag), pivots kept in a global scratch column per lane
__device__ __forceinline__ float blk_kl_tridiag(int D, const float* td, const float* te, const float* wt, float* scratch,
                                                float eta) {
    const int lane = threadIdx.x;
    const float inv = 1.f / eta;
    float* dcol = scratch + lane;
    float dprev = 1.f, cprev = 0.f, logdet = 0.f;
    bool ok = true;
    for (int i = 0; i < D; ++i) {
        const float a = fmaf(td[i], inv, 1.f);
        float d = a, c = wt[i];
        if (i > 0) {
            const float b = te[i - 1] * inv;
            const float r = b / dprev;
            d = fmaf(-b, r, a);
            c = fmaf(-r, cprev, c);
        }
        ok = ok && (d > 0.f);
        logdet += __logf(d);
        dcol[(size_t)i * 64] = d;
        dcol[(size_t)(D + i) * 64] = c;
        dprev = d; cprev = c;
    }
    float dnext = 1.f, ynext = 0.f, tr = 0.f, yy = 0.f;
    for (int i = D - 1; i >= 0; --i) {
        const float a = fmaf(td[i], inv, 1.f);
        const float d = dcol[(size_t)i * 64], c = dcol[(size_t)(D + i) * 64];
        float delta = a, y = c / d;
        if (i < D - 1) {
            const float b = te[i] * inv;
            delta = fmaf(-b, b / dnext, a);
            y = (c - b * ynext) / d;
        }
        tr += 1.f / (d + delta - a);
        yy = fmaf(y, y, yy);
        dnext = delta; ynext = y;
    }
    float kl = 0.5f * (logdet - (float)D + tr + yy * inv * inv);
    if (!ok || !(kl == kl)) kl = FLT_MAX;
    return kl;
}

// bracketing search (:335-429) on the tridiagonal form, 63 nodes of the bisection tree per round (update_kl.hip);
// state[k] = (eta*, success, KL(eta*), probes)
__global__ __launch_bounds__(64) void blk_search_kernel(int D, const float* __restrict__ td_all, const float* __restrict__ te_all,
                                                        const float* __restrict__ wt_all, const float* __restrict__ stepsizes,
                                                        const float* __restrict__ last_eta, float temperature,
                                                        float* __restrict__ scratch_all, float* __restrict__ state) {
    extern __shared__ float sm[];
    float* td = sm;
    float* te = sm + D;
    float* wt = sm + 2 * D;
    const int k = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < D; i += 64) {
        td[i] = td_all[(size_t)k * D + i];
        te[i] = te_all[(size_t)k * D + i];
        wt[i] = wt_all[(size_t)k * D + i];
    }
    __syncthreads();
    // pivot columns of the 64 lanes: in LDS when they fit (2 D x 64 floats: D <= 310 -- the backward recurrence reads them back
    // one dependent load per step, 262 us per launch from global memory at D = 300), else in global scratch
    float* scratch = scratch_all != nullptr ? scratch_all + (size_t)k * 2 * D * 64 : sm + 3 * D;
    const float eps = stepsizes[k];
    const float last = last_eta[k];
    float lb, ub;
    if (last < 0.f) { lb = -20.f; ub = 80.f; }
    else { lb = fmaxf(0.f, __logf(last) - 3.f); ub = __logf(last) + 3.f; }
    bool ub_ok = false, done = false;
    int probes = 0, iters = 0;
    float kl_ub = -1.f;
    while (!done && iters < 1000) {
        float nlb = lb, nub = ub;
        if (t >= 2) {
            const int depth = 31 - __clz(t);
            for (int b = depth - 1; b >= 0; --b) {
                const float mid = 0.5f * (nub + nlb);
                if ((t >> b) & 1) nlb = mid; else nub = mid;
            }
        }
        const float neta = 0.5f * (nub + nlb);
        const float e_eta = expf(neta);
        const float ndiff = fminf(expf(nub) - e_eta, e_eta - expf(nlb));
        const float nkl = (t >= 1) ? blk_kl_tridiag(D, td, te, wt, scratch, e_eta) : 0.f;
        int n = 1;
        for (int level = 0; level < 6 && !done && iters < 1000; ++level, ++iters) {
            const float diff = __shfl(ndiff, n), klv = __shfl(nkl, n), eta = __shfl(neta, n);
            if (diff < 1e-1f) { done = true; break; }
            ++probes;
            if (fabsf(eps - klv) < 1e-1f * eps) { lb = ub = eta; kl_ub = klv; done = true; break; }
            if (eps > klv) { ub = eta; ub_ok = true; kl_ub = klv; n = 2 * n; }
            else { lb = eta; n = 2 * n + 1; }
        }
    }
    if (ub_ok) lb = ub;
    const float lo = expf(lb), hi = expf(ub);
    const float eta_star = fmaxf(lo, temperature);
    bool success = (lo == hi);
    float kl_val = -1.f;
    if (success) {
        kl_val = (eta_star == lo && kl_ub >= 0.f) ? kl_ub : __shfl(blk_kl_tridiag(D, td, te, wt, scratch, eta_star), 0);
        success = kl_val < FLT_MAX;
    }
    if (t == 0) {
        state[4 * k] = eta_star;
        state[4 * k + 1] = success ? 1.f : 0.f;
        state[4 * k + 2] = kl_val;
        state[4 * k + 3] = (float)probes;
    }
}

// mode 0 (KL-constrained): B = I + Mc/eta* = U U^T through the Cholesky factor C of the index-reversed matrix (U = J C J);
// L' = L U^-T and z = U^-1 w by forward substitution with C on the reversed rows of L (one thread per row, thread D: w);
// mu' = mu - L' z / eta*.  Commits means / chols on success and does the bookkeeping of :518-524.
// modes 1 / 2 (direct :97-141, iBLR :160-223): Mc holds the new precision Q' (lower triangle read, as tf.linalg.cholesky
// does); Q' = U U^T the same way, the new factor is L' = U^-T (rows of the identity as right-hand sides), the new mean
// L' (U^-1 lin') for the direct update (w = new linear term), the precomputed mean (w) for iBLR.
__global__ __launch_bounds__(704) void blk_upd_final_kernel(int D, int mode, const float* __restrict__ Mc_all,
                                                            const float* __restrict__ w_all, float* __restrict__ W_all,
                                                            float* __restrict__ X_all, const float* __restrict__ state,
                                                            float* __restrict__ means, float* __restrict__ chols,
                                                            float l2_init, float* __restrict__ last_eta, float* __restrict__ l2,
                                                            float* __restrict__ num_updates, int32_t* __restrict__ success_out,
                                                            float* __restrict__ kl_out, int32_t* __restrict__ nprobes_out) {
    extern __shared__ __align__(16) float dyn[];
    __shared__ int anybad;
    const int k = blockIdx.x, t = threadIdx.x;
    const float eta_star = mode == 0 ? state[4 * k] : 1.f;
    bool success = mode == 0 ? state[4 * k + 1] != 0.f : true;
    const float kl_val = mode == 0 ? state[4 * k + 2] : 0.f;
    const float inv = 1.f / eta_star;
    const float diag_add = mode == 0 ? 1.f : 0.f;
    float* L = chols + (size_t)k * D * D;
    float* mu = means + (size_t)k * D;
    if (success) {
        const float* Mc = Mc_all + (size_t)k * D * D;
        float* W = W_all + (size_t)k * D * D;
        success = blk_cholesky(D, [Mc, D, inv, diag_add](int i, int j) {
            return (i == j ? diag_add : 0.f) + Mc[(size_t)(D - 1 - j) * D + (D - 1 - i)] * inv; }, W, dyn);
        if (success) {
            const float* w = w_all + (size_t)k * D;
            float* X = X_all + (size_t)k * D * (D + 1);
            const int ldx = D + 1;
            if (mode == 0)
                blk_trsm<true>(D, W, D + 1,
                               [L, w, D](int i, int tt) { return tt < D ? L[(size_t)tt * D + (D - 1 - i)] : w[D - 1 - i]; },
                               t < D ? D - 1 - t : 0, X, ldx, dyn);
            else
                blk_trsm<true>(D, W, mode == 1 ? D + 1 : D,
                               [w, D](int i, int tt) { return tt < D ? (tt == D - 1 - i ? 1.f : 0.f) : w[D - 1 - i]; },
                               t < D ? D - 1 - t : 0, X, ldx, dyn);
            if (t == 0) anybad = 0;
            __syncthreads();
            float new_mu = 0.f;
            if (t < D) {
                float acc = 0.f;
                bool bad = false;
                for (int i = 0; i < D; ++i) {
                    const float x = X[(size_t)i * ldx + t];
                    if (mode != 2) acc = fmaf(x, X[(size_t)i * ldx + D], acc);
                    bad |= !(x == x);
                }
                bad |= !(X[(size_t)(D - 1 - t) * ldx + t] > 0.f);          // diagonal of L'
                new_mu = mode == 0 ? mu[t] - acc * inv : (mode == 1 ? acc : w[t]);
                bad |= !(new_mu == new_mu);
                if (bad) anybad = 1;
            }
            __syncthreads();
            success = anybad == 0;                                                         // :493 is_nan(new_chol)
            if (success) {
                for (int e = t; e < D * D; e += blockDim.x) {
                    const int r = e / D, c = e % D;
                    L[e] = (c <= r) ? X[(size_t)(D - 1 - c) * ldx + r] : 0.f;
                }
                if (t < D) mu[t] = new_mu;
            }
        }
    }
    if (t == 0) {
        if (mode == 0) {
            last_eta[k] = success ? eta_star : -1.f;
            if (kl_out) kl_out[k] = success ? kl_val : -1.f;
            if (nprobes_out) nprobes_out[k] = (int32_t)state[4 * k + 3];
        }
        const float old = l2[k];
        l2[k] = success ? fmaxf(0.5f * old, l2_init) : fminf(1e-6f, 10.f * old);
        num_updates[k] += 1.f;
        if (success_out) success_out[k] = success ? 1 : 0;
    }
}

// direct: Qp = Q + step R, vec = Q mu + step (R mu - g_neg).   iBLR: Qp = Q + step (R + step/2 T2) with T2 = R Sigma R,
// vec = mu - step Sigma g_neg (mu itself on the component's first update, :184-186).
__global__ __launch_bounds__(256) void blk_plain_prep_kernel(int mode, int D, const float* __restrict__ Q,
                                                             const float* __restrict__ H_neg, const float* __restrict__ g_neg,
                                                             const float* __restrict__ means, const float* __restrict__ stepsizes,
                                                             const float* __restrict__ num_updates, const float* __restrict__ Sig,
                                                             const float* __restrict__ T2, float* __restrict__ Qp,
                                                             float* __restrict__ vec) {
    const int k = blockIdx.x;
    const size_t o = (size_t)k * D * D;
    const float step = stepsizes[k];
    for (int e = blockIdx.y * 256 + threadIdx.x; e < D * D; e += gridDim.y * 256)
        Qp[o + e] = Q[o + e] + step * (mode == 1 ? H_neg[o + e] : H_neg[o + e] + 0.5f * step * T2[o + e]);
    const float* mu = means + (size_t)k * D;
    const float* g = g_neg + (size_t)k * D;
    for (int t = blockIdx.y * 256 + threadIdx.x; t < D; t += gridDim.y * 256) {
        float a = 0.f;
        if (mode == 1) {
            float b = 0.f;
            for (int j = 0; j < D; ++j) { a = fmaf(Q[o + (size_t)t * D + j], mu[j], a); b = fmaf(H_neg[o + (size_t)t * D + j], mu[j], b); }
            vec[(size_t)k * D + t] = a + step * (b - g[t]);
        } else {
            for (int j = 0; j < D; ++j) a = fmaf(Sig[o + (size_t)t * D + j], g[j], a);
            vec[(size_t)k * D + t] = (num_updates[k] == 0.f) ? mu[t] : mu[t] - step * a;
        }
    }
}

}  // namespace

int gmmvi_blocked_update_kl(gmmvi_ctx* ctx, int K, int D, float* means, float* chols, const float* H_neg, const float* g_neg,
                            const float* stepsizes, float temperature, float l2_init, float* last_eta, float* l2,
                            float* num_updates, int32_t* success_out, float* kl_out, int32_t* nprobes_out, float* packed_out) {
    const size_t DD = (size_t)D * D;
    const size_t f_mat = (size_t)K * DD;
    const size_t f_x = (size_t)K * D * (D + 1), f_vec = (size_t)K * D, f_scr = (size_t)K * 2 * D * 64, f_state = (size_t)K * 4;
    BLK_TRY(gmmvi_ws_reserve(ctx, (4 * f_mat + f_x + 5 * f_vec + f_scr + f_state) * sizeof(float)));
    float* Rs = (float*)ctx->ws;          // R_sym, later the row-major Cholesky factor
    float* T1 = Rs + f_mat;               // R_sym L, later the column-major Cholesky factor
    float* M = T1 + f_mat;                // L^T R L, tridiagonalised in place
    float* Mc = M + f_mat;                // copy of M for the final factorisation
    float* Xs = Mc + f_mat;
    float* gt = Xs + f_x;
    float* w = gt + f_vec;
    float* wt = w + f_vec;
    float* td = wt + f_vec;
    float* te = td + f_vec;
    float* scratch = te + f_vec;
    float* state = scratch + f_scr;
    GMMVI_PROF(ctx, "blocked_update_kl");
    const int slices = (D * D + 256 * 16 - 1) / (256 * 16);          // ~16 elements per thread
    hipLaunchKernelGGL(blk_upd_prep_kernel, dim3(K, slices), dim3(256), 0, ctx->stream, D, H_neg, g_neg, means, Rs, gt);
    GMMVI_LAUNCH_CHECK(ctx);
    {
        BG g = bg_zero();
        g.A = Rs; g.lda = D; g.sA = (long long)DD; g.a_kmajor = 0;
        g.B = chols; g.ldb = D; g.sB = (long long)DD; g.b_kmajor = 1; g.tri = 2;      // opB(k = c, n = j) = L[c][j], zero for c < j
        g.C = T1; g.ldc = D; g.sC = (long long)DD;
        g.M = D; g.N = D; g.Kd = D;
        BLK_TRY(bgemm(ctx, g, K));
    }
    {
        BG g = bg_zero();
        g.A = chols; g.lda = D; g.sA = (long long)DD; g.a_kmajor = 1;                 // opA(m = i, k = c) = L[c][i]
        g.B = T1; g.ldb = D; g.sB = (long long)DD; g.b_kmajor = 1;
        g.C = M; g.ldc = D; g.sC = (long long)DD;
        g.M = D; g.N = D; g.Kd = D;
        BLK_TRY(bgemm(ctx, g, K));
    }
    hipLaunchKernelGGL(blk_upd_sym_kernel, dim3(K, slices), dim3(256), 0, ctx->stream, D, chols, gt, M, Mc, w, wt);
    GMMVI_LAUNCH_CHECK(ctx);
    {
        const int DT = blk_threads(D);
        int G = 1024 / DT;
        if (G < 1) G = 1;
        static const bool no_reg = getenv("GMMVI_BLOCKED_TRIDIAG_GLOBAL") != nullptr;         // experiments
        const int inst = no_reg ? 0 : blk_tridiag_instance(D);
        if (inst > 0) {
            const int br = inst / 8, bc = inst % 8;
            const size_t shmem = ((size_t)4 * 64 * bc + (size_t)24 * ((br + 3) / 4 * 4) + (size_t)8 * 64 * bc + 48) * sizeof(float);
#define BLK_TRIDIAG_REG(BRV, BCV)                                                                                         \
    if (br == BRV && bc == BCV)                                                                                           \
        hipLaunchKernelGGL((blk_tridiag_reg_kernel<8, BRV, BCV>), dim3(K), dim3(512), shmem, ctx->stream, D, M, wt, td, te)
            BLK_TRIDIAG_REG(8, 1); BLK_TRIDIAG_REG(16, 2); BLK_TRIDIAG_REG(24, 3); BLK_TRIDIAG_REG(32, 4);
            BLK_TRIDIAG_REG(38, 5); BLK_TRIDIAG_REG(40, 5);
#undef BLK_TRIDIAG_REG
        } else {
            const size_t shmem = ((size_t)4 * D + (size_t)G * DT + 48) * sizeof(float);
            hipLaunchKernelGGL(blk_tridiag_kernel, dim3(K), dim3(G * DT), shmem, ctx->stream, D, DT, G, M, wt, td, te);
        }
        GMMVI_LAUNCH_CHECK(ctx);
    }
    {
        const size_t lds_cols = ((size_t)3 * D + (size_t)2 * D * 64) * sizeof(float);
        const bool in_lds = lds_cols <= 156 * 1024;           // (the kernel's static LDS words need room beside it)
        if (in_lds && !(ctx->func_attr_done & 1u)) {           // the attribute is per DEVICE: remembered per context
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blk_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     156 * 1024));
            ctx->func_attr_done |= 1u;
        }
        hipLaunchKernelGGL(blk_search_kernel, dim3(K), dim3(64), in_lds ? lds_cols : (size_t)3 * D * sizeof(float), ctx->stream, D,
                           td, te, wt, stepsizes, last_eta, temperature, in_lds ? nullptr : scratch, state);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    {
        if (!(ctx->func_attr_done & 2u)) {                     // per device: remembered per context
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blk_upd_final_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            ctx->func_attr_done |= 2u;
        }
        const size_t a = blk_trsm_lds_floats(D), b = blk_chol_lds_floats(D);
        hipLaunchKernelGGL(blk_upd_final_kernel, dim3(K), dim3(blk_threads_mm(D + 1)), (a > b ? a : b) * sizeof(float), ctx->stream, D,
                           0, Mc, w, T1, Xs, state, means, chols, l2_init, last_eta, l2, num_updates, success_out, kl_out,
                           nprobes_out);
    }
    GMMVI_LAUNCH_CHECK(ctx);
    if (packed_out) BLK_TRY(gmmvi_blocked_pack(ctx, GMMVI_GAUSS, 0.f, K, D, means, chols, packed_out, nullptr));
    return GMMVI_OK;
}

// DirectNgBasedComponentUpdater (:97-141, mode 0) and NgBasedComponentUpdaterIblr (:160-223, mode 1) on the blocked path
int gmmvi_blocked_update_plain(gmmvi_ctx* ctx, int mode, int K, int D, float* means, float* chols, const float* H_neg,
                               const float* g_neg, const float* stepsizes, float l2_init, float* l2, float* num_updates,
                               int32_t* success_out) {
    const size_t DD = (size_t)D * D, ps = gmmvi_blocked_stride(D);
    const size_t f_mat = (size_t)K * DD, f_pk = (size_t)K * ps, f_x = (size_t)K * D * (D + 1), f_vec = (size_t)K * D;
    const int nmat = mode == 0 ? 3 : 6;
    BLK_TRY(gmmvi_ws_reserve(ctx, (f_pk + nmat * f_mat + f_x + f_vec) * sizeof(float)));
    float* pk = (float*)ctx->ws;          // [mu | const | L^-1] of the current components
    float* Q = pk + f_pk;                 // old precision L^-T L^-1
    float* Qp = Q + f_mat;                // new precision
    float* W = Qp + f_mat;                // column-major Cholesky factor
    float* Sig = W + f_mat;               // iBLR: Sigma, R Sigma, R Sigma R
    float* T = Sig + (mode == 0 ? 0 : f_mat);
    float* T2 = T + (mode == 0 ? 0 : f_mat);
    float* Xs = W + (size_t)(nmat - 2) * f_mat;
    float* vec = Xs + f_x;
    GMMVI_PROF(ctx, "blocked_update_plain");
    BLK_TRY(gmmvi_blocked_pack(ctx, GMMVI_GAUSS, 0.f, K, D, means, chols, pk, nullptr));
    const float* Linv = pk + gmmvi_blocked_linv_ofs(D);
    {
        BG g = bg_zero();
        g.A = Linv; g.lda = D; g.sA = (long long)ps; g.a_kmajor = 1;          // opA(m = i, k = c) = L^-1[c][i]
        g.B = Linv; g.ldb = D; g.sB = (long long)ps; g.b_kmajor = 1;
        g.C = Q; g.ldc = D; g.sC = (long long)DD;
        g.M = D; g.N = D; g.Kd = D;
        BLK_TRY(bgemm(ctx, g, K));
    }
    if (mode == 1) {
        BG g = bg_zero();
        g.A = chols; g.lda = D; g.sA = (long long)DD; g.a_kmajor = 0;
        g.B = chols; g.ldb = D; g.sB = (long long)DD; g.b_kmajor = 0;         // opB(k = c, n = j) = L[j][c]
        g.C = Sig; g.ldc = D; g.sC = (long long)DD;
        g.M = D; g.N = D; g.Kd = D;
        BLK_TRY(bgemm(ctx, g, K));
        g.A = H_neg; g.B = Sig; g.b_kmajor = 1; g.C = T;                       // R Sigma
        BLK_TRY(bgemm(ctx, g, K));
        g.A = T; g.B = H_neg; g.C = T2;                                        // (R Sigma) R
        BLK_TRY(bgemm(ctx, g, K));
    }
    const int slices = (D * D + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(blk_plain_prep_kernel, dim3(K, slices), dim3(256), 0, ctx->stream, mode == 0 ? 1 : 2, D, Q, H_neg, g_neg,
                       means, stepsizes, num_updates, Sig, T2, Qp, vec);
    GMMVI_LAUNCH_CHECK(ctx);
    {
        if (!(ctx->func_attr_done & 2u)) {                     // per device: remembered per context
            GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)blk_upd_final_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            ctx->func_attr_done |= 2u;
        }
        const size_t a = blk_trsm_lds_floats(D), b = blk_chol_lds_floats(D);
        hipLaunchKernelGGL(blk_upd_final_kernel, dim3(K), dim3(blk_threads_mm(D + 1)), (a > b ? a : b) * sizeof(float), ctx->stream, D,
                           mode == 0 ? 1 : 2, Qp, vec, W, Xs, nullptr, means, chols, l2_init, nullptr, l2, num_updates,
                           success_out, nullptr, nullptr);
        GMMVI_LAUNCH_CHECK(ctx);
    }
    return GMMVI_OK;
}
