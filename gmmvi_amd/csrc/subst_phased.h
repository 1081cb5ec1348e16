// Scalar-fed triangular substitutions with SOFTWARE-PIPELINED block loads (density.hip, padded D <= 24).
//
// A wave's component pass reads ~420 wave-uniform floats of the packed block through scalar loads.  Left to the compiler the
// pass is a chain of "s_load_dwordx16 -> s_waitcnt lgkmcnt(0) -> a few multiply-adds" segments: ~24 exposed L2 round trips of
// ~215 cycles, 3 000 of the 5 400 cycles of a pass at the north star (profiles/r03_notes.md).  Here the triangle is walked as a
// stream of 32-float pieces and piece p + 1 is in flight while piece p is multiplied:
//   * scalar loads return out of order, so every wait is lgkmcnt(0): the wait for a piece has to come BEFORE the loads of
//     the next piece are issued -- an empty asm statement that takes the piece's registers as operands forces it there;
//   * scheduling barriers keep the loads from sinking back to their first use, and an empty asm over the lane's register
//     array at the end of a piece keeps the optimiser from sinking the multiply-adds behind the following loads;
//   * the block is read through a CONSTANT-address-space pointer: with `asm volatile` statements in the loop a uniform
//     GLOBAL load would become a per-lane vector load (the asm counts as a possible writer of global memory);
//   * both substitutions are in AXPY form (forward by the columns of L, backward by the rows): the multiply-adds fed by one
//     final z_j are independent, where the dot form is a chain of dependent v_fma_f32 at ~10 cycles a link;
//   * 1 / diag L is copied to VGPRs once per pass (it is needed all through both substitutions; 2 x 32 piece registers plus
//     20 more would not fit the SGPR file).
#pragma once
#include "common.h"
#include <type_traits>

typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) sp_f32x4* sp_block_ptr;
typedef const __attribute__((address_space(4))) float* sp_const_f32;

__device__ __forceinline__ sp_block_ptr sp_block(const float* blk) { return (sp_block_ptr)(uintptr_t)blk; }
__device__ __forceinline__ float sp_at(sp_block_ptr p, int idx) { const sp_f32x4 v = p[idx >> 2]; return v[idx & 3]; }

template <int B, int E, typename F>
__device__ __forceinline__ void sp_for(F&& f) {                      // f(integral_constant<B>), ..., f(integral_constant<E - 1>)
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sp_for<B + 1, E>(f);
    }
}
template <int B, int E, typename F>
__device__ __forceinline__ void sp_for_down(F&& f) {                 // E - 1, E - 2, ..., B
    if constexpr (B < E) {
        f(std::integral_constant<int, E - 1>{});
        sp_for_down<B, E - 1>(f);
    }
}

#define SP_S4(a, o) "s"(a[o]), "s"(a[(o) + 1]), "s"(a[(o) + 2]), "s"(a[(o) + 3])
#define SP_V4(a, o) "+v"(a[o]), "+v"(a[(o) + 1]), "+v"(a[(o) + 2]), "+v"(a[(o) + 3])
// the first NL registers of a piece hold their data from here on (forces the wait for the piece to this point)
template <int NL, int O = 0>
__device__ __forceinline__ void sp_landed(const float (&a)[32]) {
    static_assert(NL % 4 == 0 && NL <= 32, "piece length");
    if constexpr (O < NL) {
        asm volatile("" ::SP_S4(a, O));
        sp_landed<NL, O + 4>(a);
    }
}
template <int N, int O = 0>
__device__ __forceinline__ void sp_landed_head(const float (&a)[N]) {
    static_assert(N % 4 == 0, "head length");
    if constexpr (O < N) {
        asm volatile("" ::SP_S4(a, O));
        sp_landed_head<N, O + 4>(a);
    }
}
// the values of a lane's register array are complete here
template <int N, int O = 0>
__device__ __forceinline__ void sp_pin(float (&a)[N]) {
    if constexpr (O + 4 <= N) {
        asm volatile("" : SP_V4(a, O));
        sp_pin<N, O + 4>(a);
    } else if constexpr (O + 2 <= N) {
        asm volatile("" : "+v"(a[O]), "+v"(a[O + 1]));
        sp_pin<N, O + 2>(a);
    } else if constexpr (O < N) {
        asm volatile("" : "+v"(a[O]));
    }
}

// A stream = the floats [BASE, BASE + LEN) of a block, fetched in pieces of 32 floats from the 16-byte boundary below BASE.
template <int BASE, int LEN>
struct SpStream {
    static constexpr int A = BASE / 4 * 4;
    static constexpr int SPAN = BASE + LEN - A;
    static constexpr int NP = (SPAN + 31) / 32;
    __host__ __device__ static constexpr int nl(int p) { return ((SPAN - 32 * p < 32 ? SPAN - 32 * p : 32) + 3) / 4 * 4; }   // floats fetched (whole 16-byte words: the tail lies in the block)
    __host__ __device__ static constexpr int first(int p) { return A + 32 * p - BASE > 0 ? A + 32 * p - BASE : 0; }          // elements of piece p: [first, last)
    __host__ __device__ static constexpr int last(int p) { return A + 32 * (p + 1) - BASE < LEN ? A + 32 * (p + 1) - BASE : LEN; }
    __host__ __device__ static constexpr int slot(int e) { return (BASE + e - A) % 32; }
    template <int P>
    __device__ __forceinline__ static void fetch(sp_block_ptr blk, float (&a)[32]) {
#pragma unroll
        for (int u = 0; u < nl(P); ++u) a[u] = sp_at(blk, A + 32 * P + u);
    }
};

// strict lower triangle, column-major (Pack<DP>::LCOL): element e = (row i > column j)
template <int DP>
__host__ __device__ constexpr int sp_col_of(int e) {
    int j = 0;
    while ((j + 1) * (DP - 1) - (j + 1) * j / 2 <= e) ++j;         // colofs(j + 1) <= e
    return j;
}
template <int DP>
__host__ __device__ constexpr int sp_row_in_col(int e) { return e - (sp_col_of<DP>(e) * (DP - 1) - sp_col_of<DP>(e) * (sp_col_of<DP>(e) - 1) / 2) + sp_col_of<DP>(e) + 1; }
// strict lower triangle, row-major (Pack<DP>::LROW): element e = (row i, column j < i)
__host__ __device__ constexpr int sp_row_of(int e) { int i = 1; while ((i + 1) * i / 2 <= e) ++i; return i; }
__host__ __device__ constexpr int sp_col_in_row(int e) { return e - sp_row_of(e) * (sp_row_of(e) - 1) / 2; }

// Forward substitution z = L^-1 (x - mu), q = |z|^2, by the columns of L; leaves 1 / diag L in vrd (VGPR copies) and the
// LAST piece of the row stream in flight for the backward substitution when PREFETCH_BACKWARD (it lands in pc[...] below).
template <int DP>
struct SpPass {
    using PK = Pack<DP>;
    static constexpr int T = PK::T;
    using Fwd = SpStream<PK::LCOL, (T > 0 ? T : 1)>;
    using Bwd = SpStream<PK::LROW, (T > 0 ? T : 1)>;

    __device__ __forceinline__ static void forward(sp_block_ptr blk, const float (&x)[DP], float (&z)[DP], float (&vrd)[DP], float& q,
                                                   float& cst, float (&pc)[2][32]) {
        static_assert(DP % 2 == 0, "padded dimensions are even");
        constexpr int HN = (2 * DP + 3) / 4 * 4;                    // mu | 1 / diag (the tail of the last word: first entries of the triangle)
        float head[HN];
#pragma unroll
        for (int u = 0; u < HN; ++u) head[u] = sp_at(blk, u);
        cst = sp_at(blk, PK::CONST);                                // the log-normaliser rides with the first wait
        if constexpr (T > 0) Fwd::template fetch<0>(blk, pc[0]);
        __builtin_amdgcn_sched_barrier(0);
        sp_landed_head<HN>(head);
        asm volatile("" ::"s"(cst));
        if constexpr (T > 0) sp_landed<Fwd::nl(0)>(pc[0]);
#pragma unroll
        for (int i = 0; i < DP; ++i) {
            z[i] = x[i] - head[PK::MU + i];
            asm volatile("v_mov_b32 %0, %1" : "=v"(vrd[i]) : "s"(head[PK::RD + i]));
        }
        q = 0.f;
        if constexpr (T > 0) {
            sp_for<0, Fwd::NP>([&](auto PCE) {
                constexpr int p = PCE;
                if constexpr (p + 1 < Fwd::NP) Fwd::template fetch<p + 1>(blk, pc[(p + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                sp_for<Fwd::first(p), Fwd::last(p)>([&](auto EL) {
                    constexpr int e = EL, j = sp_col_of<DP>(e), i = sp_row_in_col<DP>(e);
                    if constexpr (i == j + 1) {                     // first entry of column j: z_j is final
                        z[j] *= vrd[j];
                        q = fmaf(z[j], z[j], q);
                    }
                    z[i] = fmaf(-pc[p & 1][Fwd::slot(e)], z[j], z[i]);
                });
                sp_pin<DP>(z);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (p + 1 < Fwd::NP) sp_landed<Fwd::nl(p + 1)>(pc[(p + 1) & 1]);
            });
        }
        z[DP - 1] *= vrd[DP - 1];
        q = fmaf(z[DP - 1], z[DP - 1], q);
    }

    // issue the loads of the piece the backward substitution starts with (call before the log-sum-exp arithmetic)
    __device__ __forceinline__ static void backward_prefetch(sp_block_ptr blk, float (&pc)[2][32]) {
        if constexpr (T > 0) {
            Bwd::template fetch<Bwd::NP - 1>(blk, pc[(Bwd::NP - 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // y = L^-T z in place, by the rows of L in descending order: y_i = z_i / L_ii, then z_j -= L_ij y_i (j < i)
    __device__ __forceinline__ static void backward(sp_block_ptr blk, float (&z)[DP], const float (&vrd)[DP], float (&pc)[2][32]) {
        if constexpr (T > 0) {
            __builtin_amdgcn_sched_barrier(0);
            sp_landed<Bwd::nl(Bwd::NP - 1)>(pc[(Bwd::NP - 1) & 1]);
            sp_for_down<0, Bwd::NP>([&](auto PCE) {
                constexpr int p = PCE;
                if constexpr (p > 0) Bwd::template fetch<p - 1>(blk, pc[(p - 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                sp_for_down<Bwd::first(p), Bwd::last(p)>([&](auto EL) {
                    constexpr int e = EL, i = sp_row_of(e), j = sp_col_in_row(e);
                    if constexpr (j == i - 1) z[i] *= vrd[i];        // last entry of row i comes first: y_i is final
                    z[j] = fmaf(-pc[p & 1][Bwd::slot(e)], z[i], z[j]);
                });
                sp_pin<DP>(z);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (p > 0) sp_landed<Bwd::nl(p - 1)>(pc[(p - 1) & 1]);
            });
        }
        z[0] *= vrd[0];
    }
};
