// Scalar-fed triangular substitutions for TWO samples per lane (density.hip: mixture_eval_pk_kernel, padded D <= 24).
//
// Why: a wave64 v_fma_f32 with a scalar multiplier issues once per 4 cycles per SIMD on gfx950 whatever the number of resident
// waves (68 TFLOP/s over the chip), v_pk_fma_f32 issues at the same rate and does two multiply-adds per lane (130 TFLOP/s;
// tools/probe/pk_issue.hip, profiles/r04_pk_issue.txt).  The one-sample kernel's component pass is ~640 vector instructions
// and its launch is bound by exactly that issue rate (profiles/r04_notes.md), so the pass is written for register PAIRS:
// lane l owns samples l and l + 64 of a 128-sample tile, z_i = (z_i of sample A, z_i of sample B) sits in an aligned VGPR pair
// and every multiply-add of the substitution is one v_pk_fma_f32 whose scalar operand is broadcast to both halves (op_sel).
//
// The block is read from its SWEEP STREAM (common.h Pack<DP>::SWH / SWF / SWB): mean and log-normaliser, then the triangle
// by columns with 1 / L_jj in front of column j, then by rows with 1 / L_ii behind row i -- the reciprocals arrive in the
// 32-float pieces exactly when they are needed, so no pass-long copy of them occupies registers (the one-sample kernel
// keeps 20 VGPR copies, the compiler spilled scalar registers into VGPR lanes around them: 45 v_readlane / v_writelane a pass).
// Loads are software-pipelined as in subst_phased.h (piece p + 1 in flight while piece p is multiplied).
#pragma once
#include "subst_phased.h"

typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_f32x2 pk_splat(float s) { return pk_f32x2{s, s}; }
__device__ __forceinline__ pk_f32x2 pk_fma(pk_f32x2 a, pk_f32x2 b, pk_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Packed operations with a SCALAR operand taken from one half of an aligned scalar-register pair and broadcast to both halves
// of the result (op_sel / op_sel_hi select the half).  Written as asm: from a splat of the pair's HIGH element the compiler
// does not form the op_sel encoding but copies the value into the low half of a fresh pair first (one s_mov_b32 per
// multiply-add with an odd stream index: 187 per component pass at D = 20).
//   pk_fnma_s<HI>(sp, b, c) = c - sp[HI] * b      pk_mul_s<HI>(sp, a) = sp[HI] * a      pk_sub_s<HI>(a, sp) = a - sp[HI]
template <int HI>
__device__ __forceinline__ pk_f32x2 pk_fnma_s(pk_f32x2 sp, pk_f32x2 b, pk_f32x2 c) {
    if constexpr (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(c) : "s"(sp), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(c) : "s"(sp), "v"(b));
    return c;
}
template <int HI>
__device__ __forceinline__ pk_f32x2 pk_mul_s(pk_f32x2 sp, pk_f32x2 a) {
    if constexpr (HI) asm("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(a) : "s"(sp));
    else asm("v_pk_mul_f32 %0, %1, %0 op_sel:[0,0] op_sel_hi:[0,1]" : "+v"(a) : "s"(sp));
    return a;
}
template <int HI>
__device__ __forceinline__ pk_f32x2 pk_sub_s(pk_f32x2 a, pk_f32x2 sp) {
    if constexpr (HI) asm("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(a) : "s"(sp));
    else asm("v_pk_add_f32 %0, %1, %0 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(a) : "s"(sp));
    return a;
}
// the aligned pair of a 32-float piece (or of the head) that holds element `slot`
template <int SLOT, int N>
__device__ __forceinline__ pk_f32x2 pk_pair(const float (&a)[N]) { return pk_f32x2{a[SLOT & ~1], a[SLOT | 1]}; }

template <int N, int O = 0>
__device__ __forceinline__ void pk_pin(pk_f32x2 (&a)[N]) {
    if constexpr (O + 2 <= N) {
        asm volatile("" : "+v"(a[O]), "+v"(a[O + 1]));
        pk_pin<N, O + 2>(a);
    } else if constexpr (O < N) {
        asm volatile("" : "+v"(a[O]));
    }
}

// sweep stream, forward part: element e -> column j, offset r within the column (r = 0: the reciprocal of the diagonal entry)
template <int DP>
__host__ __device__ constexpr int pk_fcol(int e) {
    int j = 0;
    while ((j + 1) * DP - (j + 1) * j / 2 <= e) ++j;
    return j;
}
template <int DP>
__host__ __device__ constexpr int pk_frow(int e) { return e - (pk_fcol<DP>(e) * DP - pk_fcol<DP>(e) * (pk_fcol<DP>(e) - 1) / 2) + pk_fcol<DP>(e); }
// backward part: element e -> row i, column c <= i (c = i: the reciprocal of the diagonal entry)
__host__ __device__ constexpr int pk_brow(int e) { int i = 0; while ((i + 1) * (i + 2) / 2 <= e) ++i; return i; }
__host__ __device__ constexpr int pk_bcol(int e) { return e - pk_brow(e) * (pk_brow(e) + 1) / 2; }

template <int DP>
struct PkPass {
    using PK = Pack<DP>;
    static constexpr int TD = PK::TD;
    static constexpr int HN = (DP + 1 + 3) / 4 * 4;
    using Fwd = SpStream<PK::SWF, TD>;
    using Bwd = SpStream<PK::SWB, TD>;

    // z holds the two samples' x on entry, L^-1 (x - mu) on exit; q = |z|^2 per sample; cst = the log-normaliser
    __device__ __forceinline__ static void forward(sp_block_ptr blk, pk_f32x2 (&z)[DP], pk_f32x2& q, float& cst, float (&pc)[2][32]) {
        float head[HN];
#pragma unroll
        for (int u = 0; u < HN; ++u) head[u] = sp_at(blk, PK::SWH + u);
        Fwd::template fetch<0>(blk, pc[0]);
        __builtin_amdgcn_sched_barrier(0);
        sp_landed_head<HN>(head);
        sp_landed<Fwd::nl(0)>(pc[0]);
        cst = head[DP];
        sp_for<0, DP>([&](auto IC) {
            constexpr int i = IC;
            z[i] = pk_sub_s<(i & 1)>(z[i], pk_pair<i>(head));
        });
        q = pk_splat(0.f);
        sp_for<0, Fwd::NP>([&](auto PCE) {
            constexpr int p = PCE;
            if constexpr (p + 1 < Fwd::NP) Fwd::template fetch<p + 1>(blk, pc[(p + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            sp_for<Fwd::first(p), Fwd::last(p)>([&](auto EL) {
                constexpr int e = EL, j = pk_fcol<DP>(e), i = pk_frow<DP>(e);
                constexpr int sl = Fwd::slot(e);
                const pk_f32x2 sp = pk_pair<sl>(pc[p & 1]);
                if constexpr (i == j) {                             // 1 / L_jj: z_j is final
                    z[j] = pk_mul_s<(sl & 1)>(sp, z[j]);
                    q = pk_fma(z[j], z[j], q);
                } else {
                    z[i] = pk_fnma_s<(sl & 1)>(sp, z[j], z[i]);
                }
            });
            pk_pin<DP>(z);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (p + 1 < Fwd::NP) sp_landed<Fwd::nl(p + 1)>(pc[(p + 1) & 1]);
        });
    }

    // issue the loads of the piece the backward substitution starts with (call before the log-sum-exp arithmetic)
    __device__ __forceinline__ static void backward_prefetch(sp_block_ptr blk, float (&pc)[2][32]) {
        Bwd::template fetch<Bwd::NP - 1>(blk, pc[(Bwd::NP - 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
    }

    // y = L^-T z in place, by the rows of L in descending order
    __device__ __forceinline__ static void backward(sp_block_ptr blk, pk_f32x2 (&z)[DP], float (&pc)[2][32]) {
        __builtin_amdgcn_sched_barrier(0);
        sp_landed<Bwd::nl(Bwd::NP - 1)>(pc[(Bwd::NP - 1) & 1]);
        sp_for_down<0, Bwd::NP>([&](auto PCE) {
            constexpr int p = PCE;
            if constexpr (p > 0) Bwd::template fetch<p - 1>(blk, pc[(p - 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            sp_for_down<Bwd::first(p), Bwd::last(p)>([&](auto EL) {
                constexpr int e = EL, i = pk_brow(e), c = pk_bcol(e);
                constexpr int sl = Bwd::slot(e);
                const pk_f32x2 sp = pk_pair<sl>(pc[p & 1]);
                if constexpr (c == i) z[i] = pk_mul_s<(sl & 1)>(sp, z[i]);     // 1 / L_ii comes first: y_i is final
                else z[c] = pk_fnma_s<(sl & 1)>(sp, z[i], z[c]);
            });
            pk_pin<DP>(z);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (p > 0) sp_landed<Bwd::nl(p - 1)>(pc[(p - 1) & 1]);
        });
    }
};
