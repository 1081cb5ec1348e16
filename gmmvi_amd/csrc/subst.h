// Register-resident triangular substitutions against one packed component block (layout: common.h Pack<DP>).
// The block is addressed through a float4-typed pointer so that, once the loops are unrolled, every parameter fetch is
// a 16-byte access (ds_read_b128 broadcast from a wave-private LDS copy, or s_load_dwordx4+ from global memory); the
// compiler merges the repeated fetches of one float4.
#pragma once
#include "common.h"

struct PackRef {
    const float4* p;
#ifdef GMMVI_ME_FAKE_BLOCK         // experiment builds: no block loads at all (a uniform scalar per entry): what the pass costs without its feed
    float fake;
    __device__ __forceinline__ float operator[](int idx) const { return fake * (float)(idx % 7 + 1); }
#else
    __device__ __forceinline__ float operator[](int idx) const {
        const float4 v = p[idx >> 2];
        const int c = idx & 3;
        return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
    }
#endif
};

// The two substitutions below in AXPY form: once z_j (y_i) is final, the D - 1 - j (i) updates it feeds are independent
// multiply-adds.  The dot forms further down accumulate every row in ONE register: a chain of dependent v_fma_f32, which
// costs ~10 cycles per link on gfx950 where independent ones issue every 2 .. 5 (tools/probe/valu_issue.hip,
// profiles/r03_valu_issue.txt) -- the scalar-fed density sweep was bound by exactly that.
// z = L^-1 (x - mu) by the COLUMNS of L, q = |z|^2
template <int DP>
__device__ __forceinline__ void forward_subst_axpy(const PackRef P, const float (&x)[DP], float (&z)[DP], float& q) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = 0; i < DP; ++i) z[i] = x[i] - P[PK::MU + i];
    q = 0.f;
#pragma unroll
    for (int j = 0; j < DP; ++j) {
        z[j] *= P[PK::RD + j];
        q = fmaf(z[j], z[j], q);
#pragma unroll
        for (int i = j + 1; i < DP; ++i) z[i] = fmaf(-P[PK::LCOL + PK::colofs(j) + (i - j - 1)], z[j], z[i]);
    }
}

// y = L^-T z in place, by the ROWS of L in descending order: y_i = z_i / L_ii, then z_j -= L_ij y_i (j < i)
template <int DP>
__device__ __forceinline__ void backward_subst_axpy(const PackRef P, float (&z)[DP]) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = DP - 1; i >= 0; --i) {
        z[i] *= P[PK::RD + i];
#pragma unroll
        for (int j = i - 1; j >= 0; --j) z[j] = fmaf(-P[PK::LROW + PK::rowofs(i) + j], z[i], z[j]);
    }
}

// z = L^-1 (x - mu), q = |z|^2  (dot form)
template <int DP>
__device__ __forceinline__ void forward_subst(const PackRef P, const float (&x)[DP], float (&z)[DP], float& q) {
    using PK = Pack<DP>;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        float t = x[i] - P[PK::MU + i];
#pragma unroll
        for (int j = 0; j < i; ++j) t = fmaf(-P[PK::LROW + PK::rowofs(i) + j], z[j], t);
        z[i] = t * P[PK::RD + i];
        q = fmaf(z[i], z[i], q);
    }
}

// y = L^-T z  (Sigma^-1 (x - mu) when z = L^-1 (x - mu))
template <int DP>
__device__ __forceinline__ void backward_subst(const PackRef P, const float (&z)[DP], float (&y)[DP]) {
    using PK = Pack<DP>;
#pragma unroll
    for (int i = DP - 1; i >= 0; --i) {
        float t = z[i];
#pragma unroll
        for (int j = i + 1; j < DP; ++j) t = fmaf(-P[PK::LCOL + PK::colofs(i) + (j - i - 1)], y[j], t);
        y[i] = t * P[PK::RD + i];
    }
}
