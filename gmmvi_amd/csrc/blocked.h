// Blocked path for 64 < D <= GMMVI_BLOCKED_MAX_DIM (config C5: D = 300): the per-component triangular solves become
// dense fp32 contractions on the matrix cores (v_mfma_f32_32x32x2_f32, LDS-tiled), the D x D factorisations run one
// workgroup per component on L2-resident matrices.  Same entry points, same results; the register-resident kernels keep
// D <= 64.  Component block of this path (floats): [mu (D) | log-normaliser | pad to 4 | L^-1 dense row-major (D x D)].
#pragma once
#include "common.h"
#include <cstdlib>

#define GMMVI_BLOCKED_MAX_DIM GMMVI_MAX_DIM_BLOCKED

// Dimensions above this threshold take the blocked path.  Default 50: the register-resident kernels are instantiated for the
// padded dimensions ..., 40, 50, 64, so every D in 51..63 would run the 64-wide instance, which spills (mixture_eval: 300
// VGPRs to scratch; Stein: 1400 SGPRs to VGPR lanes) and takes 3.1 ms per iteration at K = 100, N = 10^4 against 1.86 ms on
// the blocked kernels; at D = 50 the register path wins (1.22 against 1.60 ms), at D = 40 / 32 clearly (profiles/
// r01_notes.md).  D = 64 cannot run register-resident at all (Stein needs D + 1 <= 64 lanes).  GMMVI_BLOCKED_ABOVE=<16..64>
// moves the threshold.
inline int gmmvi_blocked_above() {
    static const int v = [] {
        const char* s = getenv("GMMVI_BLOCKED_ABOVE");
        const int t = s ? atoi(s) : 50;
        return t < 16 ? 16 : (t > GMMVI_MAX_DIM ? GMMVI_MAX_DIM : t);
    }();
    return v;
}
inline bool gmmvi_is_blocked_dim(int D) { return D > gmmvi_blocked_above() && D <= GMMVI_BLOCKED_MAX_DIM; }
inline int gmmvi_blocked_linv_ofs(int D) { return ((D + 1 + 3) / 4) * 4; }
inline size_t gmmvi_blocked_stride(int D) { return (size_t)gmmvi_blocked_linv_ofs(D) + (size_t)D * D; }

int gmmvi_blocked_pack(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* means, const float* chols,
                       float* packed, float* inv_chols);
int gmmvi_blocked_cholesky(gmmvi_ctx* ctx, int K, int D, const float* covs, float* chols, int32_t* ok);
int gmmvi_blocked_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed, const float* logw,
                               const float* logw2, const float* X, int N, float* ld, float* lp, float* grad, float* lp2);
int gmmvi_blocked_sample(gmmvi_ctx* ctx, int K, int D, const float* means, const float* chols, const int32_t* offsets, int N,
                         int max_per_component, uint64_t seed, uint64_t first_index, int stream_id, const float* eps,
                         float* X_out, int32_t* mapping_out);
int gmmvi_blocked_stein(gmmvi_ctx* ctx, int K, int D, const float* packed, const float* X, int N, const float* ld,
                        const float* qgrad, const float* bg, const float* tgrad, const int32_t* mapping, int map_offset,
                        int flags, float* H_neg, float* g_neg);
int gmmvi_blocked_update_kl(gmmvi_ctx* ctx, int K, int D, float* means, float* chols, const float* H_neg, const float* g_neg,
                            const float* stepsizes, float temperature, float l2_init, float* last_eta, float* l2,
                            float* num_updates, int32_t* success_out, float* kl_out, int32_t* nprobes_out, float* packed_out);
int gmmvi_blocked_update_plain(gmmvi_ctx* ctx, int mode, int K, int D, float* means, float* chols, const float* H_neg,
                               const float* g_neg, const float* stepsizes, float l2_init, float* l2, float* num_updates,
                               int32_t* success_out);
