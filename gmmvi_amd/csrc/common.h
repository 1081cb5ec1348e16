// Shared host-side plumbing for libgmmvi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/gmmvi_hip.h"
#include "iter_prep.h"

// a merge of component-chunk partials (combine.h) that has not been launched yet: the single-call iteration lets the next
// launch carry it (density.hip defers, the target / expected-log-ratio launches take it)
struct CombineJob {
    int R = 0, N = 0, D = 0;
    long part_stride = 0;                 // floats between the parts of consecutive chunks (0: N log values, N * D gradient entries)
    int first_block = 0, blocks = 0;      // set by the carrying launcher
    const float* lp_parts = nullptr;
    const float* grad_parts = nullptr;
    const float* lp2_parts = nullptr;
    float* lp_out = nullptr;
    float* grad_out = nullptr;
    float* lp2_out = nullptr;
};

// work that rides as extra workgroups in a density launch of the single-call iteration (riders.h)
struct SampleJob {
    int K = 0, D = 0, uniform_count = 0;
    const float* means = nullptr; const float* chols = nullptr; const int32_t* offsets = nullptr;
    unsigned long long seed = 0, first_index = 0;
    float* X = nullptr; int32_t* mapping = nullptr; int32_t mapping_base = 0;
};

struct Riders {
    int first_block = 0;          // the carrying launch's own workgroups and its carried merge end here
    int prep_blocks = 0;          // workgroups first_block .. : iter_prep_block
    int sample_blocks = 0;        // then: sample_block, workgroup i = (component i % K, 256-sample chunk i / K)
    PrepArgs prep;
    SampleJob sample;
};

inline size_t riders_lds_bytes(const Riders& r) {
    return r.sample_blocks > 0 ? ((size_t)r.sample.D * r.sample.D + r.sample.D + 256 * (size_t)(r.sample.D | 1)) * sizeof(float) : 0;
}

struct gmmvi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scratch reused by multi-kernel entry points (grown on demand, never shrunk)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    uint64_t ws_epoch = 0;       // bumped by every gmmvi_ws_reserve: the scratch contents belong to the call that reserved them
    void* arena = nullptr;       // persistent scratch of the single-call iteration (fused.hip)
    size_t arena_bytes = 0;
    // small host -> device copies go through a ring of pinned staging slots and do not wait for the stream (api.hip: gmmvi_upload)
    static constexpr int UP_SLOTS = 64;
    static constexpr size_t UP_SLOT_BYTES = 64 * 1024;
    unsigned char* up_ring = nullptr;
    hipEvent_t up_event[UP_SLOTS] = {};
    bool up_used[UP_SLOTS] = {};
    int up_next = 0;
    void* bimg = nullptr;        // split bf16 images of the B operands of the blocked contractions (blocked.hip), grown on demand
    size_t bimg_bytes = 0;
    bool defer_combine = false;  // set by fused.hip around a sweep whose merge the next launch carries
    CombineJob pending;          // R > 0: partials in defer_ws wait for their merge
    Riders riders;               // prep_blocks / sample_blocks > 0: riders wait for the next density launch (riders.h)
    void* defer_ws = nullptr;    // partials of a deferred merge (ctx->ws is reused by the launches in between)
    size_t defer_bytes = 0;
    void* comm = nullptr;        // ncclComm_t
    int n_ranks = 1, rank = 0;
    int num_cus = 256;
    unsigned func_attr_done = 0; // bits: per-device kernel attributes (dynamic LDS above 64 KB) already set for this context's device
    // optional per-kernel HIP-event timing (bench.py roofline leg): events are recorded on ctx->stream
    bool prof = false;
    const char* prof_tag = nullptr;   // set by a caller that knows which sweep of the iteration a density launch is (fused.hip)
    struct ProfRec { const char* name; hipEvent_t start, stop; double units; };   // units: (sample, component) pairs of the launch
    std::vector<ProfRec> prof_recs;
};

struct GmmviProfScope {
    gmmvi_ctx* ctx; hipEvent_t stop = nullptr;
    GmmviProfScope(gmmvi_ctx* c, const char* name, double units = 0.0) : ctx(c) {
        if (!c->prof) return;
        hipEvent_t start;
        if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) { stop = nullptr; return; }
        (void)hipEventRecord(start, c->stream);
        c->prof_recs.push_back({name, start, stop, units});
    }
    ~GmmviProfScope() { if (stop) (void)hipEventRecord(stop, ctx->stream); }
};
#define GMMVI_PROF(ctx, name) GmmviProfScope prof_scope__(ctx, name)
#define GMMVI_PROF_UNITS(ctx, name, units) GmmviProfScope prof_scope__(ctx, name, units)

extern std::string g_gmmvi_global_err;

inline int gmmvi_fail(gmmvi_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_gmmvi_global_err = msg;
    return code;
}

#define GMMVI_HIP_CHECK(ctx, call)                                                                    \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return gmmvi_fail(ctx, GMMVI_ERR_HIP,                                                     \
                              std::string(#call) + ": " + hipGetErrorString(e__) + " (" + __FILE__ +  \
                                  ":" + std::to_string(__LINE__) + ")");                              \
    } while (0)

#define GMMVI_ARG_CHECK(ctx, cond)                                                                    \
    do {                                                                                              \
        if (!(cond))                                                                                  \
            return gmmvi_fail(ctx, GMMVI_ERR_ARG, std::string("invalid argument: ") + #cond + " (" +  \
                                                      __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define GMMVI_LAUNCH_CHECK(ctx) GMMVI_HIP_CHECK(ctx, hipGetLastError())

int gmmvi_ws_reserve(gmmvi_ctx* ctx, size_t nbytes);
// density.hip: component blocks in the register-path layout (Pack<padded D>, D <= 64) regardless of the blocked threshold
int gmmvi_pack_register_layout(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev, float* packed_dev);
// sampling.hip: gmmvi_sample_components with a caller-known bound on the samples per component (fewer empty workgroups)
int gmmvi_sample_components_bounded(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                                    const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed,
                                    uint64_t first_index, int stream_id, const float* eps_dev, float* X_out_dev,
                                    int32_t* mapping_out_dev);
int gmmvi_sample_components_prep(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                                 const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed, uint64_t first_index,
                                 float* X_out_dev, int32_t* mapping_out_dev, int32_t mapping_base, const PrepArgs& prep);
// comm.hip: gmmvi_combine_partials plus an optional second set of log-value partials
int gmmvi_combine_partials_internal(gmmvi_ctx* ctx, int R, int N, int D, const float* lp_parts_dev,
                                    const float* grad_parts_dev, float* lp_out_dev, float* grad_out_dev,
                                    const float* lp2_parts_dev, float* lp2_out_dev, long part_stride = 0);
// comm.hip: deferred merges.  reserve: room for the partials (no merge may be pending); flush: launch a pending merge on its
// own; take: hand a pending merge to a launch of `threads` threads per workgroup whose own workgroups end at `first_block`
int gmmvi_defer_reserve(gmmvi_ctx* ctx, size_t nbytes);
int gmmvi_flush_pending_combine(gmmvi_ctx* ctx);
CombineJob gmmvi_take_pending_combine(gmmvi_ctx* ctx, int threads, int first_block, bool light = false);
// comm.hip: riders.  take: hand the pending riders to a density launch whose own and merge workgroups end at `first_block`;
// flush: launch pending riders on their own (nothing took them)
Riders gmmvi_take_pending_riders(gmmvi_ctx* ctx, int first_block, int threads, bool prep_only = false);
int gmmvi_flush_pending_riders(gmmvi_ctx* ctx);
// stein.hip / update_kl.hip: the Stein estimate split at the partial slab (single-call iteration, fused.hip)
struct SteinSlab;
int gmmvi_stein_partials(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N, const float* ld_dev,
                         const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev, int flags, SteinSlab* slab);
int gmmvi_stein_finalize_slab(gmmvi_ctx* ctx, int K, int D, const SteinSlab& slab, int N, int flags, const float* packed_dev,
                              float* H_neg_out_dev, float* g_neg_out_dev);
int gmmvi_update_components_kl_from_slab(gmmvi_ctx* ctx, int K, int D, const SteinSlab& slab, int N, int stein_flags,
                                         const float* packed_old_dev, float* H_neg_dev, float* g_neg_dev, float* means_dev,
                                         float* chols_dev, const float* stepsizes_dev, float temperature, float l2_init,
                                         float* last_eta_dev, float* l2_dev, float* num_received_updates_dev,
                                         int32_t* success_out_dev, float* packed_out_dev);
// weights.hip: trust-region (mode 0) / direct (mode 1) weight update; exp_out (optional) receives exp(new log weights)
int gmmvi_update_weights_internal(gmmvi_ctx* ctx, int mode, int K, float* logw_dev, const float* E_dev,
                                  const float* stepsize_dev, float beta, float* kl_eta_out_dev, float* exp_out_dev);

// ---- padded dimensions the register-resident kernels are instantiated for --------------------------------
// A problem of dimension D runs in the smallest DP >= D; padded coordinates carry x = mu = 0, L_ii = 1, so
// they contribute nothing (z = y = 0, log L_ii = 0).
inline int gmmvi_padded_dim(int D) {
    static const int dps[] = {2, 4, 8, 10, 12, 16, 20, 24, 32, 40, 50, 64};
    for (int dp : dps) if (D <= dp) return dp;
    return -1;
}

#define GMMVI_DISPATCH_DP(DPVAL, ...)                  \
    switch (DPVAL) {                                   \
        case 2:  { constexpr int DP = 2;  __VA_ARGS__; } break;  \
        case 4:  { constexpr int DP = 4;  __VA_ARGS__; } break;  \
        case 8:  { constexpr int DP = 8;  __VA_ARGS__; } break;  \
        case 10: { constexpr int DP = 10; __VA_ARGS__; } break;  \
        case 12: { constexpr int DP = 12; __VA_ARGS__; } break;  \
        case 16: { constexpr int DP = 16; __VA_ARGS__; } break;  \
        case 20: { constexpr int DP = 20; __VA_ARGS__; } break;  \
        case 24: { constexpr int DP = 24; __VA_ARGS__; } break;  \
        case 32: { constexpr int DP = 32; __VA_ARGS__; } break;  \
        case 40: { constexpr int DP = 40; __VA_ARGS__; } break;  \
        case 50: { constexpr int DP = 50; __VA_ARGS__; } break;  \
        case 64: { constexpr int DP = 64; __VA_ARGS__; } break;  \
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "unsupported dimension (D must be <= 64)"); \
    }

// Padded dimension from which the density sweeps run on the matrix cores with the explicit inverse (density.hip).  Measured at
// K = 100, N = 10^4 (profiles/r02_notes.md): D = 50 dual sweep + target 368 -> 178 us, post-update sweep 100 -> 61 us; D = 20 and D = 10
// are slower there (a 16-sample sub-tile pays the log-sum-exp tail four times as often per pair): they keep the scalar-fed
// substitution kernel and the smaller block.
#ifndef GMMVI_MFMA_DENSITY_FROM_DP
#define GMMVI_MFMA_DENSITY_FROM_DP 32
#endif

// ---- packed component block layout (floats) ---------------------------------------------------------------
// [mu | 1/diag L | strict lower triangle of L by rows | the same by columns | log-normaliser | pad | L^-1 as matrix-core
// operand fragments].  The fragments feed v_mfma_f32_16x16x4_f32 directly (A operand: lane l holds A[i = l & 15][k = l >> 4]),
// one 64-float fragment per (16-row tile mt, 4-column step s):
//   forward  z = L^-1 d:    FWD + 64 fwd_index(mt, s) + l  =  Linv[16 mt + (l & 15)][4 s + (l >> 4)],   s < nf(mt)  (lower triangle)
//   backward y = L^-T z:    BWD + 64 bwd_index(mt, s) + l  =  Linv[4 s + (l >> 4)][16 mt + (l & 15)],   s >= 4 mt
// (entries outside D x D are zero), so a wave fetches a fragment with one coalesced 256-byte load.
template <int DP>
struct Pack {
    static constexpr int T = DP * (DP - 1) / 2;
    static constexpr int MU = 0;              // mu[DP]
    static constexpr int RD = DP;             // 1 / L_ii
    static constexpr int LROW = 2 * DP;       // strict lower triangle, row-major: (i, j<i) at i(i-1)/2 + j
    static constexpr int LCOL = 2 * DP + T;   // same entries column-major: (j>i, i) at colofs(i) + j-i-1
    static constexpr int CONST = 2 * DP + 2 * T;   // log-normaliser
    static constexpr bool FRAGS = DP >= GMMVI_MFMA_DENSITY_FROM_DP;      // smaller blocks carry no fragments (scalar-fed kernel)
    static constexpr int MT = (DP + 15) / 16;      // 16-row tiles
    static constexpr int KS = (DP + 3) / 4;        // 4-column steps
    __host__ __device__ static constexpr int nf(int mt) { return 4 * mt + 4 < KS ? 4 * mt + 4 : KS; }     // forward steps 0 .. nf-1
    __host__ __device__ static constexpr int nbk(int mt) { return KS - 4 * mt > 0 ? KS - 4 * mt : 0; }    // backward steps 4 mt .. KS-1
    __host__ __device__ static constexpr int fwd_base(int mt) { int a = 0; for (int m = 0; m < mt; ++m) a += nf(m); return a; }
    __host__ __device__ static constexpr int bwd_base(int mt) { int a = 0; for (int m = 0; m < mt; ++m) a += nbk(m); return a; }
    __host__ __device__ static constexpr int fwd_index(int mt, int s) { return fwd_base(mt) + s; }
    __host__ __device__ static constexpr int bwd_index(int mt, int s) { return bwd_base(mt) + (s - 4 * mt); }
    static constexpr int NF = FRAGS ? fwd_base(MT) : 0, NB = FRAGS ? bwd_base(MT) : 0;
    static constexpr int FWD = FRAGS ? ((CONST + 1 + 63) / 64) * 64 : ((CONST + 1 + 3) / 4) * 4;
    static constexpr int BWD = FWD + 64 * NF;
    // every block ends in the SWEEP STREAM: the floats a component pass of the packed
    // two-samples-per-lane sweep reads, in the order it reads them (density.hip, subst_pk.h), each part 16-byte aligned:
    //   SWH  mu[DP], log-normaliser
    //   SWF  forward substitution by columns, the reciprocal of the diagonal entry in front of its column:
    //        column j = [1 / L_jj, L_{j+1,j}, ..., L_{DP-1,j}]  at SWF + swf_col(j)
    //   SWB  backward substitution by rows (read from the end), the reciprocal of the diagonal entry behind its row:
    //        row i = [L_{i,0}, ..., L_{i,i-1}, 1 / L_ii]  at SWB + swb_row(i)
    static constexpr int TD = T + DP;
    static constexpr int SWH = BWD + 64 * NB;
    static constexpr int SWF = SWH + ((DP + 1 + 3) / 4) * 4;
    static constexpr int SWB = SWF + ((TD + 3) / 4) * 4;
    static constexpr int STRIDE = SWB + ((TD + 3) / 4) * 4;
    __host__ __device__ static constexpr int swf_col(int j) { return j * DP - j * (j - 1) / 2; }
    __host__ __device__ static constexpr int swb_row(int i) { return i * (i + 1) / 2; }
    __host__ __device__ static constexpr int rowofs(int i) { return i * (i - 1) / 2; }
    __host__ __device__ static constexpr int colofs(int i) { return i * (DP - 1) - i * (i - 1) / 2; }
};

// (run-time twin of Pack<DP>::STRIDE / ::FWD / ::BWD for code that takes the padded dimension as an argument)
struct PackDims { int fwd, bwd, stride, mt, ks, nf_total, nb_total, swh, swf, swb; };   // swh < 0: no sweep stream
inline __host__ __device__ PackDims gmmvi_pack_dims(int dp) {
    PackDims d{};
    const int T = dp * (dp - 1) / 2;
    d.mt = (dp + 15) / 16; d.ks = (dp + 3) / 4;
    const bool frags = dp >= GMMVI_MFMA_DENSITY_FROM_DP;
    for (int m = 0; frags && m < d.mt; ++m) {
        d.nf_total += 4 * m + 4 < d.ks ? 4 * m + 4 : d.ks;
        d.nb_total += d.ks - 4 * m > 0 ? d.ks - 4 * m : 0;
    }
    d.fwd = frags ? ((2 * dp + 2 * T + 1 + 63) / 64) * 64 : ((2 * dp + 2 * T + 1 + 3) / 4) * 4;
    d.bwd = d.fwd + 64 * d.nf_total;
    d.stride = d.bwd + 64 * d.nb_total;
    d.swh = d.stride;
    d.swf = d.swh + ((dp + 1 + 3) / 4) * 4;
    d.swb = d.swf + ((T + dp + 3) / 4) * 4;
    d.stride = d.swb + ((T + dp + 3) / 4) * 4;
    return d;
}
inline size_t gmmvi_packed_stride_dp(int dp) { return (size_t)gmmvi_pack_dims(dp).stride; }

// The sweep stream of one block (Pack<DP>::SWH / SWF / SWB) from a lower-triangular factor L[i * ld + j] (rows / columns >= D:
// identity) and the mean, written by the calling threads tid, tid + nthreads, ...; the log-normaliser (element SWH + dp) is
// left to the caller's thread that holds it
__device__ __forceinline__ void gmmvi_write_sweep_stream(float* __restrict__ out, int dp, int D, const float* L, int ld,
                                                         const float* mu, int tid, int nthreads) {
    const PackDims pd = gmmvi_pack_dims(dp);
    if (pd.swh < 0) return;
    for (int i = tid; i < pd.swf - pd.swh; i += nthreads)
        if (i != dp) out[pd.swh + i] = i < D ? mu[i] : 0.f;
    for (int e = tid; e < dp * dp; e += nthreads) {
        const int i = e / dp, j = e - i * dp;
        if (j > i) continue;
        const float v = i == j ? (i < D ? 1.f / L[i * ld + i] : 1.f) : (i < D ? L[i * ld + j] : 0.f);
        out[pd.swf + j * dp - j * (j - 1) / 2 + (i - j)] = v;
        out[pd.swb + i * (i + 1) / 2 + j] = v;
    }
    const int td = dp * (dp - 1) / 2 + dp, tdr = (td + 3) / 4 * 4;
    for (int e = td + tid; e < tdr; e += nthreads) { out[pd.swf + e] = 0.f; out[pd.swb + e] = 0.f; }
}

// The L^-1 fragments of one block from a dense row-major inverse Linv[i * ld + j] (rows / columns >= D read as zero), written
// by the calling threads tid, tid + nthreads, ...
__device__ __forceinline__ void gmmvi_write_inverse_fragments(float* __restrict__ out, int dp, int D, const float* Linv, int ld,
                                                              int tid, int nthreads) {
    const PackDims pd = gmmvi_pack_dims(dp);
    for (int e = tid; e < 64 * (pd.nf_total + pd.nb_total); e += nthreads) {
        const int f = e >> 6, l = e & 63;
        int mt = 0, s = 0, row, col;
        if (f < pd.nf_total) {
            int rem = f;
            for (;; ++mt) { const int n = 4 * mt + 4 < pd.ks ? 4 * mt + 4 : pd.ks; if (rem < n) break; rem -= n; }
            s = rem;
            row = 16 * mt + (l & 15); col = 4 * s + (l >> 4);
        } else {
            int rem = f - pd.nf_total;
            for (;; ++mt) { const int n = pd.ks - 4 * mt; if (rem < n) break; rem -= n; }
            s = 4 * mt + rem;
            row = 4 * s + (l >> 4); col = 16 * mt + (l & 15);
        }
        out[pd.fwd + e] = (row < D && col <= row) ? Linv[row * ld + col] : 0.f;
    }
}
