// Shared host-side plumbing for libgmmvi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/gmmvi_hip.h"

struct gmmvi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scratch reused by multi-kernel entry points (grown on demand, never shrunk)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // Every gmmvi_ws_reserve starts a new epoch of the scratch: contents survive from one entry point to the next only by
    // the explicit hand-over below.  Blocked path: the whitened samples Z of the last single-chunk density sweep stay at the
    // start of ws; the Stein estimate that follows reuses them when the scratch is untouched and 64-bit content hashes of
    // (component blocks, samples) match (blocked.hip).
    uint64_t ws_epoch = 0;
    struct ZCache { bool valid = false; uint64_t epoch = 0; const void* ws = nullptr; int K = 0, N = 0, D = 0, ldz = 0; } zc;
    unsigned long long* zc_hash = nullptr;       // device: [0,1] recorded with Z, [2,3] current call, [4] match flag
    void* arena = nullptr;       // persistent scratch of the single-call iteration (fused.hip)
    size_t arena_bytes = 0;
    void* comm = nullptr;        // ncclComm_t
    int n_ranks = 1, rank = 0;
    int num_cus = 256;
    // optional per-kernel HIP-event timing (bench.py roofline leg): events are recorded on ctx->stream
    bool prof = false;
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
// sampling.hip: gmmvi_sample_components with a caller-known bound on the samples per component (fewer empty workgroups)
int gmmvi_sample_components_bounded(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                                    const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed,
                                    uint64_t first_index, int stream_id, const float* eps_dev, float* X_out_dev,
                                    int32_t* mapping_out_dev);
struct PrepArgs;
int gmmvi_sample_components_prep(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                                 const int32_t* offsets_dev, int N, int max_per_component, uint64_t seed, uint64_t first_index,
                                 float* X_out_dev, int32_t* mapping_out_dev, int32_t mapping_base, const PrepArgs& prep);
// comm.hip: gmmvi_combine_partials plus an optional second set of log-value partials
int gmmvi_combine_partials_internal(gmmvi_ctx* ctx, int R, int N, int D, const float* lp_parts_dev,
                                    const float* grad_parts_dev, float* lp_out_dev, float* grad_out_dev,
                                    const float* lp2_parts_dev, float* lp2_out_dev);
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

// ---- packed component block layout (floats) ---------------------------------------------------------------
template <int DP>
struct Pack {
    static constexpr int T = DP * (DP - 1) / 2;
    static constexpr int MU = 0;              // mu[DP]
    static constexpr int RD = DP;             // 1 / L_ii
    static constexpr int LROW = 2 * DP;       // strict lower triangle, row-major: (i, j<i) at i(i-1)/2 + j
    static constexpr int LCOL = 2 * DP + T;   // same entries column-major: (j>i, i) at colofs(i) + j-i-1
    static constexpr int CONST = 2 * DP + 2 * T;   // log-normaliser
    static constexpr int STRIDE = ((2 * DP + 2 * T + 1 + 3) / 4) * 4;
    __host__ __device__ static constexpr int rowofs(int i) { return i * (i - 1) / 2; }
    __host__ __device__ static constexpr int colofs(int i) { return i * (DP - 1) - i * (i - 1) / 2; }
};

inline size_t gmmvi_packed_stride_dp(int dp) { return (size_t)((2 * dp + dp * (dp - 1) + 1 + 3) / 4) * 4; }
