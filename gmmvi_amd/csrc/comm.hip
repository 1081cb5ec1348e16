// Multi-GPU exchange for component shards: one process per GPU, RCCL over xGMI.
// The reference has no collective; these are the E1/E2/E3 exchanges the sharded design introduces.
#include "common.h"
#include <rccl/rccl.h>

#define GMMVI_NCCL_CHECK(ctx, call)                                                                        \
    do {                                                                                                   \
        ncclResult_t r__ = (call);                                                                         \
        if (r__ != ncclSuccess)                                                                            \
            return gmmvi_fail(ctx, GMMVI_ERR_RCCL, std::string(#call) + ": " + ncclGetErrorString(r__));   \
    } while (0)

// lp[n] = LSE_r lp_r[n];  grad[n,:] = sum_r exp(lp_r[n] - lp[n]) grad_r[n,:]
// lp[n] = log sum_r exp(lp_r[n]);  grad[n, :] = sum_r exp(lp_r[n] - lp[n]) grad_r[n, :].
// Thread = one element of the [N, D] gradient (coalesced over r-major partial arrays); the thread of column 0 also writes lp.
// Fixed summation order over r.
__global__ __launch_bounds__(256) void combine_partials_kernel(int R, int N, int D, const float* __restrict__ lp_parts,
                                                               const float* __restrict__ grad_parts, float* __restrict__ lp_out,
                                                               float* __restrict__ grad_out, const float* __restrict__ lp2_parts,
                                                               float* __restrict__ lp2_out) {
    const bool with_grad = grad_out != nullptr && grad_parts != nullptr;
    const int width = with_grad ? D : 1;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * width) return;
    const int n = (int)(e / width), i = (int)(e - (long)n * width);
    float m = -3.0e38f;
    for (int r = 0; r < R; ++r) m = fmaxf(m, lp_parts[(size_t)r * N + n]);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += __expf(lp_parts[(size_t)r * N + n] - m);
    const float lp = m + __logf(s);
    if (lp_out && i == 0) lp_out[n] = lp;
    if (lp2_out && i == (width > 1 ? 1 : 0)) {          // second mixture over the same components (log values only)
        float m2 = -3.0e38f;
        for (int r = 0; r < R; ++r) m2 = fmaxf(m2, lp2_parts[(size_t)r * N + n]);
        float s2 = 0.f;
        for (int r = 0; r < R; ++r) s2 += __expf(lp2_parts[(size_t)r * N + n] - m2);
        lp2_out[n] = m2 + __logf(s2);
    }
    if (with_grad) {
        float g = 0.f;
        for (int r = 0; r < R; ++r)
            g = fmaf(__expf(lp_parts[(size_t)r * N + n] - lp), grad_parts[((size_t)r * N + n) * D + i], g);
        grad_out[(size_t)n * D + i] = g;
    }
}

// C++ linkage (common.h): gmmvi_combine_partials plus an optional second set of log-value partials (dual mixture sweep)
int gmmvi_combine_partials_internal(gmmvi_ctx* ctx, int R, int N, int D, const float* lp_parts_dev,
                                    const float* grad_parts_dev, float* lp_out_dev, float* grad_out_dev,
                                    const float* lp2_parts_dev, float* lp2_out_dev) {
    GMMVI_ARG_CHECK(ctx, R >= 1 && N >= 0 && D >= 1 && lp_parts_dev);
    if (N == 0) return GMMVI_OK;
    const long elems = (long)N * ((grad_out_dev && grad_parts_dev) ? D : 1);
    hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, ctx->stream, R, N, D,
                       lp_parts_dev, grad_parts_dev, lp_out_dev, grad_out_dev, lp2_parts_dev,
                       lp2_parts_dev ? lp2_out_dev : nullptr);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

extern "C" {

int gmmvi_comm_unique_id(char* out_id_128) {
    if (!out_id_128) return gmmvi_fail(nullptr, GMMVI_ERR_ARG, "gmmvi_comm_unique_id: NULL buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return gmmvi_fail(nullptr, GMMVI_ERR_RCCL, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    memcpy(out_id_128, &id, 128);
    return GMMVI_OK;
}

int gmmvi_comm_init(gmmvi_ctx* ctx, const char* unique_id_128, int n_ranks, int rank) {
    GMMVI_ARG_CHECK(ctx, unique_id_128 && n_ranks >= 1 && rank >= 0 && rank < n_ranks);
    if (ctx->comm) return gmmvi_fail(ctx, GMMVI_ERR_STATE, "communicator already initialised");
    ncclUniqueId id;
    memcpy(&id, unique_id_128, 128);
    GMMVI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm;
    GMMVI_NCCL_CHECK(ctx, ncclCommInitRank(&comm, n_ranks, id, rank));
    ctx->comm = (void*)comm;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return GMMVI_OK;
}

int gmmvi_comm_destroy(gmmvi_ctx* ctx) {
    if (ctx && ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
        ctx->n_ranks = 1;
        ctx->rank = 0;
    }
    return GMMVI_OK;
}

int gmmvi_allgather_f32(gmmvi_ctx* ctx, const float* send_dev, float* recv_dev, size_t count_per_rank) {
    GMMVI_ARG_CHECK(ctx, send_dev && recv_dev);
    if (ctx->n_ranks == 1 || !ctx->comm) {
        if (send_dev != recv_dev)
            GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(recv_dev, send_dev, count_per_rank * sizeof(float),
                                                hipMemcpyDeviceToDevice, ctx->stream));
        return GMMVI_OK;
    }
    GMMVI_NCCL_CHECK(ctx, ncclAllGather(send_dev, recv_dev, count_per_rank, ncclFloat, (ncclComm_t)ctx->comm,
                                        ctx->stream));
    return GMMVI_OK;
}

int gmmvi_allreduce_f32(gmmvi_ctx* ctx, float* buf_dev, size_t count, int op) {
    GMMVI_ARG_CHECK(ctx, buf_dev && (op == 0 || op == 1));
    if (ctx->n_ranks == 1 || !ctx->comm) return GMMVI_OK;
    GMMVI_NCCL_CHECK(ctx, ncclAllReduce(buf_dev, buf_dev, count, ncclFloat, op == 0 ? ncclSum : ncclMax,
                                        (ncclComm_t)ctx->comm, ctx->stream));
    return GMMVI_OK;
}

int gmmvi_combine_partials(gmmvi_ctx* ctx, int R, int N, int D, const float* lp_parts_dev,
                           const float* grad_parts_dev, float* lp_out_dev, float* grad_out_dev) {
    return gmmvi_combine_partials_internal(ctx, R, N, D, lp_parts_dev, grad_parts_dev, lp_out_dev, grad_out_dev, nullptr,
                                           nullptr);
}

}  // extern "C"
