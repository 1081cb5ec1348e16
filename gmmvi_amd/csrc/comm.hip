// Multi-GPU exchange for component shards: one process per GPU, RCCL over xGMI.
// The reference has no collective; these are the E1/E2/E3 exchanges the sharded design introduces.
#include "common.h"
#include "combine.h"
#include <rccl/rccl.h>

#define GMMVI_NCCL_CHECK(ctx, call)                                                                        \
    do {                                                                                                   \
        ncclResult_t r__ = (call);                                                                         \
        if (r__ != ncclSuccess)                                                                            \
            return gmmvi_fail(ctx, GMMVI_ERR_RCCL, std::string(#call) + ": " + ncclGetErrorString(r__));   \
    } while (0)

// stand-alone merge launch (combine.h has the arithmetic)
__global__ __launch_bounds__(256) void combine_partials_kernel(CombineJob j) {
    combine_element(j, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// C++ linkage (common.h): gmmvi_combine_partials plus an optional second set of log-value partials (dual mixture sweep)
int gmmvi_combine_partials_internal(gmmvi_ctx* ctx, int R, int N, int D, const float* lp_parts_dev,
                                    const float* grad_parts_dev, float* lp_out_dev, float* grad_out_dev,
                                    const float* lp2_parts_dev, float* lp2_out_dev, long part_stride) {
    GMMVI_ARG_CHECK(ctx, R >= 1 && N >= 0 && D >= 1 && lp_parts_dev && part_stride >= 0);
    if (N == 0) return GMMVI_OK;
    CombineJob j;
    j.R = R; j.N = N; j.D = D; j.part_stride = part_stride;
    j.lp_parts = lp_parts_dev; j.grad_parts = grad_parts_dev; j.lp2_parts = lp2_parts_dev;
    j.lp_out = lp_out_dev; j.grad_out = grad_out_dev; j.lp2_out = lp2_parts_dev ? lp2_out_dev : nullptr;
    const long elems = (long)N * ((grad_out_dev && grad_parts_dev) ? D : 1);
    hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, ctx->stream, j);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_defer_reserve(gmmvi_ctx* ctx, size_t nbytes) {
    if (ctx->pending.R > 0) return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_defer_reserve: a merge is still pending");
    if (nbytes <= ctx->defer_bytes) return GMMVI_OK;
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->defer_ws) GMMVI_HIP_CHECK(ctx, hipFree(ctx->defer_ws));
    ctx->defer_ws = nullptr;
    ctx->defer_bytes = 0;
    const size_t want = nbytes + nbytes / 2;
    GMMVI_HIP_CHECK(ctx, hipMalloc(&ctx->defer_ws, want));
    ctx->defer_bytes = want;
    return GMMVI_OK;
}

int gmmvi_flush_pending_combine(gmmvi_ctx* ctx) {
    if (ctx->pending.R == 0) return GMMVI_OK;
    const CombineJob j = ctx->pending;
    ctx->pending = CombineJob();
    GMMVI_PROF(ctx, "mixture_combine");
    return gmmvi_combine_partials_internal(ctx, j.R, j.N, j.D, j.lp_parts, j.grad_parts, j.lp_out, j.grad_out, j.lp2_parts,
                                           j.lp2_out);
}

CombineJob gmmvi_take_pending_combine(gmmvi_ctx* ctx, int threads, int first_block, bool light) {
    CombineJob j = ctx->pending;
    ctx->pending = CombineJob();
    if (j.R == 0) return j;
    const long elems = (long)j.N * ((j.grad_out && j.grad_parts) ? j.D : 1);
    long blocks;
    if (light) {
        // a carrying kernel with a small footprint (no LDS, small workgroups): two elements per thread
        blocks = (elems + 2L * threads - 1) / (2L * threads);
        if (blocks > 16L * ctx->num_cus) blocks = 16L * ctx->num_cus;
    } else {
        // the carried workgroups have the resource footprint of the carrying kernel: give them the CUs its own workgroups
        // leave idle in their last round rather than a further round
        blocks = ctx->num_cus - first_block % ctx->num_cus;
        if (blocks < ctx->num_cus / 4) blocks += ctx->num_cus;
        const long useful = (elems + threads - 1) / threads;
        if (blocks > useful) blocks = useful;
    }
    if (blocks < 1) blocks = 1;
    j.first_block = first_block;
    j.blocks = (int)blocks;
    return j;
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
