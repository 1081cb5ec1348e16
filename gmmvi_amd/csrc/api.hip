// Context, error, memory and event entry points of the C ABI (include/gmmvi_hip.h).
#include "common.h"
#include <cstring>
#include "blocked.h"

std::string g_gmmvi_global_err;

int gmmvi_ws_reserve(gmmvi_ctx* ctx, size_t nbytes) {
    ++ctx->ws_epoch;
    if (nbytes <= ctx->ws_bytes) return GMMVI_OK;
    // A grow frees the old block; earlier kernels on the stream may still read it, so drain first.
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) GMMVI_HIP_CHECK(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = nbytes + nbytes / 2;
    GMMVI_HIP_CHECK(ctx, hipMalloc(&ctx->ws, want));
    ctx->ws_bytes = want;
    return GMMVI_OK;
}

extern "C" {

int gmmvi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gmmvi_ctx_create(gmmvi_ctx** out, int device) {
    if (!out) return gmmvi_fail(nullptr, GMMVI_ERR_ARG, "gmmvi_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return gmmvi_fail(nullptr, GMMVI_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return gmmvi_fail(nullptr, GMMVI_ERR_ARG, "device index out of range");
    gmmvi_ctx* ctx = new gmmvi_ctx();
    ctx->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_gmmvi_global_err = std::string("gmmvi_ctx_create: ") + hipGetErrorString(e);
        delete ctx;
        return GMMVI_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    *out = ctx;
    return GMMVI_OK;
}

void gmmvi_ctx_destroy(gmmvi_ctx* ctx) {
    if (!ctx) return;
    gmmvi_comm_destroy(ctx);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->bimg) (void)hipFree(ctx->bimg);
    if (ctx->up_ring) {
        for (int i = 0; i < gmmvi_ctx::UP_SLOTS; ++i) (void)hipEventDestroy(ctx->up_event[i]);
        (void)hipHostFree(ctx->up_ring);
    }
    if (ctx->defer_ws) (void)hipFree(ctx->defer_ws);
    delete ctx;
}

const char* gmmvi_last_error(gmmvi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_gmmvi_global_err.c_str(); }

int gmmvi_sync(gmmvi_ctx* ctx) {
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GMMVI_OK;
}

int gmmvi_malloc(gmmvi_ctx* ctx, size_t nbytes, void** out_dev) {
    GMMVI_ARG_CHECK(ctx, out_dev != nullptr);
    *out_dev = nullptr;
    if (nbytes == 0) nbytes = 4;
    GMMVI_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GMMVI_HIP_CHECK(ctx, hipMalloc(out_dev, nbytes));
    return GMMVI_OK;
}

int gmmvi_free(gmmvi_ctx* ctx, void* dev) {
    if (!dev) return GMMVI_OK;
    // hipFree synchronises the device, so kernels still using the block have finished.
    GMMVI_HIP_CHECK(ctx, hipFree(dev));
    return GMMVI_OK;
}

int gmmvi_upload(gmmvi_ctx* ctx, void* dst_dev, const void* src_host, size_t nbytes) {
    if (nbytes == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && src_host);
    if (nbytes <= gmmvi_ctx::UP_SLOT_BYTES) {
        // small copies (weights, offsets, index lists: several per iteration of the sample-reuse path): through a pinned staging
        // slot, stream-ordered, WITHOUT waiting for the work already queued -- a synchronous copy drains the stream every time
        // (46 us each behind a queued iteration).  A slot is reused only after the copy that last read it has completed.
        if (ctx->up_ring == nullptr) {
            GMMVI_HIP_CHECK(ctx, hipHostMalloc((void**)&ctx->up_ring, gmmvi_ctx::UP_SLOTS * gmmvi_ctx::UP_SLOT_BYTES, hipHostMallocDefault));
            for (int i = 0; i < gmmvi_ctx::UP_SLOTS; ++i)
                GMMVI_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->up_event[i], hipEventDisableTiming));
        }
        const int slot = ctx->up_next;
        ctx->up_next = (slot + 1) % gmmvi_ctx::UP_SLOTS;
        if (ctx->up_used[slot]) GMMVI_HIP_CHECK(ctx, hipEventSynchronize(ctx->up_event[slot]));
        unsigned char* stage = ctx->up_ring + (size_t)slot * gmmvi_ctx::UP_SLOT_BYTES;
        memcpy(stage, src_host, nbytes);
        GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(dst_dev, stage, nbytes, hipMemcpyHostToDevice, ctx->stream));
        GMMVI_HIP_CHECK(ctx, hipEventRecord(ctx->up_event[slot], ctx->stream));
        ctx->up_used[slot] = true;
        return GMMVI_OK;
    }
    // pageable host memory: the runtime stages the copy, the call returns when the source may be reused
    GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(dst_dev, src_host, nbytes, hipMemcpyHostToDevice, ctx->stream));
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GMMVI_OK;
}

int gmmvi_download(gmmvi_ctx* ctx, void* dst_host, const void* src_dev, size_t nbytes) {
    if (nbytes == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_host && src_dev);
    GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GMMVI_OK;
}

// a device -> host copy that does NOT wait: dst must be pinned host memory (gmmvi_host_alloc); the data is there once an event
// recorded behind it has been reached (gmmvi_event_synchronize)
int gmmvi_download_async(gmmvi_ctx* ctx, void* dst_pinned_host, const void* src_dev, size_t nbytes) {
    if (nbytes == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_pinned_host && src_dev);
    GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(dst_pinned_host, src_dev, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    return GMMVI_OK;
}

int gmmvi_host_alloc(gmmvi_ctx* ctx, size_t nbytes, void** out_host) {
    GMMVI_ARG_CHECK(ctx, out_host != nullptr && nbytes > 0);
    GMMVI_HIP_CHECK(ctx, hipHostMalloc(out_host, nbytes, hipHostMallocDefault));
    return GMMVI_OK;
}

int gmmvi_host_free(gmmvi_ctx* ctx, void* host) {
    if (host) GMMVI_HIP_CHECK(ctx, hipHostFree(host));
    return GMMVI_OK;
}

// ---- grow-in-place device buffers (the sample database) ---------------------------------------------------------------------
// A buffer that doubles by hipMalloc + copy + hipFree costs seconds once it holds gigabytes (measured at the D = 300 shard:
// 1.5 - 3.7 s of host time per doubling of the five database arrays, 0.9 - 2.6 s of it with the GPU idle); on a 288 GB part the
// database is meant to grow that far.  Here the ADDRESS range is reserved once and physical memory is mapped behind the used
// part chunk by chunk (HIP virtual memory management): growing never moves a byte and never frees one.
int gmmvi_vmm_reserve(gmmvi_ctx* ctx, size_t max_bytes, void** base_out, size_t* chunk_bytes_out) {
    GMMVI_ARG_CHECK(ctx, base_out != nullptr && chunk_bytes_out != nullptr && max_bytes > 0);
    int supported = 0;
    GMMVI_HIP_CHECK(ctx, hipDeviceGetAttribute(&supported, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device));
    if (!supported) return gmmvi_fail(ctx, GMMVI_ERR_HIP, "gmmvi_vmm_reserve: virtual memory management is not supported on this device");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->device;
    size_t gran = 0;
    GMMVI_HIP_CHECK(ctx, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t chunk = (size_t)256 << 20;                        // one map call per 256 MiB of growth (tens of microseconds)
    if (chunk % gran) chunk = (chunk / gran + 1) * gran;
    const size_t total = (max_bytes + chunk - 1) / chunk * chunk;
    GMMVI_HIP_CHECK(ctx, hipMemAddressReserve(base_out, total, gran, nullptr, 0));
    *chunk_bytes_out = chunk;
    return GMMVI_OK;
}

// physical memory behind [mapped_bytes, new_mapped_bytes) of a reserved range (both multiples of the chunk size)
int gmmvi_vmm_grow(gmmvi_ctx* ctx, void* base, size_t chunk_bytes, size_t mapped_bytes, size_t new_mapped_bytes) {
    GMMVI_ARG_CHECK(ctx, base != nullptr && chunk_bytes > 0 && mapped_bytes % chunk_bytes == 0 && new_mapped_bytes % chunk_bytes == 0 &&
                         new_mapped_bytes >= mapped_bytes);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->device;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = ctx->device; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t at = mapped_bytes; at < new_mapped_bytes; at += chunk_bytes) {
        hipMemGenericAllocationHandle_t h;
        GMMVI_HIP_CHECK(ctx, hipMemCreate(&h, chunk_bytes, &prop, 0));
        hipError_t e = hipMemMap((char*)base + at, chunk_bytes, 0, h, 0);
        if (e == hipSuccess) e = hipMemSetAccess((char*)base + at, chunk_bytes, &acc, 1);
        (void)hipMemRelease(h);                              // the mapping keeps the memory alive until it is unmapped
        GMMVI_HIP_CHECK(ctx, e);
    }
    return GMMVI_OK;
}

int gmmvi_vmm_release(gmmvi_ctx* ctx, void* base, size_t chunk_bytes, size_t mapped_bytes, size_t reserved_bytes) {
    if (base == nullptr) return GMMVI_OK;
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));     // launches that still read the buffer
    for (size_t at = 0; at < mapped_bytes; at += chunk_bytes) GMMVI_HIP_CHECK(ctx, hipMemUnmap((char*)base + at, chunk_bytes));
    const size_t total = (reserved_bytes + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
    GMMVI_HIP_CHECK(ctx, hipMemAddressFree(base, total));
    return GMMVI_OK;
}

int gmmvi_copy(gmmvi_ctx* ctx, void* dst_dev, const void* src_dev, size_t nbytes) {
    if (nbytes == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && src_dev);
    GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(dst_dev, src_dev, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return GMMVI_OK;
}

__global__ void fill_f32_kernel(float* dst, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = v;
}

int gmmvi_fill_f32(gmmvi_ctx* ctx, float* dst_dev, float value, size_t count) {
    if (count == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev != nullptr);
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dst_dev, value, count);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int32_t* __restrict__ idx, int n_rows,
                                   int row_words, uint32_t* __restrict__ dst) {
    const long total = (long)n_rows * row_words;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / row_words, c = e % row_words;
        dst[e] = src[(long)idx[r] * row_words + c];
    }
}

int gmmvi_gather_rows(gmmvi_ctx* ctx, const void* src_dev, const int32_t* idx_dev, int n_rows, int row_words,
                      void* dst_dev) {
    GMMVI_ARG_CHECK(ctx, n_rows >= 0 && row_words >= 1);
    if (n_rows == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, src_dev && idx_dev && dst_dev);
    long total = (long)n_rows * row_words;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)src_dev, idx_dev,
                       n_rows, row_words, (uint32_t*)dst_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

struct CopyBatch { uint32_t* dst[8]; const uint32_t* src[8]; unsigned long long words[8]; int n; };

__global__ void copy_batch_kernel(CopyBatch b) {
    const int a = blockIdx.y;
    if (a >= b.n) return;
    const unsigned long long n = b.words[a];
    uint32_t* __restrict__ d = b.dst[a];
    const uint32_t* __restrict__ s = b.src[a];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        d[i] = s[i];
}

int gmmvi_copy_batch(gmmvi_ctx* ctx, int n, void* const* dst_dev, const void* const* src_dev, const size_t* nbytes) {
    GMMVI_ARG_CHECK(ctx, n >= 0 && n <= 8);
    if (n == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && src_dev && nbytes);
    CopyBatch b;
    b.n = n;
    unsigned long long mx = 0;
    for (int i = 0; i < 8; ++i) {
        b.dst[i] = i < n ? (uint32_t*)dst_dev[i] : nullptr;
        b.src[i] = i < n ? (const uint32_t*)src_dev[i] : nullptr;
        b.words[i] = i < n ? nbytes[i] / 4 : 0;
        if (i < n) {
            GMMVI_ARG_CHECK(ctx, nbytes[i] % 4 == 0 && (nbytes[i] == 0 || (dst_dev[i] && src_dev[i])));
            if (b.words[i] > mx) mx = b.words[i];
        }
    }
    if (mx == 0) return GMMVI_OK;
    int bx = (int)((mx + 1023) / 1024);
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    GMMVI_PROF(ctx, "copy_batch");
    hipLaunchKernelGGL(copy_batch_kernel, dim3(bx, n), dim3(256), 0, ctx->stream, b);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

struct UnpackArgs {
    const uint32_t* src; int n_ranks; unsigned long long chunk; int n_seg;
    unsigned long long off[8], words[8]; uint32_t* dst[8];
};
__global__ void unpack_gathered_kernel(UnpackArgs a) {
    const int j = blockIdx.y;
    if (j >= a.n_seg) return;
    const unsigned long long total = a.words[j] * (unsigned long long)a.n_ranks;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long r = i / a.words[j], e = i - r * a.words[j];
        a.dst[j][i] = a.src[r * a.chunk + a.off[j] + e];
    }
}

int gmmvi_unpack_gathered(gmmvi_ctx* ctx, const void* src_dev, int n_ranks, size_t chunk_words, int n_seg,
                          const size_t* seg_words, void* const* dst_dev) {
    GMMVI_ARG_CHECK(ctx, src_dev && n_ranks >= 1 && n_seg >= 1 && n_seg <= 8 && seg_words && dst_dev);
    UnpackArgs a{};
    a.src = (const uint32_t*)src_dev; a.n_ranks = n_ranks; a.chunk = chunk_words; a.n_seg = n_seg;
    unsigned long long off = 0, mx = 0;
    for (int j = 0; j < n_seg; ++j) {
        GMMVI_ARG_CHECK(ctx, dst_dev[j] != nullptr || seg_words[j] == 0);
        a.off[j] = off; a.words[j] = seg_words[j]; a.dst[j] = (uint32_t*)dst_dev[j];
        off += seg_words[j];
        if (seg_words[j] > mx) mx = seg_words[j];
    }
    GMMVI_ARG_CHECK(ctx, off == chunk_words);
    if (mx == 0) return GMMVI_OK;
    unsigned long long bx = (mx * (unsigned long long)n_ranks + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(unpack_gathered_kernel, dim3((unsigned)bx, n_seg), dim3(256), 0, ctx->stream, a);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

__global__ void fill_strided_kernel(float* dst, size_t stride, size_t count, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += step) dst[i * stride] = v;
}

int gmmvi_fill_strided_f32(gmmvi_ctx* ctx, float* dst_dev, size_t stride, size_t count, float value) {
    if (count == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev != nullptr && stride >= 1);
    int blocks = (int)((count + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(fill_strided_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dst_dev, stride, count, value);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// one thread per row: shift columns idx+1 .. K-1 one to the left (ascending, so the in-place move is safe)
__global__ void remove_column_kernel(float* a, int rows, size_t stride, int K, int idx) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float* row = a + (size_t)r * stride;
    for (int c = idx; c + 1 < K; ++c) row[c] = row[c + 1];
}

int gmmvi_remove_column_f32(gmmvi_ctx* ctx, float* a_dev, int rows, size_t stride, int K, int idx) {
    GMMVI_ARG_CHECK(ctx, a_dev && rows >= 0 && K >= 1 && idx >= 0 && idx < K && stride >= (size_t)K);
    if (rows == 0 || idx == K - 1) return GMMVI_OK;
    hipLaunchKernelGGL(remove_column_kernel, dim3((rows + 127) / 128), dim3(128), 0, ctx->stream, a_dev, rows, stride, K,
                       idx);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

__global__ void add_scalar_i32_kernel(int32_t* dst, const int32_t* src, int32_t v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i] + v;
}

int gmmvi_add_scalar_i32(gmmvi_ctx* ctx, int32_t* dst_dev, const int32_t* src_dev, int32_t value, size_t count) {
    if (count == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && src_dev);
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_scalar_i32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dst_dev, src_dev, value, count);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

__global__ void exp_f32_kernel(float* dst, const float* src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = expf(src[i]);
}

// GMM.replace_weights (models/gmm.py:173-181): out = lw - logsumexp(lw), the log-sum-exp in fp64 as the host form of it
// (optimization on the adaptive path: adding / removing a component no longer reads the weights back).  One workgroup.
__global__ __launch_bounds__(256) void normalize_logw_kernel(const float* __restrict__ in, int n, float* __restrict__ out) {
    __shared__ double red[256];
    const int t = threadIdx.x;
    double m = -1.0e300;
    for (int i = t; i < n; i += 256) m = fmax(m, (double)in[i]);
    red[t] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmax(red[t], red[t + o]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    double s = 0.0;
    for (int i = t; i < n; i += 256) s += exp((double)in[i] - m);
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const double lse = log(red[0]) + m;
    for (int i = t; i < n; i += 256) out[i] = (float)((double)in[i] - lse);
}

int gmmvi_normalize_logw(gmmvi_ctx* ctx, const float* logw_in_dev, int n, float* logw_out_dev) {
    GMMVI_ARG_CHECK(ctx, n >= 1 && logw_in_dev && logw_out_dev);
    hipLaunchKernelGGL(normalize_logw_kernel, dim3(1), dim3(256), 0, ctx->stream, logw_in_dev, n, logw_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// VipsComponentAdaptation.add_at_best_location (component_adaptation.py:192-226): the candidate with the largest
//   reward_n = log p~(x_n) - max(max_m log q(x_m) - threshold, log q(x_n)),
// first index on ties (argmax), evaluated in fp64 as the host form; index_out[0] = that candidate.  One workgroup.
__global__ __launch_bounds__(1024) void add_heuristic_argmax_kernel(const float* __restrict__ model_ld, const float* __restrict__ tlp,
                                                                    int n, double threshold, int32_t* __restrict__ index_out) {
    __shared__ double rv[1024];
    __shared__ int ri[1024];
    const int t = threadIdx.x;
    double m = -1.0e300;
    for (int i = t; i < n; i += 1024) m = fmax(m, (double)model_ld[i]);
    rv[t] = m;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) rv[t] = fmax(rv[t], rv[t + o]); __syncthreads(); }
    const double floor_ld = rv[0] - threshold;
    __syncthreads();
    double best = -1.0e300;
    int bi = 0x7fffffff;
    for (int i = t; i < n; i += 1024) {
        const double r = (double)tlp[i] - fmax(floor_ld, (double)model_ld[i]);
        if (bi == 0x7fffffff || r > best) { best = r; bi = i; }    // ascending i per thread: the first maximum stays
    }
    rv[t] = best; ri[t] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) {
            const double b2 = rv[t + o]; const int i2 = ri[t + o];
            if (i2 != 0x7fffffff && (ri[t] == 0x7fffffff || b2 > rv[t] || (b2 == rv[t] && i2 < ri[t]))) { rv[t] = b2; ri[t] = i2; }
        }
        __syncthreads();
    }
    if (t == 0) index_out[0] = ri[0] == 0x7fffffff ? 0 : ri[0];
}

int gmmvi_add_heuristic_argmax(gmmvi_ctx* ctx, const float* model_ld_dev, const float* target_lnpdfs_dev, int n, double threshold,
                               int32_t* index_out_dev) {
    GMMVI_ARG_CHECK(ctx, n >= 1 && model_ld_dev && target_lnpdfs_dev && index_out_dev);
    hipLaunchKernelGGL(add_heuristic_argmax_kernel, dim3(1), dim3(1024), 0, ctx->stream, model_ld_dev, target_lnpdfs_dev, n,
                       threshold, index_out_dev);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_exp_f32(gmmvi_ctx* ctx, float* dst_dev, const float* src_dev, size_t count) {
    if (count == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && src_dev);
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(exp_f32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dst_dev, src_dev, count);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

__global__ void logaddexp_f32_kernel(float* __restrict__ dst, const float* __restrict__ a, float ca,
                                     const float* __restrict__ b, float cb, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = a[i] + ca, y = b[i] + cb;
        const float m = fmaxf(x, y);
        dst[i] = m + logf(expf(x - m) + expf(y - m));
    }
}

int gmmvi_logaddexp_f32(gmmvi_ctx* ctx, float* dst_dev, const float* a_dev, float ca, const float* b_dev, float cb,
                        size_t count) {
    if (count == 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && a_dev && b_dev);
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(logaddexp_f32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dst_dev, a_dev, ca, b_dev, cb, count);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

// out[g * out_stride + col0 + n] = log sum_{j in [off[g], off[g+1])} exp(logw[j] + ld[j * N + n])
__global__ void segment_lse_kernel(const int32_t* __restrict__ off, const float* __restrict__ logw, const float* __restrict__ ld,
                                   int N, float* __restrict__ out, size_t out_stride, int col0) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (n >= N) return;
    // eight independent loads in flight per round (a load per dependent step made this kernel 36 us for 3 000 components x 300
    // samples), one rescale of the running sum per round
    float m = -3.0e38f, sum = 0.f;
    const int j1 = off[g + 1];
    for (int j = off[g]; j < j1; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = j + u < j1 ? logw[j + u] + ld[(size_t)(j + u) * N + n] : -3.0e38f;
        float mr = v[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) mr = fmaxf(mr, v[u]);
        const float mn = fmaxf(m, mr);
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) part += expf(v[u] - mn);
        sum = sum * expf(m - mn) + part;
        m = mn;
    }
    out[(size_t)g * out_stride + col0 + n] = m + logf(sum);
}

int gmmvi_segment_lse_f32(gmmvi_ctx* ctx, int G, const int32_t* offsets_dev, const float* logw_dev, const float* ld_dev, int N,
                          float* out_dev, size_t out_stride, int col0) {
    if (G <= 0 || N <= 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, offsets_dev && logw_dev && ld_dev && out_dev && col0 >= 0 && out_stride >= (size_t)col0 + N);
    hipLaunchKernelGGL(segment_lse_kernel, dim3((N + 255) / 256, G), dim3(256), 0, ctx->stream, offsets_dev, logw_dev, ld_dev, N,
                       out_dev, out_stride, col0);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

__global__ void copy_2d_f32_kernel(float* __restrict__ dst, size_t dst_stride, const float* __restrict__ src, size_t src_stride,
                                   int cols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c < cols) dst[(size_t)r * dst_stride + c] = src[(size_t)r * src_stride + c];
}

int gmmvi_copy_2d_f32(gmmvi_ctx* ctx, float* dst_dev, size_t dst_stride, const float* src_dev, size_t src_stride, int rows,
                      int cols) {
    if (rows <= 0 || cols <= 0) return GMMVI_OK;
    GMMVI_ARG_CHECK(ctx, dst_dev && src_dev && dst_stride >= (size_t)cols && src_stride >= (size_t)cols);
    hipLaunchKernelGGL(copy_2d_f32_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, ctx->stream, dst_dev, dst_stride, src_dev,
                       src_stride, cols);
    GMMVI_LAUNCH_CHECK(ctx);
    return GMMVI_OK;
}

int gmmvi_event_create(gmmvi_ctx* ctx, void** out_event) {
    GMMVI_ARG_CHECK(ctx, out_event != nullptr);
    hipEvent_t ev;
    GMMVI_HIP_CHECK(ctx, hipEventCreate(&ev));
    *out_event = (void*)ev;
    return GMMVI_OK;
}

int gmmvi_event_destroy(gmmvi_ctx* ctx, void* event) {
    if (event) GMMVI_HIP_CHECK(ctx, hipEventDestroy((hipEvent_t)event));
    return GMMVI_OK;
}

int gmmvi_event_record(gmmvi_ctx* ctx, void* event) {
    GMMVI_HIP_CHECK(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return GMMVI_OK;
}

int gmmvi_event_synchronize(gmmvi_ctx* ctx, void* event) {
    GMMVI_ARG_CHECK(ctx, event != nullptr);
    GMMVI_HIP_CHECK(ctx, hipEventSynchronize((hipEvent_t)event));
    return GMMVI_OK;
}

int gmmvi_event_elapsed_ms(gmmvi_ctx* ctx, void* start, void* stop, float* out_ms) {
    GMMVI_ARG_CHECK(ctx, out_ms != nullptr);
    GMMVI_HIP_CHECK(ctx, hipEventSynchronize((hipEvent_t)stop));
    GMMVI_HIP_CHECK(ctx, hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return GMMVI_OK;
}

int gmmvi_profile_enable(gmmvi_ctx* ctx, int on) {
    ctx->prof = on != 0;
    return GMMVI_OK;
}

int gmmvi_profile_report(gmmvi_ctx* ctx, char* buf, size_t buf_size) {
    GMMVI_ARG_CHECK(ctx, buf && buf_size > 0);
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<std::string> names;
    std::vector<double> total, units;
    std::vector<long> count;
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) ms = 0.f;
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
        size_t i = 0;
        for (; i < names.size(); ++i) if (names[i] == r.name) break;
        if (i == names.size()) { names.push_back(r.name); total.push_back(0.0); units.push_back(0.0); count.push_back(0); }
        total[i] += ms;
        units[i] += r.units;
        count[i] += 1;
    }
    ctx->prof_recs.clear();
    std::string out;
    for (size_t i = 0; i < names.size(); ++i)
        out += names[i] + " " + std::to_string(count[i]) + " " + std::to_string(total[i]) + " " + std::to_string(units[i]) + "\n";
    if (out.size() + 1 > buf_size) return gmmvi_fail(ctx, GMMVI_ERR_ARG, "gmmvi_profile_report: buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return GMMVI_OK;
}

size_t gmmvi_packed_stride(int D) {
    if (gmmvi_is_blocked_dim(D)) return gmmvi_blocked_stride(D);
    int dp = gmmvi_padded_dim(D);
    return dp < 0 ? 0 : gmmvi_packed_stride_dp(dp);
}

}  // extern "C"
