// Probe: issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 with a SCALAR multiplier (the form the scalar-fed density
// sweep issues), explicit asm so that the compiler can neither pack nor unpack anything; 1 / 2 / 4 waves per SIMD.
// Also: the duration of an empty launch behind another one on the same stream (the fixed cost of a kernel boundary).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/pk_issue.hip -o tools/probe/pk_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 16 independent v_fma_f32 v, s, v, v            (16 lane-FMAs x 64 lanes per round)
// MODE 1: 16 independent v_pk_fma_f32 v[2], s[2], v[2], v[2]   (32 lane-FMAs x 64 lanes per round), natural halves
// MODE 2: as 1 with the low scalar broadcast to both halves (op_sel_hi:[0,1,1])
// MODE 3: 8 v_pk_fma_f32 + 8 v_fma_f32 interleaved
template <int MODE>
__global__ void issue(const float* __restrict__ in, float* out, uint64_t* cycles, int reps) {
    f32x2 a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { a[c].x = in[threadIdx.x + 64 * c]; a[c].y = in[threadIdx.x + 64 * c + 32]; }
    f32x2 m;
    m.x = __builtin_amdgcn_readfirstlane(__float_as_int(in[1024])) ? in[1024] : 0.5f;
    m.y = in[1025];
    float mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m.x)));
    float my = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m.y)));
    f32x2 b; b.x = in[4096 + threadIdx.x]; b.y = in[4160 + threadIdx.x];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[c].x) : "s"(mx), "v"(b.x));
            } else if (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[c]) : "s"(m), "v"(b));
            } else if (MODE == 2) {
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(a[c]) : "s"(m), "v"(b));
            } else {
                if (c & 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[c].x) : "s"(my), "v"(b.x));
                else asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[c]) : "s"(m), "v"(b));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += a[c].x + a[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void empty_kernel(float* out) {
    if (out == nullptr) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f;
}

template <int MODE>
void run(const float* in, float* out, uint64_t* cyc, const char* name, int fma_per_instr_x2) {
    const int reps = 40000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((issue<MODE>), dim3(256), dim3(threads), 0, 0, in, out, cyc, reps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((issue<MODE>), dim3(256), dim3(threads), 0, 0, in, out, cyc, reps);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const int nw = 256 * threads / 64;
        std::vector<uint64_t> c(nw);
        hipMemcpy(c.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += v;
        const double per_wave = s / nw / (reps * 16.0);
        const double lane_fma = 256.0 * threads * reps * 16.0 * fma_per_instr_x2 / 2.0;
        printf("%-52s waves/SIMD %d : %.2f cycles per instruction per wave, %.2f per SIMD; %.1f TFLOP/s by the wall clock (%.1f us)\n",
               name, waves_per_simd, per_wave, per_wave / waves_per_simd, 2.0 * lane_fma / (ms * 1e-3) / 1e12, ms * 1e3);
    }
}

int main() {
    float *in, *out; uint64_t* cyc;
    hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8 * 65536);
    std::vector<float> h(8192, 0.999f);
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<0>(in, out, cyc, "v_fma_f32 v, s, v, v", 2);
    run<1>(in, out, cyc, "v_pk_fma_f32 v[2], s[2], v[2], v[2]", 4);
    run<2>(in, out, cyc, "v_pk_fma_f32, low scalar broadcast (op_sel_hi 0)", 4);
    run<3>(in, out, cyc, "8 v_pk_fma_f32 + 8 v_fma_f32 interleaved", 3);

    // kernel boundary: n empty launches back to back on one stream, by grid size
    for (int grid : {1, 256, 512, 2048}) {
        for (int threads : {64, 512}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int n = 200;
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, 0, (float*)nullptr);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, 0, (float*)nullptr);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("empty kernel, grid %4d x %3d threads: %.2f us per launch (back to back on one stream)\n", grid, threads, ms * 1e3 / n);
        }
    }
    // the same through a graph of 8 kernel nodes
    {
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 0, st, (float*)nullptr);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("graph of 8 empty kernels (512 x 512): %.2f us per kernel node\n", ms * 1e3 / (50 * 8));
    }
    return 0;
}
