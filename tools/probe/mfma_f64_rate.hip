// Probe: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (what bounds the Gram contraction of the MORE estimator, csrc/more.hip).
// One wave per SIMD (256 threads per workgroup, one workgroup per CU), N back-to-back MFMAs on 1 / 4 / 8 independent accumulators;
// wall_clock64 ticks at 100 MHz.  Build: hipcc -O2 --offload-arch=gfx950 tools/probe/mfma_f64_rate.hip -o tools/probe/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(int reps, double* out) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(double* out) {
    const int reps = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, 2000, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, reps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)reps * 8;               // per wave
    const double flops = mfmas * 2048.0 * 1024.0;          // 1024 waves on the chip
    printf("%d independent accumulators: %.1f ns per MFMA per SIMD, %.1f TFLOP/s chip-wide\n", NACC, ms * 1e6 / mfmas, flops / (ms * 1e-3) / 1e12);
}
int main() {
    double* out; hipMalloc(&out, 256 * 256 * 8);
    run<1>(out); run<4>(out); run<8>(out);
    return 0;
}
