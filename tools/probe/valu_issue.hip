// Probe: issue cost of v_fma_f32 on gfx950 -- dependent chain vs independent accumulators, SGPR vs VGPR multiplier,
// 1 / 2 / 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/probe/valu_issue.hip -o tools/probe/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int CHAINS, bool SGPR>
__global__ void fma_chain(const float* __restrict__ in, float* out, uint64_t* cycles, int reps) {
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = in[threadIdx.x + 64 * c];
    // 16 multipliers: wave-uniform (scalar registers) or per-lane
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = SGPR ? in[1024 + i] : in[2048 + threadIdx.x + 64 * i];
    const float b = in[4096 + threadIdx.x];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) a[c] = fmaf(m[i], a[c], b);      // CHAINS independent accumulators
        asm volatile("" ::: "memory");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS, bool SGPR>
void run(const float* in, float* out, uint64_t* cyc, const char* name) {
    const int reps = 200;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU: 4 SIMDs x waves
        hipLaunchKernelGGL((fma_chain<CHAINS, SGPR>), dim3(256), dim3(threads), 0, 0, in, out, cyc, reps);
        hipDeviceSynchronize();
        const int nw = 256 * threads / 64;
        std::vector<uint64_t> c(nw);
        hipMemcpy(c.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += v;
        const double per_wave = s / nw / (reps * 16.0 * CHAINS);
        printf("%-44s waves/SIMD %d : %.2f cycles per v_fma per wave, %.2f per SIMD\n", name, waves_per_simd, per_wave,
               per_wave / waves_per_simd);
    }
}

int main() {
    float *in, *out; uint64_t* cyc;
    hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8 * 65536);
    std::vector<float> h(8192, 0.999f);
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<1, true>(in, out, cyc, "1 dependent chain, SGPR multiplier");
    run<1, false>(in, out, cyc, "1 dependent chain, VGPR multiplier");
    run<2, true>(in, out, cyc, "2 independent chains, SGPR multiplier");
    run<4, true>(in, out, cyc, "4 independent chains, SGPR multiplier");
    run<8, true>(in, out, cyc, "8 independent chains, SGPR multiplier");
    run<8, false>(in, out, cyc, "8 independent chains, VGPR multiplier");
    return 0;
}
