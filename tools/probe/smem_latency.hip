// Probe: latency of scalar loads (s_load_dword / s_load_dwordx16) on gfx950 -- dependent chains over a footprint that fits the
// scalar cache (1 KB), the L2 (1 MB) or neither (256 MB), with 1 .. 16 waves per CU running the same chase.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/smem_latency.hip -o tools/probe/smem_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void chase(const uint32_t* __restrict__ buf, int steps, uint64_t* cycles, uint32_t* sink, int wide) {
    const __attribute__((address_space(4))) uint32_t* p = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)buf;
    uint32_t idx = __builtin_amdgcn_readfirstlane(blockIdx.x * 64 % 256);
    uint32_t acc = 0;
    // warm
    for (int i = 0; i < 8; ++i) idx = p[idx];
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (!wide) {
        for (int i = 0; i < steps; ++i) idx = p[idx];
    } else {
        for (int i = 0; i < steps; ++i) {
            // a 64-byte load whose first dword is the next index
            typedef uint32_t u16v __attribute__((ext_vector_type(16)));
            const u16v v = *(const __attribute__((address_space(4))) u16v*)(p + idx);
            idx = v[0];
            acc += v[15];
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; sink[blockIdx.x] = idx + acc; }
}

int main() {
    const size_t max_words = (size_t)64 << 20;     // 256 MB
    uint32_t* d; uint64_t* cyc; uint32_t* sink;
    hipMalloc(&d, max_words * 4); hipMalloc(&cyc, 8 * 8192); hipMalloc(&sink, 4 * 8192);
    std::vector<uint32_t> h(max_words);
    const int steps = 2000;
    const size_t foot[] = {256, 4096, 262144, max_words};          // words: 1 KB, 16 KB, 1 MB, 256 MB
    for (size_t f : foot) {
        // stride-16-word (64-byte line) cyclic chain with a large co-prime step so that consecutive hops change lines
        const size_t lines = f / 16;
        size_t step = lines > 16 ? (lines / 2 + 1) | 1 : 1;
        for (size_t l = 0; l < lines; ++l) h[l * 16] = (uint32_t)(((l + step) % lines) * 16);
        hipMemcpy(d, h.data(), f * 4, hipMemcpyHostToDevice);
        for (int wide = 0; wide < 2; ++wide)
            for (int waves_per_cu : {1, 4, 16}) {
                const int blocks = 256 * waves_per_cu;
                hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, steps, cyc, sink, wide);
                hipDeviceSynchronize();
                std::vector<uint64_t> c(blocks);
                hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
                double s = 0; for (auto v : c) s += v;
                printf("footprint %8zu KB  %s  waves/CU %2d : %.1f memtime ticks per dependent load (100 MHz ticks: x10 ns)\n", f * 4 / 1024,
                       wide ? "s_load_dwordx16" : "s_load_dword   ", waves_per_cu, s / blocks / steps);
            }
    }
    return 0;
}
