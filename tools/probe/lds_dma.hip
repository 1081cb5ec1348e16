// Probe: global_load_lds_dwordx4 on gfx950 -- where does lane l's 16 bytes land?  Expect LDS[M0 + 16 l].
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_dma.hip -o gpurun_out/lds_dma && gpurun_out/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* src, float* out) {
    __shared__ __align__(16) unsigned char lds[8192];
    const unsigned lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((float*)lds)[i] = -1.f;
    __syncthreads();
    const unsigned voff = lane * 16;
    const unsigned ldsaddr = (unsigned)(size_t)(lds + 1024);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsaddr), "v"(voff), "s"(src) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = ((float*)lds)[i];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 8192);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const unsigned char*)d, o);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2048; ++i) {
        const float want = (i >= 256 && i < 512) ? (float)(i - 256) : -1.f;
        if (r[i] != want) { if (bad < 8) printf("lds float %d = %g, want %g\n", i, r[i], want); ++bad; }
    }
    printf("%s (%d mismatches)\n", bad ? "UNEXPECTED LAYOUT" : "lane l -> LDS[M0 + 16 l]: ok", bad);
    return bad != 0;
}
