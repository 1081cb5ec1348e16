// Probe: HIP virtual memory management on this stack -- reserve a large address range, map physical chunks into it on demand
// (what a grow-in-place sample database needs), time the calls.  Build: hipcc -O2 --offload-arch=gfx950 tools/probe/vmm.hip -o tools/probe/vmm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(float* p, size_t n, float v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void sum(const float* p, size_t n, float* out) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && p[i] != 1.f) atomicAdd(out, 1.f); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu bytes\n", gran);
    const size_t total = (size_t)16 << 30, chunk = (size_t)64 << 20;
    void* base = nullptr;
    double t0 = now(); CK(hipMemAddressReserve(&base, total, gran, nullptr, 0)); printf("reserve 16 GiB: %.1f us\n", (now() - t0) * 1e6);
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    float* out; CK(hipMalloc(&out, 4)); CK(hipMemset(out, 0, 4));
    for (int c = 0; c < 6; ++c) {
        hipMemGenericAllocationHandle_t h;
        double a = now(); CK(hipMemCreate(&h, chunk, &prop, 0)); double b = now();
        CK(hipMemMap((char*)base + c * chunk, chunk, 0, h, 0)); double d = now();
        CK(hipMemSetAccess((char*)base + c * chunk, chunk, &acc, 1)); double e = now();
        printf("chunk %d (64 MiB): create %.0f us, map %.0f us, set access %.0f us\n", c, (b - a) * 1e6, (d - b) * 1e6, (e - d) * 1e6);
        const size_t n = chunk / 4;
        hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, 0, (float*)((char*)base + c * chunk), n, 1.f);
        CK(hipDeviceSynchronize());
    }
    const size_t n = 6 * chunk / 4;
    hipLaunchKernelGGL(sum, dim3((n + 255) / 256), dim3(256), 0, 0, (float*)base, n, out);
    float bad = -1; CK(hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost));
    printf("contiguous range over 6 chunks reads back with %g wrong values\n", bad);
    // for comparison: a plain allocation + copy of the same size
    void *p1, *p2; double a = now(); CK(hipMalloc(&p1, 6 * chunk)); double b = now(); CK(hipMalloc(&p2, 12 * chunk)); double c2 = now();
    CK(hipMemcpy(p2, p1, 6 * chunk, hipMemcpyDeviceToDevice)); CK(hipDeviceSynchronize()); double d = now();
    printf("hipMalloc 384 MiB: %.0f us, hipMalloc 768 MiB: %.0f us, copy 384 MiB: %.0f us\n", (b - a) * 1e6, (c2 - b) * 1e6, (d - c2) * 1e6);
    return 0;
}
