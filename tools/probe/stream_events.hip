// Probe: cost of cross-stream event dependencies on gfx950.  A chain of ~5 us kernels on ONE stream against the same chain
// alternating between two streams with hipEventRecord / hipStreamWaitEvent hand-overs, and a fork-join pattern (two
// independent 5 us kernels on two streams between single-stream kernels).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/stream_events.hip -o tools/probe/stream_events
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = 1;
}

int main() {
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e[2]; hipEventCreateWithFlags(&e[0], hipEventDisableTiming); hipEventCreateWithFlags(&e[1], hipEventDisableTiming);
    int* sink; hipMalloc(&sink, 4096);
    const long long T = 500;      // 5 us in 100 MHz ticks
    const int n = 2000;
    auto run = [&](int mode) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < n; ++i) {
            if (mode == 0) {                              // one stream: A, B, C
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, sink);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, sink);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, sink);
            } else if (mode == 1) {                       // A on s1; then B on s2 || C on s1; join
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, sink);
                hipEventRecord(e[0], s1);
                hipStreamWaitEvent(s2, e[0], 0);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, T, sink);
                hipEventRecord(e[1], s2);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, sink);
                hipStreamWaitEvent(s1, e[1], 0);
            }
        }
        hipDeviceSynchronize();
        auto t1 = std::chrono::high_resolution_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    };
    run(0); run(1);
    printf("three 5 us kernels in a row on one stream: %.2f us per round\n", run(0));
    printf("A; then B on a second stream beside C; join: %.2f us per round (ideal 10 + hand-over costs)\n", run(1));
    return 0;
}
