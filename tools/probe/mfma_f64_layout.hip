// Probe: register layout of v_mfma_f64_16x16x4_f64 on gfx950 (prints the (i, j) of every (lane, reg) of D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* out) {
    const int l = threadIdx.x;
    // assumed operand layout: A[i = l%16][k = l/16], B[k = l/16][j = l%16]
    double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + i * 0.37 + (i % 5) * 1.1; hB[i] = 2 + i * 0.11 + (i % 7) * 0.9; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; l += 7) for (int r = 0; r < 4; ++r) {
        int fi = -1, fj = -1;
        for (int e = 0; e < 256; ++e) if (fabs(ref[e] - hD[l * 4 + r]) < 1e-9) { fi = e / 16; fj = e % 16; }
        printf("lane %2d reg %d -> (i=%d, j=%d)   guess (4*(l/16)+r, l%%16) = (%d, %d)\n", l, r, fi, fj, 4 * (l / 16) + r, l % 16);
        if (fi != 4 * (l / 16) + r || fj != l % 16) ok = 0;
    }
    printf("guess %s\n", ok ? "CONFIRMED" : "WRONG");
    return 0;
}
