// Probe: register layout of v_mfma_f32_16x16x4_f32 on gfx950 (prints the (i, j) of every (lane, reg) of D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* out) {
    const int l = threadIdx.x;
    // assumed operand layout: A[i = l%16][k = l/16], B[k = l/16][j = l%16]
    float a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + i * 0.37f + (i % 5) * 1.1f; hB[i] = 2 + i * 0.11f + (i % 7) * 0.9f; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += (double)hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = (float)s; }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const float v = hD[l * 4 + r];
        const float a = ref[(4 * (l / 16) + r) * 16 + l % 16], b = ref[(4 * r + l / 16) * 16 + l % 16];
        if (fabs(v - a) > 1e-5f * fabs(a)) okA = 0;
        if (fabs(v - b) > 1e-5f * fabs(b)) okB = 0;
    }
    printf("layout i = 4*(l/16)+r: %s;  layout i = 4*r + l/16: %s\n", okA ? "YES" : "no", okB ? "YES" : "no");
    return 0;
}
