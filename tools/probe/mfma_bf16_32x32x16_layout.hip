// Probe: operand / result layout of v_mfma_f32_32x32x16_bf16 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __bf16 to_bf16(float v) { uint32_t b = __float_as_uint(v); uint16_t h = (uint16_t)(b >> 16); __bf16 r; __builtin_memcpy(&r, &h, 2); return r; }
__global__ void k(const float* A, const float* B, float* out) {   // A[32][16], B[16][32] (values exactly representable in bf16)
    const int l = threadIdx.x;
    bf16x8 a, b;
    // hypothesis: A operand lane l: row i = l % 32, k = 8 * (l / 32) + e;  B operand lane l: col j = l % 32, k = 8 * (l / 32) + e
    for (int e = 0; e < 8; ++e) { a[e] = to_bf16(A[(l % 32) * 16 + 8 * (l / 32) + e]); b[e] = to_bf16(B[(8 * (l / 32) + e) * 32 + l % 32]); }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
    float hA[512], hB[512], hD[1024], ref[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l / 32), j = l % 32;
        if (hD[l * 16 + r] != ref[i * 32 + j]) ok = 0;
    }
    printf("operands (row/col = l %% 32, k = 8 (l / 32) + e), result i = (r & 3) + 8 (r >> 2) + 4 (l / 32), j = l %% 32: %s\n", ok ? "YES" : "no");
    return 0;
}
