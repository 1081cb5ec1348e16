// f32 values as sums of three bf16 values (blocked.hip: the split-operand route of the D > 50 contractions; stein.hip: the
// moment contraction).  x = x1 + x2 + x3 up to 2^-27 |x| (round-to-nearest splits: |x2| <= 2^-9 |x|, |x3| <= 2^-18 |x|); a product
// x y is the sum of the six partial products x_i y_j with i + j <= 4, each exact in f32, accumulated by the bf16 matrix-core
// instructions -- the error of an f32 contraction (tests/test_hip_blocked.py measures both routes against fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
// (a, b) -> three packed bf16 pairs, a in the low half
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(ra, rb);
    p3 = cvt_pk_bf16(ra - __uint_as_float(p2 << 16), rb - __uint_as_float(p2 & 0xffff0000u));
}
