// fp32 products on the bf16 matrix instructions of gfx950 ("bf16x3"): the fp32 matrix instructions of this chip
// (v_mfma_f32_16x16x4_f32, 32 cycles for 2 kFLOP) run at 1/16 of the bf16 rate (v_mfma_f32_16x16x32_bf16, 16 cycles
// for 16 kFLOP), so an fp32 convolution is bound by them at 0.4-0.5 of a 157 TF peak while the activations' bytes
// would allow 3-4 x more.  Every fp32 value is split into three bf16 values,
//     a = a1 + a2 + a3,   a1 = bf16(a),  a2 = bf16(a - a1),  a3 = bf16(a - a1 - a2)   (round to nearest even).
// The split is EXACT: the two subtractions are exact in fp32, |a2| <= 2^-8 |a|, and what is left after a2 is a multiple
// of a's last bit below 2^-16 |a| -- at most 8 significant bits, a bf16 value (|a3| <= 2^-16 |a|; 2^-17 measured).
// A product a * b is the sum of the six partial products
//     a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2)
// -- each exact in fp32 (8 x 8 significant bits), accumulated in fp32 by the matrix instruction.  What is dropped is
// a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b| in the worst case; over 4 M random pairs (tests/test_host_cpu.py restates the
// arithmetic with torch's bfloat16) the six-term sum is off by 2^-24.2 at most and 2^-28.1 on average -- below what ONE
// fp32 rounding costs (2^-24 at most, 2^-25.5 on average), and every accumulation step of either arithmetic commits
// such a rounding of the running sum.  Six bf16 instructions of 16 cycles replace eight fp32 instructions of 32 cycles for
// the same 32-deep slice of the reduction (96 vs 256 cycles).  bf16 has fp32's exponent range, so the split needs no
// scaling; parts below the smallest normal fp32 flush to zero (absolute error < 2^-126 there).
// RESTRICTION (finite operands below the bf16 overflow threshold): for |a| >= ~3.39e38 (and +-Inf) bf16(a) rounds to
// Inf, a - a1 is Inf - Inf = NaN, and the split kernels emit NaN where the fp32 matrix instructions propagate Inf or a
// finite product.  The opt-in arithmetic is for finite activations / weights (qpwc.h says so by the *_x3_fwd
// prototypes); the default fp32 path has no such restriction.
// tests/test_gpu_x3.py compares both arithmetic forms with float64 on the same inputs.
#pragma once
#include <hip/hip_runtime.h>

namespace qpwc {

typedef __bf16 bf16x8e __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2e __attribute__((ext_vector_type(2)));
typedef float f32x2e __attribute__((ext_vector_type(2)));

// two fp32 values -> one dword of two bf16 (a in the low half), round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{a, b}, bf16x2e));
}

// (a, b) -> the three bf16 pairs of the split above
__device__ __forceinline__ void split2_bf16x3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(ra, rb);
    p3 = pk_bf16(ra - __uint_as_float(p2 << 16), rb - __uint_as_float(p2 & 0xffff0000u));
}

// eight consecutive fp32 values -> three 16-byte operands (8 bf16 each)
__device__ __forceinline__ void split8_bf16x3(const float4& lo, const float4& hi, uint4& p1, uint4& p2, uint4& p3) {
    split2_bf16x3(lo.x, lo.y, p1.x, p2.x, p3.x);
    split2_bf16x3(lo.z, lo.w, p1.y, p2.y, p3.y);
    split2_bf16x3(hi.x, hi.y, p1.z, p2.z, p3.z);
    split2_bf16x3(hi.z, hi.w, p1.w, p2.w, p3.w);
}

typedef float f32x4s __attribute__((ext_vector_type(4)));

#ifndef QPWC_X3_TERMS
#define QPWC_X3_TERMS 6
#endif

// acc += A * B for one 32-deep slice: A = (a1, a2, a3), B = (b1, b2, b3) as split above; smallest terms first
__device__ __forceinline__ f32x4s mfma_bf16x3(const uint4& a1, const uint4& a2, const uint4& a3, const uint4& b1,
                                              const uint4& b2, const uint4& b3, f32x4s acc) {
#define QPWC_X3_MFMA(A, B) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8e, A), __builtin_bit_cast(bf16x8e, B), acc, 0, 0, 0)
#if QPWC_X3_TERMS == 9   // all nine partial products: every product exact (A/B build; 144 instead of 96 cycles per slice)
    QPWC_X3_MFMA(a3, b3);
    QPWC_X3_MFMA(a3, b2);
    QPWC_X3_MFMA(a2, b3);
#endif
    QPWC_X3_MFMA(a2, b2);
    QPWC_X3_MFMA(a3, b1);
    QPWC_X3_MFMA(a1, b3);
    QPWC_X3_MFMA(a2, b1);
    QPWC_X3_MFMA(a1, b2);
    QPWC_X3_MFMA(a1, b1);
#undef QPWC_X3_MFMA
    return acc;
}

}  // namespace qpwc
