// The register DFTs of ksa_fft.hpp on PAIRS of independent transforms (gfx950 packed fp32).
//
// A plain v_fma_f32 occupies a SIMD for the same issue slot whether 64 or 128 products are formed: v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 do two fp32 operations per lane (the chip's 157 TFLOP/s fp32 figure counts them).
// Measured (tools/valu_rate.hip, 2-3 waves per SIMD): 2.0-2.1 ns per packed wave-instruction against 1.38 ns per
// plain one, i.e. 25 % less VALU time for the same arithmetic -- if the two operations of an instruction need no
// shuffling.  Packing the re/im parts of one complex value does (round 1: op_sel / neg forms, -9 %); packing the
// SAME element of TWO transforms does not: every butterfly operation is elementwise across the pair, twiddles are
// shared (hipcc broadcasts a scalar operand into both halves with op_sel_hi, no extra instruction), and results
// are bit-identical to the unpaired code (same operations in the same order per transform).
//
// cx2 = one complex element of two transforms: re = (re_A, re_B), im = (im_A, im_B) -- four consecutive VGPRs,
// moved through LDS as one 16-byte element.  Function names and operation order mirror ksa_fft.hpp one to one.
#pragma once
#include <hip/hip_runtime.h>

#include "ksa_fft.hpp"

namespace ksa {

typedef float v2f __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) cx2 {
  v2f x, y;   // re pair, im pair
};

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float s) { return v2f{s, s}; }

__device__ __forceinline__ cx2 cadd(cx2 a, cx2 b) { return cx2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cx2 csub(cx2 a, cx2 b) { return cx2{a.x - b.x, a.y - b.y}; }
// a * w, w shared by the pair
__device__ __forceinline__ cx2 cmul(cx2 a, float2 w) {
  return cx2{fma2(-a.y, splat(w.y), a.x * splat(w.x)), fma2(a.y, splat(w.x), a.x * splat(w.y))};
}
// a + u*b
__device__ __forceinline__ cx2 cfma(cx2 a, float2 u, cx2 b) {
  return cx2{fma2(-splat(u.y), b.y, fma2(splat(u.x), b.x, a.x)), fma2(splat(u.y), b.x, fma2(splat(u.x), b.y, a.y))};
}
__device__ __forceinline__ cx2 twice_minus(cx2 a, cx2 s) {
  return cx2{fma2(splat(2.0f), a.x, -s.x), fma2(splat(2.0f), a.y, -s.y)};
}

template <int M>
__device__ __forceinline__ cx2 mul_w16(cx2 v) {
  if constexpr (M == 0) return v;
  else if constexpr (M == 1) return cmul(v, make_float2(kCosPi8, -kSinPi8));
  else if constexpr (M == 2) return cx2{(v.x + v.y) * splat(kSqrtHalf), (v.y - v.x) * splat(kSqrtHalf)};
  else if constexpr (M == 3) return cmul(v, make_float2(kSinPi8, -kCosPi8));
  else if constexpr (M == 4) return cx2{v.y, -v.x};
  else if constexpr (M == 6) return cx2{(v.y - v.x) * splat(kSqrtHalf), -(v.x + v.y) * splat(kSqrtHalf)};
  else if constexpr (M == 9) return cmul(v, make_float2(-kCosPi8, kSinPi8));
  else { static_assert(M < 0, "unsupported W16 exponent"); return v; }
}

template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft2(cx2 (&v)[SZ]) {
  cx2 a = v[BASE], b = v[BASE + STRIDE];
  v[BASE] = cadd(a, b);
  v[BASE + STRIDE] = csub(a, b);
}

template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft4(cx2 (&v)[SZ]) {
  cx2 a0 = v[BASE], a1 = v[BASE + STRIDE], a2 = v[BASE + 2 * STRIDE], a3 = v[BASE + 3 * STRIDE];
  cx2 s0 = cadd(a0, a2), d0 = csub(a0, a2);
  cx2 s1 = cadd(a1, a3), d1 = csub(a1, a3);
  v[BASE] = cadd(s0, s1);
  v[BASE + 2 * STRIDE] = csub(s0, s1);
  v[BASE + STRIDE] = cx2{d0.x + d1.y, d0.y - d1.x};      // d0 - j*d1
  v[BASE + 3 * STRIDE] = cx2{d0.x - d1.y, d0.y + d1.x};  // d0 + j*d1
}

template <int BASE, int SZ>
__device__ __forceinline__ void dft8(cx2 (&v)[SZ]) {
  dft4<BASE + 0, 2>(v);
  dft4<BASE + 1, 2>(v);
  v[BASE + 3] = mul_w16<2>(v[BASE + 3]);
  v[BASE + 5] = mul_w16<4>(v[BASE + 5]);
  v[BASE + 7] = mul_w16<6>(v[BASE + 7]);
  dft2<BASE + 0, 1>(v);
  dft2<BASE + 2, 1>(v);
  dft2<BASE + 4, 1>(v);
  dft2<BASE + 6, 1>(v);
}

__device__ __forceinline__ void dft16(cx2 (&v)[16]) {
  dft4<0, 4>(v);
  dft4<1, 4>(v);
  dft4<2, 4>(v);
  dft4<3, 4>(v);
  v[5] = mul_w16<1>(v[5]);
  v[6] = mul_w16<2>(v[6]);
  v[7] = mul_w16<3>(v[7]);
  v[9] = mul_w16<2>(v[9]);
  v[10] = mul_w16<4>(v[10]);
  v[11] = mul_w16<6>(v[11]);
  v[13] = mul_w16<3>(v[13]);
  v[14] = mul_w16<6>(v[14]);
  v[15] = mul_w16<9>(v[15]);
  dft4<0, 1>(v);
  dft4<4, 1>(v);
  dft4<8, 1>(v);
  dft4<12, 1>(v);
}

// radix-4 of (a0, u1*a1, u2*a2, u3*a3), 24 packed operations for two transforms
template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft4_tw(cx2 (&v)[SZ], float2 u1, float2 u2, float2 u3) {
  const cx2 a0 = v[BASE], a1 = v[BASE + STRIDE], a2 = v[BASE + 2 * STRIDE], a3 = v[BASE + 3 * STRIDE];
  const cx2 s02 = cfma(a0, u2, a2);
  const cx2 d02 = twice_minus(a0, s02);
  const cx2 t1 = cmul(a1, u1);
  const cx2 s13 = cfma(t1, u3, a3);
  const cx2 d13 = twice_minus(t1, s13);
  v[BASE] = cadd(s02, s13);
  v[BASE + 2 * STRIDE] = csub(s02, s13);
  v[BASE + STRIDE] = cx2{d02.x + d13.y, d02.y - d13.x};
  v[BASE + 3 * STRIDE] = cx2{d02.x - d13.y, d02.y + d13.x};
}

__device__ __forceinline__ void dft16_fused(cx2 (&v)[16], const float2 (&tw)[15]) {
  dft4_tw<0, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<1, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<2, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<3, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<0, 1>(v, tw[3], tw[4], tw[5]);
  dft4_tw<4, 1>(v, tw[6], tw[7], tw[8]);
  dft4_tw<8, 1>(v, tw[9], tw[10], tw[11]);
  dft4_tw<12, 1>(v, tw[12], tw[13], tw[14]);
}

// radix-16 of (w^t * v[t]) from six twiddles w^1, w^2, w^3, w^4, w^8, w^12 (12 registers instead of 30): the
// W16^(n2*k1) factors of the second level are applied as constant rotations
__device__ __forceinline__ void dft16_tw6(cx2 (&v)[16], float2 w1, float2 w2, float2 w3, float2 w4, float2 w8, float2 w12) {
  dft4_tw<0, 4>(v, w4, w8, w12);
  dft4_tw<1, 4>(v, w4, w8, w12);
  dft4_tw<2, 4>(v, w4, w8, w12);
  dft4_tw<3, 4>(v, w4, w8, w12);
  dft4_tw<0, 1>(v, w1, w2, w3);
  v[5] = mul_w16<1>(v[5]);
  v[6] = mul_w16<2>(v[6]);
  v[7] = mul_w16<3>(v[7]);
  dft4_tw<4, 1>(v, w1, w2, w3);
  v[9] = mul_w16<2>(v[9]);
  v[10] = mul_w16<4>(v[10]);
  v[11] = mul_w16<6>(v[11]);
  dft4_tw<8, 1>(v, w1, w2, w3);
  v[13] = mul_w16<3>(v[13]);
  v[14] = mul_w16<6>(v[14]);
  v[15] = mul_w16<9>(v[15]);
  dft4_tw<12, 1>(v, w1, w2, w3);
}

template <int R0>
__device__ __forceinline__ void dft_first(cx2 (&v)[16]) {
  if constexpr (R0 == 16) {
    dft16(v);
  } else if constexpr (R0 == 8) {
    dft8<0>(v);
    dft8<8>(v);
  } else if constexpr (R0 == 4) {
    dft4<0, 1>(v);
    dft4<4, 1>(v);
    dft4<8, 1>(v);
    dft4<12, 1>(v);
  } else {
    dft2<0, 1>(v); dft2<2, 1>(v); dft2<4, 1>(v); dft2<6, 1>(v);
    dft2<8, 1>(v); dft2<10, 1>(v); dft2<12, 1>(v); dft2<14, 1>(v);
  }
}

}  // namespace ksa
