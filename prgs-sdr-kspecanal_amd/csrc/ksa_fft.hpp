// Register-resident small DFTs and the per-size plan of the single-workgroup LDS FFT (gfx950).
//
// Replaces numpy.fft.fft at python/kspecanal.py:391 (forward DFT, e^{-2*pi*i*n*k/N}, unnormalised).
//
// Plan: N = R0 * 16^(M-1).  Every thread owns 16 complex points in VGPRs (L = N/16 threads per
// transform).  Pass 0 is radix R0 (16/R0 butterflies per thread, no twiddles, inputs straight
// from HBM/L2 with the window multiply fused); passes 1..M-1 are radix 16 with one butterfly per
// thread.  Between passes the 16 results are exchanged through LDS (Stockham autosort, so the
// last pass leaves natural-order bins l + L*t in the registers of thread l).
#pragma once
#include <hip/hip_runtime.h>

namespace ksa {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * w
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
  return make_float2(fmaf(-a.y, w.y, a.x * w.x), fmaf(a.y, w.x, a.x * w.y));
}

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kCosPi8 = 0.92387953251128675613f;
constexpr float kSinPi8 = 0.38268343236508977173f;

// v *= W16^M (forward: e^{-2*pi*i*M/16}); only the exponents a 4x4 split needs.
template <int M>
__device__ __forceinline__ float2 mul_w16(float2 v) {
  if constexpr (M == 0) return v;
  else if constexpr (M == 1) return cmul(v, make_float2(kCosPi8, -kSinPi8));
  else if constexpr (M == 2) return make_float2((v.x + v.y) * kSqrtHalf, (v.y - v.x) * kSqrtHalf);
  else if constexpr (M == 3) return cmul(v, make_float2(kSinPi8, -kCosPi8));
  else if constexpr (M == 4) return make_float2(v.y, -v.x);
  else if constexpr (M == 6) return make_float2((v.y - v.x) * kSqrtHalf, -(v.x + v.y) * kSqrtHalf);
  else if constexpr (M == 9) return cmul(v, make_float2(-kCosPi8, kSinPi8));
  else { static_assert(M < 0, "unsupported W16 exponent"); return v; }
}

// v *= W16^M for any M in 0..7 (general constant form where mul_w16 has no special case)
template <int M>
__device__ __forceinline__ float2 mul_w16_any(float2 v) {
  if constexpr (M == 0 || M == 1 || M == 2 || M == 3 || M == 4 || M == 6) return mul_w16<M>(v);
  else if constexpr (M == 5) return cmul(v, make_float2(-kSinPi8, -kCosPi8));
  else if constexpr (M == 7) return cmul(v, make_float2(-kCosPi8, -kSinPi8));
  else { static_assert(M < 0, "unsupported W16 exponent"); return v; }
}

template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft2(float2 (&v)[SZ]) {
  float2 a = v[BASE], b = v[BASE + STRIDE];
  v[BASE] = cadd(a, b);
  v[BASE + STRIDE] = csub(a, b);
}

// natural-order in-place radix-4 on v[BASE + STRIDE*{0,1,2,3}]
template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft4(float2 (&v)[SZ]) {
  float2 a0 = v[BASE], a1 = v[BASE + STRIDE], a2 = v[BASE + 2 * STRIDE], a3 = v[BASE + 3 * STRIDE];
  float2 s0 = cadd(a0, a2), d0 = csub(a0, a2);
  float2 s1 = cadd(a1, a3), d1 = csub(a1, a3);
  v[BASE] = cadd(s0, s1);
  v[BASE + 2 * STRIDE] = csub(s0, s1);
  v[BASE + STRIDE] = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - j*d1
  v[BASE + 3 * STRIDE] = make_float2(d0.x - d1.y, d0.y + d1.x);  // d0 + j*d1
}

// radix-8 on v[BASE..BASE+7]; position BASE+P ends up holding X[perm8(P)]
template <int BASE, int SZ>
__device__ __forceinline__ void dft8(float2 (&v)[SZ]) {
  dft4<BASE + 0, 2>(v);
  dft4<BASE + 1, 2>(v);
  v[BASE + 3] = mul_w16<2>(v[BASE + 3]);  // W8^1
  v[BASE + 5] = mul_w16<4>(v[BASE + 5]);  // W8^2
  v[BASE + 7] = mul_w16<6>(v[BASE + 7]);  // W8^3
  dft2<BASE + 0, 1>(v);
  dft2<BASE + 2, 1>(v);
  dft2<BASE + 4, 1>(v);
  dft2<BASE + 6, 1>(v);
}

// radix-16 on v[0..15] as 4x4; position P ends up holding X[perm16(P)]
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  dft4<0, 4>(v);
  dft4<1, 4>(v);
  dft4<2, 4>(v);
  dft4<3, 4>(v);
  // position 4*k1 + n2 holds y[n2][k1]; multiply by W16^(n2*k1)
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

// radix-16 of (w^t * v[t]) with the input twiddles folded into the 4x4 split: t = 4*n1 + n2, so
// w^t = w^(4*n1) * w^n2 -- three twiddles before the first radix-4 level, three after it.  Needs
// only w^1, w^2, w^3, w^4, w^8, w^12 (6 instead of 15 live twiddles) for 9 extra complex multiplies.
__device__ __forceinline__ void dft16_tw(float2 (&v)[16], float2 w1, float2 w2, float2 w3, float2 w4,
                                         float2 w8, float2 w12) {
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    v[4 + n2] = cmul(v[4 + n2], w4);
    v[8 + n2] = cmul(v[8 + n2], w8);
    v[12 + n2] = cmul(v[12 + n2], w12);
  }
  dft4<0, 4>(v);
  dft4<1, 4>(v);
  dft4<2, 4>(v);
  dft4<3, 4>(v);
  v[1] = cmul(v[1], w1);
  v[2] = cmul(v[2], w2);
  v[3] = cmul(v[3], w3);
  v[5] = cmul(mul_w16<1>(v[5]), w1);
  v[6] = cmul(mul_w16<2>(v[6]), w2);
  v[7] = cmul(mul_w16<3>(v[7]), w3);
  v[9] = cmul(mul_w16<2>(v[9]), w1);
  v[10] = cmul(mul_w16<4>(v[10]), w2);
  v[11] = cmul(mul_w16<6>(v[11]), w3);
  v[13] = cmul(mul_w16<3>(v[13]), w1);
  v[14] = cmul(mul_w16<6>(v[14]), w2);
  v[15] = cmul(mul_w16<9>(v[15]), w3);
  dft4<0, 1>(v);
  dft4<4, 1>(v);
  dft4<8, 1>(v);
  dft4<12, 1>(v);
}

// ---- FMA-fused forms -------------------------------------------------------------------------------
// a + u*b as four FMAs; the matching a - u*b is 2a - (a + u*b): two more instead of four.
__device__ __forceinline__ float2 cfma(float2 a, float2 u, float2 b) {
  return make_float2(fmaf(-u.y, b.y, fmaf(u.x, b.x, a.x)), fmaf(u.y, b.x, fmaf(u.x, b.y, a.y)));
}
__device__ __forceinline__ float2 twice_minus(float2 a, float2 s) {
  return make_float2(fmaf(2.0f, a.x, -s.x), fmaf(2.0f, a.y, -s.y));
}

// radix-4 of (a0, u1*a1, u2*a2, u3*a3) in 24 VALU ops (plain: 12 for the products + 16)
template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft4_tw(float2 (&v)[SZ], float2 u1, float2 u2, float2 u3) {
  const float2 a0 = v[BASE], a1 = v[BASE + STRIDE], a2 = v[BASE + 2 * STRIDE], a3 = v[BASE + 3 * STRIDE];
  const float2 s02 = cfma(a0, u2, a2);
  const float2 d02 = twice_minus(a0, s02);
  const float2 t1 = cmul(a1, u1);
  const float2 s13 = cfma(t1, u3, a3);
  const float2 d13 = twice_minus(t1, s13);
  v[BASE] = cadd(s02, s13);
  v[BASE + 2 * STRIDE] = csub(s02, s13);
  v[BASE + STRIDE] = make_float2(d02.x + d13.y, d02.y - d13.x);      // d02 - j*d13
  v[BASE + 3 * STRIDE] = make_float2(d02.x - d13.y, d02.y + d13.x);  // d02 + j*d13
}

// radix-16 of (w^t * v[t]), t = 4*n1 + n2, with all twiddles folded into the two radix-4 levels:
// level A uses w^4, w^8, w^12; level B uses c[n2][k1] = w^n2 * W16^(n2*k1).  tw[] holds them in the
// order {w4, w8, w12, c10, c20, c30, c11, c21, c31, c12, c22, c32, c13, c23, c33} (c<n2><k1>).
// 192 VALU ops instead of 256 for "multiply then transform".  Position P ends up holding X[perm16(P)].
__device__ __forceinline__ void dft16_fused(float2 (&v)[16], const float2 (&tw)[15]) {
  dft4_tw<0, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<1, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<2, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<3, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<0, 1>(v, tw[3], tw[4], tw[5]);
  dft4_tw<4, 1>(v, tw[6], tw[7], tw[8]);
  dft4_tw<8, 1>(v, tw[9], tw[10], tw[11]);
  dft4_tw<12, 1>(v, tw[12], tw[13], tw[14]);
}

// ---- 32 points per thread (ksa_kernels32.hpp) -----------------------------------------------------------
// The radix-16 forms above on a 16-element slice v[BASE .. BASE+15] of a larger register array.
template <int BASE, int SZ>
__device__ __forceinline__ void dft16_at(float2 (&v)[SZ]) {
  dft4<BASE + 0, 4>(v);
  dft4<BASE + 1, 4>(v);
  dft4<BASE + 2, 4>(v);
  dft4<BASE + 3, 4>(v);
  v[BASE + 5] = mul_w16<1>(v[BASE + 5]);
  v[BASE + 6] = mul_w16<2>(v[BASE + 6]);
  v[BASE + 7] = mul_w16<3>(v[BASE + 7]);
  v[BASE + 9] = mul_w16<2>(v[BASE + 9]);
  v[BASE + 10] = mul_w16<4>(v[BASE + 10]);
  v[BASE + 11] = mul_w16<6>(v[BASE + 11]);
  v[BASE + 13] = mul_w16<3>(v[BASE + 13]);
  v[BASE + 14] = mul_w16<6>(v[BASE + 14]);
  v[BASE + 15] = mul_w16<9>(v[BASE + 15]);
  dft4<BASE + 0, 1>(v);
  dft4<BASE + 4, 1>(v);
  dft4<BASE + 8, 1>(v);
  dft4<BASE + 12, 1>(v);
}

// radix-16 of (w^t * v[BASE+t]) with all 15 twiddles folded into the radix-4 levels (layout of dft16_fused)
template <int BASE, int SZ>
__device__ __forceinline__ void dft16_fused_at(float2 (&v)[SZ], const float2 (&tw)[15]) {
  dft4_tw<BASE + 0, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<BASE + 1, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<BASE + 2, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<BASE + 3, 4>(v, tw[0], tw[1], tw[2]);
  dft4_tw<BASE + 0, 1>(v, tw[3], tw[4], tw[5]);
  dft4_tw<BASE + 4, 1>(v, tw[6], tw[7], tw[8]);
  dft4_tw<BASE + 8, 1>(v, tw[9], tw[10], tw[11]);
  dft4_tw<BASE + 12, 1>(v, tw[12], tw[13], tw[14]);
}

// radix-16 of (w^t * v[BASE+t]) from six twiddles (layout of dft16_tw: w^1, w^2, w^3, w^4, w^8, w^12)
template <int BASE, int SZ>
__device__ __forceinline__ void dft16_tw_at(float2 (&v)[SZ], float2 w1, float2 w2, float2 w3, float2 w4, float2 w8, float2 w12) {
  dft4_tw<BASE + 0, 4>(v, w4, w8, w12);
  dft4_tw<BASE + 1, 4>(v, w4, w8, w12);
  dft4_tw<BASE + 2, 4>(v, w4, w8, w12);
  dft4_tw<BASE + 3, 4>(v, w4, w8, w12);
  dft4_tw<BASE + 0, 1>(v, w1, w2, w3);
  v[BASE + 5] = mul_w16<1>(v[BASE + 5]);
  v[BASE + 6] = mul_w16<2>(v[BASE + 6]);
  v[BASE + 7] = mul_w16<3>(v[BASE + 7]);
  dft4_tw<BASE + 4, 1>(v, w1, w2, w3);
  v[BASE + 9] = mul_w16<2>(v[BASE + 9]);
  v[BASE + 10] = mul_w16<4>(v[BASE + 10]);
  v[BASE + 11] = mul_w16<6>(v[BASE + 11]);
  dft4_tw<BASE + 8, 1>(v, w1, w2, w3);
  v[BASE + 13] = mul_w16<3>(v[BASE + 13]);
  v[BASE + 14] = mul_w16<6>(v[BASE + 14]);
  v[BASE + 15] = mul_w16<9>(v[BASE + 15]);
  dft4_tw<BASE + 12, 1>(v, w1, w2, w3);
}

// v *= W32^M, the exponents the first (radix-2) level of a 32-point butterfly needs (M = 1..15)
template <int M>
__device__ __forceinline__ float2 mul_w32(float2 v) {
  if constexpr (M % 2 == 0) return mul_w16_any<M / 2>(v);
  else {
    constexpr float c[8] = {0.98078528040323044913f, 0.83146961230254523708f, 0.55557023301960222474f, 0.19509032201612826785f,
                            -0.19509032201612826785f, -0.55557023301960222474f, -0.83146961230254523708f, -0.98078528040323044913f};
    constexpr float s[8] = {0.19509032201612826785f, 0.55557023301960222474f, 0.83146961230254523708f, 0.98078528040323044913f,
                            0.98078528040323044913f, 0.83146961230254523708f, 0.55557023301960222474f, 0.19509032201612826785f};
    return cmul(v, make_float2(c[M / 2], -s[M / 2]));   // e^{-2 pi i M/32}
  }
}

template <int T>
__device__ __forceinline__ void unroll_w32(float2 (&v)[32]) {
  if constexpr (T < 16) {
    const float2 a = v[T], b = v[T + 16];
    v[T] = cadd(a, b);
    if constexpr (T == 0) v[T + 16] = csub(a, b);
    else v[T + 16] = mul_w32<T>(csub(a, b));
    unroll_w32<T + 1>(v);
  }
}

// 32-point butterfly, natural order in, position P ends up holding X[perm32(P)] = X[2*perm16(P & 15) + (P >> 4)]:
// one radix-2 level (decimation in frequency) in front of two radix-16.
__device__ __forceinline__ void dft32(float2 (&v)[32]) {
  unroll_w32<0>(v);
  dft16_at<0>(v);
  dft16_at<16>(v);
}

// 32-point butterfly of (w^t * v[t]): the radix-2 level takes w^16, the two radix-16 take the folded twiddles of
// base w (even outputs) and of base w * W32 (odd outputs).
__device__ __forceinline__ void dft32_fused(float2 (&v)[32], float2 w16, const float2 (&twa)[15], const float2 (&twb)[15]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float2 s = cfma(v[t], w16, v[t + 16]);     // v[t] + w^16 * v[t+16]
    v[t + 16] = twice_minus(v[t], s);                // v[t] - w^16 * v[t+16]
    v[t] = s;
  }
  dft16_fused_at<0>(v, twa);
  dft16_fused_at<16>(v, twb);
}

__host__ __device__ constexpr int perm32(int p) { return 2 * (((p & 15) >> 2) | ((p & 3) << 2)) + (p >> 4); }

// Second radix-4 level of an untwiddled radix-16: group k1 takes its inputs times W16^(n2*k1).  (The FMA-fused form
// with the constants as twiddles -- 24 instead of 28 instructions for groups 1 and 3 -- measured 1 % SLOWER at config 2:
// longer dependent chains, constants through SGPRs)
__device__ __forceinline__ void dft16_level_b(float2 (&v)[16]) {
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

// ---- first pass with the window multiply folded in ---------------------------------------------------------
// The first radix-4 level of the first pass works on windowed samples r*w (K:391: tSamples*win): (r0 w0 +- r2 w2) as
// one product and two FMAs per component.  hipcc's FMA fusion reaches the same instruction count from the separate
// multiply (PMC: unchanged), but the explicit form schedules better: +1.6 % at config 2.
// v holds the RAW samples, w the matching taps.
template <int BASE, int STRIDE, int SZ>
__device__ __forceinline__ void dft4_win(float2 (&v)[SZ], const float (&w)[SZ]) {
  const float2 a0 = v[BASE], a1 = v[BASE + STRIDE], a2 = v[BASE + 2 * STRIDE], a3 = v[BASE + 3 * STRIDE];
  const float w0 = w[BASE], w1 = w[BASE + STRIDE], w2 = w[BASE + 2 * STRIDE], w3 = w[BASE + 3 * STRIDE];
  const float2 t0 = make_float2(a0.x * w0, a0.y * w0), t1 = make_float2(a1.x * w1, a1.y * w1);
  const float2 s0 = make_float2(fmaf(a2.x, w2, t0.x), fmaf(a2.y, w2, t0.y));
  const float2 d0 = make_float2(fmaf(-a2.x, w2, t0.x), fmaf(-a2.y, w2, t0.y));
  const float2 s1 = make_float2(fmaf(a3.x, w3, t1.x), fmaf(a3.y, w3, t1.y));
  const float2 d1 = make_float2(fmaf(-a3.x, w3, t1.x), fmaf(-a3.y, w3, t1.y));
  v[BASE] = cadd(s0, s1);
  v[BASE + 2 * STRIDE] = csub(s0, s1);
  v[BASE + STRIDE] = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - j*d1
  v[BASE + 3 * STRIDE] = make_float2(d0.x - d1.y, d0.y + d1.x);  // d0 + j*d1
}

// dft_first<R0> on raw samples and their taps (R0 = 16 or 4); same output positions as dft_first
template <int R0>
__device__ __forceinline__ void dft_first_win(float2 (&v)[16], const float (&w)[16]) {
  static_assert(R0 == 16 || R0 == 4, "windowed first pass: radix 16 or 4");
  if constexpr (R0 == 16) {
    dft4_win<0, 4>(v, w);
    dft4_win<1, 4>(v, w);
    dft4_win<2, 4>(v, w);
    dft4_win<3, 4>(v, w);
    dft16_level_b(v);
  } else {
    dft4_win<0, 1>(v, w);
    dft4_win<4, 1>(v, w);
    dft4_win<8, 1>(v, w);
    dft4_win<12, 1>(v, w);
  }
}

// output index held at register position P after the in-place transforms above
template <int R>
__host__ __device__ constexpr int perm(int p) {
  return R == 16 ? ((p >> 2) | ((p & 3) << 2)) : R == 8 ? ((p >> 1) | ((p & 1) << 2)) : p;
}

// all B = 16/R0 first-pass butterflies of one thread; registers are v[b*R0 + t]
template <int R0>
__device__ __forceinline__ void dft_first(float2 (&v)[16]) {
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

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

// Static plan for one transform size.
template <int N>
struct Plan {
  static_assert(N >= 16 && (N & (N - 1)) == 0, "N must be a power of two >= 16");
  static constexpr int LOG2N = ilog2(N);
  static constexpr int M = (LOG2N + 3) / 4;                 // passes
  static constexpr int R0 = 1 << (LOG2N - 4 * (M - 1));     // first-pass radix: 2, 4, 8 or 16
  static constexpr int B0 = 16 / R0;                        // first-pass butterflies per thread
  static constexpr int L = N / 16;                          // threads per transform
  static constexpr int T = L < 64 ? 64 : L;                 // workgroup size
  static constexpr int S = T / L;                           // transforms in flight per workgroup
  // ---- LDS layout of the exchanges (elements = float2; banking per MI355X_MICROARCH.md, LDS: ds_write_b64 is served in
  //      4 groups of 16 lanes over 32 banks, ds_read_b64 in 2 groups of 32 lanes over 64 banks) ------------------------
  // Exchange 1 (pass 0 -> pass 1) is stored TRANSPOSED: output c of first-pass butterfly i sits at c*ST1 + i, so the 16
  // lanes of a store group write 16 consecutive elements, and the row stride ST1 = NB + 32/R0 (NB + 1 where a transform
  // has only 16 first-pass butterflies) spreads the reads -- element l + L*t = (c = l % R0, i = l / R0 + (L/R0)*t) -- of a
  // 32-lane group over all 64 banks.  Exchange 2 (M == 3: pass 1 -> pass 2) keeps the natural Stockham order with K2 = R0
  // pad elements per 16*R0: its stores are runs of R0 elements 16*R0 apart, its reads 32 consecutive elements.  Both
  // sides of both exchanges are conflict-free (tools/lds_layout.py replays every instruction against the bank model); the
  // one-pad-per-16 layout of rounds 1-3 paid a 2-way conflict on every read (19 % of the LDS cycles at N = 4096) and 4-way
  // on the stores of N = 64 (49 %).  M > 3 (experiments builds only) keeps the padded natural order.
#ifndef KSA_XLAYOUT
#define KSA_XLAYOUT 1      // 0 (A/B builds): the one-pad-per-16 natural order everywhere
#endif
  static constexpr bool XLAYOUT = KSA_XLAYOUT && (M == 2 || M == 3);
  static constexpr int NB = N / R0;                         // first-pass butterflies per transform
  static constexpr int ST1 = NB == 16 ? 17 : NB + 32 / R0;
  static constexpr int SH2 = 4 + ilog2(R0);                 // exchange 2: K2 pads per 2^SH2 elements
  static constexpr int K2 = R0 & 15;
  static constexpr int X1_SIZE = (R0 - 1) * ST1 + NB;
  static constexpr int X2_SIZE = M == 3 ? N + K2 * ((N - 1) >> SH2) : 0;
  static constexpr int NPAD_X = ((X1_SIZE > X2_SIZE ? X1_SIZE : X2_SIZE) + 1) & ~1;
  static constexpr int N_PLUS_PAD16 = N + N / 16;           // the one-pad-per-16 natural order (padi)
  static constexpr int NPAD = XLAYOUT ? NPAD_X : N_PLUS_PAD16; // LDS complex elements per transform
  // middle-pass twiddle tables (passes 1..M-2): 15*p entries each, p = R0*16^(s-1)
  static constexpr int mid_entries() {
    int tot = 0, p = R0;
    for (int s = 1; s < M - 1; ++s) { tot += 15 * p; p *= 16; }
    return tot;
  }
  static constexpr int MID = mid_entries();
  static constexpr int P_LAST = N / 16;                     // p of the last pass (M >= 2)
  static constexpr int LDS_BYTES = (S * NPAD + MID) * 8;
};

__host__ __device__ constexpr int padi(int i) { return i + (i >> 4); }

}  // namespace ksa
