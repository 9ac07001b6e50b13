// spectrum_pair_kernel<N>: spectrum_kernel (ksa_kernels.hpp) with every workgroup walking TWO frames in lockstep, the
// two transforms packed element by element into the halves of v_pk_*_f32 instructions (ksa_fft_pair.hpp).
//
// Same rows of SURVEY.md section 8 (A0, A4-A9, A12; replaces numpy.fft.fft + the fold at python/kspecanal.py:385-396)
// and the same arithmetic per transform, except that the last pass builds its twiddles from 6 table entries
// instead of 15 (18 fewer VGPRs: no spills at 255).  Differences in shape: LDS holds the pair as 16-byte elements (re_A, re_B, im_A, im_B), moved
// with ds_write_b128 / ds_read_b128; the window taps live in 16 VGPRs (shared by both frames); two workgroups of
// Plan<N>::T threads per CU (71.5 KB of LDS each at N = 4096) = four transforms in flight per CU with 256 VGPRs per
// wave.  Used for large batches where it measures faster (N = 1024: +14..+30 %; not at 2048 / 4096);
// small batches keep spectrum_kernel's window-split mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "ksa_fft_pair.hpp"
#include "ksa_kernels.hpp"

namespace ksa {

template <int N>
struct PlanPair {
  using P = Plan<N>;
  static_assert(P::S == 1 && P::T <= 256 && P::M >= 2, "pair kernel: one transform per workgroup, N = 1024 .. 4096");
  // LDS layout of the exchanges for 16-byte pair elements (ds_write_b128: 8 groups of 8 lanes over 32 banks; ds_read_b128: 4
  // groups of 16 lanes over 64 banks): the transposed / natural scheme of Plan<N> with the row stride re-tuned for the wider
  // element (tools/lds_layout.py: 544 instead of 720 LDS cycles per wave and pair of transforms).  N = 1024 only -- the one
  // size the product instantiates; the experiments builds of 2048 / 4096 keep the one-pad-per-16 order.
  static constexpr bool XLAYOUT = KSA_XLAYOUT && N == 1024;
  static constexpr int ST1 = 260, K2 = 4, SH2 = 6;
  static constexpr int NPAD = XLAYOUT ? 1084 : P::N_PLUS_PAD16;
  static constexpr int LDS_BYTES = NPAD * 16 + P::MID * 8;
  static_assert(4 * N * 4 <= NPAD * 16, "output planes must fit the exchange buffer");
};

template <int N, int FMT, int RM, int CM>
__global__ __launch_bounds__(Plan<N>::T, 2) void spectrum_pair_kernel(const SpecParams p) {
  using P = Plan<N>;
  using PP = PlanPair<N>;
  constexpr int L = P::L, T = P::T, M = P::M, R0 = P::R0, B0 = P::B0, NPAD = PP::NPAD;
  constexpr int SB = FMT == FMT_C64 ? 8 : 2;
  extern __shared__ __attribute__((aligned(16))) cx2 lds2[];
  cx2* const my = lds2;
  float2* const tw_lds = reinterpret_cast<float2*>(lds2 + NPAD);
  const int l = threadIdx.x;

  float win[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) win[q] = p.window[l + L * q] * (FMT == FMT_U8 ? p.u8_inv_scale : 1.0f);
#ifndef KSAP_TW6
#define KSAP_TW6 1   // last pass from 6 twiddles (12 VGPRs) instead of the 15 folded ones (30 VGPRs)
#endif
  // last pass: k = l.  Rows of the folded table that are plain powers: c10 = w^1, c20 = w^2, c30 = w^3, then w^4, w^8, w^12
  constexpr int NTWL = KSAP_TW6 ? 6 : 15;
  float2 twl[NTWL];
  if constexpr (KSAP_TW6) {
    constexpr int rows[6] = {3, 4, 5, 0, 1, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) twl[e] = p.tw_last[rows[e] * P::P_LAST + l];
  } else {
#pragma unroll
    for (int e = 0; e < 15; ++e) twl[e] = p.tw_last[e * P::P_LAST + l];
  }
  if constexpr (P::MID > 0) {
    for (int i = l; i < P::MID; i += T) tw_lds[i] = p.tw_mid[i];
  }

  const int nm1 = p.nwin - 1;
  typedef typename std::conditional<FMT == FMT_C64, u32x2, unsigned short>::type raw_t;
  raw_t ra[16], rb[16];
  const int start0 = p.starts[0];
  // 16 samples l + L*q of one window of both frames (8 B/lane, wave-uniform descriptors, range-checked); with sample
  // reuse (RM > 0) only the RM new samples per thread after the first window
  auto issue_loads = [&](int fa, int fb, int k, int q0) {
    const char* ba = reinterpret_cast<const char*>(p.iq) + (long long)fa * p.frame_stride * SB;
    const char* bb = reinterpret_cast<const char*>(p.iq) + (long long)fb * p.frame_stride * SB;
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ba), 0, p.frame_len * SB, 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bb), 0, p.frame_len * SB, 0x00020000);
    const int start = RM > 0 ? start0 + k * (RM * L) : p.starts[k];
    const int voff = (start + l) * SB;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < q0) continue;
      if constexpr (FMT == FMT_C64) {
        ra[q] = __builtin_amdgcn_raw_buffer_load_b64(rsa, voff, L * q * SB, 0);
        rb[q] = __builtin_amdgcn_raw_buffer_load_b64(rsb, voff, L * q * SB, 0);
      } else {
        ra[q] = __builtin_amdgcn_raw_buffer_load_b16(rsa, voff, L * q * SB, 0);
        rb[q] = __builtin_amdgcn_raw_buffer_load_b16(rsb, voff, L * q * SB, 0);
      }
    }
  };

  const int npairs = (p.nframes + 1) / 2;
  for (int pf = blockIdx.x; pf < npairs; pf += gridDim.x) {
    const int fa = 2 * pf;
    const bool has_b = fa + 1 < p.nframes;
    const int fb = has_b ? fa + 1 : fa;          // an odd batch: the last workgroup transforms its frame twice
    v2f acc[16];
    const float init = p.cumu == CUMU_MIN ? __builtin_inff() : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = splat(init);

    for (int k = 0; k < p.nwin; ++k) {
      issue_loads(fa, fb, k, (RM > 0 && k > 0) ? 16 - RM : 0);
      cx2 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        cx2 e;
        if constexpr (FMT == FMT_C64) {
          const unsigned ar = ra[q].x, ai = ra[q].y, br = rb[q].x, bi = rb[q].y;
          e.x.x = __uint_as_float(ar) * win[q]; e.x.y = __uint_as_float(br) * win[q];
          e.y.x = __uint_as_float(ai) * win[q]; e.y.y = __uint_as_float(bi) * win[q];
        } else {
          const unsigned short xa = ra[q], xb = rb[q];
          e.x.x = ((float)(xa & 0xff) - p.u8_offset) * win[q]; e.x.y = ((float)(xb & 0xff) - p.u8_offset) * win[q];
          e.y.x = ((float)(xa >> 8) - p.u8_offset) * win[q];   e.y.y = ((float)(xb >> 8) - p.u8_offset) * win[q];
        }
        v[(q % B0) * R0 + (q / B0)] = e;
      }
      if constexpr (RM > 0) {
#pragma unroll
        for (int q = 0; q + RM < 16; ++q) { ra[q] = ra[q + RM]; rb[q] = rb[q + RM]; }
      }
      dft_first<R0>(v);
      __syncthreads();   // the previous window's (pair's) LDS reads are done
#pragma unroll
      for (int b = 0; b < B0; ++b) {
        const int i = l + b * L;
#pragma unroll
        for (int t = 0; t < R0; ++t) {
          if constexpr (PP::XLAYOUT) my[perm<R0>(t) * PP::ST1 + i] = v[b * R0 + t];      // exchange 1 transposed: [output][butterfly]
          else my[padi(i * R0 + perm<R0>(t))] = v[b * R0 + t];
        }
      }
      __syncthreads();
      int pp = R0, tw_off = 0;
#pragma unroll
      for (int s = 1; s < M; ++s) {
        if constexpr (!PP::XLAYOUT) {
#pragma unroll
          for (int t = 0; t < 16; ++t) v[t] = my[padi(l + L * t)];
        } else if (s == 1) {
          const cx2* const src = my + (l % R0) * PP::ST1 + l / R0;          // element l + L*t = output l % R0 of butterfly l / R0 + (L/R0)*t
#pragma unroll
          for (int t = 0; t < 16; ++t) v[t] = src[(L / R0) * t];
        } else {
          const cx2* const src = my + l;                                     // natural order, K2 pads per 2^SH2 = L elements
#pragma unroll
          for (int t = 0; t < 16; ++t) v[t] = src[(L + PP::K2) * t];
        }
        if (s < M - 1) {
          const float2* tw = tw_lds + tw_off + (l & (pp - 1));
          float2 tm[15];
#pragma unroll
          for (int e = 0; e < 15; ++e) tm[e] = tw[e * pp];
          dft16_fused(v, tm);
          __syncthreads();
          const int kk = l & (pp - 1);
          const int j = (l - kk) * 16 + kk;
          if constexpr (PP::XLAYOUT) {
            cx2* const dst = my + j + PP::K2 * (l >> ilog2(R0));
#pragma unroll
            for (int t = 0; t < 16; ++t) dst[perm<16>(t) * R0] = v[t];
          } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) my[padi(j + perm<16>(t) * pp)] = v[t];
          }
          __syncthreads();
          tw_off += 15 * pp;
          pp *= 16;
        } else {
          if constexpr (KSAP_TW6) dft16_tw6(v, twl[0], twl[1], twl[2], twl[3], twl[4], twl[5]);
          else dft16_fused(v, reinterpret_cast<const float2(&)[15]>(twl));
        }
      }
      // ---- |X| and the fold over the block's windows (K:391-395), both frames at once ---------------
      const int cm = CM == 0 ? p.cumu : CM;
      if (cm == CUMU_AVG) {
        const int e = k == 0 ? nm1 : nm1 - k + 1;       // closed form of the (a+x)/2 recursion
        const v2f w = splat(ldexpf(1.0f, -e));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const v2f m2 = fma2(v[i].x, v[i].x, v[i].y * v[i].y);
          acc[i] = fma2(w, v2f{__builtin_amdgcn_sqrtf(m2.x), __builtin_amdgcn_sqrtf(m2.y)}, acc[i]);
        }
      } else if (cm == CUMU_MAX) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const v2f m2 = fma2(v[i].x, v[i].x, v[i].y * v[i].y);
          acc[i] = v2f{nan_max_nonneg(acc[i].x, m2.x), nan_max_nonneg(acc[i].y, m2.y)};
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const v2f m2 = fma2(v[i].x, v[i].x, v[i].y * v[i].y);
          acc[i] = v2f{nan_min(acc[i].x, m2.x), nan_min(acc[i].y, m2.y)};
        }
      }
    }

    // ---- natural bin order through LDS (frame A at floats [0, N), frame B at [2N, 3N); each is followed by the
    //      scratch plane finish_frame may use), then the common output stage, one frame after the other
    float* const red = reinterpret_cast<float*>(lds2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      red[l + L * perm<16>(i)] = acc[i].x;
      red[2 * N + l + L * perm<16>(i)] = acc[i].y;
    }
    __syncthreads();
    finish_frame<N, T, 1, CM>(p, red, fa, l);
    if (has_b) finish_frame<N, T, 1, CM>(p, red + 2 * N, fb, l);
    // (the next pair's first exchange barrier orders these LDS reads before its writes)
  }
}

}  // namespace ksa
