// spectrum32_kernel<N>: the single-workgroup spectrum kernel with 32 points per thread (N = 8192, 16384).
//
// Same rows of SURVEY.md section 8 as spectrum_kernel (A0, A4-A9, A12; replaces numpy.fft.fft + the fold at
// python/kspecanal.py:385-396), same Stockham autosort scheme, but N/32 threads per transform and passes of radix
// 32 / 32 / 16 (16384) or 32 / 16 / 16 (8192): three passes and TWO exchanges through LDS where the 16-point plan
// needs four passes and three exchanges (16384 = 4*16^3, 8192 = 2*16^3).  The workgroup is N/32 = 512 / 256
// threads, LDS holds one transform (135 / 68 KB), so a SIMD hosts 2 waves and every wave may use 256 VGPRs: 64
// data registers, 64 landing registers of the IQ loads, 32 tap and 32 fold registers, 24 last-pass twiddles (the
// 1024-thread plan is capped at 128 VGPRs).  Twiddles are the FMA-folded radix-4 forms of ksa_fft.hpp.
//
//   pass s: butterflies i = l + b*L (b < 32/R_s), inputs x[i + (N/R_s)*t] = register b*R_s + t, k = i mod p_s,
//           outputs to (i - k)*R_s + k + t'*p_s;  p_0 = 1, p_1 = 32, p_2 = 32*R_1.
//   After the last pass register (b, P) of thread l holds bin l + L*(b + 2*perm16(P)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "ksa_kernels.hpp"

namespace ksa {

template <int N>
struct Plan32 {
  static_assert(N == 8192 || N == 16384, "32-point plan: N = 8192 or 16384");
  static constexpr int L = N / 32, T = L;
  static constexpr int R1 = N == 16384 ? 32 : 16;           // middle pass radix; first is 32, last is 16
  static constexpr int B1 = 32 / R1;
  static constexpr int P2 = 32 * R1;                        // p of the last pass = N / 16
  static constexpr int NPAD = N + N / 32;                   // one pad element per 32: stride-32 writes of pass 0 are conflict-free
  static constexpr int MID_ROWS = R1 == 32 ? 31 : 15;       // folded twiddles per middle-pass butterfly
  static constexpr int MID = MID_ROWS * 32;                 // [rows][p = 32]
  static constexpr int LDS_BYTES = (NPAD + MID) * 8;
  static constexpr int WPS = 2;                             // waves per SIMD the allocator must leave room for (256 VGPRs)
};

__host__ __device__ constexpr int pad32(int i) { return i + (i >> 5); }

// Tuning switches.  Defaults = what measured best on MI355X (fmScan step, N=16384 kaiser, 71 windows; M FFT/s):
//   no prefetch, 6-twiddle last pass, taps from L2            31.4   <- default (16-point plan: 29.7)
//   no prefetch, 15 folded last-pass twiddles (60 VGPRs)      30.4   (17 spills)
//   prefetch after the middle pass, taps from L2              29.1   (the tap loads' L2 latency stays exposed)
//   taps in VGPRs (32), with or without prefetch              14-21  (38-57 spills: every spill costs scratch traffic
//                                                                     inside the window loop)
// The -D overrides are for tools/variants.sh only.
#ifndef KSA32_PREFETCH
#define KSA32_PREFETCH 0   // 1: issue the next window's IQ loads during this window's transform (after the middle pass); 2: and its taps
#endif
#ifndef KSA32_TW6
#define KSA32_TW6 1        // last pass from 6 twiddles per butterfly (24 VGPRs) instead of 15 folded ones (60 VGPRs)
#endif
#ifndef KSA32_TWL_DYN
#define KSA32_TWL_DYN 0    // last pass, second butterfly: this many of its 6 twiddles are re-read from the L2-resident table per
                           // window instead of living in VGPRs (2 VGPRs each) -- the explicit form of what the register allocator
                           // otherwise does with scratch memory (9-13 spilled registers reloaded per window)
#endif
#ifndef KSA32_WIN_REGS
#define KSA32_WIN_REGS 0   // 1: the thread's 32 window taps live in VGPRs (0: re-read from the L2-resident table per window)
#endif

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load
// (s_waitcnt vmcnt(0)), which would drain the prefetched IQ loads at the first exchange of every window.
__device__ __forceinline__ void lds_barrier() {
#ifdef KSA32_ABL_NOLDS     // timing-only ablation build (no exchange): wrong results by construction
  return;
#endif
#if KSA32_PREFETCH
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

#ifdef KSA32_ABL_NOLDS
#define K32_ST(dst, val) do {} while (0)
#define K32_LD(dst, src) do {} while (0)
#else
#define K32_ST(dst, val) (dst) = (val)
#define K32_LD(dst, src) (dst) = (src)
#endif

template <int N, int FMT, int CM>
__global__ __launch_bounds__(Plan32<N>::T, Plan32<N>::WPS) void spectrum32_kernel(const SpecParams p) {
  using P = Plan32<N>;
  constexpr int L = P::L, T = P::T, R1 = P::R1, B1 = P::B1;
  constexpr int SB = FMT == FMT_C64 ? 8 : 2;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2* const my = lds;
  float2* const tw_lds = lds + P::NPAD;
  const int l = threadIdx.x;

  // last pass (radix 16, two butterflies i = l + b*L, k = i): twiddles in VGPRs for the whole run
  constexpr int NTWL = KSA32_TW6 ? 6 : 15;
  float2 twl[2][NTWL];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if constexpr (KSA32_TW6) {
      // rows of the folded table that are plain powers: c10 = w^1, c20 = w^2, c30 = w^3 (k1 = 0), then w^4, w^8, w^12
      constexpr int rows[6] = {3, 4, 5, 0, 1, 2};
#pragma unroll
      for (int e = 0; e < 6; ++e)
        if (!(b == 1 && e >= 6 - KSA32_TWL_DYN)) twl[b][e] = p.tw_last[(b * 15 + rows[e]) * L + l];
    } else {
#pragma unroll
      for (int e = 0; e < 15; ++e) twl[b][e] = p.tw_last[(b * 15 + e) * L + l];
    }
  }
  for (int i = l; i < P::MID; i += T) tw_lds[i] = p.tw_mid[i];

  const int nm1 = p.nwin - 1;
  const int NP = p.parts > 1 ? p.parts : 1;
  const int total = p.nframes * NP;
  typedef typename std::conditional<FMT == FMT_C64, u32x2, unsigned short>::type raw_t;
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.window), 0, N * 4, 0x00020000);
  const auto twrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.tw_last), 0, 30 * L * 8, 0x00020000);
  // Taps of the thread's 32 samples l + L*q.  The host hands this kernel the window table re-ordered as [8][L][4]
  // (element (q4, l, j) = w[l + L*(4*q4 + j)]): eight 16-byte loads per window, each wave-instruction one contiguous KiB,
  // instead of thirty-two 4-byte ones (round 4: the kernel issues 40 instead of 64 vector-memory instructions per window).
  float win[32];
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef KSA32_TAPS_X4
#define KSA32_TAPS_X4 1   // 0 (A/B builds, with the natural-order table): thirty-two 4-byte tap loads
#endif
  auto load_taps = [&]() {
#if !KSA32_TAPS_X4
#pragma unroll
    for (int q = 0; q < 32; ++q)
      win[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, l * 4, L * q * 4, 0)) * (FMT == FMT_U8 ? p.u8_inv_scale : 1.0f);
    return;
#endif
#pragma unroll
    for (int q4 = 0; q4 < 8; ++q4) {
#ifdef KSA32_ABL_NOLOAD
      for (int j = 0; j < 4; ++j) win[4 * q4 + j] = (float)(l + 4 * q4 + j) * p.u8_inv_scale;
#else
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, l * 16, q4 * L * 16, 0);
      const float sc = FMT == FMT_U8 ? p.u8_inv_scale : 1.0f;
      win[4 * q4 + 0] = __uint_as_float(w.x) * sc;
      win[4 * q4 + 1] = __uint_as_float(w.y) * sc;
      win[4 * q4 + 2] = __uint_as_float(w.z) * sc;
      win[4 * q4 + 3] = __uint_as_float(w.w) * sc;
#endif
    }
  };
  if constexpr (KSA32_WIN_REGS) load_taps();

  // 32 samples l + L*q of one window (8 B/lane, 512 B per wave-instruction); the descriptor spans exactly this
  // frame, so every load is range-checked by the hardware
  raw_t raw[32];
  auto issue_loads = [&](int vf, int k) {
    const int frame = vf / NP;
    const char* fbase = reinterpret_cast<const char*>(p.iq) + (long long)frame * p.frame_stride * SB;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(fbase), 0, p.frame_len * SB, 0x00020000);
    const int voff = (p.starts[k] + l) * SB;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
#ifdef KSA32_ABL_NOLOAD    // timing-only ablation build: wrong results by construction
      if constexpr (FMT == FMT_C64) { raw[q].x = voff + q; raw[q].y = voff * q; }
      else raw[q] = voff + q;
#else
      if constexpr (FMT == FMT_C64) raw[q] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, L * q * SB, 0);
      else raw[q] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, L * q * SB, 0);
#endif
    }
  };
  auto k_lo_of = [&](int vf) { return (int)((long long)p.nwin * (vf % NP) / NP); };
  auto k_hi_of = [&](int vf) { return (int)((long long)p.nwin * (vf % NP + 1) / NP); };

  // The next window's IQ loads are issued after the middle pass's butterflies (its 62 twiddle registers are dead
  // by then): in flight under the second exchange, the last pass and the fold, consumed at the top of the next
  // window.  Unconditional (the frame's last window is simply loaded again): a conditional issue keeps the 64
  // landing registers formally live through the whole iteration and the allocator spills 100+.
#define PREFETCH_NEXT()                                      \
  do {                                                       \
    if constexpr (KSA32_PREFETCH) {                          \
      __builtin_amdgcn_sched_barrier(0);                     \
      issue_loads(vf, k + 1 < k_hi ? k + 1 : k);             \
      if constexpr (KSA32_PREFETCH == 2) load_taps();        \
      __builtin_amdgcn_sched_barrier(0);                     \
    }                                                        \
  } while (0)

#ifdef KSA_STAMPS   // diagnostic build: where a wave's time goes (segments named at the KSA_STAMP calls below)
  unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
  for (int vf = blockIdx.x; vf < total; vf += gridDim.x) {
    const int frame = vf / NP;
    const int k_lo = k_lo_of(vf), k_hi = k_hi_of(vf);
    float acc[32];
    const float init = p.cumu == CUMU_MIN ? __builtin_inff() : 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = init;
    if constexpr (KSA32_PREFETCH) issue_loads(vf, k_lo);
    if constexpr (KSA32_PREFETCH == 2) load_taps();

    for (int k = k_lo; k < k_hi; ++k) {
      // ---- samples of window (vf, k): window multiply (rows A0, A4) ------------------------------------
      if constexpr (!KSA32_PREFETCH) issue_loads(vf, k);
      if constexpr (!KSA32_WIN_REGS && KSA32_PREFETCH != 2) load_taps();
      float2 v[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        if constexpr (FMT == FMT_C64) {
          const unsigned xr = raw[q].x, xi = raw[q].y;
          v[q] = make_float2(__uint_as_float(xr) * win[q], __uint_as_float(xi) * win[q]);
        } else {
          const unsigned short x = raw[q];
          v[q] = make_float2(((float)(x & 0xff) - p.u8_offset) * win[q], ((float)(x >> 8) - p.u8_offset) * win[q]);
        }
      }
      KSA_STAMP(0);    // issue + wait for the 32 IQ loads and 8 tap loads, window multiply
      // ---- pass 0: radix 32, no twiddles; butterfly i = l, outputs to 32*l + t' ----------------------
      dft32(v);
      KSA_STAMP(1);    // pass 0 butterflies
      lds_barrier();   // the previous window's (frame's) LDS reads are done
      KSA_STAMP(2);    // barrier in front of exchange 1
#pragma unroll
      for (int P0 = 0; P0 < 32; ++P0) K32_ST(my[pad32(l * 32 + perm32(P0))], v[P0]);
      lds_barrier();
      KSA_STAMP(3);    // exchange 1: stores + barrier
      // ---- pass 1: p = 32, k = l mod 32 ---------------------------------------------------------------
#pragma unroll
      for (int q = 0; q < 32; ++q) K32_LD(v[(q % B1) * R1 + q / B1], my[pad32(l + L * q)]);
      {
        const int kk = l & 31;
        const float2* tw = tw_lds + kk;
        if constexpr (R1 == 32) {
          float2 ta[15], tb[15];
#pragma unroll
          for (int e = 0; e < 15; ++e) { ta[e] = tw[(1 + e) * 32]; tb[e] = tw[(16 + e) * 32]; }
          dft32_fused(v, tw[0], ta, tb);
          KSA_STAMP(4);  // exchange-1 reads, 31 twiddle reads, pass 1 butterflies
          lds_barrier();
          const int j = (l - kk) * 32 + kk;
#pragma unroll
          for (int P1 = 0; P1 < 32; ++P1) K32_ST(my[pad32(j + perm32(P1) * 32)], v[P1]);
        } else {
          float2 tm[15];
#pragma unroll
          for (int e = 0; e < 15; ++e) tm[e] = tw[e * 32];
          dft16_fused_at<0>(v, tm);      // i = l      (k = l mod 32)
          dft16_fused_at<16>(v, tm);     // i = l + L  (L is a multiple of 32: same k)
          KSA_STAMP(4);
          lds_barrier();
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int j = (l + b * L - kk) * 16 + kk;
#pragma unroll
            for (int P1 = 0; P1 < 16; ++P1) K32_ST(my[pad32(j + perm<16>(P1) * 32)], v[b * 16 + P1]);
          }
        }
      }
      PREFETCH_NEXT();   // v is dead (written to LDS): the landing registers are free from here to the next window's top
      lds_barrier();
      KSA_STAMP(5);      // exchange 2: barrier, stores, barrier
      // ---- pass 2: radix 16, p = N/16, butterflies i = l + b*L with k = i, one after the other (16 live data
      //      registers instead of 32), each followed by |X| and the fold over this block's windows (K:391-395)
      const int cm = CM == 0 ? p.cumu : CM;
      const int ew = k == 0 ? nm1 : nm1 - k + 1;         // closed form of the (a+x)/2 recursion: weight 2^-ew
      const float wgt = ldexpf(1.0f, -ew);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float2 u[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
#ifdef KSA32_ABL_NOLDS
          u[t] = v[b * 16 + t];
#else
          u[t] = my[pad32(l + L * (b + 2 * t))];
#endif
        }
        if constexpr (KSA32_TW6) {
          float2 tb[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            if (b == 1 && e >= 6 - KSA32_TWL_DYN) {
              constexpr int rows[6] = {3, 4, 5, 0, 1, 2};
              const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(twrsrc, l * 8, (15 + rows[e]) * L * 8, 0);
              tb[e] = make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
            } else {
              tb[e] = twl[b][e];
            }
          }
          dft16_tw_at<0>(u, tb[0], tb[1], tb[2], tb[3], tb[4], tb[5]);
        } else dft16_fused_at<0>(u, reinterpret_cast<const float2(&)[15]>(twl[b]));
        if (cm == CUMU_AVG) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[b * 16 + i] = fmaf(wgt, __builtin_amdgcn_sqrtf(fmaf(u[i].x, u[i].x, u[i].y * u[i].y)), acc[b * 16 + i]);
        } else if (cm == CUMU_MAX) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[b * 16 + i] = nan_max_nonneg(acc[b * 16 + i], fmaf(u[i].x, u[i].x, u[i].y * u[i].y));
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[b * 16 + i] = nan_min(acc[b * 16 + i], fmaf(u[i].x, u[i].x, u[i].y * u[i].y));
        }
      }
      KSA_STAMP(6);      // exchange-2 reads, pass 2 (two radix-16 butterflies) and the fold
    }

    // ---- natural bin order through LDS, then the common output stage --------------------------------
    float* const red = reinterpret_cast<float*>(lds);
    __syncthreads();   // (also drains the frame's last, redundant prefetch)
#pragma unroll
    for (int i = 0; i < 32; ++i) red[l + L * ((i >> 4) + 2 * perm<16>(i & 15))] = acc[i];
    __syncthreads();
    if (NP == 1) {
      finish_frame<N, T, 1, CM>(p, red, frame, l);
    } else {
      float4* const dst = reinterpret_cast<float4*>(p.part_out + (long long)vf * N);
      const float4* red4 = reinterpret_cast<const float4*>(red);
      for (int q = l; q < N / 4; q += T) dst[q] = red4[q];
    }
    // (the next frame's first exchange barrier orders these LDS reads before its writes)
    KSA_STAMP(8);        // per-frame output stage
  }
#ifdef KSA_STAMPS
  if (p.dbg && (l & 63) == 0) {
    for (int i = 0; i < 12; ++i) p.dbg[((long long)blockIdx.x * (T / 64) + l / 64) * 12 + i] = seg[i];
  }
#endif
}

#undef PREFETCH_NEXT
#undef K32_ST
#undef K32_LD

}  // namespace ksa
