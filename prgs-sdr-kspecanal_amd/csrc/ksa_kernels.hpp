// HIP kernels of the spectrum / waterfall path for gfx950 (MI355X).
//
//  spectrum_kernel<N,FMT>   rows A0,A4-A9,A12 of SURVEY.md section 8: unpack, window, FFT, |X|,
//                           fold over the block's windows, fftshift, dB, waterfall row.
//                           One workgroup walks frames (persistent, grid-stride); the transform
//                           lives in VGPRs + LDS, the block of IQ is read from HBM once.
//  accumulate_*             row A10: Max/Min/Avg/Cur over a batch of frames (K:470-476)
//  scan_stitch_kernel       row A13: band stitch + per-pass accumulate (K:622-668)
//  rowmax_kernel            row A12 for the scan: per-pass waterfall row from Fft.Avg (K:696-697)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "ksa_fft.hpp"

namespace ksa {

enum { CUMU_AVG = 1, CUMU_MAX = 2, CUMU_MIN = 3 };
enum { OUT_LINEAR = 0, OUT_DB = 1, OUT_DB_CLIP = 2 };
enum { FMT_C64 = 0, FMT_U8 = 1 };
constexpr int HM_ROWS = 128;

struct SpecParams {
  const void* iq;           // float2[] or uchar2[]; frame f starts at sample f*frame_stride
  long long frame_stride;
  int frame_len;            // samples readable from a frame's start (range-checked by the buffer loads)
  int nframes;
  int nwin;
  const int* starts;        // [nwin] sample offsets inside a frame
  const float* window;      // [N]
  const float2* tw_mid;     // concatenated middle-pass tables
  const float2* tw_last;    // [15][N/16]
  float scale;              // 2*winAdj/N
  int cumu;
  int out_mode;
  float gain, min_amp;
  float u8_offset, u8_inv_scale;
  float* out;               // [nframes][N]
  int hm_w;                 // 0: no waterfall row
  const float* adj;         // [N] or null
  float* hm_rows;           // [nframes][hm_w] or null
  float* hm_ring;           // [128][hm_w] or null
  int hm_index0;            // ring row of frame 0
  int hm_first;             // first frame stored in the ring
  int parts;                // > 1: each frame's windows are split over `parts` workgroups (small batches)
  float* part_out;          // [nframes*parts][N] partial folds (natural bin order), combined by combine_parts_kernel
  unsigned long long* dbg;  // diagnostic builds only (-DKSA_STAMPS): [grid][16] cycle sums; null otherwise
};

// Shader clock held under the spectrum stage (bench.py `shader_clock_ghz_live`, ksa_prof_clock): a stamp kernel of CLK_WGS
// single-wave workgroups runs on the engine's stream directly before and directly after a PROFILED spectrum stage; each wave
// reads s_memtime (shader cycles) and s_memrealtime (the constant 100 MHz counter) and stores the pair under the identity of
// the hardware it ran on (XCC_ID and the SE / SH / CU fields of HW_ID), so that the host only ever subtracts two readings of
// the SAME counter: d(memtime) / d(memrealtime) x 100 MHz per (XCD, SE, CU) seen at both ends, median over those and over the
// launches (MI355X_MICROARCH.md, 'DVFS give-back' item 6).  (Keyed by XCC_ID alone the quotients scattered from 0.1 to 1000
// GHz on short stages: s_memtime readings taken on different shader engines are not comparable.)  The spectrum kernels
// themselves carry NO stamp: their register files are full, and two stamps inside spectrum_kernel<4096,c64,8,AVG> cost
// three more spilled VGPRs (2 -> 5; measured, removed).  The interval includes the two launch gaps (a few us).
constexpr int CLK_WGS = 2048;     // single-wave workgroups per stamp launch: several per CU, so that most CUs are hit at both ends
constexpr int CLK_KEYS = 4096;    // XCC_ID[3:0] << 8 | HW_ID[15:8] (CU_ID[3:0], SH_ID, SE_ID[2:0])
constexpr int CLK_SLOTS = 16;     // profiled launches whose stamps are kept (ring)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
// out = [which = 0 before | 1 after][key] = {memtime, memrealtime}, one 16-byte store; waves that share a key overwrite each
// other (any of them is as good as the other: they run within a microsecond)
__global__ __launch_bounds__(64) void clock_stamp_kernel(u64x2* out, int which) {
  unsigned long long c, r;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));   // hwreg(HW_REG_XCC_ID, 0, 4)
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | ((8 - 1) << 11));     // hwreg(HW_REG_HW_ID, 8, 8): CU, SH, SE
  if (threadIdx.x == 0) {
    u64x2 v;
    v.x = c;
    v.y = r;
    out[(size_t)which * CLK_KEYS + (((xcc & 15u) << 8) | (hw & 255u))] = v;
  }
}

// 10*log10(x) - gain through the hardware log2 (v_log_f32, 1 ulp): 10*log10(2) * log2(x) - gain.  libm's log10f
// costs ~12 more VALU instructions per bin (1.2 % of the config-2 kernel); 0 -> -inf and NaN -> NaN are kept,
// denormal magnitudes (< 1.2e-38, below -380 dB) read as zero.
__device__ __forceinline__ float db_of(float lin, float gain) { return fmaf(__builtin_amdgcn_logf(lin), 3.01029995663981195f, -gain); }

// np.max / np.min semantics (K:141-143, K:195): a NaN on either side wins.  v_max_f32 / v_min_f32 return the other operand.
__device__ __forceinline__ float nan_max(float a, float b) { return __builtin_isunordered(a, b) ? __builtin_nanf("") : fmaxf(a, b); }
__device__ __forceinline__ float nan_min(float a, float b) { return __builtin_isunordered(a, b) ? __builtin_nanf("") : fminf(a, b); }
// The MAX fold of |X|^2 (K:141): both operands are +0 .. +inf or NaN, and on those the unsigned order of the bit patterns
// is the float order with every NaN (either sign) above +inf -- one v_max_u32 is the NaN-propagating maximum.
__device__ __forceinline__ float nan_max_nonneg(float a, float b) {
  const unsigned x = __float_as_uint(a), y = __float_as_uint(b);
  return __uint_as_float(x > y ? x : y);
}
// Clip2MinAmp (np.clip, K:100-101): NaN stays NaN (fmaxf would return min_amp)
__device__ __forceinline__ float clip_min(float lin, float min_amp) { return lin < min_amp ? min_amp : lin; }
// dB value of one bin in out_mode units: zeroSpan keeps -inf (K:469), the scan replaces +-inf by 0 (infTo = 0, K:641 -> K:110-111)
__device__ __forceinline__ float out_db(float lin, int out_mode, float gain, float min_amp) {
  if (out_mode == OUT_DB_CLIP) lin = clip_min(lin, min_amp);
  float o = db_of(lin, gain);
  if (out_mode == OUT_DB_CLIP && fabsf(o) == __builtin_inff()) o = 0.0f;
  return o;
}

// Output stage of one frame (rows A7 tail, A8, A9, A12): slot combine, 2*winAdj/N scale, fftshift,
// LogNoGain / Clip2MinAmp, store, waterfall cell max.  red = [S][N] floats in LDS (natural bin order).
// Each thread takes 4 consecutive bins per step: one ds_read_b128, one 16-byte store, and the
// waterfall cell (g = N/W consecutive bins) needs log2(g/4) shuffle steps instead of log2(g).
// red row stride of slot s (floats).  N = 32 / 64 hold 32 / 16 transforms per wave and their unskewed rows (stride = 0 mod 32
// banks) made every staging store 16- / 8-way conflicted (25 % of config 4's LDS cycles): rows skewed by 4 floats (16-byte
// alignment kept) spread the 32 lanes of a store over the banks.  Measured on one box (15 windows per frame): N = 64 +6 %,
// N = 32 +20 %; N = 16 / 128 / 256 -2 ... +1 % (noise or worse), so those keep the plain stride; config 4 (71 windows) +0.3 %.
#ifndef KSA_RED_SKEW
#define KSA_RED_SKEW 1
#endif
template <int N, int S>
struct RedStride { static constexpr int value = (KSA_RED_SKEW && (N == 32 || N == 64)) ? N + 4 : N; };

// lane ^ 1 / lane ^ 2 inside a DPP quad (quad_transpose4 below)
__device__ __forceinline__ float dpp_quad_xor1(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
}
// The common shapes of the output stage (one transform per workgroup, AVG fold, dB output, no waterfall cell or one that a
// shuffle tree inside a wave reduces: 4 <= g <= 256) as a loop whose body has NO run-time mode branch: the generic loop
// below decides fold mode, output unit and cell path per element with ~25 scalar branches per step, and the cycle stamps put
// 8.3 % of a config-2 wave's time into it (profiles/r05_c2_stamps.txt).  Same arithmetic in the same order: identical rows.
#ifndef KSA_FINISH_FAST
#define KSA_FINISH_FAST 1
#endif
template <int N, int T, int S, int OM, int HM>   // HM: 0 no waterfall cell, 1 one bin per cell (g == 1), 2 shuffle tree (4 <= g <= 256)
__device__ __forceinline__ void finish_rows_avg(const SpecParams& p, const float* red, float* orow, float* hm_row, float* hm_ring,
                                                int g, int tid) {
  constexpr int RS = RedStride<N, S>::value;
  const float4* red4 = reinterpret_cast<const float4*>(red);
  const int lanes = g >> 2;                   // lanes per waterfall cell (HM only): 1 .. 64
  const int cell_shift = 31 - __builtin_clz(g | 1);
  // (issuing the next step's LDS read before this step's values are used, and lane ^ 1 / lane ^ 2 as DPP quad permutes instead of
  //  ds_bpermute, were both measured on top: +-0 and -0.9 % at config 2, profiles/r05_ab_finish.txt -- the plain loop stays)
#pragma unroll 1
  for (int q = tid; q < N / 4; q += T) {
    float4 r = red4[q];
    if constexpr (S > 1) {       // slot combine in slot order, five slots' loads in flight (see finish_frame)
      constexpr int CH = (S - 1) < 5 ? (S - 1) : 5;
#pragma unroll 1
      for (int s0 = 1; s0 < S; s0 += CH) {
        float4 x[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (s0 + u < S) x[u] = red4[(s0 + u) * (RS / 4) + q];
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (s0 + u < S) { r.x += x[u].x; r.y += x[u].y; r.z += x[u].z; r.w += x[u].w; }
      }
    }
    float o[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = OM != OUT_LINEAR ? out_db(o[u] * p.scale, OM, p.gain, p.min_amp) : o[u] * p.scale;
    const int sh = (4 * q + N / 2) & (N - 1);
    *reinterpret_cast<float4*>(orow + sh) = make_float4(o[0], o[1], o[2], o[3]);
    if constexpr (HM != 0) {
      if (p.adj) {
        const float4 a = *reinterpret_cast<const float4*>(p.adj + sh);
        o[0] -= a.x; o[1] -= a.y; o[2] -= a.z; o[3] -= a.w;
      }
    }
    if constexpr (HM == 1) {
      if (hm_row) *reinterpret_cast<float4*>(hm_row + sh) = make_float4(o[0], o[1], o[2], o[3]);
      if (hm_ring) *reinterpret_cast<float4*>(hm_ring + sh) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if constexpr (HM == 2) {
      float hv = fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3]));
      const bool bad = __builtin_isunordered(o[0], o[1]) | __builtin_isunordered(o[2], o[3]);
      for (int m = 1; m < lanes; m <<= 1) hv = fmaxf(hv, __shfl_xor(hv, m));
      const unsigned long long nb = __ballot(bad);
      if ((tid & (lanes - 1)) == 0) {
        const unsigned long long cell = lanes >= 64 ? ~0ull : (((1ull << lanes) - 1ull) << (tid & 63));
        if (nb & cell) hv = __builtin_nanf("");
        if (hm_row) hm_row[sh >> cell_shift] = hv;
        if (hm_ring) hm_ring[sh >> cell_shift] = hv;
      }
    }
  }
}

template <int N, int T, int S, int CM = 0>   // CM: the kernel's compile-time fold mode (0 = decided at run time)
__device__ __forceinline__ void finish_frame(const SpecParams& p, float* red, int frame, int tid) {
  constexpr int RS = RedStride<N, S>::value;
  const int g = p.hm_w > 0 ? N / p.hm_w : 0;  // bins per waterfall cell
  const bool hm_fast = g > 0 && g <= 256 && g <= 4 * T;
  float* const orow = p.out + (long long)frame * N;
  float* const hm_row = p.hm_rows ? p.hm_rows + (long long)frame * p.hm_w : nullptr;
  float* const hm_ring = (p.hm_ring && frame >= p.hm_first)
                             ? p.hm_ring + ((p.hm_index0 + frame) % HM_ROWS) * p.hm_w : nullptr;
#ifndef KSA_FINISH_FAST_SMALL
#define KSA_FINISH_FAST_SMALL 1   // the same for the small transforms (S > 1 slots per workgroup)
#endif
  if constexpr (KSA_FINISH_FAST && (S == 1 ? N >= 1024 : KSA_FINISH_FAST_SMALL) && (CM == 0 || CM == CUMU_AVG)) {
    if (p.cumu == CUMU_AVG && p.out_mode == OUT_LINEAR && g == 0) {      // sdr_curscan's linear spectra; the second stage of N >= 32768
      finish_rows_avg<N, T, S, OUT_LINEAR, 0>(p, red, orow, hm_row, hm_ring, g, tid);
      return;
    }
    if (p.cumu == CUMU_AVG && p.out_mode != OUT_LINEAR && (g <= 1 || (hm_fast && g >= 4))) {
      if (p.out_mode == OUT_DB) {
        if (g == 0) finish_rows_avg<N, T, S, OUT_DB, 0>(p, red, orow, hm_row, hm_ring, g, tid);
        else if (g == 1) finish_rows_avg<N, T, S, OUT_DB, 1>(p, red, orow, hm_row, hm_ring, g, tid);
        else finish_rows_avg<N, T, S, OUT_DB, 2>(p, red, orow, hm_row, hm_ring, g, tid);
      } else {      // (the scan's clipped output never rides with a waterfall cell: its rows come from Fft.Avg per pass, K:696-697)
        if (g == 0) finish_rows_avg<N, T, S, OUT_DB_CLIP, 0>(p, red, orow, hm_row, hm_ring, g, tid);
        else if (g == 1) finish_rows_avg<N, T, S, OUT_DB_CLIP, 1>(p, red, orow, hm_row, hm_ring, g, tid);
        else finish_rows_avg<N, T, S, OUT_DB_CLIP, 2>(p, red, orow, hm_row, hm_ring, g, tid);
      }
      return;
    }
  }
  const float4* red4 = reinterpret_cast<const float4*>(red);
#pragma unroll 1
  for (int q = tid; q < N / 4; q += T) {
    float4 r = red4[q];
    if constexpr (S > 1) {
      // slot combine, in slot order (the order fixes the rounding of the AVG sum).  The loads of KSA_SLOT_CH slots are issued
      // before the first is used: as a rolled loop of dependent load -> combine steps this was 15 LDS round trips per frame at
      // N = 64 -- with the stores and the dB math 15.6 % of a wave's time at config 4 (profiles/r05_c4_stamps.txt).  Measured
      // (profiles/r05_ab_slot.txt): config 4 +3.5 %, N = 64 / 32 at 50 % overlap +24 % / +23 %; 5 slots in flight keep N = 64 at
      // 121 VGPRs (four waves per SIMD; 15 in flight: 153).  A variant with one bin per lane (all 64 lanes busy instead of N/4)
      // on top measured +0.4 % at config 4 and -3 % at N = 64 / 50 %: not kept.
#ifndef KSA_SLOT_CH
#define KSA_SLOT_CH 5
#endif
      constexpr int CH = (S - 1) < KSA_SLOT_CH ? (S - 1) : KSA_SLOT_CH;
#pragma unroll 1
      for (int s0 = 1; s0 < S; s0 += CH) {
        float4 x[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (s0 + u < S) x[u] = red4[(s0 + u) * (RS / 4) + q];
        if (p.cumu == CUMU_AVG) {
#pragma unroll
          for (int u = 0; u < CH; ++u)
            if (s0 + u < S) { r.x += x[u].x; r.y += x[u].y; r.z += x[u].z; r.w += x[u].w; }
        } else if (p.cumu == CUMU_MAX) {
#pragma unroll
          for (int u = 0; u < CH; ++u)
            if (s0 + u < S) { r.x = nan_max(r.x, x[u].x); r.y = nan_max(r.y, x[u].y); r.z = nan_max(r.z, x[u].z); r.w = nan_max(r.w, x[u].w); }
        } else {
#pragma unroll
          for (int u = 0; u < CH; ++u)
            if (s0 + u < S) { r.x = nan_min(r.x, x[u].x); r.y = nan_min(r.y, x[u].y); r.z = nan_min(r.z, x[u].z); r.w = nan_min(r.w, x[u].w); }
        }
      }
    }
    float o[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float lin = p.cumu == CUMU_AVG ? o[u] : __builtin_amdgcn_sqrtf(o[u]);
      lin *= p.scale;
      o[u] = p.out_mode != OUT_LINEAR ? out_db(lin, p.out_mode, p.gain, p.min_amp) : lin;
    }
    const int sh = (4 * q + N / 2) & (N - 1);   // fftshift keeps runs of 4 together
    *reinterpret_cast<float4*>(orow + sh) = make_float4(o[0], o[1], o[2], o[3]);
    if (g > 0) {
      if (p.adj) {
        const float4 a = *reinterpret_cast<const float4*>(p.adj + sh);
        o[0] -= a.x; o[1] -= a.y; o[2] -= a.z; o[3] -= a.w;
      }
      // the cell maximum is np.max (K:195 through K:480): a NaN bin -- -inf minus a -inf baseline, K:405 -- makes the cell NaN
      if (!hm_fast) {
        *reinterpret_cast<float4*>(red + S * RS + sh) = make_float4(o[0], o[1], o[2], o[3]);  // second plane
      } else if (g == 1) {
        if (hm_row) *reinterpret_cast<float4*>(hm_row + sh) = make_float4(o[0], o[1], o[2], o[3]);
        if (hm_ring) *reinterpret_cast<float4*>(hm_ring + sh) = make_float4(o[0], o[1], o[2], o[3]);
      } else if (g == 2) {
        const float2 c = make_float2(nan_max(o[0], o[1]), nan_max(o[2], o[3]));
        if (hm_row) *reinterpret_cast<float2*>(hm_row + sh / 2) = c;
        if (hm_ring) *reinterpret_cast<float2*>(hm_ring + sh / 2) = c;
      } else {
        // g/4 consecutive lanes hold the g consecutive bins of one cell; NaN travels as a flag beside the v_max_f32 tree
        float hv = fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3]));
        const bool bad = __builtin_isunordered(o[0], o[1]) | __builtin_isunordered(o[2], o[3]);
        for (int m = 1; m < g / 4; m <<= 1) hv = fmaxf(hv, __shfl_xor(hv, m));
        const unsigned long long nb = __ballot(bad);
        if ((tid & (g / 4 - 1)) == 0) {
          const unsigned long long cell = g >= 256 ? ~0ull : (((1ull << (g / 4)) - 1ull) << (tid & 63));
          if (nb & cell) hv = __builtin_nanf("");
          if (hm_row) hm_row[sh / g] = hv;
          if (hm_ring) hm_ring[sh / g] = hv;
        }
      }
    }
  }
  if (g > 0 && !hm_fast) {
    __syncthreads();
    const float* hmbuf = red + S * RS;
#pragma unroll 1
    for (int cell = tid; cell < p.hm_w; cell += T) {
      float hv = hmbuf[cell * g];
      bool bad = hv != hv;
      for (int i = 1; i < g; ++i) {
        const float x = hmbuf[cell * g + i];
        bad |= x != x;
        hv = fmaxf(hv, x);
      }
      if (bad) hv = __builtin_nanf("");
      if (hm_row) hm_row[cell] = hv;
      if (hm_ring) hm_ring[cell] = hv;
    }
  }
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// In-kernel cycle stamps (diagnostic build only; never compiled into the shipped library).
#ifdef KSA_STAMPS
#define KSA_STAMP(i)                                                                        \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    unsigned long long _t;                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    seg[i] += _t - t_last;                                                                  \
    t_last = _t;                                                                            \
  } while (0)
#else
#define KSA_STAMP(i) do {} while (0)
#endif

#ifdef KSA_ABL_NOLDS   // timing-only ablation build (no exchange): wrong results by construction
#define KSA_SYNC() do {} while (0)
#define KSA_LDS_ST(dst, val) do {} while (0)
#define KSA_LDS_LD(dst, src) do {} while (0)
#else
// Exchange barriers of the window loop.  __syncthreads() also waits for every outstanding global access
// (s_waitcnt vmcnt(0)): after a frame's output stage that means the first exchange of the next frame stalls until the
// frame's dB row has reached memory.  The LDS-only form orders what the exchanges need (LDS) and lets stores drain.
#ifdef KSA_LDS_BARRIER
#define KSA_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define KSA_SYNC() __syncthreads()
#endif
#define KSA_LDS_ST(dst, val) (dst) = (val)
// Exchange reads as single ds_read_b64: hipcc otherwise merges pairs of them into ds_read2_b64 / ds_read2st64_b64, which
// move 128 B/clk instead of 256 and bank over 16-lane groups (MI355X_MICROARCH.md, LDS table) -- the conflict-free layout
// of Plan<N> is built for ds_read_b64's 32-lane groups.  A relaxed atomic 64-bit load is an ordinary ds_read_b64 that the
// load / store optimizer leaves alone.
#ifndef KSA_LDS_ATOMIC_LD
#define KSA_LDS_ATOMIC_LD 1
#endif
__device__ __forceinline__ float2 lds_ld64(const float2* p) {
#if KSA_LDS_ATOMIC_LD
  const unsigned long long v = __atomic_load_n(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED);
  return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
#else
  return *p;
#endif
}
#define KSA_LDS_LD(dst, src) (dst) = lds_ld64(&(src))
#endif

// Per-size tuning, every choice measured A/B on MI355X.  WPS = waves per SIMD the
// register allocator must leave room for: LDS caps a CU at three to four 4096-point transforms in flight
// whatever the register count, so for T <= 256 three spill-free waves (168 VGPRs) beat four spilling ones;
// 1024-thread workgroups need 4 waves per SIMD (128 VGPRs) by construction.  The -D overrides exist for
// tools/variants.sh experiments only.
template <int N>
struct Tune {
#ifdef KSA_WAVES_PER_SIMD
  static constexpr int WPS = KSA_WAVES_PER_SIMD;
#else
  static constexpr int WPS = Plan<N>::T >= 512 ? 4 : 3;
#endif
  // radix-16 passes with all 15 twiddles folded into FMA-fused radix-4 levels (192 instead of 256 VALU
  // ops per pass) where the 30 twiddle registers fit without spills; 6-twiddle form otherwise.
#ifdef KSA_FUSED
  static constexpr bool FUSED = KSA_FUSED;
#else
  static constexpr bool FUSED = Plan<N>::T <= 256;
#endif
  // the last pass keeps its twiddles in VGPRs: 30 for the fused form, 12 for the 6-twiddle form
#ifdef KSA_FUSED_LAST
  static constexpr bool FUSED_LAST = KSA_FUSED_LAST;
#else
  static constexpr bool FUSED_LAST = FUSED;
#endif
  // window taps: in LDS ([4][L][4] floats, four ds_read_b128 per window) where three workgroups per CU
  // leave the room (T <= 256), in VGPRs otherwise.  Frees 16 VGPRs for the fused-twiddle radix-16.
#ifdef KSA_WIN_LDS
  static constexpr bool WIN_LDS = KSA_WIN_LDS;
#else
  static constexpr bool WIN_LDS = Plan<N>::T <= 256;
#endif
  // T >= 512 (LDS full, 128-VGPR cap): taps are re-read from the L2-resident table with every window's
  // IQ loads instead of living in 16 VGPRs
#ifdef KSA_WIN_GLOBAL
  static constexpr bool WIN_GLOBAL = KSA_WIN_GLOBAL;
#else
  static constexpr bool WIN_GLOBAL = Plan<N>::T >= 512;
#endif
  // N = 64: a transform is held by the four lanes of one DPP quad, so the one exchange between its two passes is a
  // 4x4 transpose inside the quad (quad_perm moves + selects, no LDS, no barrier).  Measured against the LDS
  // exchange on MI355X at the quickFullScan shape: see DESIGN.md section 4.1 (-DKSA_QUAD=0/1 for the A/B).
#ifndef KSA_QUAD
#define KSA_QUAD 0
#endif
  static constexpr bool QUAD_XCHG = KSA_QUAD && N == 64;
  // window multiply folded into the first radix-4 level of pass 0 (dft_first_win): first-pass radix 16 or 4
#ifndef KSA_WIN_FUSED
#define KSA_WIN_FUSED 1
#endif
  static constexpr bool WIN_FUSED = KSA_WIN_FUSED && (Plan<N>::R0 == 16 || Plan<N>::R0 == 4);
  static constexpr int LDS_BYTES = Plan<N>::LDS_BYTES + (WIN_LDS ? N * 4 : 0);
  // Fold mode (AVG/MAX/MIN) as a template constant of the kernel instead of a branch inside the window loop:
  // with the branch, hipcc copies the 16 accumulators to and from the branch's registers in every window (32
  // v_mov of ~700 VALU instructions) and schedules around three merge points.  Measured over N = 16..16384 x
  // hops 0.5/0.25/0.1 (tools/fold_sweep.sh): constant wins everywhere (+2..+58 %, e.g. N=1024 75 % overlap
  // 1.23 -> 0.78 ms, N=64 0.60 -> 0.45 ms) except N=4096 on the general path, where the constant form makes
  // hipcc serialise the loads (155 VGPRs, 1.95 vs 1.75 ms) -- that one keeps the run-time branch.
#if defined(KSA_FOLD_GENERIC)     // tools/fold_sweep.sh: the run-time branch everywhere
  static constexpr bool fold_const(int) { return false; }
#elif defined(KSA_FOLD_CONST_ALL) // ... and the template constant everywhere
  static constexpr bool fold_const(int) { return true; }
#else
  static constexpr bool fold_const(int rm) { return !(N == 4096 && rm == 0); }
#endif
};

// 4x4 transpose of r[0..3] across the four lanes of a DPP quad: afterwards r[j] of lane q holds what r[q] of lane j
// held.  Two butterfly stages (lane ^ 1 on register pairs (0,1), (2,3); lane ^ 2 on (0,2), (1,3)); per pair one select
// of the element to hand over, one quad_perm move and two selects: the wave-local form of a Stockham exchange.
__device__ __forceinline__ void quad_transpose4(float& r0, float& r1, float& r2, float& r3, bool odd1, bool odd2) {
  {
    const float g = dpp_quad_xor1(odd1 ? r0 : r1);
    if (odd1) r0 = g; else r1 = g;
    const float h = dpp_quad_xor1(odd1 ? r2 : r3);
    if (odd1) r2 = h; else r3 = h;
  }
  {
    const float g = dpp_quad_xor2(odd2 ? r0 : r2);
    if (odd2) r0 = g; else r2 = g;
    const float h = dpp_quad_xor2(odd2 ? r1 : r3);
    if (odd2) r1 = h; else r3 = h;
  }
}

// RM > 0: consecutive windows are exactly RM*L samples apart (L = N/16 threads), so thread l's samples
// l + L*q of window k+1 are its samples q+RM of window k: the raw values stay in VGPRs and only RM new
// samples per thread are loaded per window (50 % overlap: RM = 8, 75 %: RM = 4).  Each IQ sample is then
// read from HBM exactly once.  RM = 0 is the general path (fractional hops, K:386).
template <int N, int FMT, int RM, int CM>
__global__ __launch_bounds__(Plan<N>::T, Tune<N>::WPS) void spectrum_kernel(const SpecParams p) {
  static_assert(RM == 0 || Plan<N>::S == 1, "sample reuse needs one transform per workgroup");
  using P = Plan<N>;
  constexpr int L = P::L, T = P::T, S = P::S, M = P::M, R0 = P::R0, B0 = P::B0, NPAD = P::NPAD;
  constexpr int SB = FMT == FMT_C64 ? 8 : 2;  // bytes per IQ sample
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2* const tw_lds = lds + S * NPAD;

  const int tid = threadIdx.x;
  const int slot = S == 1 ? 0 : tid / L;   // S == 1: constant, keeps window indices wave-uniform (scalar loads)
  const int l = tid - slot * L;
  float2* const my = lds + slot * NPAD;

  // ---- per-thread constants: window taps (VGPRs or LDS) and last-pass twiddles (VGPRs) --------
  constexpr bool WIN_LDS = Tune<N>::WIN_LDS;
  float* const win_lds = reinterpret_cast<float*>(tw_lds + P::MID);   // [4][L][4] floats, shared by the slots
  float win[16];
  if constexpr (WIN_LDS) {
    for (int i = tid; i < N; i += T) {       // tap of sample n = l' + L*q lives at ((q>>2)*L + l')*4 + (q&3)
      const int q = i / L, ll = i - q * L;
      win_lds[((q >> 2) * L + ll) * 4 + (q & 3)] = p.window[i] * (FMT == FMT_U8 ? p.u8_inv_scale : 1.0f);
    }
  } else if constexpr (!Tune<N>::WIN_GLOBAL) {
#pragma unroll
    for (int q = 0; q < 16; ++q) win[q] = p.window[l + L * q] * (FMT == FMT_U8 ? p.u8_inv_scale : 1.0f);
  }
  // last pass: k = l.  FUSED: the 15 folded twiddles of dft16_fused (rows of the [15][N/16] table);
  // otherwise rows t = 1,2,3,4,8,12 of the plain w^t table for dft16_tw.
  constexpr bool FUSED = Tune<N>::FUSED, FUSED_LAST = Tune<N>::FUSED_LAST;
  float2 twl[FUSED_LAST ? 15 : 6];
  if constexpr (M >= 2) {
    if constexpr (FUSED_LAST) {
#pragma unroll
      for (int e = 0; e < 15; ++e) twl[e] = p.tw_last[e * P::P_LAST + l];
    } else {
      constexpr int rows[6] = {0, 1, 2, 3, 7, 11};
#pragma unroll
      for (int e = 0; e < 6; ++e) twl[e] = p.tw_last[rows[e] * P::P_LAST + l];
    }
  }
  // M == 3: the 15 folded middle-pass twiddles of a thread (they depend on l mod R0 only) live in VGPRs as well -- the
  // transposed exchange layout freed ~30 registers (136 instead of 168 at N = 4096), which is exactly what they need.
  // Measured at config 2 on one box: re-read from LDS per window 5.76 ms, in VGPRs 5.62 ms, with the prefetch below 5.51 ms.
  // (M > 3, experiments builds only, keeps the tables of its middle passes in LDS.)
#ifndef KSA_TWM_REGS
#define KSA_TWM_REGS 1
#endif
  constexpr bool TWM_REGS = KSA_TWM_REGS && M == 3 && FUSED;
  float2 twm[15];
  if constexpr (TWM_REGS) {
#pragma unroll
    for (int e = 0; e < 15; ++e) twm[e] = p.tw_mid[e * R0 + (l & (R0 - 1))];
  } else if constexpr (P::MID > 0) {
    for (int i = tid; i < P::MID; i += T) tw_lds[i] = p.tw_mid[i];
  }

  if constexpr (WIN_LDS) __syncthreads();   // taps are read before the first exchange barrier

  const int nm1 = p.nwin - 1;
  const int NP = p.parts > 1 ? p.parts : 1;   // window split: latency mode for batches smaller than the GPU

  // Raw IQ of one window per thread: 16 samples l + L*q, loaded at the top of the window (8 B/lane, 512 B per
  // wave-instruction).  The buffer descriptor is built from scalars only (a per-lane descriptor makes hipcc wrap
  // every load in a readfirstlane "waterfall" loop) and spans exactly this frame: every load is range-checked
  // by the hardware.  Prefetching the next window into a second register set was measured and dropped
  // (spills: 104 -> 165 M FFT/s without it at the 0.1 hop; 2.5 -> 1.6 ms with sample reuse).
  typedef typename std::conditional<FMT == FMT_C64, u32x2, unsigned short>::type raw_t;
  raw_t raw[16];
  const int start0 = p.starts[0];
  auto issue_loads = [&](int fr, int k, int q0, auto rotc) {
    constexpr int ROT = decltype(rotc)::value;   // ping-pong form: sample q lands in raw[(q + ROT) & 15]
    const char* fbase = reinterpret_cast<const char*>(p.iq) + (long long)fr * p.frame_stride * SB;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(fbase), 0, p.frame_len * SB, 0x00020000);
    // reuse path: hops are constant (RM*L samples), so the start is arithmetic -- no dependent scalar load
    const int start = RM > 0 ? start0 + k * (RM * L) : p.starts[k];
    const int voff = (start + l) * SB;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < q0) continue;
#ifdef KSA_ABL_NOLOAD   // timing-only ablation build: wrong results by construction
      if constexpr (FMT == FMT_C64) { raw[(q + ROT) & 15].x = voff + q; raw[(q + ROT) & 15].y = voff * q; }
      else raw[(q + ROT) & 15] = voff + q;
#else
#ifndef KSA_LOAD_AUX
#define KSA_LOAD_AUX 0   // cache policy of the IQ loads (experiments: 2 = nt)
#endif
      if constexpr (FMT == FMT_C64) raw[(q + ROT) & 15] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, L * q * SB, KSA_LOAD_AUX);
      else raw[(q + ROT) & 15] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, L * q * SB, 0);
#endif
    }
  };
  auto shift_raw = [&]() {
#pragma unroll
    for (int q = 0; q + RM < 16; ++q) raw[q] = raw[q + RM];
  };

#ifdef KSA_STAMPS
  unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
#ifndef KSA_PF
#define KSA_PF 1   // reuse path: the RM new samples of window k+1 are requested while window k is transformed.  Fits since the
                   // transposed exchange layout of round 4: alone 155 VGPRs and no spill; together with the middle twiddles in
                   // VGPRs (TWM_REGS) the N = 4096 reuse kernels sit at the 168-VGPR cap; what the rolled loop spills (2-8 registers:
                   // 75 % overlap, MAX / MIN folds) is stored before the window loop and reloaded in the output stage -- never
                   // inside the loop (tests/test_isa_regression.py holds the compiler to that).  +1.9 % at config 2, +0.6 % at 75 % overlap,
                   // +1..3 % at N = 2048; N = 1024 would spill 6-8 registers and large batches run the pair kernel there anyway
#endif
  // (requesting the first window of the workgroup's NEXT frame before this frame's output stage, with LDS-only barriers
  //  around the staging stores so that the loads stay in flight, measured 2.5 % SLOWER at config 2: profiles/r04_ab_prefetch_twiddles.txt)
  constexpr bool PF = KSA_PF && RM > 0 && N >= 2048;
  // Ping-pong form of the 50 %-overlap kernels of N = 4096 (round 5): at RM = 8 a window's new half is the next window's old half, so
  // instead of moving registers every window (shift the carried half down, copy the prefetched half in: 24 v_mov_b64 per window in
  // the rolled loop) the loop holds TWO windows and the halves of raw[] swap roles with the window's parity (ksa_window_body.inc is
  // included twice, PAR = 0 / 1: sample q sits in raw[(q + 8*PAR) & 15], the prefetch lands in the half that has just been converted).
  // 8 moves per window are left, the AVG kernel spills nothing any more (8 registers in the rolled form), and since every VALU
  // instruction of this kernel costs its full issue time (profiles/r05_sensitivity_c2.txt) that is time: config 2 +1.8 %, uint8 input
  // +3..5 % (profiles/r05_ab_pp.txt).  The body is an included file and not a lambda on purpose: as a generic lambda the same code made
  // hipcc spill 9-12 registers in kernels that do not use it (N = 2048: 78 with it).
#ifndef KSA_PP
#define KSA_PP 1
#endif
  constexpr bool PP = KSA_PP && PF && RM == 8 && N == 4096;
  // (General path, RM == 0: letting the raw-sample registers take the NEXT round's 16 loads as soon as a round has converted
  //  them was measured at N = 64, round 5: 168 instead of 121 VGPRs = three instead of four waves per SIMD, config 4 14.9 vs
  //  17.7 G FFT/s (-16 %; uint8 -15 %); held to 128 VGPRs the same code spills 98-110 registers, with the 6-twiddle last pass
  //  as well: profiles/r05_ab_pf0.txt.  Removed.)
  for (int vf = blockIdx.x; vf < p.nframes * NP; vf += gridDim.x) {
    const int frame = vf / NP, part = vf - frame * NP;
    // this workgroup's contiguous share of the frame's windows (contiguous keeps the sample reuse valid)
    const int k_lo = (int)((long long)p.nwin * part / NP), k_hi = (int)((long long)p.nwin * (part + 1) / NP);
    const int rounds = (k_hi - k_lo + S - 1) / S;
    float acc[16];
    const float init = p.cumu == CUMU_MIN ? __builtin_inff() : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = init;

    if constexpr (PP) {
      for (int rd0 = 0; rd0 < rounds; rd0 += 2) {
        {
          const int rd = rd0;
          constexpr int PAR = 0;
#include "ksa_window_body.inc"
        }
        if (rd0 + 1 < rounds) {
          const int rd = rd0 + 1;
          constexpr int PAR = 1;
#include "ksa_window_body.inc"
        }
      }
    } else {
      for (int rd = 0; rd < rounds; ++rd) {
        constexpr int PAR = 0;
#include "ksa_window_body.inc"
      }
    }

    // ---- combine the slots, scale, fftshift, dB, waterfall row ------------------------------
    // register position i of thread (slot,l) is bin l + L*perm16(i)   (M == 1: N == 16, L == 1).
    // The fold result goes through LDS once per frame so that the (rolled, branchy) output stage
    // does not share registers with the transform loop.
    float* const red = reinterpret_cast<float*>(lds);  // [S][N] floats, inside the data region
    __syncthreads();
    KSA_STAMP(9);    // output stage, part 1: the barrier behind the last window (slowest wave, outstanding loads)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[slot * RedStride<N, S>::value + l + L * perm<16>(i)] = acc[i];
    __syncthreads();
    KSA_STAMP(10);   // part 2: fold -> LDS staging + barrier
#ifdef KSA_ABL_NOFIN   // timing-only ablation build: one store per thread keeps the fold alive
    if (red[tid] == 123.456f) p.out[tid] = red[tid];
#else
    if (NP == 1) {
      finish_frame<N, T, S, CM>(p, red, frame, tid);
    } else {
      // partial fold of this share, slots combined, natural bin order; combine_parts_kernel finishes the frame
      float4* const dst = reinterpret_cast<float4*>(p.part_out + (long long)vf * N);
      const float4* red4 = reinterpret_cast<const float4*>(red);
      for (int q = tid; q < N / 4; q += T) {
        float4 r = red4[q];
        if constexpr (S > 1) {
          for (int s2 = 1; s2 < S; ++s2) {
            const float4 x = red4[s2 * (RedStride<N, S>::value / 4) + q];
            if (p.cumu == CUMU_AVG) { r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w; }
            else if (p.cumu == CUMU_MAX) { r.x = nan_max(r.x, x.x); r.y = nan_max(r.y, x.y); r.z = nan_max(r.z, x.z); r.w = nan_max(r.w, x.w); }
            else { r.x = nan_min(r.x, x.x); r.y = nan_min(r.y, x.y); r.z = nan_min(r.z, x.z); r.w = nan_min(r.w, x.w); }
          }
        }
        dst[q] = r;
      }
    }
#endif
    KSA_STAMP(8);
  }
#ifdef KSA_STAMPS
  if (p.dbg && (tid & 63) == 0) {
    for (int i = 0; i < 12; ++i) p.dbg[((long long)blockIdx.x * (T / 64) + tid / 64) * 12 + i] = seg[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Window-split (latency) mode: a frame's windows were folded by `parts` workgroups; this combines their
// partial folds in part order and applies the same scale / fftshift / dB as finish_frame.  One thread =
// 4 consecutive bins, blockIdx.y = frame.  Waterfall rows follow from rowmax_batch.
__global__ void combine_parts_kernel(const SpecParams p, int n) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q * 4 >= n) return;
  const int frame = blockIdx.y;
  const float4* src = reinterpret_cast<const float4*>(p.part_out + (long long)frame * p.parts * n) + q;
  float4 r = src[0];
  for (int part = 1; part < p.parts; ++part) {
    const float4 x = src[(long long)part * (n / 4)];
    if (p.cumu == CUMU_AVG) { r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w; }
    else if (p.cumu == CUMU_MAX) { r.x = nan_max(r.x, x.x); r.y = nan_max(r.y, x.y); r.z = nan_max(r.z, x.z); r.w = nan_max(r.w, x.w); }
    else { r.x = nan_min(r.x, x.x); r.y = nan_min(r.y, x.y); r.z = nan_min(r.z, x.z); r.w = nan_min(r.w, x.w); }
  }
  float o[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float lin = p.cumu == CUMU_AVG ? o[u] : __builtin_amdgcn_sqrtf(o[u]);
    lin *= p.scale;
    o[u] = p.out_mode != OUT_LINEAR ? out_db(lin, p.out_mode, p.gain, p.min_amp) : lin;
  }
  const int sh = (4 * q + n / 2) & (n - 1);
  *reinterpret_cast<float4*>(p.out + (long long)frame * n + sh) = make_float4(o[0], o[1], o[2], o[3]);
}

// Waterfall cells (K:480) of finished dB rows: blockIdx.y = frame of the batch.  Used where the cell reduction does not
// ride on the output stage: the window-split (latency) mode and cells wider than the large-transform finish tile.
__global__ void rowmax_batch(const SpecParams p, int n) {
  const int frame = blockIdx.y;
  const int g = n / p.hm_w;
  const float* row = p.out + (long long)frame * n;
  for (int cell = blockIdx.x * blockDim.x + threadIdx.x; cell < p.hm_w; cell += gridDim.x * blockDim.x) {
    float hv = -__builtin_inff();
    bool nan = false;
    for (int i = 0; i < g; ++i) {
      float v = row[cell * g + i];
      if (p.adj) v -= p.adj[cell * g + i];
      nan |= v != v;
      hv = fmaxf(hv, v);
    }
    if (nan) hv = __builtin_nanf("");
    if (p.hm_rows) p.hm_rows[(long long)frame * p.hm_w + cell] = hv;
    if (p.hm_ring && frame >= p.hm_first) p.hm_ring[((p.hm_index0 + frame) % HM_ROWS) * p.hm_w + cell] = hv;
  }
}

// ------------------------------------------------------------------------------------------------
// zeroSpanPlay: dB + waterfall row of spectra that are already linear magnitudes (K:469, K:480)
struct DbRowParams {
  const float* lin;  // [nframes][N]
  float* out;        // [nframes][N] dB
  int n, nframes;
  float gain;
  int hm_w;
  const float* adj;
  float* hm_rows;
  float* hm_ring;
  int hm_index0, hm_first;
};

__global__ void db_rows_kernel(const DbRowParams p) {
  const int frame = blockIdx.y;
  const int g = p.hm_w > 0 ? p.n / p.hm_w : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x)
    p.out[(long long)frame * p.n + i] = db_of(p.lin[(long long)frame * p.n + i], p.gain);
  if (g == 0) return;
  for (int cell = blockIdx.x * blockDim.x + threadIdx.x; cell < p.hm_w; cell += gridDim.x * blockDim.x) {
    float hv = -__builtin_inff();
    bool nan = false;
    for (int i = 0; i < g; ++i) {
      const int b = cell * g + i;
      float v = db_of(p.lin[(long long)frame * p.n + b], p.gain);
      if (p.adj) v -= p.adj[b];
      nan |= v != v;
      hv = fmaxf(hv, v);
    }
    if (nan) hv = __builtin_nanf("");
    if (p.hm_rows) p.hm_rows[(long long)frame * p.hm_w + cell] = hv;
    if (p.hm_ring && frame >= p.hm_first)
      p.hm_ring[((p.hm_index0 + frame) % HM_ROWS) * p.hm_w + cell] = hv;
  }
}

// ------------------------------------------------------------------------------------------------
// Row A10.  Max / Min are elementwise; Avg is the closed form of the reference's EMA:
// after frames 0..n:  avg = x0*2^-n + sum_{k>=1} x_k*2^-(n-k+1)   (K:137-139 applied at K:476).
// Non-finite terms (the -inf of a zero magnitude, K:469) poison the sum whatever their weight,
// as they do in the recursion.
struct AccParams {
  const float* db;   // [nframes][N]
  int n, nframes;
  long long first_index, total_frames;  // position of this batch in the logical run
  int has_prev;      // Fft.Avg already holds frames (it is seeded by copy otherwise: data_cumu(None), K:133-134)
  int chunk;         // frames per partial
  float* part;       // [chunks][3][N] : max, min, sum
};

// One thread = 4 consecutive bins (16-byte loads), one blockIdx.y = one chunk of frames.
__global__ void accumulate_partial_kernel(const AccParams p) {
  const int b4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (b4 * 4 >= p.n) return;
  const int c = blockIdx.y;
  const int f0 = c * p.chunk;
  const int f1 = min(p.nframes, f0 + p.chunk);
  const float ninf = -__builtin_inff(), pinf = __builtin_inff();
  float mx[4] = {ninf, ninf, ninf, ninf}, mn[4] = {pinf, pinf, pinf, pinf}, sum[4] = {0.f, 0.f, 0.f, 0.f};
  const float4* src = reinterpret_cast<const float4*>(p.db) + b4;
  const long long stride4 = p.n / 4;
#pragma unroll 4
  for (int f = f0; f < f1; ++f) {
    const float4 x4 = src[(long long)f * stride4];
    const long long kg = p.first_index + f;
    long long e = p.total_frames - kg;            // 2^-(n-k+1) with n = total-1
    if (kg == 0 && !p.has_prev) e = p.total_frames - 1;
    const float w = e > 160 ? 0.f : ldexpf(1.0f, -(int)e);
    const float x[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      mx[u] = nan_max(mx[u], x[u]);
      mn[u] = nan_min(mn[u], x[u]);
      sum[u] += fabsf(x[u]) < pinf ? w * x[u] : x[u];
    }
  }
  float4* o = reinterpret_cast<float4*>(p.part + (long long)c * 3 * p.n) + b4;
  o[0] = make_float4(mx[0], mx[1], mx[2], mx[3]);
  o[stride4] = make_float4(mn[0], mn[1], mn[2], mn[3]);
  o[2 * stride4] = make_float4(sum[0], sum[1], sum[2], sum[3]);
}

// partial block layout: [max | cur-or--inf | -min | sum], N floats each.
// Workgroup = 64 bins x 16 chunk groups (1024 threads): every thread folds chunks/16 partials, the 16
// groups are combined through LDS in chunk order, so the short dependent chains run in parallel.
__global__ __launch_bounds__(1024) void accumulate_reduce_kernel(const float* part, int chunks, int n, const float* last_db,
                                                                 int owns_last, float* partial) {
  __shared__ float sh[3][16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int bin = blockIdx.x * 64 + lane;
  float mx = -__builtin_inff(), mn = __builtin_inff(), sum = 0.f;
  const int per = (chunks + 15) / 16;
  const int c0 = grp * per, c1 = min(chunks, c0 + per);
  if (bin < n) {
    for (int c = c0; c < c1; ++c) {
      const float* o = part + (long long)c * 3 * n;
      mx = nan_max(mx, o[bin]);
      mn = nan_min(mn, o[n + bin]);
      sum += o[2 * n + bin];
    }
  }
  sh[0][grp][lane] = mx;
  sh[1][grp][lane] = mn;
  sh[2][grp][lane] = sum;
  __syncthreads();
  if (grp == 0 && bin < n) {
    for (int g = 1; g < 16; ++g) {
      mx = nan_max(mx, sh[0][g][lane]);
      mn = nan_min(mn, sh[1][g][lane]);
      sum += sh[2][g][lane];
    }
    partial[bin] = mx;
    partial[n + bin] = owns_last ? last_db[bin] : -__builtin_inff();
    partial[2 * n + bin] = -mn;   // negated: one MAX all-reduce then covers max | cur | -min
    partial[3 * n + bin] = sum;
  }
}

// Multi-GPU merge (no reference counterpart; algebra of SURVEY.md 8(e)): `g` is the all-gather of every rank's
// exchange block [4][N] partial | [128][W] ring, `stride` floats apart, in rank order.  Thread i < N reduces the
// partial rows (max | cur | -min by NaN-propagating max, the weighted sum in rank order: deterministic, the same
// bits on every rank); thread c < 128*W picks ring cell c from the rank whose chunk holds the newest frame that
// maps to that ring slot (global frame f lives in slot (idx0 + f) % 128; rank r holds frames [r*fpr, (r+1)*fpr)).
__global__ void merge_gathered_kernel(const float* g, int world, long long stride, int n, float* partial, float* hm,
                                      int hm_w, int idx0, int fpr, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float mx = g[i], cur = g[n + i], nmn = g[2 * n + i], sum = g[3 * n + i];
    for (int r = 1; r < world; ++r) {
      const float* s = g + r * stride;
      mx = nan_max(mx, s[i]);
      cur = nan_max(cur, s[n + i]);
      nmn = nan_max(nmn, s[2 * n + i]);
      sum += s[3 * n + i];
    }
    partial[i] = mx; partial[n + i] = cur; partial[2 * n + i] = nmn; partial[3 * n + i] = sum;
  }
  if (hm && i < (long long)HM_ROWS * hm_w) {
    const int slot = (int)(i / hm_w);
    const long long last = total - 1;
    const long long back = ((idx0 + last - slot) % HM_ROWS + HM_ROWS) % HM_ROWS;
    const long long f = last - back;      // newest frame of this run stored in `slot`, or < 0: none
    if (f >= 0) hm[i] = g[(f / fpr) * stride + 4ll * n + i];
  }
}

// state layout: [cur | max | min | avg].  has_max / has_min / has_avg: that curve already holds frames; a curve
// that was switched off so far (bDataMax/Min/Avg, K:471-476) is still None in the reference and is seeded by copy
// the first time its flag is on (data_cumu(None) -> np.copy, K:133-134) -- per curve, not per engine.
__global__ void commit_kernel(const float* partial, float* state, int n, int has_max, int has_min, int has_avg,
                              long long total_frames, int b_max, int b_min, int b_avg) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= n) return;
  const float mx = partial[bin], cur = partial[n + bin], mn = -partial[2 * n + bin], sum = partial[3 * n + bin];
  state[bin] = cur;
  if (b_max) state[n + bin] = has_max ? nan_max(state[n + bin], mx) : mx;
  if (b_min) state[2 * n + bin] = has_min ? nan_min(state[2 * n + bin], mn) : mn;
  if (b_avg) {
    float a = sum;
    if (has_avg) {
      const float prev = state[3 * n + bin];
      const float w = total_frames > 160 ? 0.f : ldexpf(1.0f, -(int)total_frames);
      a += (fabsf(prev) < __builtin_inff()) ? prev * w : prev;
    }
    state[3 * n + bin] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// Row A13.  Element e of the stitched range is covered by the steps i with i*hop <= e < i*hop + N.
// The reference copies the first covering step (RAW, K:644) and halves-in every later one (AVG,
// K:649); after the last covering step the value is final and is what Max/Min/Avg see (K:657-668).
struct StitchParams {
  const float* step_db;  // [npasses][own_steps][N]: the bands [step_lo, step_lo + own_steps) of every pass
  const float* halo_db;  // [nhalo][npasses][N] (band major): bands [step_lo - nhalo, step_lo), or null
  int n, nsteps, hop, total;
  int step_lo, own_steps, nhalo;   // single engine: 0, nsteps, 0
  int own_band_major;              // the own block is [own_steps][npasses][N] (one strided spectrum launch per band) instead of [npasses][own_steps][N]
  int e_lo, e_hi;                  // elements of the stitched range this launch owns (single engine: 0, total)
  float* state;          // [4][total] : cur, max, min, avg
  int first_pass;        // the first pass of this call is pass 0 of the run: it seeds Avg by copy (K:615-618)
  int b_max, b_min;
  int base_is_raw;       // bScanRangeBaseDataIsRaw (K:651-656): Max/Min/Avg see every covering step's own spectrum
  int npasses;           // passes folded by this call, in order (a batch of captured passes resident in HBM)
  float* avg_rows;       // [npasses - avg_row0][total] Fft.Avg after each of the last passes (waterfall source), or null
  int avg_row0;
};

// One thread = one element of the stitched range, walking the passes of the batch in order: exactly the
// sequence of updates the reference applies pass after pass, with the state in registers in between.
// A band-sharded scan launches it over the elements one engine owns; bands in front of its own ones that still
// cover those elements come from the halo block its left neighbours sent.
__global__ __launch_bounds__(256) void scan_stitch_kernel(const StitchParams p) {
  const int e = p.e_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.e_hi) return;
  int i0 = (e - p.n + p.hop) / p.hop;  // ceil((e-n+1)/hop) for e-n+1 > 0
  if (e - p.n + 1 <= 0) i0 = 0;
  int i1 = e / p.hop;
  if (i1 > p.nsteps - 1) i1 = p.nsteps - 1;
  const bool covered = i0 <= i1;       // not covered: the state keeps its value
  const int tot = p.total;
  float cur = p.state[e], mx = p.state[tot + e], mn = p.state[2 * tot + e], av = p.state[3 * tot + e];
  // spectrum of band i in pass 0 and the distance between passes (own block or halo block)
  auto band = [&](int i, long long& pass_stride) -> const float* {
    if (i >= p.step_lo) {
      if (p.own_band_major) {
        pass_stride = p.n;
        return p.step_db + (long long)(i - p.step_lo) * p.npasses * p.n;
      }
      pass_stride = (long long)p.own_steps * p.n;
      return p.step_db + (long long)(i - p.step_lo) * p.n;
    }
    pass_stride = p.n;
    return p.halo_db + (long long)(i - (p.step_lo - p.nhalo)) * p.npasses * p.n;
  };
  // stitched value of this element in pass ps (K:643-650): the first covering step raw, every later one halved in
  auto stitched = [&](int ps) {
    long long st;
    const float* b0 = band(i0, st);
    float c = b0[ps * st + (e - i0 * p.hop)];
    for (int i = i0 + 1; i <= i1; ++i) {
      const float* bi = band(i, st);
      c = (c + bi[ps * st + (e - i * p.hop)]) * 0.5f;
    }
    return c;
  };
  constexpr int AHEAD = 32;  // passes whose loads are in flight per thread (a batch has few elements but many passes)
  float nxt[AHEAD];
  const int ncover = i1 - i0 + 1;                       // steps covering this element: 1 or 2 at the usual hop of N/2
  long long sa = 0, sb = 0;
  const float* pa = p.step_db;
  const float* pb = p.step_db;
  if (covered) {
    pa = band(i0, sa) + (e - i0 * p.hop);
    if (ncover >= 2) pb = band(i0 + 1, sb) + (e - (i0 + 1) * p.hop);
  }
  for (int ps0 = 0; ps0 < p.npasses; ps0 += AHEAD) {
    if (covered && !p.base_is_raw) {
      if (ncover <= 2) {
        // straight-line loads (no inner loop): all 2*AHEAD are in flight before the first is used
        // (running pointers: the distance between passes differs between the own block and the halo block, so it is
        //  a per-thread value -- one 64-bit add per load instead of a 64-bit multiply.  A whole group of AHEAD passes
        //  -- every group but the batch's last -- takes the branch-free form.)
        float a[AHEAD], b[AHEAD];
        if (ps0 + AHEAD <= p.npasses) {
#pragma unroll
          for (int u = 0; u < AHEAD; ++u) { a[u] = *pa; pa += sa; }
          if (ncover == 2) {
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) { b[u] = *pb; pb += sb; }
          } else {
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) b[u] = 0.f;
          }
        } else {
#pragma unroll
          for (int u = 0; u < AHEAD; ++u) {
            const bool in = ps0 + u < p.npasses;
            a[u] = in ? *pa : 0.f;
            b[u] = in && ncover == 2 ? *pb : 0.f;
            pa += sa;
            pb += sb;
          }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) nxt[u] = ncover == 2 ? (a[u] + b[u]) * 0.5f : a[u];
      } else {
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) nxt[u] = ps0 + u < p.npasses ? stitched(ps0 + u) : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const int ps = ps0 + u;
      if (covered && ps < p.npasses) {   // (no break: the loop must unroll fully, or nxt[] goes to scratch memory)
        const bool first = p.first_pass && ps == 0;
        if (!p.base_is_raw) {
          cur = nxt[u];
          if (p.b_max) mx = nan_max(mx, cur);
          if (p.b_min) mn = nan_min(mn, cur);
          av = first ? cur : (av + cur) * 0.5f;
        } else {
          // every covering step, in order, folds its own spectrum in (pass 0: Avg is overwritten by each step)
          cur = stitched(ps);
          for (int i = i0; i <= i1; ++i) {
            long long st;
            const float* bi = band(i, st);
            const float x = bi[ps * st + (e - i * p.hop)];
            if (p.b_max) mx = nan_max(mx, x);
            if (p.b_min) mn = nan_min(mn, x);
            av = first ? x : (av + x) * 0.5f;
          }
        }
      }
      if (p.avg_rows && ps >= p.avg_row0 && ps < p.npasses) p.avg_rows[(long long)(ps - p.avg_row0) * tot + e] = av;
    }
  }
  if (covered) {
    p.state[e] = cur;
    p.state[tot + e] = mx;
    p.state[2 * tot + e] = mn;
    p.state[3 * tot + e] = av;
  }
}

// Plot-side decimation on the device (data_plotcompress / _data_plotcompress, K:168-221): curve c of `src`
// ([ncurves][n]) is cut into `cells` groups of g = n/cells bins and reduced with AVG / MAX / MIN, minus the
// optional Fft.Adj baseline (K:400-411), so only cells-sized arrays cross PCIe for the Levels plot.
// mode: 0 AVG, 1 MAX, 2 MIN.  blockIdx.y = curve.
__global__ void levels_kernel(const float* src, const float* adj, int n, int cells, int mode, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  const int g = n / cells;
  const float* row = src + (long long)blockIdx.y * n + (long long)c * g;
  const float* arow = adj ? adj + (long long)c * g : nullptr;
  float acc = mode == 0 ? 0.f : (mode == 1 ? -__builtin_inff() : __builtin_inff());
  double sum = 0.0;   // np.average accumulates in float64 (pairwise); float64 here keeps the order irrelevant
  bool nan = false;
  for (int i = 0; i < g; ++i) {
    float v = row[i];
    if (arow) v -= arow[i];
    nan |= v != v;
    if (mode == 0) sum += (double)v;
    else if (mode == 1) acc = fmaxf(acc, v);
    else acc = fminf(acc, v);
  }
  if (mode == 0) acc = (float)(sum / (double)g);
  out[(long long)blockIdx.y * cells + c] = nan ? __builtin_nanf("") : acc;
}

// Peak markers of plot_highs (K:243-272) on the device.  The reference walks the points of the plotted curve from
// the highest level down (ascending argsort read backwards, K:253-258) and marks a point unless an already marked
// one lies closer than delta4Marking along the frequency axis (K:261-262), until numMarkers are marked (K:268).
// That greedy walk is "count" rounds of: arg-max over the points not excluded by the marks so far.  The axis of the
// plotted curve is uniform, so the spacing rule is |c - m| < min_sep in cells (min_sep = delta / cell width, host
// computed).  Two details of the reference are kept: NaN sorts above +inf (numpy's argsort puts NaN last, so the
// walk meets it first), and the walk stops one short of the lowest point (K:258: i = -1 .. -(len-1)), which is
// therefore never marked.  Equal levels: the higher index first.  One workgroup; lv = [cells] decimated curve.
struct HighsParams {
  const float* lv;
  int cells;
  double min_sep;
  int count;        // <= HIGHS_MAX
  int* idx;         // [count] marked cell, in marking order
  float* lvl;       // [count]
  int* found;       // number marked (< count when the curve runs out of eligible points)
};
constexpr int HIGHS_MAX = 64;

__device__ __forceinline__ unsigned order_bits(float v) {
  if (v != v) return 0xffffffffu;                       // NaN above everything
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);    // monotone map of the float order onto unsigned
}

__device__ __forceinline__ unsigned long long wg_reduce_max(unsigned long long k, unsigned long long* sh, bool want_min) {
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned hi = __shfl_xor((unsigned)(k >> 32), m), lo = __shfl_xor((unsigned)k, m);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    k = want_min ? (o < k ? o : k) : (o > k ? o : k);
  }
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wave] = k;
  __syncthreads();
  unsigned long long r = sh[0];
  for (int w = 1; w < nw; ++w) r = want_min ? (sh[w] < r ? sh[w] : r) : (sh[w] > r ? sh[w] : r);
  return r;
}

__global__ __launch_bounds__(1024) void highs_kernel(const HighsParams p) {
  __shared__ unsigned long long sh[16];
  __shared__ int marks[HIGHS_MAX];
  const int tid = threadIdx.x;
  // the lowest point (lowest index among equals; NaN never counts as low): not part of the walk
  unsigned long long kmin = ~0ull;
  for (int c = tid; c < p.cells; c += blockDim.x) {
    const unsigned long long k = ((unsigned long long)order_bits(p.lv[c]) << 32) | (unsigned)c;
    kmin = k < kmin ? k : kmin;
  }
  const int lowest = (int)(unsigned)wg_reduce_max(kmin, sh, true);
  int nm = 0;
  for (; nm < p.count; ++nm) {
    unsigned long long best = 0;   // 0 = nothing eligible (real keys are stored +1)
    for (int c = tid; c < p.cells; c += blockDim.x) {
      if (c == lowest) continue;
      bool free_ = true;
      for (int j = 0; j < nm; ++j) free_ &= c != marks[j] && fabs((double)(c - marks[j])) >= p.min_sep;   // (each point is visited once)
      if (!free_) continue;
      const unsigned long long k = (((unsigned long long)order_bits(p.lv[c]) << 32) | (unsigned)c) + 1ull;
      best = k > best ? k : best;
    }
    best = wg_reduce_max(best, sh, false);
    if (best == 0) break;
    const int c = (int)(unsigned)(best - 1ull);
    if (tid == 0) { marks[nm] = c; p.idx[nm] = c; p.lvl[nm] = p.lv[c]; }
    __syncthreads();
  }
  if (tid == 0) *p.found = nm;
}

__global__ void fill_kernel(float* dst, long long n, float v) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}

// Scan waterfall rows of a batch of passes (K:696-697): row r of `src` ([rows][cells*g], Fft.Avg after a pass)
// -> ring row (row0 + r) % 128.  One wave per cell (lanes stride over the cell's g bins: coalesced), blockIdx.y = r.
__global__ __launch_bounds__(64) void rowmax_rows_kernel(const float* src, const float* adj, int cells, int g, float* ring, int row0) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const float* row = src + (long long)blockIdx.y * cells * g + (long long)c * g;
  const float* arow = adj ? adj + (long long)c * g : nullptr;
  float hv = -__builtin_inff();
  int nan = 0;
  for (int i = lane; i < g; i += 64) {
    float v = row[i];
    if (arow) v -= arow[i];
    nan |= v != v;
    hv = fmaxf(hv, v);
  }
  for (int m = 32; m >= 1; m >>= 1) {
    hv = fmaxf(hv, __shfl_xor(hv, m));
    nan |= __shfl_xor(nan, m);
  }
  if (lane == 0) ring[(long long)((row0 + blockIdx.y) % HM_ROWS) * cells + c] = nan ? __builtin_nanf("") : hv;
}

// Band-sharded scan: the same row reduction over the elements [e_lo, e_hi) this engine owns only -- a PARTIAL row
// (-inf where the engine owns nothing of a cell); scan_merge_rows_kernel combines the engines' partial rows.
__global__ __launch_bounds__(64) void rowmax_rows_range_kernel(const float* src, const float* adj, int cells, int g, int e_lo,
                                                               int e_hi, float* rows) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const long long total = (long long)cells * g;
  const float* row = src + (long long)blockIdx.y * total;
  const int lo = max(c * g, e_lo), hi = min((c + 1) * g, e_hi);
  float hv = -__builtin_inff();
  int nan = 0;
  for (int i = lo + lane; i < hi; i += 64) {
    float v = row[i];
    if (adj) v -= adj[i];
    nan |= v != v;
    hv = fmaxf(hv, v);
  }
  for (int m = 32; m >= 1; m >>= 1) {
    hv = fmaxf(hv, __shfl_xor(hv, m));
    nan |= __shfl_xor(nan, m);
  }
  if (lane == 0) rows[(long long)blockIdx.y * cells + c] = nan ? __builtin_nanf("") : hv;
}

// gathered = [world][rows][cells] partial rows in rank order -> ring row (row0 + r) % 128, NaN-propagating max
__global__ void scan_merge_rows_kernel(const float* gathered, int world, int rows, int cells, float* ring, int row0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cells) return;
  float v = gathered[i];
  for (int w = 1; w < world; ++w) v = nan_max(v, gathered[(long long)w * rows * cells + i]);
  const int r = i / cells, c = i - r * cells;
  ring[(long long)((row0 + r) % HM_ROWS) * cells + c] = v;
}

// out[c] = max_{i<g} (src[c*g+i] - adj[c*g+i])
__global__ void rowmax_kernel(const float* src, const float* adj, int cells, int g, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  float hv = -__builtin_inff();
  bool nan = false;
  for (int i = 0; i < g; ++i) {
    float v = src[(long long)c * g + i];
    if (adj) v -= adj[(long long)c * g + i];
    nan |= v != v;
    hv = fmaxf(hv, v);
  }
  out[c] = nan ? __builtin_nanf("") : hv;
}

}  // namespace ksa
