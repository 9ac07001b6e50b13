// spectrum64_kernel: the 64-point transform (quickFullScan's fftSize, K:916-921) as 8 x 8 with ADJACENT samples per lane.
//
// Same rows of SURVEY.md section 8 as spectrum_kernel (A4-A9, A12; replaces numpy.fft.fft + the fold at
// python/kspecanal.py:385-396) and the same workgroup shape (one wave = 16 transforms of 4 lanes, 16 points per lane), but a
// different split of the DFT.  spectrum_kernel<64> is 4 x 16: lane l owns samples l + 4q and needs sixteen 8-byte loads per
// round whose quads point at 16 different places; the cycle stamps put half of a wave's time into issuing and awaiting them
// (profiles/r05_c4_stamps_issue.txt), and tools/ta_probe.hip prices that shape at 2.48 us per round per CU against 1.70 us
// for eight 16-byte loads of adjacent samples (profiles/r05_ta_probe.txt).  Here lane l owns samples n = 8j + 2l + c
// (j = 0..7, c = 0,1 -- each load is two adjacent samples), and with k = k1 + 8 k2:
//
//   X[k1 + 8 k2] = sum_m W8^(m k2) * { W64^(m k1) * sum_j x[8j + m] W8^(j k1) },   m = 2l + c
//
//   A  two radix-8 butterflies per lane over j (no twiddles, window multiply in front),
//   B  the 14 twiddles W64^(m k1) of a lane (per-lane constants, float64-generated table),
//   -- one exchange through LDS inside the slot (element (k1, m) at k1*8 + (m ^ k1): the XOR keeps both sides off each
//      other's banks without padding rows) --
//   C  two radix-8 butterflies per lane over m (no twiddles): lane l' ends with bins (2l' + c') + 8 k2.
//
// Measured against spectrum_kernel<64> in one build (profiles/r05_ab_k64.txt): config 4 +3.6 ... +5.6 % on four boxes, N = 64 at
// hops 0.5 / 0.25 +1 ... +10 %.  Variants on top, measured and removed: 3 waves per SIMD without spills = the old kernel's speed;
// all 14 twiddles in VGPRs (19 spilled registers at the 128-VGPR cap of four waves per SIMD) -0.5 %; taps as four ds_read_b128
// up front -0.8 %; the exchange as 8 + 8 16-byte operations on plain rows +-0; odd slots one element later (conflict-free in
// the bank model; the kept layout shows a conflict ratio of 0.41 in the counters) -0.8 %: the conflicts are not on a wave's
// critical path.  Where a wave's time goes (-DKSA_STAMPS, profiles/r05_c4_stamps_k64.txt): load issue 22 %, wait + taps +
// multiply 16 %, A + B 12 %, exchange 18 %, reads + C 10 %, fold 9 %, output stage 13 %.
// uint8 input keeps spectrum_kernel<64> (its two-sample piece is a 4-byte load at 2-byte alignment).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ksa_kernels.hpp"

namespace ksa {

struct Plan64 {
  static constexpr int N = 64, T = 64, S = 16, L = 4;
  static constexpr int XS = 68;                            // complex elements per slot (64 + 4: slots off each other's banks)
  static constexpr int LDS_BYTES = S * XS * 8 + N * 4;     // exchange / staging rows + taps
};

// W1: the window table is all ones -- the reference's default (gWindow = WINDOW_ONES, K:52) and what quickFullScan runs with: x * 1.0f is x
// bit for bit, so the 32 multiplies and the tap reads of a round are left out (the host checks the table it uploads).
template <int FMT, int CM, bool W1>
__global__ __launch_bounds__(64, 4) void spectrum64_kernel(const SpecParams p) {
  static_assert(FMT == FMT_C64, "adjacent-sample loads: complex64 input (16-byte loads of two samples)");
  using P = Plan64;
  constexpr int N = P::N, T = P::T, S = P::S;
  constexpr int SB = 8;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float* const taps_lds = reinterpret_cast<float*>(lds + S * P::XS);
  const int tid = threadIdx.x, slot = tid >> 2, l = tid & 3;
  float2* const my = lds + slot * P::XS;

  // taps in the lanes' load order: lane l' needs w[8j + 2l' + c] for j = 0..7, c = 0,1 -> taps_lds[l'*16 + 2j + c]
  if constexpr (!W1) {
    const int n = tid, lp = (n & 7) >> 1, c = n & 1, j = n >> 3;
    taps_lds[lp * 16 + 2 * j + c] = p.window[n];
  }
  // step B: W64^(m k1), m = 2l + c, k1 = 1..7 (p.tw_mid = [m][k1], generated in float64 on the host)
  // only the EVEN samples' twiddles W64^(2l k1) live in VGPRs; the odd samples' are those times the constants W64^k1 below
  float2 twb[7];
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) twb[k1 - 1] = p.tw_mid[(2 * l) * 8 + k1];
  __syncthreads();

  const int nm1 = p.nwin - 1;
  const int rounds = (p.nwin + S - 1) / S;
#ifdef KSA_STAMPS
  unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#endif
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  for (int frame = blockIdx.x; frame < p.nframes; frame += gridDim.x) {
    float acc[16];
    const float init = p.cumu == CUMU_MIN ? __builtin_inff() : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = init;
    const char* fbase = reinterpret_cast<const char*>(p.iq) + (long long)frame * p.frame_stride * SB;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(fbase), 0, p.frame_len * SB, 0x00020000);

    for (int rd = 0; rd < rounds; ++rd) {
      const int k = rd * S + slot;
      const bool active = k < p.nwin;
      float2 v[16];            // v[c*8 + j] = x[8j + 2l + c] * w
      if (active) {
        const int voff = (p.starts[k] + 2 * l) * SB;
        u32x4 piece[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) piece[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 8 * j * SB, 0);
        KSA_STAMP(11);
#pragma unroll
        for (int j = 0; j < 8; ++j) {    // the two taps of a piece are read (ds_read_b64) right where the piece is converted
          const float2 w2 = W1 ? make_float2(1.0f, 1.0f) : reinterpret_cast<const float2*>(taps_lds)[l * 8 + j];
          const unsigned a = piece[j].x, b = piece[j].y, c = piece[j].z, d = piece[j].w;   // (scalar copies: see spectrum_kernel)
          if constexpr (W1) {
            v[j] = make_float2(__uint_as_float(a), __uint_as_float(b));
            v[8 + j] = make_float2(__uint_as_float(c), __uint_as_float(d));
          } else {
            v[j] = make_float2(__uint_as_float(a) * w2.x, __uint_as_float(b) * w2.x);
            v[8 + j] = make_float2(__uint_as_float(c) * w2.y, __uint_as_float(d) * w2.y);
          }
        }
        KSA_STAMP(0);
        dft8<0>(v);            // position c*8 + P holds y_m[perm8(P)]
        dft8<8>(v);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int Pp = 1; Pp < 8; ++Pp) {
            constexpr float cw[8] = {1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f,
                                     0.88192126434835502971f, 0.83146961230254523708f, 0.77301045336273696081f};
            constexpr float sw[8] = {0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f, 0.38268343236508977173f,
                                     0.47139673682599764856f, 0.55557023301960222474f, 0.63439328416364549822f};
            float2 y = v[c * 8 + Pp];
            if (c == 1) y = cmul(y, make_float2(cw[perm<8>(Pp)], -sw[perm<8>(Pp)]));      // W64^k1
            v[c * 8 + Pp] = cmul(y, twb[perm<8>(Pp) - 1]);
          }
      }
      KSA_STAMP(1);
      __syncthreads();         // (single-wave workgroup: orders the previous round's reads before these stores)
      if (active) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int Pp = 0; Pp < 8; ++Pp) {
            const int k1 = perm<8>(Pp);
            my[k1 * 8 + ((2 * l + c) ^ k1)] = v[c * 8 + Pp];
          }
      }
      __syncthreads();
      KSA_STAMP(3);
      if (active) {
        float2 u[16];          // u[c'*8 + m] = z_m[k1], k1 = 2l + c'
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int m = 0; m < 8; ++m) u[c * 8 + m] = lds_ld64(&my[(2 * l + c) * 8 + (m ^ (2 * l + c))]);

        dft8<0>(u);            // position c'*8 + P holds X[(2l + c') + 8*perm8(P)]
        dft8<8>(u);
        KSA_STAMP(6);
        const int cm = CM == 0 ? p.cumu : CM;
        if (cm == CUMU_AVG) {
          const int e = k == 0 ? nm1 : nm1 - k + 1;          // closed form of the (a+x)/2 recursion (K:137-139 at K:395)
          const float wgt = ldexpf(1.0f, -e);
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[i] = fmaf(wgt, __builtin_amdgcn_sqrtf(fmaf(u[i].x, u[i].x, u[i].y * u[i].y)), acc[i]);
        } else if (cm == CUMU_MAX) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = nan_max_nonneg(acc[i], fmaf(u[i].x, u[i].x, u[i].y * u[i].y));
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = nan_min(acc[i], fmaf(u[i].x, u[i].x, u[i].y * u[i].y));
        }
      }
      KSA_STAMP(7);
    }
    // natural bin order through LDS, then the common output stage: acc[c'*8 + P] is bin (2l + c') + 8*perm8(P)
    float* const red = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int Pp = 0; Pp < 8; ++Pp) red[slot * RedStride<N, S>::value + (2 * l + c) + 8 * perm<8>(Pp)] = acc[c * 8 + Pp];
    __syncthreads();
    KSA_STAMP(10);
    finish_frame<N, T, S, CM>(p, red, frame, tid);
    KSA_STAMP(8);
  }
#ifdef KSA_STAMPS
  if (p.dbg && tid == 0)
    for (int i = 0; i < 12; ++i) p.dbg[(long long)blockIdx.x * 12 + i] = seg[i];
#endif
}

}  // namespace ksa
