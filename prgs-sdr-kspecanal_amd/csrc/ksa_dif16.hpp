// Large transforms (N = 32768 .. 1048576) as radix-16 / 32 / 64 decimation in frequency in front of the tuned
// single-workgroup kernels (gfx950): radix 16 up to N = 262144 (second stage of N/16 <= 16384 points), radix 32 for
// 524288 and radix 64 for 1048576 (second stage: the 32-point-per-thread kernel at 16384 points).  Replaces numpy.fft.fft at python/kspecanal.py:391 for those fftSize values;
// same rows of SURVEY.md section 8 as ksa_kernels.hpp (A0, A4-A9, A12).
//
//   n = n1 + N1*q  (n1 < N1 = N/16, q < 16),   k = 16*k1 + k2
//   X[16*k1 + k2] = sum_n1 W_N1^(n1*k1) * { W_N^(n1*k2) * sum_q x[n1 + N1*q] W_16^(q*k2) }
//
//   dif16_kernel        per (frame, window, n1): window multiply, one radix-16 butterfly in registers over the 16
//                       samples N1 apart (lanes run along n1: every load and store is a coalesced 512-B run; no
//                       LDS, no barrier), twiddle W_N^(n1*k2), store Z[frame][k2][window][n1].
//   spectrum_kernel<N1> (ksa_kernels.hpp, unchanged) on the 16 "pseudo frames" (frame, k2): each is nwin
//                       back-to-back blocks of N1 points, transformed and folded over the windows exactly as a
//                       zeroSpan frame with hop N1 -> Y[frame][k2][k1] linear magnitudes.
//   dif16_finish_kernel interleave k = 16*k1 + k2 through an LDS tile, fftshift, dB / clip (K:100-112), store,
//                       waterfall cells (K:480).
//
// The intermediate Z (8 N bytes per window, written once and read once) is the price of a transform that does not
// fit one workgroup's LDS; frames are processed in chunks so that Z stays within a fixed scratch budget.
#pragma once
#include <hip/hip_runtime.h>

#include "ksa_kernels.hpp"

namespace ksa {

struct DifParams {
  const void* iq;            // float2[] or uchar2[]; frame f at sample f*frame_stride
  long long frame_stride;
  int frame0;                // first frame of this chunk
  int nwin;
  const int* starts;         // [nwin]
  const float* window;       // [N]
  const float2* tw;          // [6][N1]: W_N^(n1*e) for e = 1, 2, 3, 4, 8, 12 (radix 32 / 64: [9][N1], + e = 16, 32, 48)
  int n1;                    // N / radix
  float u8_offset, u8_inv_scale;
  float2* z;                 // [chunk_frames][16][nwin][n1]
};

// Two adjacent n1 per thread: 16-byte loads / stores per lane (1 KiB per wave-instruction), half the
// vector-memory instructions of the one-point form.
template <int FMT>
__global__ __launch_bounds__(256) void dif16_kernel(const DifParams p) {
  const int n1 = (blockIdx.x * 256 + threadIdx.x) * 2;
  const int w = blockIdx.y, fr = blockIdx.z;
  const int N1 = p.n1;
  const long long base = (long long)(p.frame0 + fr) * p.frame_stride + p.starts[w] + n1;
  float2 va[16], vb[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float2 wn = *reinterpret_cast<const float2*>(p.window + n1 + N1 * q);
    float2 xa, xb;
    if constexpr (FMT == FMT_C64) {
      // window starts are arbitrary sample offsets (K:386): 8-byte alignment only
      const float2* src = reinterpret_cast<const float2*>(p.iq) + base + (long long)N1 * q;
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const float4 x4 = *reinterpret_cast<const float4*>(src);
        xa = make_float2(x4.x, x4.y); xb = make_float2(x4.z, x4.w);
      } else {
        xa = src[0]; xb = src[1];
      }
    } else {
      const uchar2* src = reinterpret_cast<const uchar2*>(p.iq) + base + (long long)N1 * q;
      uchar4 b;
      if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
        b = *reinterpret_cast<const uchar4*>(src);
      } else {
        const uchar2 b0 = src[0], b1 = src[1];
        b = make_uchar4(b0.x, b0.y, b1.x, b1.y);
      }
      xa = make_float2(((float)b.x - p.u8_offset) * p.u8_inv_scale, ((float)b.y - p.u8_offset) * p.u8_inv_scale);
      xb = make_float2(((float)b.z - p.u8_offset) * p.u8_inv_scale, ((float)b.w - p.u8_offset) * p.u8_inv_scale);
    }
    va[q] = make_float2(xa.x * wn.x, xa.y * wn.x);
    vb[q] = make_float2(xb.x * wn.y, xb.y * wn.y);
  }
  dft16(va);   // position P holds Y[perm16(P)]
  dft16(vb);
  // output twiddles w^k2, k2 = a + 4b: w^a * w^(4b) from six table rows (generated in float64)
  float4 wa[4], wb[4];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    wa[r + 1] = *reinterpret_cast<const float4*>(p.tw + r * N1 + n1);
    wb[r + 1] = *reinterpret_cast<const float4*>(p.tw + (3 + r) * N1 + n1);
  }
  float2* const z = p.z + ((long long)fr * 16 * p.nwin + w) * N1 + n1;
#pragma unroll
  for (int P = 0; P < 16; ++P) {
    const int k2 = perm<16>(P), a = k2 & 3, b = k2 >> 2;
    float2 ya = va[P], yb = vb[P];
    if (a && b) {
      ya = cmul(ya, cmul(make_float2(wa[a].x, wa[a].y), make_float2(wb[b].x, wb[b].y)));
      yb = cmul(yb, cmul(make_float2(wa[a].z, wa[a].w), make_float2(wb[b].z, wb[b].w)));
    } else if (a) {
      ya = cmul(ya, make_float2(wa[a].x, wa[a].y));
      yb = cmul(yb, make_float2(wa[a].z, wa[a].w));
    } else if (b) {
      ya = cmul(ya, make_float2(wb[b].x, wb[b].y));
      yb = cmul(yb, make_float2(wb[b].z, wb[b].w));
    }
#ifndef KSA_DIF_NT_STORE
#define KSA_DIF_NT_STORE 1   // Z is written with the non-temporal policy: it is read once, by the next kernel (+3 % at config 5; 0: A/B)
#endif
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4* const dst = reinterpret_cast<f32x4*>(z + (long long)k2 * p.nwin * N1);
    const f32x4 val = {ya.x, ya.y, yb.x, yb.y};
    if constexpr (KSA_DIF_NT_STORE) __builtin_nontemporal_store(val, dst);
    else *dst = val;
  }
}

// ---- radix 32 / 64 first stage: one n1 per thread (8-byte accesses, 512 B per wave-instruction) ----------------------------
constexpr float kCos64[64] = {1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220882438f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254523567f, 0.77301045336273699338f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960228867f, 0.47139673682599780857f, 0.38268343236508983729f, 0.29028467725446233105f, 0.19509032201612833135f, 0.09801714032956077016f, 0.00000000000000006123f, -0.09801714032956064526f, -0.19509032201612819257f, -0.29028467725446216452f, -0.38268343236508972627f, -0.47139673682599769755f, -0.55557023301960195560f, -0.63439328416364537677f, -0.70710678118654746172f, -0.77301045336273699338f, -0.83146961230254534669f, -0.88192126434835493853f, -0.92387953251128673848f, -0.95694033573220882438f, -0.98078528040323043058f, -0.99518472667219681771f, -1.00000000000000000000f, -0.99518472667219692873f, -0.98078528040323043058f, -0.95694033573220893540f, -0.92387953251128684951f, -0.88192126434835504956f, -0.83146961230254545772f, -0.77301045336273710440f, -0.70710678118654768376f, -0.63439328416364593188f, -0.55557023301960217765f, -0.47139673682599786408f, -0.38268343236509033689f, -0.29028467725446244208f, -0.19509032201612866442f, -0.09801714032956045097f, -0.00000000000000018370f, 0.09801714032956009015f, 0.19509032201612830359f, 0.29028467725446205350f, 0.38268343236509000382f, 0.47139673682599758653f, 0.55557023301960184458f, 0.63439328416364559882f, 0.70710678118654735069f, 0.77301045336273666031f, 0.83146961230254523567f, 0.88192126434835482751f, 0.92387953251128651644f, 0.95694033573220882438f, 0.98078528040323031956f, 0.99518472667219692873f};
constexpr float kSin64[64] = {0.00000000000000000000f, 0.09801714032956060363f, 0.19509032201612824808f, 0.29028467725446233105f, 0.38268343236508978178f, 0.47139673682599764204f, 0.55557023301960217765f, 0.63439328416364548779f, 0.70710678118654746172f, 0.77301045336273699338f, 0.83146961230254523567f, 0.88192126434835493853f, 0.92387953251128673848f, 0.95694033573220893540f, 0.98078528040323043058f, 0.99518472667219681771f, 1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220893540f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254545772f, 0.77301045336273710440f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960217765f, 0.47139673682599786408f, 0.38268343236508989280f, 0.29028467725446238656f, 0.19509032201612860891f, 0.09801714032956082567f, 0.00000000000000012246f, -0.09801714032956058975f, -0.19509032201612835911f, -0.29028467725446210901f, -0.38268343236508967076f, -0.47139673682599764204f, -0.55557023301960195560f, -0.63439328416364526575f, -0.70710678118654746172f, -0.77301045336273666031f, -0.83146961230254523567f, -0.88192126434835493853f, -0.92387953251128651644f, -0.95694033573220882438f, -0.98078528040323031956f, -0.99518472667219692873f, -1.00000000000000000000f, -0.99518472667219692873f, -0.98078528040323043058f, -0.95694033573220893540f, -0.92387953251128662746f, -0.88192126434835504956f, -0.83146961230254545772f, -0.77301045336273688235f, -0.70710678118654768376f, -0.63439328416364593188f, -0.55557023301960217765f, -0.47139673682599791960f, -0.38268343236509039240f, -0.29028467725446249759f, -0.19509032201612871993f, -0.09801714032956050648f};

// 64-point DFT in registers as 4 x 16 (decimation in frequency): radix-4 over the samples 16 apart, constant twiddles
// W64^(b*d), then four 16-point DFTs.  Position 16*d + P ends up holding Y[4*perm16(P) + d].
template <int B, int D>
__device__ __forceinline__ void w64_twiddle(float2 (&v)[64]) {
  if constexpr (D < 4) {
    if constexpr (B > 0 && D > 0) {
      constexpr int m = (B * D) & 63;
      v[B + 16 * D] = cmul(v[B + 16 * D], make_float2(kCos64[m], -kSin64[m]));
    }
    if constexpr (B < 15) w64_twiddle<B + 1, D>(v);
    else w64_twiddle<0, D + 1>(v);
  }
}
template <int B>
__device__ __forceinline__ void w64_level_a(float2 (&v)[64]) {
  if constexpr (B < 16) {
    dft4<B, 16>(v);
    w64_level_a<B + 1>(v);
  }
}
__device__ __forceinline__ void dft64(float2 (&v)[64]) {
  w64_level_a<0>(v);
  w64_twiddle<1, 1>(v);
  dft16_at<0>(v);
  dft16_at<16>(v);
  dft16_at<32>(v);
  dft16_at<48>(v);
}
__host__ __device__ constexpr int perm64(int p) { return 4 * (((p & 15) >> 2) | ((p & 3) << 2)) + (p >> 4); }

template <int FMT, int R>
__global__ __launch_bounds__(256) void dif_wide_kernel(const DifParams p) {
  static_assert(R == 32 || R == 64, "wide first stage: radix 32 or 64");
  const int n1 = blockIdx.x * 256 + threadIdx.x;
  const int w = blockIdx.y, fr = blockIdx.z;
  const int N1 = p.n1;
  const long long base = (long long)(p.frame0 + fr) * p.frame_stride + p.starts[w] + n1;
  float2 v[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const float wn = p.window[n1 + N1 * q];
    float2 x;
    if constexpr (FMT == FMT_C64) {
      x = (reinterpret_cast<const float2*>(p.iq) + base)[(long long)N1 * q];
    } else {
      const uchar2 b = (reinterpret_cast<const uchar2*>(p.iq) + base)[(long long)N1 * q];
      x = make_float2(((float)b.x - p.u8_offset) * p.u8_inv_scale, ((float)b.y - p.u8_offset) * p.u8_inv_scale);
    }
    v[q] = make_float2(x.x * wn, x.y * wn);
  }
  if constexpr (R == 32) dft32(v);
  else dft64(v);
  // output twiddles w^k2, k2 = a + 4b + 16c: w^a * w^(4b) * w^(16c) from nine table rows (generated in float64)
  float2 wa[4], wb[4], wc[4];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    wa[r + 1] = p.tw[r * N1 + n1];
    wb[r + 1] = p.tw[(3 + r) * N1 + n1];
    if (r < R / 16 - 1) wc[r + 1] = p.tw[(6 + r) * N1 + n1];
  }
  float2* const z = p.z + ((long long)fr * R * p.nwin + w) * N1 + n1;
#pragma unroll
  for (int P = 0; P < R; ++P) {
    const int k2 = R == 32 ? perm32(P) : perm64(P);
    const int a = k2 & 3, b = (k2 >> 2) & 3, c = k2 >> 4;
    float2 y = v[P];
    if (a) y = cmul(y, wa[a]);
    if (b) y = cmul(y, wb[b]);
    if (c) y = cmul(y, wc[c]);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 val = {y.x, y.y};
    __builtin_nontemporal_store(val, reinterpret_cast<f32x2*>(z + (long long)k2 * p.nwin * N1));
  }
}

struct DifFinishParams {
  const float* y;            // [chunk_frames][radix][N1] linear magnitudes, each row fftshifted by N1/2 (spectrum_kernel's output)
  int n, n1;
  int radix;                 // 16, 32 or 64
  int frame0;
  int out_mode;
  float gain, min_amp;
  float* out;                // [nframes][N]
  int hm_w;
  const float* adj;
  float* hm_rows;
  float* hm_ring;
  int hm_index0, hm_first;
};

// One workgroup = 1024/radix consecutive k1 x radix k2 = 1024 consecutive output bins of one frame.
template <int R>
__global__ __launch_bounds__(256) void dif16_finish_kernel(const DifFinishParams p) {
  __shared__ float tile[1024 + 64];          // [1024/R][R + 1]
  __shared__ float cellv[1024];
  const int tid = threadIdx.x;
  const int fr = blockIdx.y;                 // frame inside the chunk
  const int frame = p.frame0 + fr;
  constexpr int KO = 1024 / R;
  const int k1_0 = blockIdx.x * KO;
  const int N = p.n, N1 = p.n1;
  const float* yf = p.y + (long long)fr * R * N1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + 256 * j, k2 = e / KO, ko = e - k2 * KO;
    tile[ko * (R + 1) + k2] = yf[(long long)k2 * N1 + ((k1_0 + ko + N1 / 2) & (N1 - 1))];
  }
  __syncthreads();
  const int g = p.hm_w > 0 ? N / p.hm_w : 0;
  const bool hm_here = g > 0 && g <= 1024;   // larger cells: rowmax_batch afterwards
  const int sh0 = (R * k1_0 + N / 2) & (N - 1);
  float* const orow = p.out + (long long)frame * N + sh0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j;             // k = R*k1_0 + i
    float lin = tile[(i / R) * (R + 1) + (i % R)];
    const float o = p.out_mode != OUT_LINEAR ? out_db(lin, p.out_mode, p.gain, p.min_amp) : lin;
    orow[i] = o;
    if (hm_here) cellv[i] = p.adj ? o - p.adj[sh0 + i] : o;
  }
  if (!hm_here) return;
  __syncthreads();
  const int cells = 1024 / g;
  for (int c = tid; c < cells; c += 256) {
    float hv = -__builtin_inff();
    bool nan = false;
    for (int i = 0; i < g; ++i) {
      const float v = cellv[c * g + ((i + c) & (g - 1))];   // rotated start: spreads the LDS banks
      nan |= v != v;
      hv = fmaxf(hv, v);
    }
    if (nan) hv = __builtin_nanf("");
    const int cell = sh0 / g + c;
    if (p.hm_rows) p.hm_rows[(long long)frame * p.hm_w + cell] = hv;
    if (p.hm_ring && frame >= p.hm_first) p.hm_ring[((p.hm_index0 + frame) % HM_ROWS) * p.hm_w + cell] = hv;
  }
}

}  // namespace ksa
