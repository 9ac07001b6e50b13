// Four-step path for transforms that do not fit one workgroup's LDS (N = N1*N2 > 16384, up to 2^20).
//
// Same arithmetic as ksa_kernels.hpp::spectrum_kernel (rows A0, A4-A9 of SURVEY.md section 8; replaces
// numpy.fft.fft at python/kspecanal.py:391 for fftSize 32768..1048576), split over two kernels:
//
//   n = N2*n1 + n2,  k = k1 + N1*k2
//   X[k1 + N1*k2] = sum_n2 W_N^(n2*k1) W_N2^(n2*k2) [ sum_n1 x[N2*n1 + n2] W_N1^(n1*k1) ]
//
//   fourstep_cols<N1>  per (frame, window, group of S adjacent columns n2): window multiply, N1-point
//                      LDS FFT down the column, twiddle W_N^(n2*k1), store Z[k1][n2] (scratch in HBM)
//   fourstep_rows<N2>  per (frame, group of S adjacent rows k1): for every window of the frame an
//                      N2-point LDS FFT along the row, |X|, fold over the windows (K:392-395), then
//                      scale / fftshift / dB and a staged store in runs of S consecutive bins
//   rowmax_batch       waterfall cells (K:480) from the finished dB rows
//
// Scratch traffic is 2 x 8 N bytes per window on top of the IQ read, so this path is HBM-streaming
// bound; frames are processed in chunks sized to a fixed scratch budget.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "ksa_kernels.hpp"

namespace ksa {

// ---- one NS-point transform per L = NS/16 threads, 16 points per thread, exchange through LDS ------
// in : v[(q % B0)*R0 + q/B0] = x[l + L*q]           (slot order of pass 0)
// out: v[i] = X[l + L*perm16(i)]
// `my` = this transform's private LDS region of Plan<NS>::NPAD (+pad) float2; tw_mid = middle-pass
// tables in LDS; w* = last-pass twiddles of thread l.  Every thread of the workgroup must call it.
template <int NS>
__device__ __forceinline__ void fft_lds(float2 (&v)[16], float2* my, const float2* tw_mid, int l,
                                        float2 w1, float2 w2, float2 w3, float2 w4, float2 w8, float2 w12) {
  using P = Plan<NS>;
  constexpr int L = P::L, M = P::M, R0 = P::R0, B0 = P::B0;
  dft_first<R0>(v);
  if constexpr (M >= 2) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B0; ++b) {
      const int i = l + b * L;
#pragma unroll
      for (int t = 0; t < R0; ++t) my[padi(i * R0 + perm<R0>(t))] = v[b * R0 + t];
    }
    __syncthreads();
    int pp = R0, tw_off = 0;
#pragma unroll
    for (int s = 1; s < M; ++s) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = my[padi(l + L * t)];
      if (s < M - 1) {
        const float2* tw = tw_mid + tw_off + (l & (pp - 1));
        dft16_tw(v, tw[0], tw[pp], tw[2 * pp], tw[3 * pp], tw[7 * pp], tw[11 * pp]);
        __syncthreads();
        const int kk = l & (pp - 1);
        const int j = (l - kk) * 16 + kk;
#pragma unroll
        for (int t = 0; t < 16; ++t) my[padi(j + perm<16>(t) * pp)] = v[t];
        __syncthreads();
        tw_off += 15 * pp;
        pp *= 16;
      } else {
        dft16_tw(v, w1, w2, w3, w4, w8, w12);
      }
    }
  }
}

template <int NS>
struct SubPlan {
  using P = Plan<NS>;
  static constexpr int T = 256;
  static constexpr int L = P::L;
  static constexpr int S = T / L;                 // transforms per workgroup
  static constexpr int STRIDE = P::NPAD | 1;      // odd region stride: slot-fastest lanes hit distinct banks
  static constexpr int LDS_BYTES = (S * STRIDE + P::MID) * 8;
};

struct FourParams {
  SpecParams sp;          // same per-run fields as the single-workgroup kernel
  int n1, n2;             // N = n1*n2
  int frame0;             // first frame of this chunk
  int chunk_frames;       // frames in this chunk
  float2* z;              // [chunk_frames][nwin][n1][n2]
  const float2* tw_big;   // [n1][n2] : W_N^(k1*n2)
  const float2* tw1_mid;  // sub-plan tables for N1
  const float2* tw1_last;
  const float2* tw2_mid;  // sub-plan tables for N2
  const float2* tw2_last;
};

template <int N1, int FMT>
__global__ __launch_bounds__(256) void fourstep_cols(const FourParams fp) {
  using SP = SubPlan<N1>;
  using P = Plan<N1>;
  constexpr int L = SP::L, S = SP::S, R0 = P::R0, B0 = P::B0;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2* const tw_lds = lds + S * SP::STRIDE;
  const SpecParams& p = fp.sp;
  const int tid = threadIdx.x;
  const int slot = tid % S;            // adjacent columns on adjacent lanes: coalesced 8*S-byte runs
  const int l = tid / S;
  float2* const my = lds + slot * SP::STRIDE;
  if constexpr (P::MID > 0) {
    for (int i = tid; i < P::MID; i += 256) tw_lds[i] = fp.tw1_mid[i];
  }
  float2 w1, w2, w3, w4, w8, w12;
  if constexpr (P::M >= 2) {
    w1 = fp.tw1_last[0 * L + l];  w2 = fp.tw1_last[1 * L + l];  w3 = fp.tw1_last[2 * L + l];
    w4 = fp.tw1_last[3 * L + l];  w8 = fp.tw1_last[7 * L + l];  w12 = fp.tw1_last[11 * L + l];
  }
  const int n2 = fp.n2;
  const int col = blockIdx.x * S + slot;
  const int w = blockIdx.y;
  const int fr = blockIdx.z;           // frame inside the chunk
  const long long base = (long long)(fp.frame0 + fr) * p.frame_stride + p.starts[w];
  float2 v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int n = n2 * (l + L * q) + col;
    const float wn = p.window[n];
    float2 x;
    if constexpr (FMT == FMT_C64) {
      x = reinterpret_cast<const float2*>(p.iq)[base + n];
    } else {
      const uchar2 b = reinterpret_cast<const uchar2*>(p.iq)[base + n];
      x = make_float2(((float)b.x - p.u8_offset) * p.u8_inv_scale, ((float)b.y - p.u8_offset) * p.u8_inv_scale);
    }
    v[(q % B0) * R0 + (q / B0)] = make_float2(x.x * wn, x.y * wn);
  }
  fft_lds<N1>(v, my, tw_lds, l, w1, w2, w3, w4, w8, w12);
  float2* const z = fp.z + ((long long)fr * p.nwin + w) * fp.n1 * n2;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k1 = l + L * perm<16>(i);
    z[(long long)k1 * n2 + col] = cmul(v[i], fp.tw_big[(long long)k1 * n2 + col]);
  }
}

template <int N2>
__global__ __launch_bounds__(256) void fourstep_rows(const FourParams fp) {
  using SP = SubPlan<N2>;
  using P = Plan<N2>;
  constexpr int L = SP::L, S = SP::S, R0 = P::R0, B0 = P::B0;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2* const tw_lds = lds + S * SP::STRIDE;
  const SpecParams& p = fp.sp;
  const int tid = threadIdx.x;
  const int slot = tid / L;            // one row per slot, lanes run along the row
  const int l = tid - slot * L;
  float2* const my = lds + slot * SP::STRIDE;
  if constexpr (P::MID > 0) {
    for (int i = tid; i < P::MID; i += 256) tw_lds[i] = fp.tw2_mid[i];
  }
  float2 w1, w2, w3, w4, w8, w12;
  if constexpr (P::M >= 2) {
    w1 = fp.tw2_last[0 * L + l];  w2 = fp.tw2_last[1 * L + l];  w3 = fp.tw2_last[2 * L + l];
    w4 = fp.tw2_last[3 * L + l];  w8 = fp.tw2_last[7 * L + l];  w12 = fp.tw2_last[11 * L + l];
  }
  const int n1 = fp.n1;
  const int k1 = blockIdx.x * S + slot;
  const int fr = blockIdx.y;
  const int nm1 = p.nwin - 1;
  float acc[16];
  const float init = p.cumu == CUMU_MIN ? __builtin_inff() : 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = init;
  for (int w = 0; w < p.nwin; ++w) {
    const float2* zrow = fp.z + (((long long)fr * p.nwin + w) * n1 + k1) * N2;
    float2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[(q % B0) * R0 + (q / B0)] = zrow[l + L * q];
    fft_lds<N2>(v, my, tw_lds, l, w1, w2, w3, w4, w8, w12);
    if (p.cumu == CUMU_AVG) {
      const int e = w == 0 ? nm1 : nm1 - w + 1;
      const float wt = ldexpf(1.0f, -e);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = fmaf(wt, __builtin_amdgcn_sqrtf(fmaf(v[i].x, v[i].x, v[i].y * v[i].y)), acc[i]);
    } else if (p.cumu == CUMU_MAX) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaxf(acc[i], fmaf(v[i].x, v[i].x, v[i].y * v[i].y));
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fminf(acc[i], fmaf(v[i].x, v[i].x, v[i].y * v[i].y));
    }
  }
  // stage [k2][slot] so that the store loop writes S consecutive bins (k = k1 + n1*k2) per k2
  float* const stage = reinterpret_cast<float*>(lds);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[(l + L * perm<16>(i)) * S + slot] = acc[i];
  __syncthreads();
  const int n = n1 * N2;
  float* const orow = p.out + (long long)(fp.frame0 + fr) * n;
  for (int i = tid; i < S * N2; i += 256) {
    const int k = blockIdx.x * S + (i % S) + n1 * (i / S);
    float lin = p.cumu == CUMU_AVG ? stage[i] : __builtin_amdgcn_sqrtf(stage[i]);
    lin *= p.scale;
    float o = lin;
    if (p.out_mode != OUT_LINEAR) {
      if (p.out_mode == OUT_DB_CLIP) lin = fmaxf(lin, p.min_amp);
      o = db_of(lin, p.gain);
    }
    orow[(k + n / 2) & (n - 1)] = o;
  }
}

// waterfall cells of finished dB rows: blockIdx.y = frame of the batch
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

// ---- host side ---------------------------------------------------------------------------------------
inline void make_twiddles(int n, std::vector<float2>& mid, std::vector<float2>& last) {
  const int log2n = ilog2(n);
  const int m = (log2n + 3) / 4;
  const int r0 = 1 << (log2n - 4 * (m - 1));
  int pcur = r0;
  for (int s = 1; s < m; ++s) {
    std::vector<float2>& dst = s < m - 1 ? mid : last;
    for (int t = 1; t < 16; ++t)
      for (int k = 0; k < pcur; ++k) {
        const double ang = -2.0 * M_PI * (double)t * (double)k / ((double)pcur * 16.0);
        dst.push_back(make_float2((float)std::cos(ang), (float)std::sin(ang)));
      }
    pcur *= 16;
  }
}

struct FourStep {
  int n = 0, n1 = 0, n2 = 0, nwin = 0;
  int threads = 256, lds_bytes = 0, vgprs = 0;
  int chunk_frames = 1;
  float2 *d_z = nullptr, *d_tw_big = nullptr, *d_tw1_mid = nullptr, *d_tw1_last = nullptr, *d_tw2_mid = nullptr,
         *d_tw2_last = nullptr;
};

inline int fs_upload(float2** dst, const std::vector<float2>& src) {
  if (hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(src.size(), 1) * sizeof(float2)) != hipSuccess) return 1;
  if (!src.empty() && hipMemcpy(*dst, src.data(), src.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

template <int NS, int FMT, bool COLS>
int fs_launch(const FourParams& fp, dim3 grid, hipStream_t stream, bool configure, int* vgprs) {
  using SP = SubPlan<NS>;
  if constexpr (COLS) {
    auto kfn = fourstep_cols<NS, FMT>;
    if (configure) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SP::LDS_BYTES) != hipSuccess) return 1;
      return 0;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), SP::LDS_BYTES, stream, fp);
  } else {
    auto kfn = fourstep_rows<NS>;
    if (configure) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SP::LDS_BYTES) != hipSuccess) return 1;
      hipFuncAttributes attr;
      if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kfn)) != hipSuccess) return 1;
      if (vgprs) *vgprs = attr.numRegs;
      return 0;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), SP::LDS_BYTES, stream, fp);
  }
  return hipGetLastError() != hipSuccess;
}

template <int FMT, bool COLS>
int fs_dispatch(int ns, const FourParams& fp, dim3 grid, hipStream_t stream, bool configure, int* vgprs) {
  switch (ns) {
    case 128: return fs_launch<128, FMT, COLS>(fp, grid, stream, configure, vgprs);
    case 256: return fs_launch<256, FMT, COLS>(fp, grid, stream, configure, vgprs);
    case 512: return fs_launch<512, FMT, COLS>(fp, grid, stream, configure, vgprs);
    case 1024: return fs_launch<1024, FMT, COLS>(fp, grid, stream, configure, vgprs);
    default: return 1;
  }
}

inline int fourstep_create(FourStep& f, int n, int nwin, int max_frames, int /*num_cu*/) {
  const int lg = ilog2(n);
  if (lg < 15 || lg > 20) return 1;
  f.n = n;
  f.n1 = 1 << (lg / 2);          // 128x256, 256x256, 256x512, 512x512, 512x1024, 1024x1024
  f.n2 = n / f.n1;
  f.nwin = nwin;
  std::vector<float2> mid, last;
  make_twiddles(f.n1, mid, last);
  if (fs_upload(&f.d_tw1_mid, mid) || fs_upload(&f.d_tw1_last, last)) return 1;
  mid.clear(); last.clear();
  make_twiddles(f.n2, mid, last);
  if (fs_upload(&f.d_tw2_mid, mid) || fs_upload(&f.d_tw2_last, last)) return 1;
  std::vector<float2> big((size_t)n);
  for (int k1 = 0; k1 < f.n1; ++k1)
    for (int c = 0; c < f.n2; ++c) {
      const double ang = -2.0 * M_PI * (double)k1 * (double)c / (double)n;
      big[(size_t)k1 * f.n2 + c] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
  if (fs_upload(&f.d_tw_big, big)) return 1;
  // scratch budget 4 GiB (at least one frame; 2.46 -> 2.31 ms at config 5 against 1 GiB, smaller chunks only lose: 32 MiB 9.7 ms)
  const size_t per_frame = (size_t)nwin * n * sizeof(float2);
  size_t budget = (size_t)4 << 30;
  if (const char* mb = getenv("KSA_FS_SCRATCH_MB")) budget = (size_t)atol(mb) << 20;   // A/B switch for measurements
  size_t cf = std::max<size_t>(1, budget / per_frame);
  f.chunk_frames = (int)std::min<size_t>(cf, (size_t)max_frames);
  if (hipMalloc(reinterpret_cast<void**>(&f.d_z), per_frame * f.chunk_frames) != hipSuccess) return 1;
  FourParams fp{};
  if (fs_dispatch<FMT_C64, true>(f.n1, fp, dim3(1), nullptr, true, nullptr)) return 1;
  if (fs_dispatch<FMT_U8, true>(f.n1, fp, dim3(1), nullptr, true, nullptr)) return 1;
  if (fs_dispatch<FMT_C64, false>(f.n2, fp, dim3(1), nullptr, true, &f.vgprs)) return 1;
  f.lds_bytes = f.n2 == 128 ? SubPlan<128>::LDS_BYTES : f.n2 == 256 ? SubPlan<256>::LDS_BYTES
              : f.n2 == 512 ? SubPlan<512>::LDS_BYTES : SubPlan<1024>::LDS_BYTES;
  return 0;
}

inline void fourstep_destroy(FourStep& f) {
  float2* ptrs[] = {f.d_z, f.d_tw_big, f.d_tw1_mid, f.d_tw1_last, f.d_tw2_mid, f.d_tw2_last};
  for (float2* p : ptrs) if (p) hipFree(p);
  f = FourStep();
}

inline int fourstep_run(FourStep& f, const SpecParams& sp, int fmt, hipStream_t stream, int /*num_cu*/) {
  FourParams fp{};
  fp.sp = sp;
  fp.n1 = f.n1;
  fp.n2 = f.n2;
  fp.z = f.d_z;
  fp.tw_big = f.d_tw_big;
  fp.tw1_mid = f.d_tw1_mid;  fp.tw1_last = f.d_tw1_last;
  fp.tw2_mid = f.d_tw2_mid;  fp.tw2_last = f.d_tw2_last;
  const int s1 = 256 / (f.n1 / 16), s2 = 256 / (f.n2 / 16);
  for (int f0 = 0; f0 < sp.nframes; f0 += f.chunk_frames) {
    const int cf = std::min(f.chunk_frames, sp.nframes - f0);
    fp.frame0 = f0;
    fp.chunk_frames = cf;
    const dim3 gcols(f.n2 / s1, sp.nwin, cf);
    const int rc = fmt == FMT_C64 ? fs_dispatch<FMT_C64, true>(f.n1, fp, gcols, stream, false, nullptr)
                                  : fs_dispatch<FMT_U8, true>(f.n1, fp, gcols, stream, false, nullptr);
    if (rc) return 1;
    if (fs_dispatch<FMT_C64, false>(f.n2, fp, dim3(f.n1 / s2, cf), stream, false, nullptr)) return 1;
  }
  if (sp.hm_w > 0) {
    hipLaunchKernelGGL(rowmax_batch, dim3((sp.hm_w + 255) / 256, sp.nframes), dim3(256), 0, stream, sp, f.n);
    if (hipGetLastError() != hipSuccess) return 1;
  }
  return 0;
}

}  // namespace ksa
