// tools/ta_probe.hip -- what the vector-memory path charges for the load shapes of the small transforms (config 4, N = 64):
// single-wave workgroups, 16 per CU, every wave walks "frames" of 512 complex64 samples in 5 rounds; per round
//   mode 0: sixteen 8-byte loads per lane, lane (slot s, l): sample start_s + l + 4q   (what spectrum_kernel<64> does)
//   mode 1: eight 16-byte loads per lane,  lane (s, l): samples start_s + 2l + 8j, +1   (a lane owns ADJACENT samples)
//   mode 2: sixteen 8-byte loads per lane, fully contiguous per instruction (64 lanes x 8 B = 512 B)   (coalescing floor)
//   mode 3: eight 16-byte loads, fully contiguous per instruction (1 KB)
//   mode 4: mode 1 + the quad shuffle that hands every sample to the lane the 4 x 16 plan wants it in (64 DPP moves per round)
//   mode 5: mode 1's loads with one FRAME per slot (probe_frames below)
// plus ~350 dependent-free FMAs per round to stand for the transform.  Prints microseconds per round per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ta_probe tools/ta_probe.hip && tools/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int FMAS>
__global__ __launch_bounds__(64, 4) void probe(const float2* iq, int nframes, float* sink) {
  const int tid = threadIdx.x, slot = tid >> 2, l = tid & 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
    const char* fbase = reinterpret_cast<const char*>(iq) + (long long)f * 512 * 8;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(fbase), 0, 512 * 8, 0x00020000);
    for (int rd = 0; rd < 5; ++rd) {
      const int k = rd * 16 + slot;
      const int start = (int)(k * 6.4);      // hops of 6 / 7 samples, as K:386 gives them at nonOverlap 0.1
      float2 v[16];
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (start + l) * 8, 4 * q * 8, 0);
          v[q] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
        }
      } else if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (start + 2 * l) * 8, 8 * j * 8, 0);
          v[2 * j] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
          v[2 * j + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
        }
      } else if (MODE == 4) {
        // lane l loads samples 8j + 2l, 8j + 2l + 1; target lane t, register q <- lane (t>>1) + 2(q&1), piece q>>1, half t&1
        u32x4 piece[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) piece[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (start + 2 * l) * 8, 8 * j * 8, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x4 pj = piece[q >> 1];
          unsigned re = 0, im = 0;
          if (q & 1) {     // quad_perm [2,2,3,3] = 0xFA
            re = __builtin_amdgcn_update_dpp(re, pj.x, 0xFA, 0xF, 0x5, false);
            re = __builtin_amdgcn_update_dpp(re, pj.z, 0xFA, 0xF, 0xA, false);
            im = __builtin_amdgcn_update_dpp(im, pj.y, 0xFA, 0xF, 0x5, false);
            im = __builtin_amdgcn_update_dpp(im, pj.w, 0xFA, 0xF, 0xA, false);
          } else {         // quad_perm [0,0,1,1] = 0x50
            re = __builtin_amdgcn_update_dpp(re, pj.x, 0x50, 0xF, 0x5, false);
            re = __builtin_amdgcn_update_dpp(re, pj.z, 0x50, 0xF, 0xA, false);
            im = __builtin_amdgcn_update_dpp(im, pj.y, 0x50, 0xF, 0x5, false);
            im = __builtin_amdgcn_update_dpp(im, pj.w, 0x50, 0xF, 0xA, false);
          }
          v[q] = make_float2(__uint_as_float(re), __uint_as_float(im));
        }
      } else if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, tid * 8, (q & 7) * 512, 0);
          v[q] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (j & 3) * 1024, 0);
          v[2 * j] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
          v[2 * j + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
        }
      }
#pragma unroll
      for (int it = 0; it < FMAS / 32; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[q & 7] = fmaf(v[q].x, 1.0001f, acc[q & 7]);
          acc[(q + 3) & 7] = fmaf(v[q].y, 0.9999f, acc[(q + 3) & 7]);
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 123.456f) sink[tid] = s;
}

template <int MODE, int FMAS>
double run(const float2* iq, int nframes, float* sink, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<MODE, FMAS>), dim3(grid), dim3(64), 0, 0, iq, nframes, sink);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<MODE, FMAS>), dim3(grid), dim3(64), 0, 0, iq, nframes, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}


// mode 5: one FRAME per slot -- a wave walks 16 consecutive frames at once, all slots at the same window k (71 rounds per 16
// frames instead of 80, no idle slot, the window start a scalar); each load instruction touches 16 frames 4 KB apart
template <int FMAS>
__global__ __launch_bounds__(64, 4) void probe_frames(const float2* iq, int nframes, float* sink) {
  const int tid = threadIdx.x, slot = tid >> 2, l = tid & 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int g = blockIdx.x; g * 16 < nframes; g += gridDim.x) {
    const char* fbase = reinterpret_cast<const char*>(iq) + (long long)g * 16 * 512 * 8;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(fbase), 0, 16 * 512 * 8, 0x00020000);
    for (int k = 0; k < 71; ++k) {
      const int start = (int)(k * 6.4);
      float2 v[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (slot * 512 + start + 2 * l) * 8, 8 * j * 8, 0);
        v[2 * j] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
        v[2 * j + 1] = make_float2(__uint_as_float(r.z), __uint_as_float(r.w));
      }
#pragma unroll
      for (int it = 0; it < FMAS / 32; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[q & 7] = fmaf(v[q].x, 1.0001f, acc[q & 7]);
          acc[(q + 3) & 7] = fmaf(v[q].y, 0.9999f, acc[(q + 3) & 7]);
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 123.456f) sink[tid] = s;
}

template <int FMAS>
double run_frames(const float2* iq, int nframes, float* sink, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe_frames<FMAS>), dim3(grid), dim3(64), 0, 0, iq, nframes, sink);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe_frames<FMAS>), dim3(grid), dim3(64), 0, 0, iq, nframes, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  const int nframes = 1255424 / 4;      // a quarter of config 4's batch: 1.3 GB
  float2* iq; float* sink;
  hipMalloc(&iq, (size_t)nframes * 512 * 8);
  hipMalloc(&sink, 4096);
  hipMemset(iq, 0, (size_t)nframes * 512 * 8);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * 16;
  const double rounds_per_cu = (double)nframes * 5 / prop.multiProcessorCount;
  const char* names[5] = {"16 x 8 B, config-4 shape (slots 6.4 samples apart)", "8 x 16 B, a lane owns adjacent samples", "16 x 8 B, contiguous per instruction", "8 x 16 B, contiguous per instruction", "8 x 16 B adjacent samples + quad shuffle to the 4 x 16 layout"};
  double t[5][2];
  t[0][0] = run<0, 0>(iq, nframes, sink, grid);   t[0][1] = run<0, 352>(iq, nframes, sink, grid);
  t[1][0] = run<1, 0>(iq, nframes, sink, grid);   t[1][1] = run<1, 352>(iq, nframes, sink, grid);
  t[2][0] = run<2, 0>(iq, nframes, sink, grid);   t[2][1] = run<2, 352>(iq, nframes, sink, grid);
  t[3][0] = run<3, 0>(iq, nframes, sink, grid);   t[3][1] = run<3, 352>(iq, nframes, sink, grid);
  t[4][0] = run<4, 0>(iq, nframes, sink, grid);   t[4][1] = run<4, 352>(iq, nframes, sink, grid);
  for (int m = 0; m < 5; ++m)
    printf("%-58s loads only %.3f ms = %.4f us per round per CU (16 waves) | + 352 FMAs per lane and round %.3f ms = %.4f us\n", names[m], t[m][0],
           t[m][0] * 1e3 / rounds_per_cu * 16, t[m][1], t[m][1] * 1e3 / rounds_per_cu * 16);
  const double f0 = run_frames<0>(iq, nframes, sink, grid), f1 = run_frames<352>(iq, nframes, sink, grid);
  const double rounds5 = (double)nframes * 71 / 16 / prop.multiProcessorCount;
  printf("%-58s loads only %.3f ms = %.4f us per round per CU (16 waves) | + 352 FMAs per lane and round %.3f ms = %.4f us   (71 rounds per 16 frames: whole batch %.3f ms against %.3f ms for the adjacent-sample shape)\n",
         "8 x 16 B adjacent samples, one FRAME per slot", f0, f0 * 1e3 / rounds5 * 16, f1, f1 * 1e3 / rounds5 * 16, f1, t[1][1]);
  return 0;
}
