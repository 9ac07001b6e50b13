// libksa: C ABI (include/ksa.h) over the gfx950 kernels in ksa_kernels.hpp.
// Host-side orchestration only: table generation, scratch, launches, state hand-off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ksa.h"
#include "ksa_dif16.hpp"
#include "ksa_kernels.hpp"
#include "ksa_kernels32.hpp"
#include "ksa_kernels64.hpp"
#include "ksa_kernels_pair.hpp"

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}

#define HIP_OK(call)                                                                      \
  do {                                                                                    \
    hipError_t _e = (call);                                                               \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// Entry points run on their engine's device and hand the caller's current device back on every exit path (a process that
// drives one engine per GPU, or torch beside libksa, keeps its own notion of "current").
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// A/B switches of the measurement builds (tools/variants.sh compiles with -DKSA_EXPERIMENTS).  The shipped library reads
// no environment variable: which kernel a user runs depends on the engine's configuration only.
#ifdef KSA_EXPERIMENTS
const char* exp_env(const char* name) { return getenv(name); }
#else
constexpr const char* exp_env(const char*) { return nullptr; }
#endif

// default first-stage scratch per chunk of frames (ksa_dif16.hpp); tuned on MI355X, see DESIGN.md 4.2
#ifndef KSA_DIF_SCRATCH_MB_DEFAULT
#define KSA_DIF_SCRATCH_MB_DEFAULT 4096
#endif

// host mirrors of ksa::Tune<N>::FUSED / FUSED_LAST (the twiddle table layouts depend on them)
bool tune_fused(int n) {
#ifdef KSA_FUSED
  (void)n;
  return KSA_FUSED;
#else
  return n <= 4096;
#endif
}
bool tune_fused_last(int n) {
#ifdef KSA_FUSED_LAST
  (void)n;
  return KSA_FUSED_LAST;
#else
  return tune_fused(n);
#endif
}

}  // namespace

struct ksa_engine {
  ksa_config cfg{};
  hipStream_t stream = nullptr;
  int num_cu = 0;
  // tables
  int* d_starts = nullptr;
  int* d_start_last = nullptr;  // RAW mode: the last window only
  float* d_window = nullptr;
  float* d_window32 = nullptr;  // 32-point plan: the same taps as [8][N/32][4] (16-byte tap loads, ksa_kernels32.hpp)
  float2* d_tw_mid = nullptr;
  float2* d_tw_last = nullptr;
  float* d_adj = nullptr;       // zeroSpan Fft.Adj or null
  float* d_scan_adj = nullptr;
  // scratch
  void* d_iq_stage = nullptr;   // one block for the host-pointer entry points
  float* d_frames = nullptr;    // [max_frames][N] per-frame spectra when the caller passes none
  float* d_part = nullptr;      // [chunks][3][N]
  float* d_xchg = nullptr;      // one block [4][N] partial | [128][hm_width] ring: what a rank sends to the others
  float* d_partial = nullptr;   // [4][N]            (= d_xchg)
  float* d_state = nullptr;     // [4][N] cur,max,min,avg
  float* d_hm = nullptr;        // [128][hm_width]   (= d_xchg + 4N)
  float* d_scan_state = nullptr;  // [4][total]
  float* d_scan_hm = nullptr;     // [128][scan_hm_width]
  float* d_scan_avg_rows = nullptr;  // [128][total] Fft.Avg after each of a batch's last passes (multi-pass stitch)
  float* d_levels = nullptr;      // [4][cells] plot-side decimation scratch
  int* d_highs = nullptr;         // peak markers: [HIGHS_MAX] cell | [HIGHS_MAX] level (float bits) | found
  float* d_parts = nullptr;       // [capacity][N] partial folds of the window-split (latency) mode
  int levels_cap = 0;
  // multi-engine merges inside one process (ksa_allreduce_state, ksa_scan_allstitch) and the host-pointer scan pass
  float* d_gather = nullptr;      // [n][4N + 128W] every engine's exchange block, or [n][rows][scan_hm_width]
  size_t gather_cap = 0;          // floats
  void* d_scan_stage = nullptr;   // [nsteps][full_size] capture blocks of ksa_scan_pass_c64 / _u8
  size_t scan_stage_cap = 0;      // bytes
  float* d_scan_rows = nullptr;   // [128][scan_hm_width] partial waterfall rows of a band-sharded batch
  int scan_rows = 0;              // rows the last ksa_scan_stitch_range_dev wrote
  int scan_rows_passes = 0;       // ... and the passes of that batch (ring advance of ksa_scan_merge_rows_dev)
  float* d_scan_halo = nullptr;   // [nhalo][npasses][N] halo bands (ksa_scan_allstitch)
  float* d_scan_send = nullptr;   // [nhalo][npasses][N] own bands packed for the right neighbours
  size_t scan_halo_cap = 0, scan_send_cap = 0;   // floats
  hipEvent_t ev_ready = nullptr, ev_copied = nullptr, ev_stream = nullptr;
  // N > 16384: radix-16 / 32 / 64 decimation in frequency in front of the single-workgroup kernel (ksa_dif16.hpp)
  int sub_n = 0;                // size of the single-workgroup transform: fft_size (path 0) or fft_size/16 (path 2)
  float2* d_dif_tw = nullptr;   // [6][N1]
  float2* d_dif_z = nullptr;    // [chunk][16][nwin][N1]
  float* d_dif_y = nullptr;     // [chunk][16][N1]
  float* d_ones = nullptr;      // [N1] taps of the second stage (the window was applied in the first)
  int* d_starts_b = nullptr;    // [nwin] w*N1
  int dif_chunk = 1;
  int dif_radix = 16;           // first-stage radix: 16 (N <= 262144), 32 (524288), 64 (1048576)
  // bookkeeping
  long long frames_seen = 0;
  int hm_index = 0;
  int pending_frames = 0;       // frames of the last uncommitted batch
  int b_max = 1, b_min = 1, b_avg = 1;
  int has_max = 0, has_min = 0, has_avg = 0;   // the curve holds frames already (seeded by copy otherwise, K:133-134)
  int scan_base_is_raw = 0;
  long long scan_passes = 0;
  int scan_hm_index = 0;
  int max_chunks = 1;
  // launch config of the spectrum kernel
  int path = 0, threads = 0, lds_bytes = 0, vgprs = 0, blocks_per_cu = 1;
  int reuse_m = 0;              // new samples per thread per window when hops are a fixed multiple of N/16
  bool plan32 = false;          // 8192 / 16384: the 32-points-per-thread kernel (ksa_kernels32.hpp)
  // 1024 .. 4096, large batches: two frames per workgroup in packed fp32 (ksa_kernels_pair.hpp)
  bool pair_ok = false;
  int pair_bpc = 0, pair_vgprs = 0, pair_lds = 0;
  // N = 64, complex64: the 8 x 8 plan with adjacent-sample loads (ksa_kernels64.hpp)
  bool k64_ok = false;
  bool win_ones = false;          // every tap of the window table is exactly 1.0f (the reference's default window, K:52)
  int k64_bpc = 0, k64_vgprs = 0;
  // profiling
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  double prof_ms = 0;
  long long prof_launches = 0;
  // clock stamps around profiled spectrum stages (ksa_prof_clock, ksa::clock_stamp_kernel)
  unsigned long long* d_clk = nullptr;      // [CLK_SLOTS][before | after][CLK_KEYS][memtime, memrealtime]
  long long clk_launches = 0;
};

namespace {

using ksa::SpecParams;

template <int N, int FMT, int RM, int CM>
int launch_spec_c(ksa_engine* e, const SpecParams& p, bool configure_only);

template <int N, int FMT, int RM, int CM>
int launch_pair_c(ksa_engine* e, const SpecParams& p, bool configure_only) {
  using PP = ksa::PlanPair<N>;
  auto kfn = ksa::spectrum_pair_kernel<N, FMT, RM, CM>;
  if (configure_only) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS_BYTES));
    hipFuncAttributes attr;
    HIP_OK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kfn)));
    int occ = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, ksa::Plan<N>::T, PP::LDS_BYTES));
    if (FMT == ksa::FMT_C64 && CM == ksa::CUMU_AVG) {
      e->pair_bpc = std::max(1, occ);
      e->pair_vgprs = attr.numRegs;
      e->pair_lds = PP::LDS_BYTES;
    }
    return 0;
  }
  const int npairs = (p.nframes + 1) / 2;
  const int grid = std::max(1, std::min(npairs, e->num_cu * e->pair_bpc));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(ksa::Plan<N>::T), PP::LDS_BYTES, e->stream, p);
  HIP_OK(hipGetLastError());
  return 0;
}

template <int N, int FMT, int RM>
int launch_pair(ksa_engine* e, const SpecParams& p, bool cfg_only) {
  if (cfg_only) {
    if (launch_pair_c<N, FMT, RM, ksa::CUMU_MAX>(e, p, true) || launch_pair_c<N, FMT, RM, ksa::CUMU_MIN>(e, p, true)) return 1;
    return launch_pair_c<N, FMT, RM, ksa::CUMU_AVG>(e, p, true);
  }
  if (p.cumu == ksa::CUMU_AVG) return launch_pair_c<N, FMT, RM, ksa::CUMU_AVG>(e, p, false);
  if (p.cumu == ksa::CUMU_MAX) return launch_pair_c<N, FMT, RM, ksa::CUMU_MAX>(e, p, false);
  return launch_pair_c<N, FMT, RM, ksa::CUMU_MIN>(e, p, false);
}

template <int N, int FMT, int RM>
int launch_spec_t(ksa_engine* e, const SpecParams& p, bool configure_only) {
  // large batches of 1024 .. 4096-point transforms: two frames per workgroup in packed fp32
#ifdef KSA_EXPERIMENTS
  constexpr bool pair_size = ksa::Plan<N>::S == 1 && ksa::Plan<N>::T <= 256 && ksa::Plan<N>::M >= 2;   // 1024 .. 4096 (KSA_PAIR_ALL)
#else
  constexpr bool pair_size = N == 1024;     // where it measured faster: the only size the product instantiates
#endif
  if constexpr (pair_size) {
    if (e->pair_ok) {
      if (configure_only) {
        // every reuse variant a later batch can pick (RAW mode runs RM = 0 whatever the hops); the one in use last,
        // so that pair_bpc / pair_vgprs describe it
        if (RM != 0 && launch_pair<N, FMT, 0>(e, p, true)) return 1;
        if (RM != 4 && launch_pair<N, FMT, 4>(e, p, true)) return 1;
        if (RM != 8 && launch_pair<N, FMT, 8>(e, p, true)) return 1;
        if (launch_pair<N, FMT, RM>(e, p, true)) return 1;
      } else if (p.nframes >= 2 * e->num_cu * e->pair_bpc) return launch_pair<N, FMT, RM>(e, p, false);
    }
  }
  // fold mode as a template constant (Tune<N>::fold_const) or as a run-time branch inside the window loop
  if constexpr (ksa::Tune<N>::fold_const(RM)) {
    if (configure_only) {
      if (launch_spec_c<N, FMT, RM, ksa::CUMU_MAX>(e, p, true) || launch_spec_c<N, FMT, RM, ksa::CUMU_MIN>(e, p, true)) return 1;
      return launch_spec_c<N, FMT, RM, ksa::CUMU_AVG>(e, p, true);
    }
    if (p.cumu == ksa::CUMU_AVG) return launch_spec_c<N, FMT, RM, ksa::CUMU_AVG>(e, p, false);
    if (p.cumu == ksa::CUMU_MAX) return launch_spec_c<N, FMT, RM, ksa::CUMU_MAX>(e, p, false);
    return launch_spec_c<N, FMT, RM, ksa::CUMU_MIN>(e, p, false);
  } else {
    return launch_spec_c<N, FMT, RM, 0>(e, p, configure_only);
  }
}

template <int N, int FMT, int RM, int CM>
int launch_spec_c(ksa_engine* e, const SpecParams& p, bool configure_only) {
  using P = ksa::Plan<N>;
  auto kfn = ksa::spectrum_kernel<N, FMT, RM, CM>;
  static const int lds_pad = exp_env("KSA_LDS_PAD_KB") ? atoi(exp_env("KSA_LDS_PAD_KB")) * 1024 : 0;   // occupancy experiments
  const int lds_bytes = ksa::Tune<N>::LDS_BYTES + lds_pad;
  if (configure_only) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipFuncAttributes attr;
    HIP_OK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kfn)));
    int occ = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, P::T, lds_bytes));
    if (FMT == ksa::FMT_C64) {
      e->threads = P::T;
      e->lds_bytes = ksa::Tune<N>::LDS_BYTES;
      e->vgprs = attr.numRegs;
      e->blocks_per_cu = std::max(1, occ);
    }
    return 0;
  }
  const int capacity = e->num_cu * e->blocks_per_cu;
  SpecParams q = p;
  // small batches (the per-frame drop-in, one scan pass): split every frame's windows over several
  // workgroups so that the GPU is filled; the partial folds are combined by a second, tiny kernel
  // (pays from N = 1024 up: 84 -> 51 us per block at N=4096, 720 -> 117 us at N=16384; tiny transforms only lose the launches)
  if (e->d_parts && N >= 1024 && !exp_env("KSA_NO_SPLIT") && p.nwin > 1 && p.nframes * 2 <= capacity) {
    q.parts = std::min(p.nwin, capacity / p.nframes);
    q.part_out = e->d_parts;
  }
  const int grid = std::max(1, std::min(q.nframes * std::max(1, q.parts), capacity));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(P::T), lds_bytes, e->stream, q);
  if (q.parts > 1) {
    hipLaunchKernelGGL(ksa::combine_parts_kernel, dim3((N / 4 + 63) / 64, q.nframes), dim3(64), 0, e->stream, q, N);
    if (q.hm_w > 0)
      hipLaunchKernelGGL(ksa::rowmax_batch, dim3((q.hm_w + 255) / 256, q.nframes), dim3(256), 0, e->stream, q, N);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// sample-reuse variants exist where one transform fills the workgroup and the carried samples fit the register
// budget (N = 1024..4096).  At T >= 512 (128-VGPR cap) the carried registers spill: measured 9..28 % slower than
// re-reading the overlap through L2 (N=8192: 42.6 vs 46.3 M FFT/s at 50 %, 40.7 vs 51.9 at 75 %; N=16384: 20.4 vs
// 23.3 and 19.7 vs 25.0), so those sizes take the general path.
template <int N, int FMT>
int launch_spec_rm(ksa_engine* e, const SpecParams& p, bool cfg_only, int rm) {
  if constexpr (ksa::Plan<N>::S == 1 && ksa::Plan<N>::T <= 256) {
    if (rm == 8) return launch_spec_t<N, FMT, 8>(e, p, cfg_only);
    if (rm == 4) return launch_spec_t<N, FMT, 4>(e, p, cfg_only);
  }
  return launch_spec_t<N, FMT, 0>(e, p, cfg_only);
}

// 32 points per thread (N = 8192, 16384): fold mode always a template constant
template <int N, int FMT, int CM>
int launch_spec32_c(ksa_engine* e, const SpecParams& p, bool configure_only) {
  using P = ksa::Plan32<N>;
  auto kfn = ksa::spectrum32_kernel<N, FMT, CM>;
  if (configure_only) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS_BYTES));
    hipFuncAttributes attr;
    HIP_OK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kfn)));
    int occ = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, P::T, P::LDS_BYTES));
    if (FMT == ksa::FMT_C64 && CM == ksa::CUMU_AVG) {
      e->threads = P::T;
      e->lds_bytes = P::LDS_BYTES;
      e->vgprs = attr.numRegs;
      e->blocks_per_cu = std::max(1, occ);
    }
    return 0;
  }
  const int capacity = e->num_cu * e->blocks_per_cu;
  SpecParams q = p;
  if (e->d_parts && !exp_env("KSA_NO_SPLIT") && p.nwin > 1 && p.nframes * 2 <= capacity) {   // window-split (latency) mode
    q.parts = std::min(p.nwin, capacity / p.nframes);
    q.part_out = e->d_parts;
  }
  int grid = std::max(1, std::min(q.nframes * std::max(1, q.parts), capacity));
  if (const char* g = exp_env("KSA_GRID")) grid = std::max(1, std::min(grid, atoi(g)));   // measurement: fewer persistent workgroups
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(P::T), P::LDS_BYTES, e->stream, q);
  if (q.parts > 1) {
    hipLaunchKernelGGL(ksa::combine_parts_kernel, dim3((N / 4 + 63) / 64, q.nframes), dim3(64), 0, e->stream, q, N);
    if (q.hm_w > 0)
      hipLaunchKernelGGL(ksa::rowmax_batch, dim3((q.hm_w + 255) / 256, q.nframes), dim3(256), 0, e->stream, q, N);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

template <int N, int FMT>
int launch_spec32(ksa_engine* e, const SpecParams& p, bool cfg_only) {
  if (cfg_only) {
    if (launch_spec32_c<N, FMT, ksa::CUMU_MAX>(e, p, true) || launch_spec32_c<N, FMT, ksa::CUMU_MIN>(e, p, true)) return 1;
    return launch_spec32_c<N, FMT, ksa::CUMU_AVG>(e, p, true);
  }
  if (p.cumu == ksa::CUMU_AVG) return launch_spec32_c<N, FMT, ksa::CUMU_AVG>(e, p, false);
  if (p.cumu == ksa::CUMU_MAX) return launch_spec32_c<N, FMT, ksa::CUMU_MAX>(e, p, false);
  return launch_spec32_c<N, FMT, ksa::CUMU_MIN>(e, p, false);
}

// N = 64, complex64 input: 8 x 8 with adjacent samples per lane (ksa_kernels64.hpp)
template <int CM, bool W1>
int launch_spec64_c(ksa_engine* e, const SpecParams& p, bool configure_only) {
  auto kfn = ksa::spectrum64_kernel<ksa::FMT_C64, CM, W1>;
  if (configure_only) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, ksa::Plan64::LDS_BYTES));
    int occ = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, ksa::Plan64::T, ksa::Plan64::LDS_BYTES));
    if (CM == ksa::CUMU_AVG) {
      hipFuncAttributes attr;
      HIP_OK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kfn)));
      e->k64_bpc = std::max(1, occ);
      e->k64_vgprs = attr.numRegs;
    }
    return 0;
  }
  const int grid = std::max(1, std::min(p.nframes, e->num_cu * e->k64_bpc));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(ksa::Plan64::T), ksa::Plan64::LDS_BYTES, e->stream, p);
  HIP_OK(hipGetLastError());
  return 0;
}

template <bool W1>
int launch_spec64_w(ksa_engine* e, const SpecParams& p, bool cfg_only) {
  if (cfg_only) {
    if (launch_spec64_c<ksa::CUMU_MAX, W1>(e, p, true) || launch_spec64_c<ksa::CUMU_MIN, W1>(e, p, true)) return 1;
    return launch_spec64_c<ksa::CUMU_AVG, W1>(e, p, true);
  }
  if (p.cumu == ksa::CUMU_AVG) return launch_spec64_c<ksa::CUMU_AVG, W1>(e, p, false);
  if (p.cumu == ksa::CUMU_MAX) return launch_spec64_c<ksa::CUMU_MAX, W1>(e, p, false);
  return launch_spec64_c<ksa::CUMU_MIN, W1>(e, p, false);
}
int launch_spec64(ksa_engine* e, const SpecParams& p, bool cfg_only) {
  // (p.window is the engine's own table here: the first-stage paths, which hand this stage a table of ones, start at N = 32768)
  return e->win_ones && !exp_env("KSA_NO_W1") ? launch_spec64_w<true>(e, p, cfg_only) : launch_spec64_w<false>(e, p, cfg_only);
}

template <int FMT>
int launch_spec_n(ksa_engine* e, const SpecParams& p, bool cfg_only) {
  const int rm = p.nwin > 1 ? e->reuse_m : 0;   // RAW mode transforms a single window: nothing to reuse
  if (e->plan32) return e->sub_n == 8192 ? launch_spec32<8192, FMT>(e, p, cfg_only) : launch_spec32<16384, FMT>(e, p, cfg_only);
  switch (e->sub_n) {
    case 16: return launch_spec_rm<16, FMT>(e, p, cfg_only, rm);
    case 32: return launch_spec_rm<32, FMT>(e, p, cfg_only, rm);
    case 64:
      if constexpr (FMT == ksa::FMT_C64) {
        if (e->k64_ok) {
          if (cfg_only) { if (launch_spec64(e, p, true)) return 1; }     // (then the general kernel's attributes too: uint8 input runs it)
          else if (!exp_env("KSA_NO_K64")) return launch_spec64(e, p, false);
        }
      }
      return launch_spec_rm<64, FMT>(e, p, cfg_only, rm);
    case 128: return launch_spec_rm<128, FMT>(e, p, cfg_only, rm);
    case 256: return launch_spec_rm<256, FMT>(e, p, cfg_only, rm);
    case 512: return launch_spec_rm<512, FMT>(e, p, cfg_only, rm);
    case 1024: return launch_spec_rm<1024, FMT>(e, p, cfg_only, rm);
    case 2048: return launch_spec_rm<2048, FMT>(e, p, cfg_only, rm);
    case 4096: return launch_spec_rm<4096, FMT>(e, p, cfg_only, rm);
#ifdef KSA_EXPERIMENTS   // the 16-point plan at these sizes is reachable through KSA_PLAN16 only
    case 8192: return launch_spec_rm<8192, FMT>(e, p, cfg_only, rm);
    case 16384: return launch_spec_rm<16384, FMT>(e, p, cfg_only, rm);
#endif
    default: return fail("fft_size %d has no single-workgroup plan", e->sub_n);
  }
}

int prof_begin(ksa_engine* e, hipEvent_t* a, hipEvent_t* b) {
  *a = *b = nullptr;
  if (!e->prof || e->prof_events.size() >= 8192) return 0;
  HIP_OK(hipEventCreate(a));
  HIP_OK(hipEventCreate(b));
  HIP_OK(hipEventRecord(*a, e->stream));
  return 0;
}
int prof_end(ksa_engine* e, hipEvent_t a, hipEvent_t b) {
  if (!a) return 0;
  HIP_OK(hipEventRecord(b, e->stream));
  e->prof_events.emplace_back(a, b);
  return 0;
}

// N = 16 * N1 (32768 .. 262144): first stage in registers, N1-point stage by the single-workgroup kernel on the
// 16 pseudo frames of every frame, interleave + dB + waterfall by dif16_finish_kernel (ksa_dif16.hpp).
int run_dif16(ksa_engine* e, const SpecParams& p, int fmt) {
  const int n = e->cfg.fft_size, n1 = e->sub_n, nwin = p.nwin;
  for (int f0 = 0; f0 < p.nframes; f0 += e->dif_chunk) {
    const int cf = std::min(e->dif_chunk, p.nframes - f0);
    ksa::DifParams a{};
    a.iq = p.iq;
    a.frame_stride = p.frame_stride;
    a.frame0 = f0;
    a.nwin = nwin;
    a.starts = p.starts;
    a.window = p.window;
    a.tw = e->d_dif_tw;
    a.n1 = n1;
    a.u8_offset = p.u8_offset;
    a.u8_inv_scale = p.u8_inv_scale;
    a.z = e->d_dif_z;
    const int R = e->dif_radix;
    if (R == 16) {
      const dim3 ga(n1 / 512, nwin, cf);     // two adjacent n1 per thread
      if (fmt == KSA_FMT_C64) hipLaunchKernelGGL(ksa::dif16_kernel<ksa::FMT_C64>, ga, dim3(256), 0, e->stream, a);
      else hipLaunchKernelGGL(ksa::dif16_kernel<ksa::FMT_U8>, ga, dim3(256), 0, e->stream, a);
    } else {
      const dim3 ga(n1 / 256, nwin, cf);     // one n1 per thread, 32 or 64 samples in registers
      if (R == 32 && fmt == KSA_FMT_C64) hipLaunchKernelGGL((ksa::dif_wide_kernel<ksa::FMT_C64, 32>), ga, dim3(256), 0, e->stream, a);
      else if (R == 32) hipLaunchKernelGGL((ksa::dif_wide_kernel<ksa::FMT_U8, 32>), ga, dim3(256), 0, e->stream, a);
      else if (fmt == KSA_FMT_C64) hipLaunchKernelGGL((ksa::dif_wide_kernel<ksa::FMT_C64, 64>), ga, dim3(256), 0, e->stream, a);
      else hipLaunchKernelGGL((ksa::dif_wide_kernel<ksa::FMT_U8, 64>), ga, dim3(256), 0, e->stream, a);
    }
    SpecParams b{};
    b.iq = e->d_dif_z;
    b.frame_stride = (long long)nwin * n1;
    b.frame_len = nwin * n1;
    b.nframes = cf * R;
    b.nwin = nwin;
    b.starts = e->d_starts_b;      // w * N1 (a single window, RAW mode, starts at 0 too)
    b.window = e->d_ones;
    b.tw_mid = p.tw_mid;
    b.tw_last = p.tw_last;
    b.scale = p.scale;
    b.cumu = p.cumu;
    b.out_mode = ksa::OUT_LINEAR;
    b.out = e->d_dif_y;
    if (launch_spec_n<ksa::FMT_C64>(e, b, false)) return 1;
    ksa::DifFinishParams c{};
    c.y = e->d_dif_y;
    c.n = n;
    c.n1 = n1;
    c.radix = R;
    c.frame0 = f0;
    c.out_mode = p.out_mode;
    c.gain = p.gain;
    c.min_amp = p.min_amp;
    c.out = p.out;
    c.hm_w = p.hm_w;
    c.adj = p.adj;
    c.hm_rows = p.hm_rows;
    c.hm_ring = p.hm_ring;
    c.hm_index0 = p.hm_index0;
    c.hm_first = p.hm_first;
    const dim3 gf(n1 / (1024 / R), cf);
    if (R == 16) hipLaunchKernelGGL(ksa::dif16_finish_kernel<16>, gf, dim3(256), 0, e->stream, c);
    else if (R == 32) hipLaunchKernelGGL(ksa::dif16_finish_kernel<32>, gf, dim3(256), 0, e->stream, c);
    else hipLaunchKernelGGL(ksa::dif16_finish_kernel<64>, gf, dim3(256), 0, e->stream, c);
  }
  if (p.hm_w > 0 && n / p.hm_w > 1024)      // cells wider than the finish kernel's tile
    hipLaunchKernelGGL(ksa::rowmax_batch, dim3((p.hm_w + 255) / 256, p.nframes), dim3(256), 0, e->stream, p, n);
  HIP_OK(hipGetLastError());
  return 0;
}

// Spectrum stage for a batch: the single-workgroup LDS FFT, behind a radix-16 / 32 / 64 first stage for N > 16384.
int run_spectrum(ksa_engine* e, const void* iq, int fmt, long long stride, int nframes, int out_mode,
                 float* out, bool with_hm, float* hm_rows) {
  const ksa_config& c = e->cfg;
  if (fmt != KSA_FMT_C64 && fmt != KSA_FMT_U8) return fail("unknown sample format %d", fmt);
  if (nframes < 1 || nframes > c.max_frames) return fail("nframes %d outside 1..max_frames(%d)", nframes, c.max_frames);
  if (stride < 0) return fail("negative frame_stride");
  if (stride > (1ll << 27)) return fail("frame_stride %lld exceeds 2^27 samples", stride);
  // the output stage stores float4 runs; IQ loads are per-sample but frames should start on sample bounds
  if ((reinterpret_cast<uintptr_t>(out) & 15) || (hm_rows && (reinterpret_cast<uintptr_t>(hm_rows) & 15)))
    return fail("device output buffers must be 16-byte aligned");
  if (reinterpret_cast<uintptr_t>(iq) & (fmt == KSA_FMT_C64 ? 7 : 1)) return fail("IQ buffer is not sample aligned");
  SpecParams p{};
  p.iq = iq;
  p.frame_stride = stride;
  p.frame_len = c.full_size;
  p.nframes = nframes;
  const bool raw = c.cumu_mode == KSA_CUMU_RAW;
  p.nwin = raw ? 1 : c.num_windows;            // RAW keeps the last window only (K:135-136)
  p.starts = raw ? e->d_start_last : e->d_starts;
#if defined(KSA32_TAPS_X4) && !KSA32_TAPS_X4
  p.window = e->d_window;
#else
  p.window = (e->plan32 && e->path == 0) ? e->d_window32 : e->d_window;
#endif
  p.tw_mid = e->d_tw_mid;
  p.tw_last = e->d_tw_last;
  p.scale = (float)c.mag_scale;
  p.cumu = (raw || c.cumu_mode == KSA_CUMU_AVG) ? ksa::CUMU_AVG : c.cumu_mode == KSA_CUMU_MAX ? ksa::CUMU_MAX : ksa::CUMU_MIN;
  p.out_mode = out_mode;
  p.gain = c.gain;
  p.min_amp = c.min_amp;
  p.u8_offset = c.u8_offset;
  p.u8_inv_scale = 1.0f / c.u8_scale;
  p.out = out;
  p.hm_w = with_hm ? c.hm_width : 0;
  p.adj = with_hm ? e->d_adj : nullptr;
  p.hm_rows = with_hm ? hm_rows : nullptr;
  p.hm_ring = with_hm ? e->d_hm : nullptr;
  p.hm_index0 = e->hm_index;
  p.hm_first = std::max(0, nframes - KSA_HM_ROWS);
#ifdef KSA_STAMPS
  static unsigned long long* dbg = nullptr;   // diagnostic build: 4096 blocks x 16 waves x 12 segments
  const size_t dbg_n = 4096 * 16 * 12;
  if (!dbg) hipMalloc(reinterpret_cast<void**>(&dbg), dbg_n * 8);
  hipMemsetAsync(dbg, 0, dbg_n * 8, e->stream);
  p.dbg = dbg;
#endif
  // profiled stage: clock stamps directly before and after it on the same stream (ksa_prof_clock), OUTSIDE the event pair
  ksa::u64x2* const clk = (e->prof && e->d_clk) ? reinterpret_cast<ksa::u64x2*>(e->d_clk) + (size_t)(e->clk_launches++ % ksa::CLK_SLOTS) * 2 * ksa::CLK_KEYS : nullptr;
  if (clk) hipLaunchKernelGGL(ksa::clock_stamp_kernel, dim3(ksa::CLK_WGS), dim3(64), 0, e->stream, clk, 0);
  hipEvent_t ea, eb;
  if (prof_begin(e, &ea, &eb)) return 1;
  int rc;
  if (e->path == 2) {
    if ((rc = run_dif16(e, p, fmt))) return rc;
  } else {
    rc = fmt == KSA_FMT_C64 ? launch_spec_n<ksa::FMT_C64>(e, p, false) : launch_spec_n<ksa::FMT_U8>(e, p, false);
    if (rc) return rc;
  }
#ifdef KSA_STAMPS
  if (const char* path = exp_env("KSA_STAMPS_FILE")) {
    hipStreamSynchronize(e->stream);
    std::vector<unsigned long long> h(dbg_n);
    hipMemcpy(h.data(), dbg, dbg_n * 8, hipMemcpyDeviceToHost);
    double sum[12] = {0};
    long long waves = 0;
    for (size_t w = 0; w < dbg_n / 12; ++w) {
      if (!h[w * 12 + 0] && !h[w * 12 + 7]) continue;
      ++waves;
      for (int i = 0; i < 12; ++i) sum[i] += (double)h[w * 12 + i];
    }
    if (FILE* f = fopen(path, "a")) {
      fprintf(f, "waves %lld nframes %d nwin %d :", waves, nframes, p.nwin);
      for (int i = 0; i < 12; ++i) fprintf(f, " %.0f", waves ? sum[i] / waves : 0.0);
      fprintf(f, "\n");
      fclose(f);
    }
  }
#endif
  if (prof_end(e, ea, eb)) return 1;
  if (clk) hipLaunchKernelGGL(ksa::clock_stamp_kernel, dim3(ksa::CLK_WGS), dim3(64), 0, e->stream, clk, 1);
  return 0;
}

int run_accumulate(ksa_engine* e, const float* db, int nframes, long long first_index, long long total) {
  const int n = e->cfg.fft_size;
  ksa::AccParams a{};
  a.db = db;
  a.n = n;
  a.nframes = nframes;
  a.first_index = first_index;
  a.total_frames = total;
  a.has_prev = e->has_avg;
  int chunks = std::min(e->max_chunks, std::max(1, nframes / 64));
  a.chunk = (nframes + chunks - 1) / chunks;
  chunks = (nframes + a.chunk - 1) / a.chunk;
  a.part = e->d_part;
  const int tb = 256, gx = (n + tb - 1) / tb;
  const int tb4 = 64, gx4 = (n / 4 + tb4 - 1) / tb4;     // 4 bins per thread
  hipLaunchKernelGGL(ksa::accumulate_partial_kernel, dim3(gx4, chunks), dim3(tb4), 0, e->stream, a);
  const int owns_last = first_index + nframes == total;
  hipLaunchKernelGGL(ksa::accumulate_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, e->stream, e->d_part, chunks, n,
                     db + (long long)(nframes - 1) * n, owns_last, e->d_partial);
  HIP_OK(hipGetLastError());
  return 0;
}

int do_commit(ksa_engine* e, long long total, int local_frames) {
  const int n = e->cfg.fft_size;
  const int tb = 256, gx = (n + tb - 1) / tb;
  hipLaunchKernelGGL(ksa::commit_kernel, dim3(gx), dim3(tb), 0, e->stream, e->d_partial, e->d_state, n,
                     e->has_max, e->has_min, e->has_avg, total, e->b_max, e->b_min, e->b_avg);
  HIP_OK(hipGetLastError());
  e->has_max |= e->b_max;
  e->has_min |= e->b_min;
  e->has_avg |= e->b_avg;
  e->frames_seen += total;
  e->hm_index = (int)((e->hm_index + local_frames) % KSA_HM_ROWS);
  e->pending_frames = 0;
  return 0;
}

template <typename T>
int upload(T** dst, const T* src, size_t count) {
  HIP_OK(hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(count, 1) * sizeof(T)));
  if (count) HIP_OK(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

int fill(ksa_engine* e, float* dst, long long n, float v) {
  if (n <= 0) return 0;
  const int tb = 256;
  const int gx = (int)std::min<long long>((n + tb - 1) / tb, 4096);
  hipLaunchKernelGGL(ksa::fill_kernel, dim3(gx), dim3(tb), 0, e->stream, dst, n, v);
  HIP_OK(hipGetLastError());
  return 0;
}

int scan_reset(ksa_engine* e) {
  const ksa_config& c = e->cfg;
  if (!c.scan_total_entries) return fail("engine was created without scan geometry");
  // K:603-608: Cur = Max = Avg = dB(minAmp4Clip), Min = dB(1.0); K:613-614 ring = minAmp4Clip (linear)
  float floor_db = (float)(10.0 * std::log10((double)c.min_amp) - (double)c.gain);
  if (std::isinf(floor_db)) floor_db = 0.f;      // infTo = 0 (K:604 -> K:110-111): minAmp4Clip 0 starts the curves at 0 dB
  const float one_db = (float)(10.0 * std::log10(1.0) - (double)c.gain);
  const long long t = c.scan_total_entries;
  if (fill(e, e->d_scan_state, t, floor_db)) return 1;
  if (fill(e, e->d_scan_state + t, t, floor_db)) return 1;
  if (fill(e, e->d_scan_state + 2 * t, t, one_db)) return 1;
  if (fill(e, e->d_scan_state + 3 * t, t, floor_db)) return 1;
  if (fill(e, e->d_scan_hm, (long long)KSA_HM_ROWS * c.scan_hm_width, c.min_amp)) return 1;
  e->scan_passes = 0;
  e->scan_hm_index = 0;
  e->scan_rows = e->scan_rows_passes = 0;      // an abandoned band-sharded batch must not block ksa_scan_allstitch for ever
  return 0;
}

size_t sample_bytes(int fmt) { return fmt == KSA_FMT_C64 ? 8 : 2; }

// Grow-only device scratch of an engine (on its device, which the caller has made current).
template <typename T>
int ensure(T** ptr, size_t* cap, size_t count) {
  if (*ptr && *cap >= count) return 0;
  if (*ptr) hipFree(*ptr);
  *ptr = nullptr;
  *cap = 0;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(ptr), std::max<size_t>(count, 1) * sizeof(T)));
  *cap = count;
  return 0;
}


}  // namespace

extern "C" {

int ksa_abi_version(void) { return KSA_ABI_VERSION; }
const char* ksa_last_error(void) { return g_err.c_str(); }

int ksa_create(const ksa_config* cfg, ksa_engine** out) {
  if (!cfg || !out) return fail("null argument");
  *out = nullptr;
  if (cfg->abi_version != KSA_ABI_VERSION) return fail("ABI version %d, library is %d", cfg->abi_version, KSA_ABI_VERSION);
  const int n = cfg->fft_size;
  if (!is_pow2(n) || n < 16 || n > (1 << 20)) return fail("fft_size %d must be a power of two in 16..1048576", n);
  if (cfg->full_size < n) return fail("full_size %d < fft_size %d", cfg->full_size, n);
  if (cfg->num_windows < 1 || !cfg->window_starts || !cfg->window) return fail("window table / starts missing");
  for (int i = 0; i < cfg->num_windows; ++i)
    if (cfg->window_starts[i] < 0 || cfg->window_starts[i] + n > cfg->full_size)
      return fail("window %d start %d runs past the block", i, cfg->window_starts[i]);
  if (cfg->cumu_mode < KSA_CUMU_RAW || cfg->cumu_mode > KSA_CUMU_MIN) return fail("unknown cumu_mode %d", cfg->cumu_mode);
  if (cfg->hm_width < 0 || (cfg->hm_width && (n % cfg->hm_width || !is_pow2(cfg->hm_width))))
    return fail("hm_width %d must be a power of two dividing fft_size", cfg->hm_width);
  if (cfg->max_frames < 1) return fail("max_frames must be >= 1");
  if (!(cfg->u8_scale != 0.f)) return fail("u8_scale must be non-zero");
  if (cfg->scan_total_entries) {
    if (cfg->scan_hop < 1 || cfg->scan_hop > n) return fail("scan_hop %d outside 1..fft_size", cfg->scan_hop);
    if (cfg->scan_hm_width < 1 || cfg->scan_total_entries % cfg->scan_hm_width)
      return fail("scan_hm_width %d must divide scan_total_entries %d", cfg->scan_hm_width, cfg->scan_total_entries);
  }
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail("device %d not present (%d visible)", cfg->device, ndev);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, cfg->device));

  ksa_engine* e = new ksa_engine();
  e->cfg = *cfg;
  e->cfg.window_starts = nullptr;
  e->cfg.window = nullptr;
  e->num_cu = prop.multiProcessorCount;
  int rc = 0;
  auto bail = [&](int r) { ksa_destroy(e); return r; };

  if ((rc = upload(&e->d_starts, cfg->window_starts, (size_t)cfg->num_windows))) return bail(rc);
  if ((rc = upload(&e->d_start_last, cfg->window_starts + cfg->num_windows - 1, 1))) return bail(rc);
  if ((rc = upload(&e->d_window, cfg->window, (size_t)n))) return bail(rc);
  e->win_ones = std::all_of(cfg->window, cfg->window + n, [](float w) { return w == 1.0f; });

  {
    e->path = n <= 16384 ? 0 : 2;
    e->dif_radix = n <= 262144 ? 16 : n == 524288 ? 32 : 64;
    const int sn = e->path == 0 ? n : n / e->dif_radix;     // the single-workgroup transform
    e->sub_n = sn;
    const int pt = 16, lpt = 4;   // 16 points per thread, radix-16 passes (an 8-point / radix-8 plan measured 20 % slower)
    const bool fused_mid = tune_fused(sn), fused_last = tune_fused_last(sn);   // layouts must match ksa::Tune<N>
    // twiddles in double, stored as float: middle passes [pt-1][p] each, last pass [pt-1][N/pt]
    const int log2n = ksa::ilog2(sn);
    const int m = (log2n + lpt - 1) / lpt;
    const int r0 = 1 << (log2n - lpt * (m - 1));
    std::vector<float2> mid, last;
    int pcur = r0;
    e->plan32 = (sn == 8192 || sn == 16384) && !exp_env("KSA_PLAN16");   // KSA_PLAN16 (experiments build): back to the 16-point plan
    // Two frames per workgroup in packed fp32 (ksa_kernels_pair.hpp).  Measured against the one-frame kernel on MI355X
    // (tools/pair_sweep.sh, hops 0.5 / 0.25 / 0.1): N = 1024 +14..+30 %, N = 2048 -3..-5 %, N = 4096 0..-5 % -- on by
    // default for 1024 only.  KSA_PAIR_ALL enables it for 1024 .. 4096, KSA_NO_PAIR disables it (A/B switches).
    e->pair_ok = e->path == 0 && !exp_env("KSA_NO_PAIR") && (sn == 1024 || (exp_env("KSA_PAIR_ALL") && sn >= 1024 && sn <= 4096));
    if (e->plan32) {
      // folded twiddles of dft16_fused for a base twiddle of `beta` turns: w^4, w^8, w^12, then w^n2 * W16^(n2*k1)
      auto fused15 = [](double beta, int e) {
        double turns;
        if (e < 3) turns = 4.0 * (e + 1) * beta;
        else { const int k1 = (e - 3) / 3, n2 = (e - 3) % 3 + 1; turns = n2 * beta + (double)(n2 * k1) / 16.0; }
        const double ang = -2.0 * M_PI * turns;
        return make_float2((float)std::cos(ang), (float)std::sin(ang));
      };
      const int r1 = sn == 16384 ? 32 : 16, lth = sn / 32;
      if (r1 == 32) {          // middle pass radix 32, p = 32: [31][32] = w^16 | fused15(k/1024) | fused15(k/1024 + 1/32)
        mid.resize((size_t)31 * 32);
        for (int k = 0; k < 32; ++k) {
          const double beta = (double)k / 1024.0;
          const double a16 = -2.0 * M_PI * 16.0 * beta;
          mid[k] = make_float2((float)std::cos(a16), (float)std::sin(a16));
          for (int ee = 0; ee < 15; ++ee) {
            mid[(size_t)(1 + ee) * 32 + k] = fused15(beta, ee);
            mid[(size_t)(16 + ee) * 32 + k] = fused15(beta + 1.0 / 32.0, ee);
          }
        }
      } else {                 // middle pass radix 16, p = 32: [15][32] = fused15(k/512)
        mid.resize((size_t)15 * 32);
        for (int k = 0; k < 32; ++k)
          for (int ee = 0; ee < 15; ++ee) mid[(size_t)ee * 32 + k] = fused15((double)k / 512.0, ee);
      }
      // last pass radix 16, two butterflies per thread, k = i = l + b*L: [(b*15 + e)][L]
      last.resize((size_t)30 * lth);
      for (int b = 0; b < 2; ++b)
        for (int ee = 0; ee < 15; ++ee)
          for (int l = 0; l < lth; ++l) last[(size_t)(b * 15 + ee) * lth + l] = fused15((double)(l + b * lth) / (double)sn, ee);
    }
    for (int s = 1; s < m && !e->plan32; ++s) {
      std::vector<float2>& dst = s < m - 1 ? mid : last;
      const bool fused = s < m - 1 ? fused_mid : fused_last;
      if (fused) {
        // rows of dft16_fused: w^4, w^8, w^12, then c[n2][k1] = w^n2 * W16^(n2*k1) for k1 = 0..3, n2 = 1..3
        for (int e = 0; e < 15; ++e)
          for (int k = 0; k < pcur; ++k) {
            const double base = (double)k / ((double)pcur * 16.0);
            double turns;
            if (e < 3) turns = 4.0 * (e + 1) * base;
            else { const int k1 = (e - 3) / 3, n2 = (e - 3) % 3 + 1; turns = n2 * base + (double)(n2 * k1) / 16.0; }
            const double ang = -2.0 * M_PI * turns;
            dst.push_back(make_float2((float)std::cos(ang), (float)std::sin(ang)));
          }
      } else {
        for (int t = 1; t < pt; ++t)
          for (int k = 0; k < pcur; ++k) {
            const double ang = -2.0 * M_PI * (double)t * (double)k / ((double)pcur * pt);
            dst.push_back(make_float2((float)std::cos(ang), (float)std::sin(ang)));
          }
      }
      pcur *= pt;
    }
    if (e->plan32) {
      // taps (or, behind a first stage, the all-ones table) in the 32-point kernel's load order: [q4][l][j] = w[l + L*(4*q4 + j)]
      const int lth = sn / 32;
      std::vector<float> w32((size_t)sn);
      for (int q4 = 0; q4 < 8; ++q4)
        for (int l = 0; l < lth; ++l)
          for (int j = 0; j < 4; ++j)
            w32[((size_t)q4 * lth + l) * 4 + j] = e->path == 0 ? cfg->window[l + lth * (4 * q4 + j)] : 1.0f;
      if ((rc = upload(&e->d_window32, w32.data(), w32.size()))) return bail(rc);
    }
    if (sn == 64 && e->path == 0) {
      // the 8 x 8 plan of N = 64 (ksa_kernels64.hpp): W64^(m k1) as [m][k1]; the 4 x 16 plan has no middle pass, the slot is free
      e->k64_ok = true;
      mid.resize(64);
      for (int mm = 0; mm < 8; ++mm)
        for (int k1 = 0; k1 < 8; ++k1) {
          const double ang = -2.0 * M_PI * (double)(mm * k1) / 64.0;
          mid[(size_t)mm * 8 + k1] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
    }
    if ((rc = upload(&e->d_tw_mid, mid.data(), mid.size()))) return bail(rc);
    if ((rc = upload(&e->d_tw_last, last.data(), last.size()))) return bail(rc);
    // constant hop of 1/2 or 1/4 of the transform: raw samples are carried over in registers
    if (e->path == 0 && cfg->num_windows > 1 && n >= 1024) {
      const int hop = cfg->window_starts[1] - cfg->window_starts[0];
      bool same = true;
      for (int i = 2; i < cfg->num_windows; ++i) same &= cfg->window_starts[i] - cfg->window_starts[i - 1] == hop;
      if (same && (hop == n / 2 || hop == n / 4)) e->reuse_m = hop / (n / pt);
    }
    if (exp_env("KSA_NO_REUSE")) e->reuse_m = 0;   // A/B switch of the experiments build
    SpecParams dummy{};
    dummy.nwin = cfg->num_windows;
    if ((rc = launch_spec_n<ksa::FMT_C64>(e, dummy, true))) return bail(rc);
    if ((rc = launch_spec_n<ksa::FMT_U8>(e, dummy, true))) return bail(rc);
    if (e->path == 2) {
      const int n1 = sn, nw = cfg->num_windows;
      // first-stage output twiddles W_N^(n1*e), e = 1,2,3,4,8,12 (float64-generated); w^k2 = w^(k2&3) * w^(k2&12)
      static const int ex[9] = {1, 2, 3, 4, 8, 12, 16, 32, 48};
      const int nrows = e->dif_radix == 16 ? 6 : 9;
      std::vector<float2> tw((size_t)nrows * n1);
      for (int r = 0; r < nrows; ++r)
        for (int k = 0; k < n1; ++k) {
          const double ang = -2.0 * M_PI * (double)ex[r] * (double)k / (double)n;
          tw[(size_t)r * n1 + k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
      if ((rc = upload(&e->d_dif_tw, tw.data(), tw.size()))) return bail(rc);
      std::vector<float> ones((size_t)n1, 1.0f);
      if ((rc = upload(&e->d_ones, ones.data(), ones.size()))) return bail(rc);
      std::vector<int> sb((size_t)nw);
      for (int w = 0; w < nw; ++w) sb[w] = w * n1;
      if ((rc = upload(&e->d_starts_b, sb.data(), sb.size()))) return bail(rc);
      // scratch: Z = 8 N bytes per window.  Chunks of <= KSA_FS_SCRATCH_MB (default below) of Z
      const size_t per_frame = (size_t)nw * n * sizeof(float2);
      // 32-bit offsets inside the kernels: a pseudo frame is nwin*N1 complex points addressed in bytes
      if ((long long)nw * n1 >= (1ll << 28)) return bail(fail("num_windows %d x fft_size/16 %d exceeds 2^28 points per frame", nw, n1));
      size_t budget = (size_t)KSA_DIF_SCRATCH_MB_DEFAULT << 20;
      if (const char* mb = exp_env("KSA_FS_SCRATCH_MB")) {       // A/B switch of the experiments build; nonsense keeps the default
        const long v = atol(mb);
        if (v >= 1 && v <= (256l << 10)) budget = (size_t)v << 20;
      }
      e->dif_chunk = (int)std::min<size_t>(std::max<size_t>(1, budget / per_frame), (size_t)cfg->max_frames);
      e->dif_chunk = std::min(e->dif_chunk, 4095);               // gridDim.z of dif16_kernel, 16*chunk pseudo frames
      hipError_t he2;
      if ((he2 = hipMalloc(reinterpret_cast<void**>(&e->d_dif_z), per_frame * e->dif_chunk)) != hipSuccess)
        return bail(fail("hipMalloc(%zu) for the first-stage scratch: %s", per_frame * e->dif_chunk, hipGetErrorString(he2)));
      if ((he2 = hipMalloc(reinterpret_cast<void**>(&e->d_dif_y), (size_t)e->dif_chunk * n * 4)) != hipSuccess)
        return bail(fail("hipMalloc for the second-stage output: %s", hipGetErrorString(he2)));
    }
  }

  const size_t nn = (size_t)n;
  e->max_chunks = (int)std::max<size_t>(1, std::min<size_t>(128, (64u << 20) / (3 * nn * 4)));
  hipError_t he;
#define ALLOC(ptr, bytes)                                                        \
  if ((he = hipMalloc(reinterpret_cast<void**>(&(ptr)), (bytes))) != hipSuccess) \
    return bail(fail("hipMalloc(%zu) for %s: %s", (size_t)(bytes), #ptr, hipGetErrorString(he)));
  ALLOC(e->d_iq_stage, (size_t)cfg->full_size * 8);
  ALLOC(e->d_frames, (size_t)cfg->max_frames * nn * 4);
  ALLOC(e->d_part, (size_t)e->max_chunks * 3 * nn * 4);
  ALLOC(e->d_xchg, (4 * nn + (size_t)KSA_HM_ROWS * cfg->hm_width) * 4);
  e->d_partial = e->d_xchg;
  if (cfg->hm_width) e->d_hm = e->d_xchg + 4 * nn;
  ALLOC(e->d_state, 4 * nn * 4);
  ALLOC(e->d_parts, (size_t)e->num_cu * e->blocks_per_cu * (size_t)e->sub_n * 4);
  if (cfg->scan_total_entries) {
    ALLOC(e->d_scan_state, (size_t)4 * cfg->scan_total_entries * 4);
    ALLOC(e->d_scan_hm, (size_t)KSA_HM_ROWS * cfg->scan_hm_width * 4);
  }
#undef ALLOC
  if (ksa_reset_state(e)) return bail(1);
  if (cfg->scan_total_entries && scan_reset(e)) return bail(1);
  if (hipStreamSynchronize(e->stream) != hipSuccess) return bail(fail("initial sync failed"));
  *out = e;
  return 0;
}

void ksa_destroy(ksa_engine* e) {
  if (!e) return;
  DeviceGuard dev_guard;
  hipSetDevice(e->cfg.device);
  hipDeviceSynchronize();
  for (auto& pr : e->prof_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  for (hipEvent_t ev : {e->ev_ready, e->ev_copied, e->ev_stream}) if (ev) hipEventDestroy(ev);
  void* ptrs[] = {e->d_clk, e->d_gather, e->d_scan_stage, e->d_scan_rows, e->d_scan_halo, e->d_scan_send,
                  e->d_starts, e->d_start_last, e->d_window, e->d_window32, e->d_tw_mid, e->d_tw_last, e->d_adj, e->d_scan_adj,
                  e->d_iq_stage, e->d_frames, e->d_part, e->d_xchg, e->d_state, e->d_scan_state, e->d_scan_hm,
                  e->d_levels, e->d_parts, e->d_highs, e->d_scan_avg_rows, e->d_dif_tw, e->d_dif_z, e->d_dif_y, e->d_ones, e->d_starts_b};
  for (void* p : ptrs) if (p) hipFree(p);
  delete e;
}

int ksa_set_stream(ksa_engine* e, void* hip_stream) {
  if (!e) return fail("null engine");
  hipStream_t ns = reinterpret_cast<hipStream_t>(hip_stream);
  if (ns == e->stream) return 0;
  // engine-owned buffers (state, scratch, rings) may still be in use on the old stream: order the new one behind it
  // (an event is recorded on the OLD stream, so a stream handed in here must stay alive until the next ksa_set_stream /
  //  ksa_destroy of this engine)
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  if (!e->ev_stream) HIP_OK(hipEventCreateWithFlags(&e->ev_stream, hipEventDisableTiming));
  HIP_OK(hipEventRecord(e->ev_stream, e->stream));
  HIP_OK(hipStreamWaitEvent(ns, e->ev_stream, 0));
  e->stream = ns;
  return 0;
}

int ksa_synchronize(ksa_engine* e) {
  if (!e) return fail("null engine");
  // (the default engine stream is the NULL stream, which means "the null stream of the CURRENT device": select the engine's)
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_curscan_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride, int32_t nframes,
                    int32_t out_mode, float* out_dev) {
  if (!e || !iq_dev || !out_dev) return fail("null argument");
  if (out_mode < KSA_OUT_LINEAR || out_mode > KSA_OUT_DB_CLIP) return fail("unknown out_mode %d", out_mode);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  return run_spectrum(e, iq_dev, fmt, frame_stride, nframes, out_mode, out_dev, false, nullptr);
}

static int curscan_host(ksa_engine* e, const void* iq_host, int fmt, float* mag_host) {
  if (!e || !iq_host || !mag_host) return fail("null argument");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipMemcpyAsync(e->d_iq_stage, iq_host, (size_t)e->cfg.full_size * sample_bytes(fmt), hipMemcpyHostToDevice, e->stream));
  if (run_spectrum(e, e->d_iq_stage, fmt, 0, 1, KSA_OUT_LINEAR, e->d_frames, false, nullptr)) return 1;
  HIP_OK(hipMemcpyAsync(mag_host, e->d_frames, (size_t)e->cfg.fft_size * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_curscan_c64(ksa_engine* e, const float* iq_host, float* mag_host) { return curscan_host(e, iq_host, KSA_FMT_C64, mag_host); }
int ksa_curscan_u8(ksa_engine* e, const uint8_t* iq_host, float* mag_host) { return curscan_host(e, iq_host, KSA_FMT_U8, mag_host); }

int ksa_frames_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride, int32_t nframes,
                   int64_t first_index, int64_t total_frames, float* cur_db_dev, float* hm_rows_dev, int32_t commit) {
  if (!e || !iq_dev) return fail("null argument");
  if (first_index < 0 || first_index + nframes > total_frames) return fail("batch [%lld,+%d) outside run of %lld frames", (long long)first_index, nframes, (long long)total_frames);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  float* db = cur_db_dev ? cur_db_dev : e->d_frames;
  if (run_spectrum(e, iq_dev, fmt, frame_stride, nframes, KSA_OUT_DB, db, e->cfg.hm_width > 0, hm_rows_dev)) return 1;
  if (run_accumulate(e, db, nframes, first_index, total_frames)) return 1;
  e->pending_frames = nframes;
  if (commit) return do_commit(e, total_frames, nframes);
  return 0;
}

static int frame_host(ksa_engine* e, const void* iq_host, int fmt) {
  if (!e || !iq_host) return fail("null argument");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipMemcpyAsync(e->d_iq_stage, iq_host, (size_t)e->cfg.full_size * sample_bytes(fmt), hipMemcpyHostToDevice, e->stream));
  if (ksa_frames_dev(e, e->d_iq_stage, fmt, 0, 1, 0, 1, nullptr, nullptr, 1)) return 1;
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_frame_c64(ksa_engine* e, const float* iq_host) { return frame_host(e, iq_host, KSA_FMT_C64); }
int ksa_frame_u8(ksa_engine* e, const uint8_t* iq_host) { return frame_host(e, iq_host, KSA_FMT_U8); }

int ksa_frame_spectrum(ksa_engine* e, const float* mag_host) {
  if (!e || !mag_host) return fail("null argument");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  const int n = e->cfg.fft_size;
  float* lin = reinterpret_cast<float*>(e->d_iq_stage);  // full_size*8 bytes >= N*4
  HIP_OK(hipMemcpyAsync(lin, mag_host, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  ksa::DbRowParams p{};
  p.lin = lin;
  p.out = e->d_frames;
  p.n = n;
  p.nframes = 1;
  p.gain = e->cfg.gain;
  p.hm_w = e->cfg.hm_width;
  p.adj = e->d_adj;
  p.hm_rows = nullptr;
  p.hm_ring = e->d_hm;
  p.hm_index0 = e->hm_index;
  p.hm_first = 0;
  hipLaunchKernelGGL(ksa::db_rows_kernel, dim3(std::max(1, std::min(64, n / 256)), 1), dim3(256), 0, e->stream, p);
  HIP_OK(hipGetLastError());
  if (run_accumulate(e, e->d_frames, 1, 0, 1)) return 1;
  if (do_commit(e, 1, 1)) return 1;
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_partial_dev(ksa_engine* e, float** partial_dev) {
  if (!e || !partial_dev) return fail("null argument");
  *partial_dev = e->d_partial;
  return 0;
}

int ksa_exchange_dev(ksa_engine* e, float** xchg_dev, int64_t* nfloats) {
  if (!e || !xchg_dev || !nfloats) return fail("null argument");
  *xchg_dev = e->d_xchg;
  *nfloats = 4ll * e->cfg.fft_size + (long long)KSA_HM_ROWS * e->cfg.hm_width;
  return 0;
}

int ksa_merge_gathered_dev(ksa_engine* e, const float* gathered_dev, int32_t world, int32_t frames_per_rank,
                           int32_t hm_index0) {
  if (!e || !gathered_dev) return fail("null argument");
  if (world < 1) return fail("world %d < 1", world);
  if (frames_per_rank < 1) return fail("frames_per_rank %d < 1", frames_per_rank);
  if (e->pending_frames != frames_per_rank)
    return fail("ksa_merge_gathered_dev: %d frames pending, frames_per_rank says %d", e->pending_frames, frames_per_rank);
  if (hm_index0 < 0 || hm_index0 >= KSA_HM_ROWS) return fail("hm_index0 %d outside 0..127", hm_index0);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  const int n = e->cfg.fft_size, w = e->cfg.hm_width;
  const long long stride = 4ll * n + (long long)KSA_HM_ROWS * w;
  const long long total = (long long)world * frames_per_rank;
  const long long cells = std::max<long long>(n, (long long)KSA_HM_ROWS * w);
  hipLaunchKernelGGL(ksa::merge_gathered_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, e->stream,
                     gathered_dev, world, stride, n, e->d_partial, e->d_hm, w, hm_index0, frames_per_rank, total);
  HIP_OK(hipGetLastError());
  if (do_commit(e, total, 0)) return 1;
  e->hm_index = (int)((hm_index0 + total) % KSA_HM_ROWS);
  return 0;
}

int ksa_commit(ksa_engine* e, int64_t total_frames) {
  if (!e) return fail("null engine");
  if (e->pending_frames <= 0) return fail("ksa_commit without a pending ksa_frames_dev(commit=0)");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  return do_commit(e, total_frames, e->pending_frames);
}

int ksa_set_flags(ksa_engine* e, int32_t b_max, int32_t b_min, int32_t b_avg) {
  if (!e) return fail("null engine");
  e->b_max = b_max != 0;
  e->b_min = b_min != 0;
  e->b_avg = b_avg != 0;
  return 0;
}

int ksa_set_adj(ksa_engine* e, int32_t scan, const float* adj_host, int32_t n) {
  if (!e) return fail("null engine");
  if (scan && !e->cfg.scan_total_entries) return fail("engine was created without scan geometry");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipStreamSynchronize(e->stream));
  float** slot = scan ? &e->d_scan_adj : &e->d_adj;
  if (*slot) hipFree(*slot);
  *slot = nullptr;
  if (!adj_host) return 0;
  const int want = scan ? e->cfg.scan_total_entries : e->cfg.fft_size;
  if (n != want) return fail("adj length %d, the %s baseline needs %d", n, scan ? "scan" : "zeroSpan", want);
  return upload(slot, adj_host, (size_t)n);
}

int ksa_reset_state(ksa_engine* e) {
  if (!e) return fail("null engine");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipMemsetAsync(e->d_state, 0, (size_t)4 * e->cfg.fft_size * 4, e->stream));
  if (e->d_hm) HIP_OK(hipMemsetAsync(e->d_hm, 0, (size_t)KSA_HM_ROWS * e->cfg.hm_width * 4, e->stream));  // np.zeros K:456
  e->frames_seen = 0;
  e->has_max = e->has_min = e->has_avg = 0;
  e->hm_index = 0;
  e->pending_frames = 0;
  return 0;
}

int ksa_read_state(ksa_engine* e, float* cur, float* max, float* min, float* avg, float* hm, int32_t* hm_index,
                   int64_t* frames_seen) {
  if (!e) return fail("null engine");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  if (hm && !e->d_hm) return fail("engine has no waterfall (hm_width 0)");      // (validated before the first copy is enqueued)
  const size_t nb = (size_t)e->cfg.fft_size * 4;
  float* dst[4] = {cur, max, min, avg};
  for (int i = 0; i < 4; ++i)
    if (dst[i]) HIP_OK(hipMemcpyAsync(dst[i], e->d_state + (size_t)i * e->cfg.fft_size, nb, hipMemcpyDeviceToHost, e->stream));
  if (hm) {
    HIP_OK(hipMemcpyAsync(hm, e->d_hm, (size_t)KSA_HM_ROWS * e->cfg.hm_width * 4, hipMemcpyDeviceToHost, e->stream));
  }
  HIP_OK(hipStreamSynchronize(e->stream));
  if (hm_index) *hm_index = e->hm_index;
  if (frames_seen) *frames_seen = e->frames_seen;
  return 0;
}

int ksa_state_dev(ksa_engine* e, float** state_dev, float** hm_ring_dev) {
  if (!e) return fail("null engine");
  if (state_dev) *state_dev = e->d_state;
  if (hm_ring_dev) *hm_ring_dev = e->d_hm;
  return 0;
}

int ksa_set_hm_index(ksa_engine* e, int32_t hm_index) {
  if (!e) return fail("null engine");
  if (hm_index < 0 || hm_index >= KSA_HM_ROWS) return fail("hm_index %d outside the ring", hm_index);
  e->hm_index = hm_index;
  return 0;
}

// Stitch + Max/Min/Avg over the elements [elem_lo, elem_hi) from the bands [step_lo, step_hi) (+ nhalo halo bands in
// front of them).  ranged == false: the whole range on one engine, the waterfall rows go straight into the ring;
// ranged == true: partial rows into d_scan_rows for ksa_scan_merge_rows_dev.
static int scan_stitch(ksa_engine* e, const float* step_db_dev, int nsteps, int npasses, bool ranged = false,
                       const float* halo_db_dev = nullptr, int nhalo = 0, int step_lo = 0, int step_hi = -1,
                       int elem_lo = 0, int elem_hi = -1, int own_band_major = 0) {
  if (!e || !step_db_dev) return fail("null argument");
  const ksa_config& c = e->cfg;
  if (!c.scan_total_entries) return fail("engine was created without scan geometry");
  if (nsteps < 1 || npasses < 1) return fail("nsteps and npasses must be >= 1");
  if (step_hi < 0) step_hi = nsteps;
  if (elem_hi < 0) elem_hi = c.scan_total_entries;
  HIP_OK(hipSetDevice(c.device));          // (the exported caller holds the DeviceGuard)
  ksa::StitchParams s{};
  s.step_db = step_db_dev;
  s.halo_db = halo_db_dev;
  s.n = c.fft_size;
  s.nsteps = nsteps;
  s.hop = c.scan_hop;
  s.total = c.scan_total_entries;
  s.step_lo = step_lo;
  s.own_steps = step_hi - step_lo;
  s.nhalo = nhalo;
  s.own_band_major = own_band_major;
  s.e_lo = elem_lo;
  s.e_hi = elem_hi;
  s.state = e->d_scan_state;
  s.first_pass = e->scan_passes == 0;
  s.b_max = e->b_max;
  s.b_min = e->b_min;
  s.base_is_raw = e->scan_base_is_raw;
  s.npasses = npasses;
  const int rows = std::min(npasses, KSA_HM_ROWS);       // only the last 128 passes of a batch reach the ring
  if (npasses > 1 || ranged) {
    // Fft.Avg after each of those passes: the waterfall row of a pass is built from it (K:696-697)
    if (!e->d_scan_avg_rows)
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_scan_avg_rows), (size_t)KSA_HM_ROWS * s.total * 4));
    s.avg_rows = e->d_scan_avg_rows;
    s.avg_row0 = npasses - rows;
  }
  const int tb = 256;
  const int elems = elem_hi - elem_lo;
  if (elems > 0) hipLaunchKernelGGL(ksa::scan_stitch_kernel, dim3((elems + tb - 1) / tb), dim3(tb), 0, e->stream, s);
  const int g = c.scan_total_entries / c.scan_hm_width;
  if (ranged) {
    if (!e->d_scan_rows) HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_scan_rows), (size_t)KSA_HM_ROWS * c.scan_hm_width * 4));
    hipLaunchKernelGGL(ksa::rowmax_rows_range_kernel, dim3(c.scan_hm_width, rows), dim3(64), 0, e->stream,
                       e->d_scan_avg_rows, e->d_scan_adj, c.scan_hm_width, g, elem_lo, elem_hi, e->d_scan_rows);
    HIP_OK(hipGetLastError());
    e->scan_rows = rows;
    e->scan_rows_passes = npasses;
    e->scan_passes += npasses;       // (the ring advances in ksa_scan_merge_rows_dev)
    return 0;
  }
  if (npasses == 1) {
    hipLaunchKernelGGL(ksa::rowmax_kernel, dim3((c.scan_hm_width + tb - 1) / tb), dim3(tb), 0, e->stream,
                       e->d_scan_state + (size_t)3 * s.total, e->d_scan_adj, c.scan_hm_width, g,
                       e->d_scan_hm + (size_t)e->scan_hm_index * c.scan_hm_width);
  } else {
    hipLaunchKernelGGL(ksa::rowmax_rows_kernel, dim3(c.scan_hm_width, rows), dim3(64), 0, e->stream,
                       e->d_scan_avg_rows, e->d_scan_adj, c.scan_hm_width, g, e->d_scan_hm,
                       (e->scan_hm_index + s.avg_row0) % KSA_HM_ROWS);
  }
  HIP_OK(hipGetLastError());
  e->scan_passes += npasses;
  e->scan_hm_index = (e->scan_hm_index + npasses) % KSA_HM_ROWS;  // K:732, once per pass
  return 0;
}

// Bands in front of step_lo that still cover elements from step_lo*hop on: ceil(N/hop) - 1, at most step_lo.
static int halo_bands(const ksa_config& c, int step_lo) { return std::min(step_lo, (c.fft_size + c.scan_hop - 1) / c.scan_hop - 1); }

int ksa_scan_stitch_range_dev(ksa_engine* e, const float* own_db_dev, int32_t own_band_major, const float* halo_db_dev,
                              int32_t nhalo, int32_t step_lo, int32_t step_hi, int32_t nsteps, int32_t npasses,
                              int32_t elem_lo, int32_t elem_hi) {
  if (!e) return fail("null engine");
  const ksa_config& c = e->cfg;
  if (!c.scan_total_entries) return fail("engine was created without scan geometry");
  if (nsteps < 1 || npasses < 1) return fail("nsteps and npasses must be >= 1");
  if (step_lo < 0 || step_hi < step_lo || step_hi > nsteps) return fail("bands [%d,%d) outside the pass of %d", step_lo, step_hi, nsteps);
  if (elem_lo < 0 || elem_hi < elem_lo || elem_hi > c.scan_total_entries)
    return fail("elements [%d,%d) outside the stitched range of %d", elem_lo, elem_hi, c.scan_total_entries);
  if (nhalo < 0 || nhalo > step_lo) return fail("nhalo %d outside 0..step_lo(%d)", nhalo, step_lo);
  if (elem_hi > elem_lo) {
    // every band that covers an owned element must be at hand: i0(elem_lo) >= step_lo - nhalo, i1(elem_hi-1) < step_hi
    const int n = c.fft_size, hop = c.scan_hop;
    const int i0 = elem_lo - n + 1 <= 0 ? 0 : (elem_lo - n + hop) / hop;
    const int i1 = std::min((elem_hi - 1) / hop, nsteps - 1);
    if (i0 <= i1 && (i0 < step_lo - nhalo || i1 >= step_hi))
      return fail("elements [%d,%d) are covered by bands %d..%d, at hand are %d..%d", elem_lo, elem_hi, i0, i1, step_lo - nhalo, step_hi - 1);
    if (step_hi > step_lo && !own_db_dev) return fail("null own_db_dev");
    if (nhalo > 0 && !halo_db_dev) return fail("null halo_db_dev");
  }
  const float* own = own_db_dev ? own_db_dev : e->d_scan_state;   // (an engine that owns no band reads nothing)
  DeviceGuard dev_guard;
  return scan_stitch(e, own, nsteps, npasses, true, halo_db_dev, nhalo, step_lo, step_hi, elem_lo, elem_hi, own_band_major != 0);
}

int ksa_scan_rows_dev(ksa_engine* e, float** rows_dev, int32_t* rows) {
  if (!e || !rows_dev || !rows) return fail("null argument");
  if (!e->d_scan_rows || e->scan_rows <= 0) return fail("no band-sharded batch pending (ksa_scan_stitch_range_dev first)");
  *rows_dev = e->d_scan_rows;
  *rows = e->scan_rows;
  return 0;
}

int ksa_scan_merge_rows_dev(ksa_engine* e, const float* gathered_dev, int32_t world, int32_t rows, int32_t npasses) {
  if (!e || !gathered_dev) return fail("null argument");
  const ksa_config& c = e->cfg;
  if (!c.scan_total_entries) return fail("engine was created without scan geometry");
  if (world < 1) return fail("world %d < 1", world);
  if (rows != e->scan_rows || npasses != e->scan_rows_passes)
    return fail("ksa_scan_merge_rows_dev: %d rows of %d passes pending, call says %d of %d", e->scan_rows, e->scan_rows_passes, rows, npasses);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(c.device));
  const int cells = rows * c.scan_hm_width;
  hipLaunchKernelGGL(ksa::scan_merge_rows_kernel, dim3((cells + 255) / 256), dim3(256), 0, e->stream, gathered_dev, world,
                     rows, c.scan_hm_width, e->d_scan_hm, (e->scan_hm_index + npasses - rows) % KSA_HM_ROWS);
  HIP_OK(hipGetLastError());
  e->scan_hm_index = (e->scan_hm_index + npasses) % KSA_HM_ROWS;
  e->scan_rows = e->scan_rows_passes = 0;
  return 0;
}

// device-to-device copy issued on `s` (a stream of dst_dev): peer copy when the devices differ
static int copy_d2d(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s) {
  if (!bytes) return 0;
  if (dst_dev == src_dev) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
  else HIP_OK(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s));
  return 0;
}

static int ensure_events(ksa_engine* e) {
  if (!e->ev_ready) HIP_OK(hipEventCreateWithFlags(&e->ev_ready, hipEventDisableTiming));
  if (!e->ev_copied) HIP_OK(hipEventCreateWithFlags(&e->ev_copied, hipEventDisableTiming));
  return 0;
}

// every engine -> every engine: engine i's stream receives block_of(j) of all j into its d_gather[j], behind the
// producers (ev_ready) ; ev_copied_i marks the end of i's copies so that no producer overwrites a block too early
static int gather_all(ksa_engine* const* h, int n, size_t nfloats, float* (*block_of)(ksa_engine*)) {
  for (int j = 0; j < n; ++j) {
    HIP_OK(hipSetDevice(h[j]->cfg.device));
    if (ensure_events(h[j])) return 1;
    HIP_OK(hipEventRecord(h[j]->ev_ready, h[j]->stream));
  }
  for (int i = 0; i < n; ++i) {
    ksa_engine* e = h[i];
    HIP_OK(hipSetDevice(e->cfg.device));
    if (ensure(&e->d_gather, &e->gather_cap, (size_t)n * nfloats)) return 1;
    for (int j = 0; j < n; ++j) {
      if (j != i) HIP_OK(hipStreamWaitEvent(e->stream, h[j]->ev_ready, 0));
      if (copy_d2d(e->d_gather + (size_t)j * nfloats, e->cfg.device, block_of(h[j]), h[j]->cfg.device, nfloats * 4, e->stream)) return 1;
    }
    HIP_OK(hipEventRecord(e->ev_copied, e->stream));
  }
  for (int i = 0; i < n; ++i) {
    HIP_OK(hipSetDevice(h[i]->cfg.device));
    for (int j = 0; j < n; ++j)
      if (j != i) HIP_OK(hipStreamWaitEvent(h[i]->stream, h[j]->ev_copied, 0));
  }
  return 0;
}

// Direct xGMI copies between the engines' GPUs where the hardware allows them (hipMemcpyPeerAsync works either way,
// through host staging otherwise).  Best effort: "already enabled" and "not supported" are not errors here.  Every
// ordered device pair is tried ONCE per process (a bit per pair), not on every multi-engine call.
static void enable_peer_access(ksa_engine* const* h, int n) {
  static std::mutex mu;
  static std::vector<unsigned char> tried;       // [a * ndev + b]
  static int ndev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!ndev) {
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { ndev = 0; (void)hipGetLastError(); return; }
    tried.assign((size_t)ndev * ndev, 0);
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const int a = h[i]->cfg.device, b = h[j]->cfg.device;
      if (a == b || a >= ndev || b >= ndev || tried[(size_t)a * ndev + b]) continue;
      tried[(size_t)a * ndev + b] = 1;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
      if (hipSetDevice(a) == hipSuccess) (void)hipDeviceEnablePeerAccess(b, 0);
      (void)hipGetLastError();
    }
}

static int check_handles(ksa_engine* const* h, int n) {
  if (!h || n < 1) return fail("need at least one engine handle");
  for (int i = 0; i < n; ++i) {
    if (!h[i]) return fail("engine handle %d is null", i);
    for (int j = 0; j < i; ++j) if (h[j] == h[i]) return fail("engine handle %d is handle %d again", i, j);
    const ksa_config &a = h[0]->cfg, &b = h[i]->cfg;
    if (a.fft_size != b.fft_size || a.hm_width != b.hm_width || a.scan_total_entries != b.scan_total_entries ||
        a.scan_hop != b.scan_hop || a.scan_hm_width != b.scan_hm_width)
      return fail("engine %d has a different geometry than engine 0", i);
  }
  enable_peer_access(h, n);
  return 0;
}

int ksa_allreduce_state(ksa_engine* const* handles, int32_t n, int32_t frames_per_rank, int32_t hm_index0) {
  DeviceGuard dev_guard;
  // everything is validated before the first copy or launch: a refusal leaves every engine as it was
  if (check_handles(handles, n)) return 1;
  if (frames_per_rank < 1) return fail("frames_per_rank %d < 1", frames_per_rank);
  if (hm_index0 < 0 || hm_index0 >= KSA_HM_ROWS) return fail("hm_index0 %d outside 0..127", hm_index0);
  for (int i = 0; i < n; ++i)
    if (handles[i]->pending_frames != frames_per_rank)
      return fail("engine %d has %d frames pending, frames_per_rank says %d", i, handles[i]->pending_frames, frames_per_rank);
  const size_t nfloats = 4 * (size_t)handles[0]->cfg.fft_size + (size_t)KSA_HM_ROWS * handles[0]->cfg.hm_width;
  if (gather_all(handles, n, nfloats, +[](ksa_engine* e) { return e->d_xchg; })) return 1;
  for (int i = 0; i < n; ++i)
    if (ksa_merge_gathered_dev(handles[i], handles[i]->d_gather, n, frames_per_rank, hm_index0)) return 1;
  return 0;
}

int ksa_scan_allstitch(ksa_engine* const* handles, int32_t n, float* const* own_db_dev, int32_t nsteps, int32_t npasses) {
  DeviceGuard dev_guard;
  if (check_handles(handles, n)) return 1;
  if (!own_db_dev) return fail("null own_db_dev");
  const ksa_config& c = handles[0]->cfg;
  if (!c.scan_total_entries) return fail("engines were created without scan geometry");
  if (nsteps < 1 || npasses < 1) return fail("nsteps and npasses must be >= 1");
  const size_t band = (size_t)npasses * c.fft_size;   // floats of one band over the batch
  const int reach = (c.fft_size + c.scan_hop - 1) / c.scan_hop - 1;   // bands in front of a band that still overlap it
  auto lo_of = [&](int r) { return (int)((long long)nsteps * r / n); };
  // 0. every pointer and range is checked before the first launch: a refusal at rank k must not leave ranks < k with
  //    their pass counters advanced and their state stitched while the ring is never merged
  for (int r = 0; r < n; ++r) {
    const int lo = lo_of(r), hi = lo_of(r + 1);
    if (hi > lo && !own_db_dev[r]) return fail("own_db_dev[%d] is null (rank %d owns bands %d..%d)", r, r, lo, hi - 1);
    if (handles[r]->scan_rows > 0) return fail("engine %d still holds unmerged partial rows of an earlier batch", r);
    const long long e_lo = (long long)lo * c.scan_hop;
    if (r < n - 1 && hi > lo && e_lo > c.scan_total_entries) return fail("rank %d's bands start past the stitched range", r);
  }
  // 1. every engine packs the bands its right neighbours need ([band][npasses][N]) behind its spectrum stage
  for (int r = 0; r < n; ++r) {
    ksa_engine* e = handles[r];
    const int lo = lo_of(r), hi = lo_of(r + 1), mine = hi - lo;
    HIP_OK(hipSetDevice(e->cfg.device));
    if (ensure_events(e)) return 1;
    const int keep = std::min(mine, reach);            // its last bands may be someone's halo
    if (keep > 0 && r + 1 < n) {
      if (ensure(&e->d_scan_send, &e->scan_send_cap, (size_t)keep * band)) return 1;
      for (int b = 0; b < keep; ++b)     // band hi-keep+b of every pass -> send[b][pass][N]
        HIP_OK(hipMemcpy2DAsync(e->d_scan_send + (size_t)b * band, (size_t)c.fft_size * 4,
                                own_db_dev[r] + (size_t)(mine - keep + b) * c.fft_size, (size_t)mine * c.fft_size * 4,
                                (size_t)c.fft_size * 4, (size_t)npasses, hipMemcpyDeviceToDevice, e->stream));
    }
    HIP_OK(hipEventRecord(e->ev_ready, e->stream));
  }
  // From here on engines are being modified: a failure must not leave partial rows behind that would make every later
  // ksa_scan_allstitch refuse ("still holds unmerged partial rows").
  auto abandon = [&]() { for (int r = 0; r < n; ++r) handles[r]->scan_rows = handles[r]->scan_rows_passes = 0; return 1; };
  // 2. halos: band j of rank r's halo comes from the rank that owns j
  for (int r = 0; r < n; ++r) {
    ksa_engine* e = handles[r];
    const int lo = lo_of(r), hi = lo_of(r + 1);
    const int nhalo = halo_bands(c, lo);
    HIP_OK(hipSetDevice(e->cfg.device));
    if (nhalo > 0 && ensure(&e->d_scan_halo, &e->scan_halo_cap, (size_t)nhalo * band)) return 1;
    for (int j = lo - nhalo; j < lo; ++j) {
      int src = r - 1;
      while (src > 0 && lo_of(src) > j) --src;
      ksa_engine* s = handles[src];
      const int shi = lo_of(src + 1), smine = shi - lo_of(src);
      const int skeep = std::min(smine, reach);
      const int b = j - (shi - skeep);                 // position of band j in src's send block
      if (b < 0) return fail("internal: band %d is not in rank %d's send block", j, src);
      HIP_OK(hipStreamWaitEvent(e->stream, s->ev_ready, 0));
      if (copy_d2d(e->d_scan_halo + (size_t)(j - (lo - nhalo)) * band, e->cfg.device, s->d_scan_send + (size_t)b * band,
                   s->cfg.device, band * 4, e->stream)) return 1;
    }
    const int elem_lo = lo * c.scan_hop, elem_hi = r == n - 1 ? c.scan_total_entries : std::min(c.scan_total_entries, hi * c.scan_hop);
    if (ksa_scan_stitch_range_dev(e, own_db_dev[r], 0, e->d_scan_halo, nhalo, lo, hi, nsteps, npasses, std::min(elem_lo, elem_hi), elem_hi)) return abandon();
  }
  // 3. partial waterfall rows: all-to-all copies, merged on every engine
  const int rows = std::min(npasses, KSA_HM_ROWS);
  const size_t nfloats = (size_t)rows * c.scan_hm_width;
  if (gather_all(handles, n, nfloats, +[](ksa_engine* e) { return e->d_scan_rows; })) return abandon();
  for (int i = 0; i < n; ++i)
    if (ksa_scan_merge_rows_dev(handles[i], handles[i]->d_gather, n, rows, npasses)) return abandon();
  return 0;
}

int ksa_scan_gather_state(ksa_engine* const* handles, int32_t n, int32_t nsteps, float* cur, float* max, float* min, float* avg) {
  DeviceGuard dev_guard;
  if (check_handles(handles, n)) return 1;
  const ksa_config& c = handles[0]->cfg;
  if (!c.scan_total_entries) return fail("engines were created without scan geometry");
  if (nsteps < 1) return fail("nsteps must be >= 1");
  float* dst[4] = {cur, max, min, avg};
  const size_t t = (size_t)c.scan_total_entries;
  for (int r = 0; r < n; ++r) {
    ksa_engine* e = handles[r];
    const long long lo = (long long)nsteps * r / n, hi = (long long)nsteps * (r + 1) / n;
    const size_t e_lo = std::min<size_t>(t, (size_t)lo * c.scan_hop), e_hi = r == n - 1 ? t : std::min<size_t>(t, (size_t)hi * c.scan_hop);
    HIP_OK(hipSetDevice(e->cfg.device));
    for (int k = 0; k < 4; ++k)
      if (dst[k] && e_hi > e_lo)
        HIP_OK(hipMemcpyAsync(dst[k] + e_lo, e->d_scan_state + k * t + e_lo, (e_hi - e_lo) * 4, hipMemcpyDeviceToHost, e->stream));
  }
  for (int r = 0; r < n; ++r) {
    HIP_OK(hipSetDevice(handles[r]->cfg.device));
    HIP_OK(hipStreamSynchronize(handles[r]->stream));
  }
  return 0;
}

static int scan_pass_host(ksa_engine* e, const void* iq_host, int fmt, int nsteps, const uint8_t* step_ok) {
  if (!e || !iq_host) return fail("null argument");
  if (!e->cfg.scan_total_entries) return fail("engine was created without scan geometry");
  if (nsteps < 1 || nsteps > e->cfg.max_frames) return fail("nsteps %d outside 1..max_frames(%d)", nsteps, e->cfg.max_frames);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  const size_t bytes = (size_t)nsteps * e->cfg.full_size * sample_bytes(fmt);
  if (ensure(reinterpret_cast<unsigned char**>(&e->d_scan_stage), &e->scan_stage_cap, bytes)) return 1;
  HIP_OK(hipMemcpyAsync(e->d_scan_stage, iq_host, bytes, hipMemcpyHostToDevice, e->stream));
  if (ksa_scan_passes_dev(e, e->d_scan_stage, fmt, e->cfg.full_size, nsteps, 1, step_ok)) return 1;
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_scan_pass_c64(ksa_engine* e, const float* iq_host, int32_t nsteps, const uint8_t* step_ok) {
  return scan_pass_host(e, iq_host, KSA_FMT_C64, nsteps, step_ok);
}
int ksa_scan_pass_u8(ksa_engine* e, const uint8_t* iq_host, int32_t nsteps, const uint8_t* step_ok) {
  return scan_pass_host(e, iq_host, KSA_FMT_U8, nsteps, step_ok);
}

int ksa_host_alloc(void** out, int64_t bytes) {
  if (!out || bytes < 1) return fail("ksa_host_alloc: null pointer or size < 1");
  *out = nullptr;
  HIP_OK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  return 0;
}
int ksa_host_free(void* p) {
  if (p) HIP_OK(hipHostFree(p));
  return 0;
}

int ksa_scan_stitch_dev(ksa_engine* e, const float* step_db_dev, int32_t nsteps) {
  DeviceGuard dev_guard;
  return scan_stitch(e, step_db_dev, nsteps, 1);
}

int ksa_scan_stitch_passes_dev(ksa_engine* e, const float* step_db_dev, int32_t nsteps, int32_t npasses) {
  DeviceGuard dev_guard;
  return scan_stitch(e, step_db_dev, nsteps, npasses);
}

// Clip2MinAmp + LogNoGain spectra of `frames` capture blocks (K:640-641) into out_dev[frames][N]; a block whose
// step_ok entry is 0 is replaced by the dummy band: ones(fftSize) through the same two steps (K:637-641)
static int scan_spectra(ksa_engine* e, const void* iq_dev, int fmt, long long frame_stride, int frames, const uint8_t* step_ok,
                        float* out_dev) {
  if (run_spectrum(e, iq_dev, fmt, frame_stride, frames, KSA_OUT_DB_CLIP, out_dev, false, nullptr)) return 1;
  if (step_ok) {
    const float v = (float)(10.0 * std::log10(std::max(1.0, (double)e->cfg.min_amp)) - (double)e->cfg.gain);
    for (long long s = 0; s < frames; ++s)
      if (!step_ok[s] && fill(e, out_dev + (size_t)s * e->cfg.fft_size, e->cfg.fft_size, v)) return 1;
  }
  return 0;
}

int ksa_scan_spectra_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride, int32_t nframes,
                         const uint8_t* step_ok, float* out_dev) {
  if (!e || !iq_dev || !out_dev) return fail("null argument");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  return scan_spectra(e, iq_dev, fmt, frame_stride, nframes, step_ok, out_dev);
}

int ksa_scan_passes_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride, int32_t nsteps,
                        int32_t npasses, const uint8_t* step_ok) {
  if (!e || !iq_dev) return fail("null argument");
  if (!e->cfg.scan_total_entries) return fail("engine was created without scan geometry");
  if (nsteps < 1 || npasses < 1) return fail("nsteps and npasses must be >= 1");
  const long long frames = (long long)nsteps * npasses;
  if (frames > e->cfg.max_frames) return fail("%d passes x %d steps exceed max_frames %d", npasses, nsteps, e->cfg.max_frames);
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  if (scan_spectra(e, iq_dev, fmt, frame_stride, (int)frames, step_ok, e->d_frames)) return 1;
  return scan_stitch(e, e->d_frames, nsteps, npasses);
}

int ksa_scan_pass_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride, int32_t nsteps,
                      const uint8_t* step_ok) {
  return ksa_scan_passes_dev(e, iq_dev, fmt, frame_stride, nsteps, 1, step_ok);
}

int ksa_scan_read_state(ksa_engine* e, float* cur, float* max, float* min, float* avg, float* hm, int32_t* hm_index,
                        int64_t* passes) {
  if (!e) return fail("null engine");
  if (!e->cfg.scan_total_entries) return fail("engine was created without scan geometry");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  const size_t t = (size_t)e->cfg.scan_total_entries;
  float* dst[4] = {cur, max, min, avg};
  for (int i = 0; i < 4; ++i)
    if (dst[i]) HIP_OK(hipMemcpyAsync(dst[i], e->d_scan_state + i * t, t * 4, hipMemcpyDeviceToHost, e->stream));
  if (hm) HIP_OK(hipMemcpyAsync(hm, e->d_scan_hm, (size_t)KSA_HM_ROWS * e->cfg.scan_hm_width * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));
  if (hm_index) *hm_index = e->scan_hm_index;
  if (passes) *passes = e->scan_passes;
  return 0;
}

int ksa_scan_state_dev(ksa_engine* e, float** state_dev, float** hm_ring_dev) {
  if (!e) return fail("null engine");
  if (state_dev) *state_dev = e->d_scan_state;
  if (hm_ring_dev) *hm_ring_dev = e->d_scan_hm;
  return 0;
}

int ksa_scan_set_base_is_raw(ksa_engine* e, int32_t on) {
  if (!e) return fail("null engine");
  e->scan_base_is_raw = on != 0;
  return 0;
}

int ksa_scan_reset(ksa_engine* e) {
  if (!e) return fail("null engine");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  return scan_reset(e);
}

static int levels_to_scratch(ksa_engine* e, int scan, int mode, int cells) {
  if (mode < 0 || mode > 2) return fail("levels mode %d (0 AVG, 1 MAX, 2 MIN)", mode);
  const int n = scan ? e->cfg.scan_total_entries : e->cfg.fft_size;
  if (scan && !n) return fail("engine was created without scan geometry");
  if (cells < 1 || n % cells) return fail("cells %d must divide %d", cells, n);
  HIP_OK(hipSetDevice(e->cfg.device));     // (the exported caller holds the DeviceGuard)
  if (!e->d_levels || e->levels_cap < cells) {
    if (e->d_levels) hipFree(e->d_levels);
    e->d_levels = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_levels), (size_t)4 * cells * 4));
    e->levels_cap = cells;
  }
  const int tb = 128;
  hipLaunchKernelGGL(ksa::levels_kernel, dim3((cells + tb - 1) / tb, 4), dim3(tb), 0, e->stream,
                     scan ? e->d_scan_state : e->d_state, scan ? e->d_scan_adj : e->d_adj, n, cells, mode, e->d_levels);
  HIP_OK(hipGetLastError());
  return 0;
}

int ksa_read_levels(ksa_engine* e, int32_t scan, int32_t mode, int32_t cells, float* out_host) {
  if (!e || !out_host) return fail("null argument");
  DeviceGuard dev_guard;
  if (levels_to_scratch(e, scan, mode, cells)) return 1;
  HIP_OK(hipMemcpyAsync(out_host, e->d_levels, (size_t)4 * cells * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_read_highs(ksa_engine* e, int32_t scan, int32_t mode, int32_t cells, int32_t curve, double min_sep_cells,
                   int32_t count, int32_t* idx_host, float* lvl_host, int32_t* found) {
  if (!e || !idx_host || !lvl_host || !found) return fail("null argument");
  if (curve < 0 || curve > 3) return fail("curve %d (0 cur, 1 max, 2 min, 3 avg)", curve);
  if (count < 1 || count > ksa::HIGHS_MAX) return fail("marker count %d outside 1..%d", count, ksa::HIGHS_MAX);
  if (!(min_sep_cells >= 0.0)) return fail("min_sep_cells must be >= 0");
  DeviceGuard dev_guard;
  if (levels_to_scratch(e, scan, mode, cells)) return 1;
  if (!e->d_highs) HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_highs), (2 * ksa::HIGHS_MAX + 1) * 4));
  ksa::HighsParams h{};
  h.lv = e->d_levels + (size_t)curve * cells;
  h.cells = cells;
  h.min_sep = min_sep_cells;
  h.count = count;
  h.idx = e->d_highs;
  h.lvl = reinterpret_cast<float*>(e->d_highs + ksa::HIGHS_MAX);
  h.found = e->d_highs + 2 * ksa::HIGHS_MAX;
  hipLaunchKernelGGL(ksa::highs_kernel, dim3(1), dim3(1024), 0, e->stream, h);
  HIP_OK(hipGetLastError());
  int host[2 * ksa::HIGHS_MAX + 1];
  HIP_OK(hipMemcpyAsync(host, e->d_highs, sizeof host, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));
  *found = host[2 * ksa::HIGHS_MAX];
  for (int i = 0; i < *found; ++i) {
    idx_host[i] = host[i];
    memcpy(&lvl_host[i], &host[ksa::HIGHS_MAX + i], 4);
  }
  return 0;
}

// rows [row0, row0 + nrows) (mod 128) of a waterfall ring -> host
static int copy_ring_rows(ksa_engine* e, int scan, int row0, int nrows, float* out_host) {
  const float* ring = scan ? e->d_scan_hm : e->d_hm;
  const int w = scan ? e->cfg.scan_hm_width : e->cfg.hm_width;
  if (!ring || w < 1) return fail("engine has no %s waterfall", scan ? "scan" : "zeroSpan");
  if (row0 < 0 || row0 >= KSA_HM_ROWS || nrows < 0 || nrows > KSA_HM_ROWS) return fail("ring rows [%d,+%d) outside 0..127", row0, nrows);
  const int first = std::min(nrows, KSA_HM_ROWS - row0);
  if (first > 0) HIP_OK(hipMemcpyAsync(out_host, ring + (size_t)row0 * w, (size_t)first * w * 4, hipMemcpyDeviceToHost, e->stream));
  if (nrows > first)
    HIP_OK(hipMemcpyAsync(out_host + (size_t)first * w, ring, (size_t)(nrows - first) * w * 4, hipMemcpyDeviceToHost, e->stream));
  return 0;
}

int ksa_read_hm_rows(ksa_engine* e, int32_t scan, int32_t row0, int32_t nrows, float* out_host) {
  if (!e || !out_host) return fail("null argument");
  if (scan && !e->cfg.scan_total_entries) return fail("engine was created without scan geometry");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  if (copy_ring_rows(e, scan, row0, nrows, out_host)) return 1;
  HIP_OK(hipStreamSynchronize(e->stream));
  return 0;
}

int ksa_read_view(ksa_engine* e, int32_t scan, int32_t mode, int32_t cells, float* levels_host, int32_t curve,
                  double min_sep_cells, int32_t count, int32_t* idx_host, float* lvl_host, int32_t* found,
                  int32_t hm_rows, float* hm_rows_host, int32_t* hm_index) {
  if (!e || !levels_host) return fail("null argument");
  if (count > 0 && (!idx_host || !lvl_host || !found)) return fail("null marker outputs");
  if (count < 0 || count > ksa::HIGHS_MAX) return fail("marker count %d outside 0..%d", count, ksa::HIGHS_MAX);
  if (count > 0 && (curve < 0 || curve > 3)) return fail("curve %d (0 cur, 1 max, 2 min, 3 avg)", curve);
  if (count > 0 && !(min_sep_cells >= 0.0)) return fail("min_sep_cells must be >= 0");
  if (hm_rows < 0 || hm_rows > KSA_HM_ROWS || (hm_rows > 0 && !hm_rows_host)) return fail("hm_rows %d outside 0..128 or null buffer", hm_rows);
  if (scan && !e->cfg.scan_total_entries) return fail("engine was created without scan geometry");
  if (hm_rows > 0 && (!(scan ? e->d_scan_hm : e->d_hm) || (scan ? e->cfg.scan_hm_width : e->cfg.hm_width) < 1))
    return fail("engine has no %s waterfall", scan ? "scan" : "zeroSpan");
  if (found) *found = 0;
  DeviceGuard dev_guard;
  if (levels_to_scratch(e, scan, mode, cells)) return 1;       // (selects the device; nothing is enqueued towards the host yet)
  // From the first asynchronous copy on, every exit synchronises the stream: the copies target caller buffers and a stack
  // array that must not be written after this function has returned.
  struct SyncOnExit {
    hipStream_t s; bool armed = true;
    ~SyncOnExit() { if (armed) (void)hipStreamSynchronize(s); }
  } drain{e->stream};
  HIP_OK(hipMemcpyAsync(levels_host, e->d_levels, (size_t)4 * cells * 4, hipMemcpyDeviceToHost, e->stream));
  int host[2 * ksa::HIGHS_MAX + 1];
  if (count > 0) {
    if (!e->d_highs) HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_highs), (2 * ksa::HIGHS_MAX + 1) * 4));
    ksa::HighsParams h{};
    h.lv = e->d_levels + (size_t)curve * cells;
    h.cells = cells;
    h.min_sep = min_sep_cells;
    h.count = count;
    h.idx = e->d_highs;
    h.lvl = reinterpret_cast<float*>(e->d_highs + ksa::HIGHS_MAX);
    h.found = e->d_highs + 2 * ksa::HIGHS_MAX;
    hipLaunchKernelGGL(ksa::highs_kernel, dim3(1), dim3(1024), 0, e->stream, h);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(host, e->d_highs, sizeof host, hipMemcpyDeviceToHost, e->stream));
  }
  const int idx = scan ? e->scan_hm_index : e->hm_index;       // the ring position AFTER the newest row
  if (hm_rows > 0 && copy_ring_rows(e, scan, ((idx - hm_rows) % KSA_HM_ROWS + KSA_HM_ROWS) % KSA_HM_ROWS, hm_rows, hm_rows_host)) return 1;
  drain.armed = false;
  HIP_OK(hipStreamSynchronize(e->stream));
  if (count > 0) {
    *found = host[2 * ksa::HIGHS_MAX];
    for (int i = 0; i < *found; ++i) {
      idx_host[i] = host[i];
      memcpy(&lvl_host[i], &host[ksa::HIGHS_MAX + i], 4);
    }
  }
  if (hm_index) *hm_index = idx;
  return 0;
}

int ksa_prof_enable(ksa_engine* e, int32_t on) {
  if (!e) return fail("null engine");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  for (auto& pr : e->prof_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  e->prof_events.clear();
  e->prof = on != 0;
  if (on) {
    const size_t bytes = (size_t)ksa::CLK_SLOTS * 2 * ksa::CLK_KEYS * 2 * sizeof(unsigned long long);
    if (!e->d_clk) HIP_OK(hipMalloc(reinterpret_cast<void**>(&e->d_clk), bytes));
    HIP_OK(hipMemsetAsync(e->d_clk, 0, bytes, e->stream));
    e->clk_launches = 0;
  }
  return 0;
}

int ksa_prof_read(ksa_engine* e, double* spectrum_ms, int64_t* launches) {
  if (!e) return fail("null engine");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipStreamSynchronize(e->stream));
  double ms = 0;
  for (auto& pr : e->prof_events) {
    float t = 0;
    HIP_OK(hipEventElapsedTime(&t, pr.first, pr.second));
    ms += t;
  }
  if (spectrum_ms) *spectrum_ms = ms;
  if (launches) *launches = (int64_t)e->prof_events.size();
  return 0;
}

int ksa_prof_clock(ksa_engine* e, double* shader_ghz, double* ghz_min, double* ghz_max, int64_t* samples) {
  if (!e || !shader_ghz) return fail("null argument");
  *shader_ghz = 0.0;
  if (ghz_min) *ghz_min = 0.0;
  if (ghz_max) *ghz_max = 0.0;
  if (samples) *samples = 0;
  if (!e->d_clk) return fail("ksa_prof_clock: profiling was never enabled on this engine");
  DeviceGuard dev_guard;
  HIP_OK(hipSetDevice(e->cfg.device));
  HIP_OK(hipStreamSynchronize(e->stream));
  const size_t per = (size_t)2 * ksa::CLK_KEYS * 2;      // qwords per profiled launch
  std::vector<unsigned long long> h((size_t)ksa::CLK_SLOTS * per);
  HIP_OK(hipMemcpy(h.data(), e->d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> ghz;
  for (int slot = 0; slot < ksa::CLK_SLOTS; ++slot)
    for (int key = 0; key < ksa::CLK_KEYS; ++key) {
      const unsigned long long* b0 = &h[slot * per + (size_t)key * 2];
      const unsigned long long* b1 = b0 + (size_t)ksa::CLK_KEYS * 2;
      const unsigned long long c0 = b0[0], r0 = b0[1], c1 = b1[0], r1 = b1[1];
      if (r0 && r1 > r0 && c1 > c0 && r1 - r0 >= 1000)    // this CU was stamped at both ends, >= 10 us apart (100 MHz ticks)
        ghz.push_back((double)(c1 - c0) / (double)(r1 - r0) * 0.1);
    }
  if (samples) *samples = (int64_t)ghz.size();
  if (ghz.empty()) return 0;
  std::sort(ghz.begin(), ghz.end());
  *shader_ghz = ghz[ghz.size() / 2];
  if (ghz_min) *ghz_min = ghz.front();
  if (ghz_max) *ghz_max = ghz.back();
  return 0;
}

int ksa_kernel_info(ksa_engine* e, int32_t* threads, int32_t* lds_bytes, int32_t* vgprs, int32_t* grid, int32_t* path) {
  if (!e) return fail("null engine");
  if (threads) *threads = e->threads;
  if (lds_bytes) *lds_bytes = e->lds_bytes;
  if (vgprs) *vgprs = e->vgprs;
  if (grid) *grid = e->num_cu * e->blocks_per_cu;
  if (path) *path = (e->path == 0 && e->plan32) ? 3 : (e->path == 0 && e->pair_ok) ? 4 : (e->path == 0 && e->k64_ok) ? 5 : e->path;
  if (e->path == 0 && e->k64_ok) {            // what complex64 input runs
    if (lds_bytes) *lds_bytes = ksa::Plan64::LDS_BYTES;
    if (vgprs) *vgprs = e->k64_vgprs;
    if (grid) *grid = e->num_cu * e->k64_bpc;
  }
  if (e->path == 0 && e->pair_ok) {           // what large batches run
    if (lds_bytes) *lds_bytes = e->pair_lds;
    if (vgprs) *vgprs = e->pair_vgprs;
    if (grid) *grid = e->num_cu * e->pair_bpc;
  }
  return 0;
}

}  // extern "C"
