// Do VALU work and LDS traffic overlap on one CU when they come from DIFFERENT workgroups that each alternate between the
// two (the structure of spectrum_kernel: exchange -> barrier -> butterflies -> barrier -> exchange ...)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vlo tools/valu_lds_overlap.hip && /tmp/vlo
// Workgroups of 256 threads (4 waves, one per SIMD), W = 1 / 2 / 3 of them per CU (dynamic LDS pads the occupancy).
// Per "transform" a wave issues V independent-chain v_fma_f32 (VALU phase) and S ds_write_b64 + S ds_read_b64 (LDS
// phase) separated by workgroup barriers, as the real kernel does.  Modes: VALU only, LDS only, both alternating.
// If both ~= max(VALU only, LDS only) the two pipes overlap across workgroups; if ~= their sum they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int V, int S>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  extern __shared__ float2 lds[];
  const int tid = threadIdx.x;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = tid * 0.001f + i;
  float2 r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = make_float2(tid + i, tid - i);
  const float c0 = 0.999f, c1 = 0.001f;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int v = 0; v < V / 16; ++v)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
    }
    if (MODE != 0) {
      __syncthreads();
#pragma unroll
      for (int s = 0; s < S; ++s) lds[tid * 17 + (s & 15)] = r[s & 15];   // 16 consecutive per thread, padded: conflict free
      __syncthreads();
#pragma unroll
      for (int s = 0; s < S; ++s) r[s & 15] = lds[tid + 256 * (s & 15) + (s >> 4)];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a[i] + r[i].x + r[i].y;
  if (acc == 12345.678f) out[tid] = acc;
}

// The real kernel's shape: three VALU blocks of V/3 instructions on CH independent chains, two exchanges of 16 stores |
// barrier | 16 loads each with a barrier in front (4 barriers per transform), optionally GL streaming global loads of
// 8 bytes per lane per transform (waited for at the top of the next one) and TR transcendental (v_sqrt) instructions.
template <int V, int CH, int GL, int TR, int PF = 1, int XL = 0, int NB = 4>
__global__ __launch_bounds__(256) void shaped(float* out, const float2* __restrict__ src, long long nsrc, int iters) {
  extern __shared__ float2 lds[];
  const int tid = threadIdx.x;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = tid * 0.001f + i;
  float2 r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = make_float2(tid + i, tid - i);
  const float c0 = 0.999f, c1 = 0.001f;
  float2 g[GL > 0 ? GL : 1];
  const long long mask = nsrc - 1;   // nsrc is a power of two
  long long pos = ((long long)blockIdx.x * 256 + tid) & mask;
#define VALU_BLOCK(n)                                                                                        \
  _Pragma("unroll") for (int v = 0; v < (n); ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v % CH]) : "v"(c0), "v"(c1))
  if (GL > 0 && PF == 2) {
#pragma unroll
    for (int q = 0; q < GL; ++q) g[q] = src[(pos + (long long)q * 256) & mask];
    pos = (pos + (long long)gridDim.x * 256 * GL) & mask;
  }
  for (int it = 0; it < iters; ++it) {
    if (GL > 0 && PF != 2) {
#pragma unroll
      for (int q = 0; q < GL; ++q) g[q] = src[(pos + (long long)q * 256) & mask];
      pos = (pos + (long long)gridDim.x * 256 * GL) & mask;
    }
    if (GL > 0 && PF != 1) {     // PF 0: the loads are needed at once (window multiply); PF 2: they were issued one transform ago
#pragma unroll
      for (int q = 0; q < GL; ++q) a[q % CH] += g[q].x * g[q].y;
      if (PF == 2) {
#pragma unroll
        for (int q = 0; q < GL; ++q) g[q] = src[(pos + (long long)q * 256) & mask];
        pos = (pos + (long long)gridDim.x * 256 * GL) & mask;
      }
    }
    VALU_BLOCK(V / 3);
    if (GL > 0 && PF == 1) {
#pragma unroll
      for (int q = 0; q < GL; ++q) a[q % CH] += g[q].x * g[q].y;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (NB == 4 || (NB == 3 && e == 0)) __syncthreads();   // NB 3: the second exchange is in place (no barrier in front of its stores); NB 2: neither
#pragma unroll
      for (int s = 0; s < 16; ++s) lds[tid * 17 + s] = r[s];
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 16; ++s) r[s] = lds[XL ? (tid + (tid >> 4)) + 272 * s + e : tid + 256 * s + e];   // XL: the real kernel's one-in-16 padding (2-way conflict per 32 lanes)
      if (XL && e == 0) {
        float2 tw[15];
#pragma unroll
        for (int s = 0; s < 15; ++s) tw[s] = lds[4352 + s * 16 + (tid & 15)];          // middle-pass twiddles: 16 distinct addresses per wave
#pragma unroll
        for (int s = 0; s < 15; ++s) { a[s % CH] += tw[s].x; a[(s + 1) % CH] += tw[s].y; }
      }
      if (XL && e == 1) {
        const float4* t4 = reinterpret_cast<const float4*>(lds + 4608);
#pragma unroll
        for (int s = 0; s < 4; ++s) { const float4 w = t4[s * 256 + tid]; a[s % CH] += w.x + w.y + w.z + w.w; }   // window taps
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      VALU_BLOCK(V / 3);
    }
#pragma unroll
    for (int t = 0; t < TR; ++t) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[t % CH]));
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a[i] + r[i].x + r[i].y;
  if (acc == 12345.678f) out[tid] = acc;
}

template <int V, int CH, int GL, int TR, int PF = 1, int XL = 0, int NB = 4>
float run_shaped(int wg_per_cu, int iters, float* out, const float2* src, long long nsrc) {
  const int lds_bytes = wg_per_cu == 1 ? 120 * 1024 : wg_per_cu == 2 ? 72 * 1024 : 48 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(shaped<V, CH, GL, TR, PF, XL, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  shaped<V, CH, GL, TR, PF, XL, NB><<<grid, 256, lds_bytes>>>(out, src, nsrc, 10);
  hipEventRecord(e0);
  shaped<V, CH, GL, TR, PF, XL, NB><<<grid, 256, lds_bytes>>>(out, src, nsrc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters / wg_per_cu;   // us per transform per CU
}

// Design-space explorer: T threads per workgroup, per wave and transform V v_fma on 8 chains, two exchanges of X stores | barrier
// | X loads (4 barriers), XT twiddle-like b64 reads, XP tap-like b128 reads from LDS, GL streaming 8-byte global loads and GT
// 16-byte global loads from a small (cache resident) table, NS v_sqrt; lds_bytes sets the workgroups per CU.
template <int T, int V, int X, int XT, int XP, int GL, int GT, int NS, int NE = 2, int STG = 0>
__global__ __launch_bounds__(T) void explore(float* out, const float2* __restrict__ src, long long nsrc, const float4* __restrict__ tab, int iters) {
  extern __shared__ float2 lds[];
  const int tid = threadIdx.x;
  constexpr int CH = 8;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = tid * 0.001f + i;
  float2 r[X];
#pragma unroll
  for (int i = 0; i < X; ++i) r[i] = make_float2(tid + i, tid - i);
  const float c0 = 0.999f, c1 = 0.001f;
  const long long mask = nsrc - 1;
  long long pos = ((long long)blockIdx.x * T + tid) & mask;
  for (int it = 0; it < iters; ++it) {
    float2 g[GL > 0 ? GL : 1];
#pragma unroll
    for (int q = 0; q < GL; ++q) g[q] = src[(pos + (long long)q * T) & mask];
    pos = (pos + (long long)gridDim.x * T * GL) & mask;
#pragma unroll
    for (int q = 0; q < GT; ++q) { const float4 w = tab[q * T + tid]; a[q % CH] += w.x + w.y + w.z + w.w; }
    if (STG) {
      // staged samples: the wave's GL loads cover the span of all its windows once; they go to LDS (aliasing the exchange
      // buffer) and every thread picks its 16 samples from there: slot s = tid / 4 starts 6.4 samples after slot s - 1
#pragma unroll
      for (int q = 0; q < GL; ++q) lds[q * 64 + (tid & 63)] = g[q];
      const int off = (int)((tid >> 2) * 6.4f) + (tid & 3);
      float2 x[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = lds[off + 4 * q];
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q % CH] += x[q].x * x[q].y;
    } else {
#pragma unroll
      for (int q = 0; q < GL; ++q) a[q % CH] += g[q].x * g[q].y;
    }
    VALU_BLOCK(V / (NE + 1));
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      __syncthreads();
#pragma unroll
      for (int s = 0; s < X; ++s) lds[tid * (X + 1) + s] = r[s];
      __syncthreads();
#pragma unroll
      for (int s = 0; s < X; ++s) r[s] = lds[(tid + (tid >> 4)) + (T + T / 16) * s + e];
      if (e == 0 && XT > 0) {
#pragma unroll
        for (int s = 0; s < XT; ++s) { const float2 tw = lds[T * (X + 2) + s * 16 + (tid & 15)]; a[s % CH] += tw.x; a[(s + 1) % CH] += tw.y; }
      }
      if (e == NE - 1) {
        const float4* t4 = reinterpret_cast<const float4*>(lds + T * (X + 2) + 256);
#pragma unroll
        for (int s = 0; s < XP; ++s) { const float4 w = t4[s * T + tid]; a[s % CH] += w.x + w.y + w.z + w.w; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      VALU_BLOCK(V / (NE + 1));
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[t % CH]));
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a[i];
#pragma unroll
  for (int i = 0; i < X; ++i) acc += r[i].x + r[i].y;
  if (acc == 12345.678f) out[tid] = acc;
}

template <int T, int V, int X, int XT, int XP, int GL, int GT, int NS, int NE = 2, int STG = 0>
float run_explore(int lds_kb, int wg_per_cu, int iters, float* out, const float2* src, long long nsrc, const float4* tab, int lds_bytes_exact = 0) {
  auto k = explore<T, V, X, XT, XP, GL, GT, NS, NE, STG>;
  const int lb = lds_bytes_exact ? lds_bytes_exact : lds_kb * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T, lb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  k<<<grid, T, lb>>>(out, src, nsrc, tab, 10);
  hipEventRecord(e0);
  k<<<grid, T, lb>>>(out, src, nsrc, tab, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("[occupancy %d WG/CU] ", occ);
  return ms * 1e3f / iters / wg_per_cu;
}

template <int MODE, int V, int S>
float run(int wg_per_cu, int iters, float* out) {
  const int lds_bytes = wg_per_cu == 1 ? 120 * 1024 : wg_per_cu == 2 ? 72 * 1024 : 48 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, V, S>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  probe<MODE, V, S><<<grid, 256, lds_bytes>>>(out, 10);
  hipEventRecord(e0);
  probe<MODE, V, S><<<grid, 256, lds_bytes>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

// Round 4: P points per thread, T threads per transform, both exchanges moved in PIECES pieces through an LDS buffer of
// T * P / PIECES elements (a transform that lives in registers and borrows LDS a piece at a time).  T == 64 is the
// WAVE-PRIVATE transform of VERDICT r03 item 3: no s_barrier at all, the exchange is ordered by s_waitcnt alone.  Per
// transform and thread: V v_fma on 16 chains in three blocks, GL streaming 8-byte loads, NS v_sqrt.
template <int T, int P, int V, int PIECES, int GL, int NS>
__global__ __launch_bounds__(T) void pieces(float* out, const float2* __restrict__ src, long long nsrc, int iters) {
  extern __shared__ float2 lds[];
  const int tid = threadIdx.x;
  constexpr int Q = P / PIECES;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = tid * 0.001f + i;
  float2 r[P];
#pragma unroll
  for (int i = 0; i < P; ++i) r[i] = make_float2(tid + i, tid - i);
  const float c0 = 0.999f, c1 = 0.001f;
  const long long mask = nsrc - 1;
  long long pos = ((long long)blockIdx.x * T + tid) & mask;
  for (int it = 0; it < iters; ++it) {
    float2 g[GL > 0 ? GL : 1];
#pragma unroll
    for (int q = 0; q < GL; ++q) g[q] = src[(pos + (long long)q * T) & mask];
    pos = (pos + (long long)gridDim.x * T * GL) & mask;
#pragma unroll
    for (int q = 0; q < GL; ++q) a[q % 16] += g[q].x * g[q].y;
#pragma unroll
    for (int v = 0; v < V / 3; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v % 16]) : "v"(c0), "v"(c1));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        if (T > 64) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < Q; ++s) lds[s * (T + 2) + tid] = r[pc * Q + s];                 // transposed layout: conflict free
        if (T > 64) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < Q; ++s) r[pc * Q + s] = lds[(tid % Q) * (T + 2) + tid / Q + (T / Q) * s];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int v = 0; v < V / 3; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v % 16]) : "v"(c0), "v"(c1));
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[t % 16]));
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a[i];
#pragma unroll
  for (int i = 0; i < P; ++i) acc += r[i].x + r[i].y;
  if (acc == 12345.678f) out[tid] = acc;
}

template <int T, int P, int V, int PIECES, int GL, int NS>
float run_pieces(int wg_per_cu, int lds_bytes, int iters, float* out, const float2* src, long long nsrc) {
  auto k = pieces<T, P, V, PIECES, GL, NS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T, lds_bytes);
  hipFuncAttributes at;
  hipFuncGetAttributes(&at, reinterpret_cast<const void*>(k));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  k<<<grid, T, lds_bytes>>>(out, src, nsrc, 10);
  hipEventRecord(e0);
  k<<<grid, T, lds_bytes>>>(out, src, nsrc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("[%d VGPRs, occupancy %d WG/CU, asked %d] ", at.numRegs, occ, wg_per_cu);
  return ms * 1e3f / iters / wg_per_cu;   // us per transform per CU
}

int main_r4(float* out, const float2* src, long long nsrc) {
  const int it = 3000;
  printf("round 4 shapes, us per transform per CU (N = 4096 unless stated; the real kernel: 1.42 after round 4, 1.50 before):\n");
  printf("  4 waves x 16 points, 600 fma, 8 loads, 16 sqrt, whole exchanges, 3 WG/CU (49 KB)   : %.3f\n", run_pieces<256, 16, 600, 1, 8, 16>(3, 49 * 1024, it, out, src, nsrc));
  printf("  wave-private: 1 wave x 64 points, 2400 fma, 32 loads, 64 sqrt, whole exchanges, 4 WG/CU (34 KB each)      : %.3f\n", run_pieces<64, 64, 2400, 1, 32, 64>(4, 34 * 1024, it / 2, out, src, nsrc));
  printf("  wave-private, exchanges in 4 pieces through 8.5 KB, 8 WG/CU (2 waves per SIMD)                            : %.3f\n", run_pieces<64, 64, 2400, 4, 32, 64>(8, 9 * 1024, it / 2, out, src, nsrc));
  printf("  wave-private, exchanges in 4 pieces, 12 WG/CU (3 waves per SIMD if the registers allow)                   : %.3f\n", run_pieces<64, 64, 2400, 4, 32, 64>(12, 9 * 1024, it / 2, out, src, nsrc));
  printf("  2 waves x 32 points, 1200 fma, 16 loads, 32 sqrt, whole exchanges, 4 WG/CU (34 KB)                        : %.3f\n", run_pieces<128, 32, 1200, 1, 16, 32>(4, 34 * 1024, it / 2, out, src, nsrc));
  printf("  2 waves x 32 points, exchanges in 2 pieces through 17 KB, 6 WG/CU                                         : %.3f\n", run_pieces<128, 32, 1200, 2, 16, 32>(6, 18 * 1024, it / 2, out, src, nsrc));
  printf("N = 16384 (config 3; the real kernel: 8.0 us per window per CU):\n");
  printf("  today: 8 waves x 32 points, 1632 fma, 64 loads, 32 sqrt, whole exchanges, 1 WG/CU (135 KB)                : %.3f\n", run_pieces<512, 32, 1632, 1, 64, 32>(1, 135 * 1024, it / 4, out, src, nsrc));
  printf("  4 waves x 64 points, 3264 fma, 128 loads, 64 sqrt, exchanges in 2 pieces through 66 KB, 2 WG/CU           : %.3f\n", run_pieces<256, 64, 3264, 2, 128, 64>(2, 67 * 1024, it / 4, out, src, nsrc));
  printf("  4 waves x 64 points, exchanges in 4 pieces through 33 KB, 2 WG/CU                                         : %.3f\n", run_pieces<256, 64, 3264, 4, 128, 64>(2, 34 * 1024, it / 4, out, src, nsrc));
  printf("  8 waves x 32 points, exchanges in 2 pieces through 66 KB, 2 WG/CU (needs 128 VGPRs)                       : %.3f\n", run_pieces<512, 32, 1632, 2, 64, 32>(2, 67 * 1024, it / 4, out, src, nsrc));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'r') {
    float* out;
    hipMalloc(&out, 4096);
    float2* src;
    const long long nsrc = 1ll << 28;
    hipMalloc(&src, nsrc * 8);
    hipMemset(src, 0, nsrc * 8);
    return main_r4(out, src, nsrc);
  }
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 20000;
  // V = 640 VALU instructions, S = 32 stores + 32 loads of 8 bytes per lane: one 4096-point transform's worth per wave
  printf("per iteration and workgroup: 640 v_fma per wave, 32 ds_write_b64 + 32 ds_read_b64 per wave; %d iterations\n", iters);
  for (int w = 1; w <= 3; ++w) {
    const float tv = run<0, 640, 32>(w, iters, out), tl = run<1, 640, 32>(w, iters, out), tb = run<2, 640, 32>(w, iters, out);
    // time per "transform" per CU: w workgroups per CU each do `iters`
    const double k = 1e3 / (double)iters / w;   // us per transform per CU
    printf("W = %d WG/CU: VALU only %.3f us  LDS only %.3f us  both %.3f us  (sum %.3f, max %.3f) per transform per CU\n", w, tv * k,
           tl * k, tb * k, (tv + tl) * k, (tv > tl ? tv : tl) * k);
  }
  // the same with twice the waves per workgroup-equivalent: 6 workgroups per CU of half the work each is not possible
  // with 48 KB each; instead 3 WG/CU with half the VALU per wave shows how the balance point moves
  for (int w = 3; w <= 3; ++w) {
    const float tv = run<0, 320, 32>(w, iters, out), tl = run<1, 320, 32>(w, iters, out), tb = run<2, 320, 32>(w, iters, out);
    const double k = 1e3 / (double)iters / w;
    printf("W = %d, half the VALU: VALU only %.3f  LDS only %.3f  both %.3f  (sum %.3f)\n", w, tv * k, tl * k, tb * k, (tv + tl) * k);
  }
  // the real kernel's shape at 3 workgroups per CU: which ingredient costs what
  float2* src;
  const long long nsrc = 1ll << 28;   // 2 GiB of float2
  hipMalloc(&src, nsrc * 8);
  hipMemset(src, 0, nsrc * 8);
  const int it2 = 5000;
  printf("shaped (3 VALU blocks, 2 exchanges, 4 barriers), us per transform per CU at 3 / 2 WG per CU:\n");
  printf("  642 fma on 16 chains                      : %.3f / %.3f\n", run_shaped<642, 16, 0, 0>(3, it2, out, src, nsrc), run_shaped<642, 16, 0, 0>(2, it2, out, src, nsrc));
  printf("  642 fma on 8 chains                       : %.3f / %.3f\n", run_shaped<642, 8, 0, 0>(3, it2, out, src, nsrc), run_shaped<642, 8, 0, 0>(2, it2, out, src, nsrc));
  printf("  642 fma on 4 chains                       : %.3f / %.3f\n", run_shaped<642, 4, 0, 0>(3, it2, out, src, nsrc), run_shaped<642, 4, 0, 0>(2, it2, out, src, nsrc));
  printf("  642 fma on 2 chains                       : %.3f / %.3f\n", run_shaped<642, 2, 0, 0>(3, it2, out, src, nsrc), run_shaped<642, 2, 0, 0>(2, it2, out, src, nsrc));
  printf("  630 fma on 16 chains + 16 v_sqrt          : %.3f / %.3f\n", run_shaped<630, 16, 0, 16>(3, it2, out, src, nsrc), run_shaped<630, 16, 0, 16>(2, it2, out, src, nsrc));
  printf("  642 fma on 16 chains + 8 global loads     : %.3f / %.3f\n", run_shaped<642, 16, 8, 0>(3, it2, out, src, nsrc), run_shaped<642, 16, 8, 0>(2, it2, out, src, nsrc));
  printf("  630 fma on 8 chains + 16 sqrt + 8 loads   : %.3f / %.3f\n", run_shaped<630, 8, 8, 16>(3, it2, out, src, nsrc), run_shaped<630, 8, 8, 16>(2, it2, out, src, nsrc));
  printf("  the same, loads needed at once            : %.3f / %.3f\n", run_shaped<630, 8, 8, 16, 0>(3, it2, out, src, nsrc), run_shaped<630, 8, 8, 16, 0>(2, it2, out, src, nsrc));
  printf("  the same, loads issued one transform ahead: %.3f / %.3f\n", run_shaped<630, 8, 8, 16, 2>(3, it2, out, src, nsrc), run_shaped<630, 8, 8, 16, 2>(2, it2, out, src, nsrc));
  printf("  8 loads at once + twiddle / tap LDS reads + padded (2-way) exchange reads: %.3f / %.3f\n", run_shaped<600, 8, 8, 16, 0, 1>(3, it2, out, src, nsrc), run_shaped<600, 8, 8, 16, 0, 1>(2, it2, out, src, nsrc));
  printf("  ... with 3 barriers per transform (second exchange in place)             : %.3f / %.3f\n", run_shaped<600, 8, 8, 16, 0, 1, 3>(3, it2, out, src, nsrc), run_shaped<600, 8, 8, 16, 0, 1, 3>(2, it2, out, src, nsrc));
  printf("  ... with 2 barriers per transform (not realisable in 53 KB; for scale)   : %.3f / %.3f\n", run_shaped<600, 8, 8, 16, 0, 1, 2>(3, it2, out, src, nsrc), run_shaped<600, 8, 8, 16, 0, 1, 2>(2, it2, out, src, nsrc));
  printf("  ... 4 barriers again                                                     : %.3f / %.3f\n", run_shaped<600, 8, 8, 16, 0, 1, 4>(3, it2, out, src, nsrc), run_shaped<600, 8, 8, 16, 0, 1, 4>(2, it2, out, src, nsrc));
  printf("  16 loads needed at once (no reuse)        : %.3f / %.3f\n", run_shaped<630, 8, 16, 16, 0>(3, it2, out, src, nsrc), run_shaped<630, 8, 16, 16, 0>(2, it2, out, src, nsrc));
  float4* tab;
  hipMalloc(&tab, 64 * 1024);
  hipMemset(tab, 0, 64 * 1024);
  printf("explorer (alternative shapes), us per transform per CU -- compare inside one run:\n");
  printf("  today: 256 thr, 600 fma, 16+16 x2, 15 tw + 4 taps in LDS, 8 loads, 3 WG/CU (51 KB)   : %.3f\n", run_explore<256, 600, 16, 15, 4, 8, 0, 16>(51, 3, it2, out, src, nsrc, tab));
  printf("  the same, 53 KB = 54272 B of LDS (the occupancy API says 3, the time says 2 WG/CU)     : %.3f\n", run_explore<256, 600, 16, 15, 4, 8, 0, 16>(53, 3, it2, out, src, nsrc, tab));
  printf("  the same, 56 KB (2 WG/CU)                                                              : %.3f\n", run_explore<256, 600, 16, 15, 4, 8, 0, 16>(56, 3, it2, out, src, nsrc, tab));
  printf("  taps from L2 (4 x 16 B loads), 38 KB of LDS, 4 WG/CU                                   : %.3f\n", run_explore<256, 600, 16, 15, 0, 8, 4, 16>(38, 4, it2, out, src, nsrc, tab));
  printf("  the same at 3 WG/CU                                                                    : %.3f\n", run_explore<256, 600, 16, 15, 0, 8, 4, 16>(38, 3, it2, out, src, nsrc, tab));
  printf("  512 thr (8 points per thread, pair butterflies): 350 fma, 8+8 x2, 8 tw + 2 taps, 4 loads, 3 WG/CU : %.3f\n", run_explore<512, 350, 8, 8, 2, 4, 0, 8>(51, 3, it2, out, src, nsrc, tab));
  printf("  the same with 380 fma                                                                  : %.3f\n", run_explore<512, 380, 8, 8, 2, 4, 0, 8>(51, 3, it2, out, src, nsrc, tab));
  printf("  today, no tap reads                                                                    : %.3f\n", run_explore<256, 600, 16, 15, 0, 8, 0, 16>(51, 3, it2, out, src, nsrc, tab));
  printf("  today, no tap and no twiddle reads                                                     : %.3f\n", run_explore<256, 600, 16, 0, 0, 8, 0, 16>(51, 3, it2, out, src, nsrc, tab));
  for (int bytes : {52224, 52736, 53120, 53248, 53760, 54272})
    printf("  today with exactly %d B of LDS: %.3f\n", bytes, run_explore<256, 600, 16, 15, 4, 8, 0, 16>(0, 3, it2, out, src, nsrc, tab, bytes));
  // the N = 64 kernel's shape (config 4): single-wave workgroups, 16 per CU, one exchange, 16 loads (90 %% overlap: all 16 samples re-read)
  printf("  N = 64 shape: 64 thr, 350 fma, 16+16 x1, 4 taps, 16 loads, 16 WG/CU (9 KB): %.4f us per wave-round per CU (real kernel: 0.232)\n", run_explore<64, 350, 16, 0, 4, 16, 0, 16, 1>(9, 16, it2 * 4, out, src, nsrc, tab));
  printf("  the same with 8 loads : %.4f\n", run_explore<64, 350, 16, 0, 4, 8, 0, 16, 1>(9, 16, it2 * 4, out, src, nsrc, tab));
  printf("  the same with 0 loads : %.4f\n", run_explore<64, 350, 16, 0, 4, 0, 0, 16, 1>(9, 16, it2 * 4, out, src, nsrc, tab));
  printf("  16 loads, no LDS exchange (X = 0 is not expressible: 1 store + 1 load): %.4f\n", run_explore<64, 350, 1, 0, 4, 16, 0, 16, 1>(9, 16, it2 * 4, out, src, nsrc, tab));
  printf("  staged: 3 loads -> LDS -> 16 ds_read_b64 per thread (334 fma)               : %.4f\n", run_explore<64, 334, 16, 0, 4, 3, 0, 16, 1, 1>(9, 16, it2 * 4, out, src, nsrc, tab));
  printf("  16 loads, 12 WG/CU     : %.4f\n", run_explore<64, 350, 16, 0, 4, 16, 0, 16, 1>(12, 12, it2 * 4, out, src, nsrc, tab));
  printf("  today again                                                                            : %.3f\n", run_explore<256, 600, 16, 15, 4, 8, 0, 16>(51, 3, it2, out, src, nsrc, tab));
  return 0;
}
