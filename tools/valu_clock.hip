// Micro-benchmark (development tool): cycles per wave64 VALU instruction per SIMD, with the shader clock measured
// in the same kernel (clock64 = s_memtime shader cycles, wall_clock64 = 100 MHz constant), at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o variants/valu_clock tools/valu_clock.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, long long* stamps, int iters, float a, float b) {
  float x[8];
  f2 y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1}; }
  f2 pa{a, a}, pb{b, b};
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(pa), "v"(pb));
        if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));  // 3 VGPR sources
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
}

template <int KIND>
void run(const char* name, int cus, float* d, long long* st) {
  for (int wps : {1, 2, 3, 4, 8}) {
    const int iters = 20000, threads = 256, blocks = cus * wps;   // 4 waves per block = 1 per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, st, 100, 1.0001f, 0.5f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, st, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
    const double instr = (double)iters * 32 * wps;             // wave-instructions issued on the SIMD in that span
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    printf("%-22s waves/SIMD %d : %.2f shader cycles per wave-instruction per SIMD, shader clock %.2f GHz, %.2f ns\n",
           name, wps, (double)h[0] / instr, ghz, (double)h[1] / 100e6 * 1e9 / instr);
  }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  float* d; hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
  long long* st; hipMalloc(&st, 16);
  run<0>("v_fma_f32 (1 vgpr src)", p.multiProcessorCount, d, st);
  run<3>("v_fma_f32 (3 vgpr src)", p.multiProcessorCount, d, st);
  run<1>("v_add_f32", p.multiProcessorCount, d, st);
  run<2>("v_pk_fma_f32", p.multiProcessorCount, d, st);
  return 0;
}
