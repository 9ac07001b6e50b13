// Micro-benchmark (development tool, not part of the product): VALU issue rate on gfx950 for the
// instruction kinds the FFT butterflies use -- plain vs packed f32 add/mul/fma -- at 1..8 waves/SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, int iters, float a, float b) {
  float x[8];
  f2 y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1}; }
  f2 pa{a, a}, pb{b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(pa), "v"(pb));
        if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(pa));
        if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(pa));
        if (KIND == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
        if (KIND == 7) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int cus, float* d) {
  for (int wps : {1, 2, 3, 4, 8}) {
    const int iters = 20000;
    const int threads = 256;          // 4 waves per block = 1 per SIMD
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 32 * wps;   // wave-instructions issued on each SIMD
    double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-14s waves/SIMD %d : %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ns_per, ns_per * 2.4);
  }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  float* d; hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
  run<0>("v_fma_f32", p.multiProcessorCount, d);
  run<1>("v_add_f32", p.multiProcessorCount, d);
  run<2>("v_mul_f32", p.multiProcessorCount, d);
  run<7>("v_sub_f32", p.multiProcessorCount, d);
  run<3>("v_pk_fma_f32", p.multiProcessorCount, d);
  run<4>("v_pk_add_f32", p.multiProcessorCount, d);
  run<5>("v_pk_mul_f32", p.multiProcessorCount, d);
  run<6>("v_sqrt_f32", p.multiProcessorCount, d);
  return 0;
}
