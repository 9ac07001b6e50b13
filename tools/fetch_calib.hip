// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the access shapes libksa uses (development tool).
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE reads exactly 1/2 of the bytes of a 16 B/lane coalesced
// stream; other widths must be calibrated on a known byte count in the kernel's own pattern.
//   calib_b64_windowed : spectrum_kernel's pattern -- 256 threads read one 32 KiB window as 16
//                        buffer_load_dwordx2 (8 B/lane, 512 B per wave-instruction, 2 KiB apart)
//   calib_b128_stream  : the guide's reference shape, 16 B/lane streaming
// Each kernel reads `bytes` exactly once from a buffer far larger than the 256 MiB Infinity Cache.
// usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib   (prints the byte counts)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_b64_windowed(const float2* in, float* out, long long windows) {
  float acc = 0;
  for (long long w = blockIdx.x; w < windows; w += gridDim.x) {
    const char* base = reinterpret_cast<const char*>(in) + w * 32768;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 32768, 0x00020000);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, threadIdx.x * 8, 2048 * q, 0);
      const unsigned a = x.x, b = x.y;
      acc += __uint_as_float(a) + __uint_as_float(b);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_b128_stream(const float4* in, float* out, long long n16) {
  float acc = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n16; i += gridDim.x * 256ll) {
    const float4 x = in[i];
    acc += x.x + x.y + x.z + x.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const long long bytes = 4ll << 30;
  void* d; float* o;
  hipMalloc(&d, bytes); hipMalloc(&o, 2048 * 256 * 4);
  hipMemset(d, 0, bytes);
  hipLaunchKernelGGL(calib_b64_windowed, dim3(1024), dim3(256), 0, 0, (const float2*)d, o, bytes / 32768);
  hipLaunchKernelGGL(calib_b128_stream, dim3(2048), dim3(256), 0, 0, (const float4*)d, o, bytes / 16);
  hipLaunchKernelGGL(calib_b64_windowed, dim3(1024), dim3(256), 0, 0, (const float2*)d, o, bytes / 32768);
  hipLaunchKernelGGL(calib_b128_stream, dim3(2048), dim3(256), 0, 0, (const float4*)d, o, bytes / 16);
  hipDeviceSynchronize();
  printf("each kernel read %lld bytes = %lld KiB exactly once\n", bytes, bytes / 1024);
  return 0;
}
