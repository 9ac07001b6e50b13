// Does a write-then-read round trip of S bytes run faster when S fits the 256 MiB Infinity Cache?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/mall_probe.hip && /tmp/mall_probe
// For each S: alternate a streaming write kernel and a streaming read kernel over the same S bytes (float4 per
// lane, grid-stride), 20 rounds, report GB/s of each half.  Decides whether chunking the first-stage scratch of
// the N > 16384 transforms (ksa_dif16.hpp) to the cache size can pay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void wr(float4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ void rd(const float4* p, size_t n, float* out) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 x = p[i]; s += x.x + x.y + x.z + x.w; }
  if (s == 1234.5f) *out = s;
}
int main() {
  float* out; hipMalloc(&out, 4);
  size_t sizes_mb[] = {8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 384, 512, 1024, 4096};
  for (size_t mb : sizes_mb) {
    size_t bytes = mb << 20, n = bytes / 16;
    float4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc %zu failed\n", mb); continue; }
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    const int rounds = 20, grid = 256 * 16;
    double tw = 0, tr = 0;
    for (int r = 0; r < rounds + 2; ++r) {
      hipEventRecord(e[0]); wr<<<grid, 256>>>(buf, n, (float)r); hipEventRecord(e[1]);
      rd<<<grid, 256>>>(buf, n, out); hipEventRecord(e[2]);
      hipEventSynchronize(e[2]);
      float a, b; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]);
      if (r >= 2) { tw += a; tr += b; }
    }
    printf("S = %5zu MiB: write %7.1f GB/s   read-back %7.1f GB/s   round trip %7.1f GB/s\n", mb, bytes * rounds / tw / 1e6, bytes * rounds / tr / 1e6,
           2.0 * bytes * rounds / (tw + tr) / 1e6);
    hipFree(buf);
  }
  return 0;
}
