// mall_bw.hip -- microbenchmark: streaming read / copy / write-then-read rates against the working-set size, to see
// what the 256 MiB Infinity Cache is worth to a kernel pair that hands a buffer over (forward transform).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mall_bw.hip -o tools/micro/mall_bw && tools/micro/mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, size_t n4, float *out) {
  float s = 0.f;
  size_t const stride = (size_t)gridDim.x * 256;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = k + u * stride < n4 ? a[k + u * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; u++) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4 *__restrict__ a, size_t n4, float v) {
  size_t const stride = (size_t)gridDim.x * 256;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride) a[k] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
  size_t const stride = (size_t)gridDim.x * 256;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = k + u * stride < n4 ? a[k + u * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; u++) if (k + u * stride < n4) b[k + u * stride] = v[u];
  }
}

int main() {
  size_t const maxb = (size_t)1 << 30;
  float4 *a, *b; float *out;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int const grid = 256 * 8, reps = 20;
  printf("%8s %12s %12s %12s %12s %14s\n", "MiB", "read GB/s", "write GB/s", "copy GB/s", "wr->rd GB/s", "copy a->b->a");
  for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
    size_t const bytes = mb << 20, n4 = bytes / 16;
    float ms; double r[5];
    for (int mode = 0; mode < 5; mode++) {
      for (int w = 0; w < 3; w++) { // warm
        if (mode == 0) k_read<<<grid, 256>>>(a, n4, out);
        if (mode == 1) k_write<<<grid, 256>>>(a, n4, 1.f);
        if (mode == 2) k_copy<<<grid, 256>>>(a, b, n4 / 2);
        if (mode == 3) { k_write<<<grid, 256>>>(a, n4, 1.f); k_read<<<grid, 256>>>(a, n4, out); }
        if (mode == 4) { k_copy<<<grid, 256>>>(a, b, n4 / 2); k_copy<<<grid, 256>>>(b, a, n4 / 2); }
      }
      CK(hipEventRecord(e0));
      for (int w = 0; w < reps; w++) {
        if (mode == 0) k_read<<<grid, 256>>>(a, n4, out);
        if (mode == 1) k_write<<<grid, 256>>>(a, n4, 1.f);
        if (mode == 2) k_copy<<<grid, 256>>>(a, b, n4 / 2);      // bytes/2 read + bytes/2 written: working set = bytes
        if (mode == 3) { k_write<<<grid, 256>>>(a, n4, 1.f); k_read<<<grid, 256>>>(a, n4, out); }
        if (mode == 4) { k_copy<<<grid, 256>>>(a, b, n4 / 2); k_copy<<<grid, 256>>>(b, a, n4 / 2); }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      double const moved = (mode == 3 || mode == 4) ? 2.0 * bytes : (double)bytes;
      r[mode] = moved * reps / (ms * 1e-3) / 1e9;
    }
    printf("%8zu %12.0f %12.0f %12.0f %12.0f %14.0f\n", mb, r[0], r[1], r[2], r[3], r[4]);
  }
  return 0;
}
