// stride_bw.hip -- microbenchmark: what the row pitch costs a column-tile pass (the access pattern of kh_fft_cols and of
// kh_fft_rows' transposed store): every workgroup moves a tile of 16 columns x 512 rows, 128-byte row segments at the
// pitch of the matrix, from src to dst.  Same bytes, same workgroups; only the pitches differ.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stride_bw.hip -o tools/micro/stride_bw && tools/micro/stride_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// grid (tiles per row, blocks); block b's matrix starts at b*rows*pitch
template <int ROWS>
__global__ __launch_bounds__(256, 4) void k_tile(const float2 *__restrict__ src, float2 *__restrict__ dst, size_t ps, size_t pd) {
  int const tid = threadIdx.x, w = tid % 16, r0 = tid / 16;
  const float2 *s = src + (size_t)blockIdx.y * ROWS * ps + blockIdx.x * 16 + w;
  float2 *d = dst + (size_t)blockIdx.y * ROWS * pd + blockIdx.x * 16 + w;
  float2 v[ROWS / 16];
#pragma unroll
  for (int a = 0; a < ROWS / 16; a++) v[a] = s[(size_t)(r0 + 16 * a) * ps];
#pragma unroll
  for (int a = 0; a < ROWS / 16; a++) d[(size_t)(r0 + 16 * a) * pd] = make_float2(v[a].y, v[a].x);
}

int main() {
  size_t const cap = (size_t)1200 << 20;
  float2 *a, *b;
  CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
  CK(hipMemset(a, 0, cap)); CK(hipMemset(b, 0, cap));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { int cols; int pad_s, pad_d; int blocks; };
  // 512 rows; cols x 8 bytes per row; 64 MiB of payload each way
  Case const cases[] = {{512, 0, 0, 32},  {512, 16, 16, 32},  {1024, 0, 0, 16}, {1024, 16, 0, 16}, {1024, 0, 16, 16}, {1024, 16, 16, 16},
                        {1024, 32, 32, 16}, {1024, 48, 48, 16}, {2048, 0, 0, 8},  {2048, 16, 16, 8},  {4096, 0, 0, 4}, {4096, 16, 16, 4},
                        // beyond the Infinity Cache: 512 MiB each way
                        {512, 0, 0, 256}, {512, 16, 16, 256}, {1024, 0, 0, 128}, {1024, 16, 16, 128}, {1024, 0, 16, 128}, {1024, 16, 0, 128},
                        {2048, 0, 0, 64}, {2048, 16, 16, 64}};
  printf("%6s %6s %6s %10s %10s\n", "cols", "pad_s", "pad_d", "us", "GB/s");
  for (Case const &c : cases) {
    size_t const ps = c.cols + c.pad_s, pd = c.cols + c.pad_d;
    if ((size_t)c.blocks * 512 * (ps > pd ? ps : pd) * 8 > cap) { printf("too big\n"); continue; }
    dim3 const grid(c.cols / 16, c.blocks);
    for (int w = 0; w < 3; w++) k_tile<512><<<grid, 256>>>(a, b, ps, pd);
    int const reps = 20;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; w++) k_tile<512><<<grid, 256>>>(a, b, ps, pd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double const bytes = 2.0 * c.blocks * 512 * c.cols * 8;
    printf("%6d %6d %6d %10.1f %10.0f\n", c.cols, c.pad_s, c.pad_d, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
  }
  return 0;
}
