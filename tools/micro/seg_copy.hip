// micro-benchmark: what a transposing pass can reach.  Every wavefront reads 512 contiguous bytes (64 lanes x float2) and
// writes them as SEG-byte segments at a stride of `pitch` bytes (the row pass's store pattern: SEG = 128 at pitch 4096),
// or the other way round (the column pass's load pattern).  67 MB in, 67 MB out.
// hipcc --offload-arch=gfx950 -O3 tools/micro/seg_copy.hip -o tools/micro/seg_copy && tools/micro/seg_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SEG_LANES, bool SCATTER_STORE>
__global__ __launch_bounds__(256) void seg_copy(const float2 *__restrict__ in, float2 *__restrict__ out, int n1, int n2, int per_thread) {
  // matrix n1 x n2 (row-major in), transposed out in tiles: a workgroup takes a tile of SEG_LANES columns x (256/SEG_LANES * per_thread) rows
  int const tid = threadIdx.x, w = tid % SEG_LANES, r0 = tid / SEG_LANES;
  int const rows_per_wg = (256 / SEG_LANES) * per_thread;
  int const tiles_x = n2 / SEG_LANES;
  int const tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  float2 v[32];
#pragma unroll
  for (int u = 0; u < 32; u++) {
    if (u < per_thread) {
      size_t const row = (size_t)ty * rows_per_wg + r0 + u * (256 / SEG_LANES);
      size_t const a = row * n2 + tx * SEG_LANES + w;          // SEG-byte segments at pitch n2*8
      size_t const b = ((size_t)blockIdx.x * per_thread + u) * 256 + tid; // contiguous
      v[u] = SCATTER_STORE ? in[b] : in[a];
    }
  }
#pragma unroll
  for (int u = 0; u < 32; u++) {
    if (u < per_thread) {
      size_t const row = (size_t)ty * rows_per_wg + r0 + u * (256 / SEG_LANES);
      size_t const a = row * n2 + tx * SEG_LANES + w;
      size_t const b = ((size_t)blockIdx.x * per_thread + u) * 256 + tid;
      if (SCATTER_STORE) out[a] = v[u]; else out[b] = v[u];
    }
  }
}
template <int SEG_LANES, bool SC>
static void run(const char *name, float2 *in, float2 *out, int n1, int n2, int blocks) {
  int const per_thread = 32;
  size_t const total = (size_t)n1 * n2 * blocks;
  int const wgs = (int)(total / (256 * per_thread));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 3; k++) seg_copy<SEG_LANES, SC><<<wgs, 256>>>(in, out, n1 * blocks, n2, per_thread);
  hipEventRecord(e0);
  int const reps = 20;
  for (int k = 0; k < reps; k++) seg_copy<SEG_LANES, SC><<<wgs, 256>>>(in, out, n1 * blocks, n2, per_thread);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s seg %4d B  %7.2f us  %6.2f TB/s (read + write)\n", name, SEG_LANES * 8, 1e3 * ms / reps, 2.0 * total * 8 / (ms / reps * 1e-3) / 1e12);
}
int main() {
  int const n1 = 512, n2 = 512, blocks = 32; // configs[1]: 32 blocks of 2^18 points
  size_t const bytes = (size_t)n1 * n2 * blocks * 8;
  float2 *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 0, bytes);
  run<16, true>("contiguous read, segmented store", in, out, n1, n2, blocks);
  run<32, true>("contiguous read, segmented store", in, out, n1, n2, blocks);
  run<64, true>("contiguous read, segmented store", in, out, n1, n2, blocks);
  run<16, false>("segmented read, contiguous store", in, out, n1, n2, blocks);
  run<32, false>("segmented read, contiguous store", in, out, n1, n2, blocks);
  run<64, false>("segmented read, contiguous store", in, out, n1, n2, blocks);
  // plain copy
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice); hipEventRecord(e0);
  for (int k = 0; k < 20; k++) hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s            %7.2f us  %6.2f TB/s\n", "hipMemcpy device to device", 1e3 * ms / 20, 2.0 * bytes / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
