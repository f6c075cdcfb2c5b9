// kh_n0.cuh -- compute_n0's second iteration (radio.c:451-462) as a device function, so that it can be a kernel of
// its own (kh_n0_pass1) or share a launch with the channel wavefronts (kh_back_a, kh_chan_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "kh_internal.h"
#include "../../include/ka9q_hip.h"

namespace kh {

constexpr int N0_GX = 64;     // workgroups (slices) per (input, block)
constexpr int N0_MAXB = 1024; // blocks per launch the smoothing pass can hold

struct N0Pass1Args {
  const float2 *spec;
  size_t spec_in_stride;
  int N, start, bincount; // radio.c:440-447: bins (start + n) mod N, n < bincount
  float *part;            // layout below
  int nblocks, ninputs, npart0, gx;
  const unsigned *sync; // direct mode (N0Fuse): sync[N0_SYNC_STRIDE pt + 1] != 0 marks the blocks whose second iteration is still due
};

struct N0FinalArgs { // kh_n0_final: partial sums -> n0 per block, smoothing (radio.c:197-202), IF power
  const float *part, *energy_part;
  int echunks, L, N, samprate, valid, have_n0;
  float *n0_state;
  kh_input_status *status;
  int nblocks, ninputs;
};

__device__ __forceinline__ float n0_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// part layout: [0, PT*N0_SLOTS) first-pass sums (left by the forward transform); then PT*N0_GX pairs {sum1, count1}
// (PT = ninputs*nblocks).  wg = slice + N0_GX*(block + nblocks*input); 128 or 256 threads; `sh` holds >= 8 floats.
// algorithmic bytes: 8*bincount per (input, block)
__device__ __forceinline__ void n0_pass1_body(N0Pass1Args const &Z, int wg, float *sh) {
  float *ps = sh, *pc = sh + 4;
  int const tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
  int const slice = wg % Z.gx, b = (wg / Z.gx) % Z.nblocks, i = wg / (Z.gx * Z.nblocks);
  const float2 *fd = Z.spec + (size_t)i * Z.spec_in_stride + (size_t)b * Z.N;
  size_t const pt = (size_t)i * Z.nblocks + b;
  if (Z.sync != nullptr && Z.sync[N0_SYNC_STRIDE * pt + 1] == 0) return; // uniform: the row pass has taken this block's iteration
  const float *p0 = Z.part + pt * N0_SLOTS;
  float *p1 = Z.part + (size_t)Z.ninputs * Z.nblocks * N0_SLOTS + pt * (2 * N0_GX);
  // radio.c:456,461: twice the mean of iteration 0 (npart0 <= 64 partials, added in a fixed order)
  float t = tid < Z.npart0 ? p0[tid] : 0.f;
  t = n0_wave_sum(t);
  if ((tid & 63) == 0) ps[tid >> 6] = t; // all partials sit in wave 0
  __syncthreads();
  float const th = (ps[0] / Z.bincount) * 2;
  __syncthreads();
  float s = 0.f, c = 0.f;
  // eight loads in flight per lane: when this shares a launch with the channel wavefronts their LDS footprint caps
  // the workgroups per CU, and one load at a time would leave the memory system idle
  int const stride = Z.gx * nt;
  for (int n0 = slice * nt + tid; n0 < Z.bincount; n0 += 8 * stride) {
    float2 x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int const n = n0 + u * stride;
      int in = Z.start + n; // radio.c:440-446
      if (in >= Z.N) in -= Z.N;
      x[u] = n < Z.bincount ? fd[in] : make_float2(INFINITY, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      float const p = x[u].x * x[u].x + x[u].y * x[u].y;
      if (p < th) {
        s += p;
        c += 1.f;
      }
    }
  }
  s = n0_wave_sum(s);
  c = n0_wave_sum(c);
  if ((tid & 63) == 0) {
    ps[tid >> 6] = s;
    pc[tid >> 6] = c;
  }
  __syncthreads();
  if (tid == 0) {
    float ss = ps[0], cc = pc[0];
    for (int w = 1; w < nw; w++) ss += ps[w], cc += pc[w];
    p1[2 * slice] = ss;
    p1[2 * slice + 1] = cc;
  }
}

// one workgroup per input, one wave per block (16 waves: 16 blocks at a time; every lane issues all of its loads
// before the first shuffle, so a block costs one memory round trip), then lane 0 applies radio.c:197-202 block by
// block from LDS
constexpr int N0_FT = 1024;
// one workgroup (any multiple of 64 threads) per input i; sv/se hold N0_MAXB floats each.  Every wave takes blocks
// wave, wave + nwaves, ...; the loads of four blocks are issued before the first shuffle.
__device__ __forceinline__ void n0_final_body(N0FinalArgs const &Z, int i, float *sv, float *se) {
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  int const nblocks = Z.nblocks;
  const float *part1 = Z.part + (size_t)Z.ninputs * nblocks * N0_SLOTS;
  constexpr int U = 4;
  for (int b0 = wave; b0 < nblocks; b0 += U * nwaves) {
    float es[U], s1[U], c1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int const b = b0 + u * nwaves;
      es[u] = s1[u] = c1[u] = 0.f;
      if (b < nblocks) {
        size_t const pt = (size_t)i * nblocks + b;
        const float *e = Z.energy_part + pt * Z.echunks;
        for (int k = lane; k < Z.echunks; k += 64) es[u] += e[k];
        if (Z.have_n0 && Z.valid) {
          float2 const sc = reinterpret_cast<const float2 *>(part1 + pt * (2 * N0_GX))[lane]; // N0_GX == 64: one per lane
          s1[u] = sc.x;
          c1[u] = sc.y;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      int const b = b0 + u * nwaves;
      float const e = n0_wave_sum(es[u]), s = n0_wave_sum(s1[u]), c = n0_wave_sum(c1[u]);
      float v = NAN;
      if (Z.have_n0) v = Z.valid ? (s / c) / ((float)Z.N * Z.samprate) : 0.f; // radio.c:433-435 ("Can this happen?"), 461, 464
      if (lane == 0 && b < nblocks) {
        se[b] = e / Z.L; // radio.c:197
        sv[b] = v;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && Z.have_n0) {
    float n0 = Z.n0_state[i];
    for (int b = 0; b < nblocks; b++) {
      float const v = sv[b];
      if (!isnan(n0))
        n0 = (float)((double)n0 + .005 * (double)(v - n0));
      else
        n0 = v;
      sv[b] = n0;
    }
    Z.n0_state[i] = n0;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    Z.status[(size_t)i * nblocks + b].if_power = se[b];
    Z.status[(size_t)i * nblocks + b].n0 = sv[b];
  }
}

} // namespace kh
