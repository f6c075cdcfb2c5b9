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
  // bracket mode: cand != nullptr; sync then holds the column pass's window-energy partials (floats), ncols of them per block
  const float *cand;
  int ncols, nrows, rwaves; // column tiles, row tiles, wavefronts per row tile
  int no_reread = 0;        // a masked spectrum (SpecMask): the row pass has finished every block whose bracket fails
};

// ---- bracket mode.  radio.c:451-462's second iteration keeps the bins below th = 2 x (mean of the counted bins), which
// is known only when every row tile of the block is done.  By Parseval the counted bins hold at most N x (energy of the
// block's N window samples), so thP = 2 N E / bincount >= th, with equality when the excluded bins (the passband around
// DC) are empty.  The row pass therefore adds up, exactly, what lies below N0_BR_LO thP, and stores the bins in
// [N0_BR_LO thP, N0_BR_HI thP) -- a few per cent of a noise-like spectrum -- as candidates; once th is known (the next
// kernel on the stream) the candidates below it are added.  The result is the exact second iteration whenever
// N0_BR_LO thP <= th < N0_BR_HI thP and no wavefront's candidates outgrew their N0_CAPW slots; otherwise (more than a
// quarter of the power inside the excluded range, or a spectrum with a step just below the threshold) the block's second
// iteration re-reads the spectrum as it does without the mode.  Which path a block takes depends on its data alone, not
// on timing: results are reproducible run to run, and no workgroup ever waits for another.
constexpr float N0_BR_LO = 0.75f, N0_BR_HI = 1.01f;
constexpr int N0_CAPW = 512; // candidates a row-pass wavefront (2048 bins) can leave
// per (input, block), K = row tiles x wavefronts per tile: [3][K] floats (candidates found, sum and count below the
// bracket), then [K][N0_CAPW] candidates
// |x|^2, the same bits wherever it is taken (the row pass compares a bin, and later stores it, from two evaluations)
__device__ __forceinline__ float n0_power(float2 x) {
#pragma clang fp contract(off)
  return x.x * x.x + x.y * x.y;
}
// one wavefront: the bracket's centre from the column pass's partials (the same operations wherever it is called)
__device__ __forceinline__ float n0_parseval_th(const float *ewin, int ncols, int N, int bincount) {
  int const lane = threadIdx.x & 63;
  float t = lane < ncols ? ewin[lane] : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  return ((t * (float)N) / bincount) * 2;
}

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

// ---- bracket mode's second step, one task per (input, block, row tile), in two halves: everything a task may need is
// requested before the first value is used, so a task costs one memory round trip (the counts and, with them, the first
// N0_PRE slots of every wavefront, filled or not; only a wavefront with more candidates costs a second trip).
constexpr int N0_PRE = 64;                 // slots per wavefront requested with the counts (filled or not); more: a second trip
constexpr int N0_MAXV = 8 * N0_PRE / 128;  // per thread: <= 8 wavefronts per tile, >= 128 threads
struct N0Early {
  float v[N0_MAXV]; // the first N0_PRE candidate slots of each of the tile's wavefronts
  float hk[4];      // the block's candidate counts (K <= 512)
  float t, ew, n, slo, clo;
};
// task = tile + gx*(block + nblocks*input)
__device__ __forceinline__ void n0_bracket_issue(N0Pass1Args const &Z, int task, N0Early &E) {
  int const tid = threadIdx.x, nt = blockDim.x;
  int const slice = task % Z.gx, b = (task / Z.gx) % Z.nblocks, i = task / (Z.gx * Z.nblocks);
  size_t const pt = (size_t)i * Z.nblocks + b;
  int const K = Z.nrows * Z.rwaves;
  const float *hdr = Z.cand + pt * ((size_t)K * (N0_CAPW + 3));
  const float *rb = hdr + 3 * (size_t)K + (size_t)slice * Z.rwaves * N0_CAPW; // the tile's wavefronts follow one another
  int const total = slice < Z.nrows ? Z.rwaves * N0_PRE : 0;
#pragma unroll
  for (int u = 0; u < N0_MAXV; u++) {
    int const q = tid + u * nt;
    E.v[u] = q < total ? rb[(q / N0_PRE) * N0_CAPW + (q & (N0_PRE - 1))] : INFINITY;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) E.hk[u] = tid + u * nt < K ? hdr[tid + u * nt] : 0.f;
  E.t = tid < Z.npart0 ? Z.part[pt * N0_SLOTS + tid] : 0.f;
  E.ew = tid < Z.ncols ? reinterpret_cast<const float *>(Z.sync)[N0_SYNC_STRIDE * pt + tid] : 0.f;
  E.n = E.slo = E.clo = 0.f;
  if (tid < Z.rwaves && total) {
    int const idx = slice * Z.rwaves + tid;
    E.n = hdr[idx];
    E.slo = hdr[K + idx]; // what the wavefront found below the bracket
    E.clo = hdr[2 * K + idx];
  }
}
// true: done; false (uniform over the workgroup, and the same for every tile of the block): the block's second iteration
// has to re-read the spectrum (n0_pass1_reread).  `sh` holds >= N0_SH floats (workgroups of up to 8 wavefronts).
constexpr int N0_SH = 24;
__device__ __forceinline__ bool n0_bracket_finish(N0Pass1Args const &Z, int task, N0Early const &E, float *sh) {
  float *ps = sh, *pc = sh + 8, *nsh = sh + 16;
  int const tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
  int const slice = task % Z.gx, b = (task / Z.gx) % Z.nblocks, i = task / (Z.gx * Z.nblocks);
  size_t const pt = (size_t)i * Z.nblocks + b;
  float *p1 = Z.part + (size_t)Z.ninputs * Z.nblocks * N0_SLOTS + pt * (2 * N0_GX);
  int const total = slice < Z.nrows ? Z.rwaves * N0_PRE : 0;
  int bad = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) bad |= E.hk[u] > (float)N0_CAPW;
  if (tid < Z.rwaves) nsh[tid] = E.n;
  if (tid < 64) { // radio.c:456,461 and the bracket's centre: both sit in wave 0 (npart0, ncols <= 64)
    float const t = n0_wave_sum(E.t), e = n0_wave_sum(E.ew);
    if (tid == 0) ps[0] = t, pc[0] = ((e * (float)Z.N) / Z.bincount) * 2; // (n0_parseval_th's operations)
  }
  __syncthreads();
  float const th = (ps[0] / Z.bincount) * 2, thP = pc[0];
  bad |= !(th >= thP * N0_BR_LO && th < thP * N0_BR_HI);
  bad = __syncthreads_or(bad);
  if (bad) return false;
  float s = E.slo, c = E.clo;
#pragma unroll
  for (int u = 0; u < N0_MAXV; u++) {
    int const q = tid + u * nt;
    if (q < total && (q & (N0_PRE - 1)) < (int)nsh[q / N0_PRE] && E.v[u] < th) {
      s += E.v[u];
      c += 1.f;
    }
  }
  if (total) { // wavefronts with more than N0_PRE candidates (a band of plain noise: ~180 of 2048 bins): the rest, now
    size_t const K = (size_t)Z.nrows * Z.rwaves;
    const float *rb = Z.cand + pt * (K * (N0_CAPW + 3)) + 3 * K + (size_t)slice * Z.rwaves * N0_CAPW;
    for (int w = 0; w < Z.rwaves; w++) {
      int const n = (int)nsh[w]; // uniform
      for (int k0 = N0_PRE + tid; k0 < n; k0 += 4 * nt) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = k0 + u * nt < n ? rb[w * N0_CAPW + k0 + u * nt] : INFINITY;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (x[u] < th) {
            s += x[u];
            c += 1.f;
          }
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
  if (slice == 0 && tid >= Z.gx && tid < N0_GX) reinterpret_cast<float2 *>(p1)[tid] = make_float2(0.f, 0.f); // kh_n0_final adds all N0_GX pairs
  __syncthreads(); // (sh may be used again)
  return true;
}

// part layout: [0, PT*N0_SLOTS) first-pass sums (left by the forward transform); then PT*N0_GX pairs {sum1, count1}
// (PT = ninputs*nblocks).  wg = slice + gx*(block + nblocks*input); 128 or 256 threads; `sh` holds >= N0_SH floats.
// algorithmic bytes: 8*bincount per (input, block)
__device__ __forceinline__ void n0_pass1_reread(N0Pass1Args const &Z, int wg, float *sh) {
  float *ps = sh, *pc = sh + 8;
  int const tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
  int const slice = wg % Z.gx, b = (wg / Z.gx) % Z.nblocks, i = wg / (Z.gx * Z.nblocks);
  const float2 *fd = Z.spec + (size_t)i * Z.spec_in_stride + (size_t)b * Z.N;
  size_t const pt = (size_t)i * Z.nblocks + b;
  const float *p0 = Z.part + pt * N0_SLOTS;
  float *p1 = Z.part + (size_t)Z.ninputs * Z.nblocks * N0_SLOTS + pt * (2 * N0_GX);
  if (slice == 0 && tid >= Z.gx && tid < N0_GX) reinterpret_cast<float2 *>(p1)[tid] = make_float2(0.f, 0.f); // kh_n0_final adds all N0_GX pairs
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
__device__ __forceinline__ void n0_pass1_body(N0Pass1Args const &Z, int wg, float *sh) {
  if (Z.cand != nullptr) { // bracket mode
    N0Early E;
    n0_bracket_issue(Z, wg, E);
    if (n0_bracket_finish(Z, wg, E, sh) || Z.no_reread) return; // (uniform)
  } else if (Z.sync != nullptr) {
    int const b = (wg / Z.gx) % Z.nblocks, i = wg / (Z.gx * Z.nblocks);
    if (Z.sync[N0_SYNC_STRIDE * ((size_t)i * Z.nblocks + b) + 1] == 0) return; // uniform: the row pass has taken this block's iteration
  }
  n0_pass1_reread(Z, wg, sh);
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
