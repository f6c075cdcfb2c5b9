// kh_fm.cuh -- fm.c's Rice-distribution SNR correction (fm.c:142-205), shared by the filter-output epilogue
// and the FM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "kh_internal.h"
#include "kh_fm_xi_cheb.h"

namespace kh {
static __device__ double fm_i0(double z) { // fm.c:152-163
  double const t = 0.25 * (z * z);
  double sum = 1 + t;
  double term = t;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * k);
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return sum;
}
static __device__ double fm_i1(double z) { // fm.c:166-179
  double const t = 0.25 * (z * z);
  double term = 0.5 * t;
  double sum = 1 + term;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * (k + 1));
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return 0.5 * z * sum;
}
static __device__ double fm_xi(double thetasq) { // fm.c:180-186
  double t = (2 + thetasq) * fm_i0(0.25 * thetasq) + thetasq * fm_i1(0.25 * thetasq);
  t *= t;
  return 2 + thetasq - (0.125 * M_PI) * exp(-0.5 * thetasq) * t;
}
// fm_xi on [0, 100] from a piecewise Chebyshev expansion (tools/fit_fm_xi.py; within 6e-13 of the series above):
// ~50 fp64 operations instead of two 40-term series with a division per term and an exp.  fm_snr's fixed-point
// iteration below runs it 3 times for a strong signal, 30 and more for a block of noise (r just above pi/(4-pi)),
// all of it on one lane's critical path.  Outside [0, 100] (never reached from fm_snr) the series is used.
static __constant__ double kh_xi_edges[KH_XI_SEGS + 1] = KH_XI_EDGES;
static __constant__ double kh_xi_coeffs[KH_XI_SEGS][KH_XI_DEG + 1] = KH_XI_COEFFS;
static __device__ double fm_xi_fast(double thetasq) {
  if (!(thetasq >= 0.0 && thetasq <= 100.0)) return fm_xi(thetasq);
  int seg = 0;
#pragma unroll
  for (int k = 1; k < KH_XI_SEGS; k++) seg += thetasq > kh_xi_edges[k];
  double const a = kh_xi_edges[seg], b = kh_xi_edges[seg + 1];
  double const x = (2.0 * thetasq - a - b) / (b - a);
  const double *c = kh_xi_coeffs[seg];
  double b1 = 0.0, b2 = 0.0;
#pragma unroll
  for (int k = KH_XI_DEG; k >= 1; k--) { // Clenshaw
    double const t = fma(2.0 * x, b1, c[k] - b2);
    b2 = b1;
    b1 = t;
  }
  return fma(x, b1, c[0] - b2);
}
static __device__ double fm_snr(double r) { // fm.c:190-205
  double thetasq = r;
  double othetasq = r + 10;
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  int guard = 0; // the reference loop has no bound; NaN input falls through both tests above and exits here
  while (fabs(thetasq - othetasq) > .001 && guard++ < 10000) {
    othetasq = thetasq;
    thetasq = fm_xi_fast(thetasq) * (1 + r) - 2;
  }
  return thetasq;
}


// block-wide sum for workgroups of NT threads (NT a multiple of 64, <= 1024); `red` holds NT/64 floats.
// Every thread gets the total.  Two barriers.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) t += red[w];
  __syncthreads();
  return t;
}

constexpr double FM_DECAY = 0.99376949; // fm.c:111,130

// fm.c:86-111 for one block that sits in LDS, for everything that does not depend on the neighbouring blocks:
// the conjugate-product discriminator and threshold-extension weight of samples 1 .. olen-1 (sample 0 needs the
// last sample of the block before and the squelch history: kh_fm_finish supplies it as part of the carry), and the
// de-emphasis integrator run from a ZERO carry-in.  y[0 .. olen) is overwritten.  Leaves (ang, s0) per sample in
// `wo`, the block's carry-out for carry-in zero in *carry_out, and returns the sum of the angles (every thread).
// The squelch decision is not known here: the result is simply not used by kh_fm_finish for a closed block.
template <int NT>
__device__ __forceinline__ float fm_front(float2 *y, int olen, float gain, int flat, float var, FmWork *__restrict__ wo,
                                          double *__restrict__ carry_out, float *red, double *w_A, double *w_B) {
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // discriminator in place: sample n reads y[n] and y[n-1] and leaves (ang, weight) in y[n].  Batches of NT samples
  // run from the top down and read before anyone writes, so a slot is overwritten only after both readers are done.
  float angsum = 0.f;
  int const nbatch = (olen + NT - 1) / NT;
#pragma unroll 1
  for (int j = nbatch - 1; j >= 0; j--) {
    int const n = tid + j * NT;
    float ang = 0.f, amp = 0.f;
    if (n >= 1 && n < olen) {
      float2 const a = y[n], b = y[n - 1];
      float2 const p = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); // y[n] * conj(y[n-1])
      ang = atan2f(p.y, p.x);       // cargf, fm.c:89
      amp = hypotf(p.x, p.y) / var; // cabsf(p) / fm_variance, fm.c:91
      if (amp > 1) amp = 1;
      angsum += ang;
    }
    __syncthreads();
    if (n < olen) y[n] = make_float2(ang, amp);
    __syncthreads();
  }
  // de-emphasis integrator from a zero carry-in: contiguous chunk per thread, affine maps scanned over the block
  int const CH = (olen + NT - 1) / NT;
  int const n0 = tid * CH < olen ? tid * CH : olen;
  int const n1 = (n0 + CH < olen) ? n0 + CH : olen;
  double A = 1.0, B = 0.0; // this chunk's map carry_out = A*carry_in + B
  if (!flat) {
    for (int n = n0; n < n1; n++) {
      float2 const aw = y[n];
      double const x = (double)aw.x * .114 * (double)gain * (double)aw.y; // fm.c:110
      B = (B + x) * FM_DECAY;
      A *= FM_DECAY;
    }
  }
  double Ai = A, Bi = B;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double const Ap = __shfl_up(Ai, o, 64), Bp = __shfl_up(Bi, o, 64);
    if (lane >= o) {
      Bi = Ai * Bp + Bi;
      Ai = Ai * Ap;
    }
  }
  if (lane == 63) {
    w_A[wave] = Ai;
    w_B[wave] = Bi;
  }
  __syncthreads();
  double c = 0.0; // zero carry-in at the block start
  for (int w = 0; w < wave; w++) c = w_A[w] * c + w_B[w];
  double const Ae = __shfl_up(Ai, 1, 64), Be = __shfl_up(Bi, 1, 64);
  double cin = lane == 0 ? c : Ae * c + Be; // carry into this thread's chunk
  if (tid == NT - 1) *carry_out = Ai * c + Bi;
  for (int n = n0; n < n1; n++) {
    float2 const aw = y[n];
    float s0;
    if (flat) {
      s0 = aw.x * gain; // fm.c:105
    } else {
      double const x = (double)aw.x * .114 * (double)gain * (double)aw.y;
      cin += x;        // samples[n] = lastaudio += ...
      s0 = (float)cin;
      cin *= FM_DECAY; // lastaudio *= 0.99376949
    }
    y[n] = make_float2(aw.x, s0);
  }
  __syncthreads();
  for (int n = tid; n < olen; n += NT) {
    float2 const v = y[n];
    wo[n] = FmWork{v.x, v.y};
  }
  return block_sum<NT>(angsum, red);
}

} // namespace kh
