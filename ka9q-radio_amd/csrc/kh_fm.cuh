// kh_fm.cuh -- fm.c's Rice-distribution SNR correction (fm.c:142-205), shared by the filter-output epilogue
// and the FM kernels.
#pragma once
#include <hip/hip_runtime.h>

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
static __device__ double fm_snr(double r) { // fm.c:190-205
  double thetasq = r;
  double othetasq = r + 10;
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  int guard = 0; // the reference loop has no bound; NaN input falls through both tests above and exits here
  while (fabs(thetasq - othetasq) > .001 && guard++ < 10000) {
    othetasq = thetasq;
    thetasq = fm_xi(thetasq) * (1 + r) - 2;
  }
  return thetasq;
}


// fm_snr evaluated by one whole wavefront (all 64 lanes call it with the same r and get the same result).
// The two Bessel series of fm.c:152-179 are 40-term products of per-term factors, i.e. a multiplicative scan
// over lanes followed by a sum.  The reference's early exit (term < 1e-12 * sum) only drops terms below
// 1e-12 of the sum, so summing all 40 terms changes the result by less than that.
static __device__ __forceinline__ double fm_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ double fm_xi_wave(double thetasq, int lane) { // fm.c:180-186 with fm.c:152-179 inlined
  double const z = 0.25 * thetasq;
  double const t = 0.25 * (z * z);
  // lane k holds the factor that takes term k-1 to term k (k < 40); lanes >= 40 hold zero terms
  double f0 = 1.0, f1 = 1.0;
  if (lane >= 1) {
    double const dk = (double)lane;
    f0 = t / (dk * dk);         // i0: term_k = term_{k-1} * t/(k*k), term_1 = t
    f1 = t / (dk * (dk + 1.0)); // i1: term_k = term_{k-1} * t/(k*(k+1)), term_1 = 0.5*t
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double const p0 = __shfl_up(f0, o, 64), p1 = __shfl_up(f1, o, 64);
    if (lane >= o) {
      f0 *= p0;
      f1 *= p1;
    }
  }
  if (lane >= 40) f0 = 0.0, f1 = 0.0;
  double const i0 = fm_wave_sum(f0);
  double const i1 = 0.5 * z * fm_wave_sum(f1);
  double tt = (2 + thetasq) * i0 + thetasq * i1;
  tt *= tt;
  return 2 + thetasq - (0.125 * M_PI) * exp(-0.5 * thetasq) * tt;
}
static __device__ double fm_snr_wave(double r, int lane) { // fm.c:190-205
  double thetasq = r;
  double othetasq = r + 10;
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  int guard = 0;
  while (fabs(thetasq - othetasq) > .001 && guard++ < 10000) {
    othetasq = thetasq;
    thetasq = fm_xi_wave(thetasq, lane) * (1 + r) - 2;
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

// per (FM channel, block) record left by the filter-output epilogue (fm.c:53-70)
struct FmBlockStat {
  float bb_power, variance, snr, pad;
};

} // namespace kh
