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
// fm_snr (fm.c:190-205): the reference's fixed-point iteration thetasq <- xi(thetasq) (1 + r) - 2, run until two
// iterates differ by at most .001.  It takes 3 rounds for a strong signal and 30 to several hundred for a block of
// noise (r just above pi/(4-pi)) -- an idle channel is the expensive case -- and every round waits for the one before,
// so what counts is the latency of one round.  xi on [0, 100] comes from piecewise degree-11 polynomials
// (tools/fit_fm_xi.py; within 2e-14 (2 + thetasq) of the series above) evaluated by Estrin's scheme: about seven
// dependent fp64 operations a round instead of two 40-term series with a division per term and an exp.  The iterates
// move slowly, so the coefficients of the current segment stay in registers and are reloaded (scalar loads: the
// argument is the same in every lane) only when an iterate leaves the segment.  Outside [0, 100] (not reached from
// fm_snr's own range checks) the series is used.
static __constant__ double kh_xi_edges[KH_XI_SEGS + 1] = KH_XI_EDGES;
static __constant__ double kh_xi_coeffs[KH_XI_SEGS][KH_XI_DEG + 1] = KH_XI_COEFFS;
static __constant__ int kh_xi_lut[KH_XI_NLUT] = KH_XI_LUT;
static_assert(KH_XI_DEG == 11, "fm_snr's Estrin evaluation is written for degree 11");

__device__ __forceinline__ double wave_uniform(double v) { // the first active lane's value, as a scalar
  int const lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// r must be the same in every active lane of the wavefront (both callers pass a wavefront-wide reduction).
static __device__ double fm_snr(double r) { // fm.c:190-205
  r = wave_uniform(r);
  double thetasq = r;
  double othetasq = r + 10;
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  double const r1 = 1 + r;
  double lo = 1.0, hi = 0.0, mid = 0.0, scale = 0.0, c[KH_XI_DEG + 1]; // empty segment: the first round loads one
#pragma unroll
  for (int k = 0; k <= KH_XI_DEG; k++) c[k] = 0.0;
  int guard = 0; // the reference loop has no bound; NaN input falls through both tests above and exits here
  while (fabs(thetasq - othetasq) > .001 && guard++ < 10000) {
    othetasq = thetasq;
    if (!(thetasq >= 0.0 && thetasq <= 100.0)) {
      thetasq = fm_xi(thetasq) * r1 - 2;
      continue;
    }
    if (!(thetasq >= lo && thetasq < hi)) {
      int g = (int)(thetasq * (1.0 / KH_XI_GRID));
      g = __builtin_amdgcn_readfirstlane(g < KH_XI_NLUT ? g : KH_XI_NLUT - 1);
      int const seg = kh_xi_lut[g];
      double const e1 = kh_xi_edges[seg + 1];
      lo = kh_xi_edges[seg];
      hi = seg == KH_XI_SEGS - 1 ? 101.0 : e1; // thetasq == 100 belongs to the last segment
      mid = 0.5 * (lo + e1);
      scale = 2.0 / (e1 - lo);
#pragma unroll
      for (int k = 0; k <= KH_XI_DEG; k++) c[k] = kh_xi_coeffs[seg][k];
    }
    double const u = (thetasq - mid) * scale;
    double const u2 = u * u, u4 = u2 * u2, u8 = u4 * u4;
    double const p01 = fma(c[1], u, c[0]), p23 = fma(c[3], u, c[2]), p45 = fma(c[5], u, c[4]);
    double const p67 = fma(c[7], u, c[6]), p89 = fma(c[9], u, c[8]), pab = fma(c[11], u, c[10]);
    double const q0 = fma(p23, u2, p01), q1 = fma(p67, u2, p45), q2 = fma(pab, u2, p89);
    double const xi = fma(q2, u8, fma(q1, u4, q0));
    thetasq = xi * r1 - 2;
  }
  return thetasq;
}


// cargf (fm.c:89) for the channel wavefronts, eight per lane: octant reduction with v_rcp_f32, atan(t) = t P(t^2) on
// [0, 1] with a degree-8 P (1.0e-7 absolute over a 200001-point grid in fp32; with the reciprocal's ulp ~2.5e-7), the
// usual reflections, and 0 for a zero argument as cargf gives.  About 25 instructions against ~50 of the library's
// atan2f; the angle feeds audio that ends up as int16 and statistics held to 1e-4.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
  float const ax = fabsf(x), ay = fabsf(y);
  float const mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __builtin_amdgcn_rcpf(mx);
  t = mx > 1e-37f ? t : 0.f;
  float const u = t * t;
  float p = 0.0027662834618240595f;
  p = fmaf(p, u, -0.015731249004602432f);
  p = fmaf(p, u, 0.04213762283325195f);
  p = fmaf(p, u, -0.07456854730844498f);
  p = fmaf(p, u, 0.10618370771408081f);
  p = fmaf(p, u, -0.14197798073291779f);
  p = fmaf(p, u, 0.1999187171459198f);
  p = fmaf(p, u, -0.333330363035202f);
  p = fmaf(p, u, 1.0f);
  float r = p * t;
  r = ay > ax ? 1.57079632679489662f - r : r;
  r = (__float_as_int(x) < 0) ? 3.14159265358979324f - r : r; // sign bit, so that -0 reflects as cargf does
  return copysignf(r, y);
}

// scaleclip: audio.c:22-28
__device__ __forceinline__ short scaleclip(float x) {
  if (x >= 1.0f) return 32767;
  if (x <= -1.0f) return -32768;
  return (short)(32767 * x); // truncation toward zero
}
__device__ __forceinline__ short pcm_word(float x, int big_endian) {
  short const v = scaleclip(x);
  if (!big_endian) return v;
  unsigned short const u = (unsigned short)v;
  return (short)(unsigned short)((u << 8) | (u >> 8)); // htons, audio.c:50,101
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
constexpr double FM_DECAY_K8 = FM_DECAY * FM_DECAY * FM_DECAY * FM_DECAY * FM_DECAY * FM_DECAY * FM_DECAY * FM_DECAY;

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
