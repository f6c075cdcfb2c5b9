// kh_ingest.cuh -- wire format conversion and NCO mixing of proc_samples (radio.c:132-191, osc.c:22-58), shared by
// kh_ingest (small transforms) and the fused load stage of the large forward transform.
#pragma once
#include <hip/hip_runtime.h>
#include "kh_internal.h"
#include "../../include/ka9q_hip.h"

namespace kh {

__device__ __forceinline__ void nco_phasor(const NcoArgs &a, unsigned n, double &c, double &s) {
  // step_osc (osc.c:39-50) returns p0 * prod_{k=1..n} step_k with step_k = s0 * ss^k:
  // phase(n) = phase0 + n*freq + rate*n(n+1)/2 cycles.  Evaluated in closed form in fp64.
  double const dn = (double)n;
  double ph = fma(dn, a.freq, a.phase0);
  if (a.rate != 0) ph = fma(a.rate, 0.5 * dn * (dn + 1.0), ph);
  ph -= rint(ph);
  sincospi(2.0 * ph, &s, &c);
}

__device__ __forceinline__ float2 load_raw(const unsigned char *raw, size_t n, int payload, float gain_factor) {
  float const SCALE16 = 1.f / 32767.f; // radio.c:34
  float const SCALE8 = 1.f / 127.f;    // radio.c:35
  switch (payload) {
  case KH_PT_IQ16LE: { // radio.c:159-167
    short2 const v = reinterpret_cast<const short2 *>(raw)[n];
    float const gain = SCALE16 * gain_factor;
    return make_float2(gain * v.x, gain * v.y);
  }
  case KH_PT_PCM16BE: { // radio.c:148-157
    uchar4 const v = reinterpret_cast<const uchar4 *>(raw)[n];
    float const gain = SCALE16 * gain_factor;
    return make_float2(gain * (short)((v.x << 8) | v.y), gain * (short)((v.z << 8) | v.w));
  }
  case KH_PT_IQ8: { // radio.c:169-176
    char2 const v = reinterpret_cast<const char2 *>(raw)[n];
    float const gain = SCALE8 * gain_factor;
    return make_float2(gain * (signed char)v.x, gain * (signed char)v.y);
  }
  case KH_PT_IQ12: { // radio.c:138-146
    const unsigned char *dp = raw + 3 * n;
    int const d0 = dp[0], d1 = dp[1], d2 = dp[2];
    float const gain = SCALE16 * gain_factor;
    return make_float2(gain * (short)(((d0 << 8) | d1) & 0xfff0), gain * (short)(((d1 << 8) | d2) << 4));
  }
  default: { // KH_PT_CF32: already float (no wire format in the reference; gain_factor still applies)
    float2 const v = reinterpret_cast<const float2 *>(raw)[n];
    return make_float2(gain_factor * v.x, gain_factor * v.y);
  }
  }
}


// x * phasor of one oscillator at launch-relative sample n (radio.c:180-187: complex float *= complex double)
__device__ __forceinline__ float2 nco_mix(float2 x, const NcoArgs &a, unsigned n) {
  double c, s;
  nco_phasor(a, n, c, s);
  double const re = (double)x.x * c - (double)x.y * s;
  double const im = (double)x.x * s + (double)x.y * c;
  return make_float2((float)re, (float)im);
}

} // namespace kh
