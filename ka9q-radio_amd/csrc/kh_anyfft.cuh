// kh_anyfft.cuh -- LDS transform of any 2,3,5,7-smooth size up to KH_MAX_SMALL_N, for the sizes the reference's
// nextfastfft (misc.c:129-148) produces: main.c:326-330 gives N = 7680 = 2^9*3*5 at 192 kS/s, N/4 = 1920 per channel.
//
// Stockham autosort passes between two LDS buffers (no in-place trick: the radices are run-time values), radix 4 / 2
// butterflies written out, radix 3 / 5 / 7 as direct DFTs with the roots taken from the same twiddle table.  The
// power-of-two sizes keep their unrolled, in-place kernels (kh_fft.cuh); this path trades speed for generality.
// Unnormalised DFT, SIGN = -1 forward / +1 backward (the FFTW contract, filter.c:81,198).
#pragma once
#include <hip/hip_runtime.h>
#include "kh_internal.h"
#include "kh_fft.cuh"

namespace kh {

template <int RDX, int SIGN>
__device__ __forceinline__ void any_pass(const float2 *__restrict__ src, float2 *__restrict__ dst, int n, int LEN, int S,
                                         const float2 *__restrict__ tw, int tid, int nt) {
  int const NB = n / RDX;      // butterflies
  int const tstep = n / LEN;   // W_LEN^k = tw[k*tstep]
  float2 root[RDX];            // W_RDX^m
#pragma unroll
  for (int m = 0; m < RDX; m++) root[m] = twiddle<SIGN>(tw, m * (n / RDX));
  for (int rest = tid; rest < NB; rest += nt) {
    int const p = rest / S, q = rest - p * S;
    float2 x[RDX], y[RDX];
#pragma unroll
    for (int k = 0; k < RDX; k++) x[k] = src[rest + k * NB]; // = q + S*(p + k*LEN/RDX)
    if constexpr (RDX == 2) {
      y[0] = cadd(x[0], x[1]);
      y[1] = csub(x[0], x[1]);
    } else if constexpr (RDX == 4) {
      float2 const apc = cadd(x[0], x[2]), amc = csub(x[0], x[2]), bpd = cadd(x[1], x[3]), bmd = csub(x[1], x[3]);
      float2 const jb = SIGN < 0 ? make_float2(bmd.y, -bmd.x) : make_float2(-bmd.y, bmd.x);
      y[0] = cadd(apc, bpd);
      y[1] = cadd(amc, jb);
      y[2] = csub(apc, bpd);
      y[3] = csub(amc, jb);
    } else {
#pragma unroll
      for (int j = 0; j < RDX; j++) {
        float2 acc = x[0];
#pragma unroll
        for (int k = 1; k < RDX; k++) acc = cadd(acc, cmul(x[k], root[(j * k) % RDX]));
        y[j] = acc;
      }
    }
    int const obase = q + S * RDX * p;
    dst[obase] = y[0];
#pragma unroll
    for (int j = 1; j < RDX; j++) {
      int idx = (p * j * tstep) % n; // W_LEN^{p j}
      dst[obase + j * S] = LEN > RDX ? cmul(y[j], twiddle<SIGN>(tw, idx)) : y[j];
    }
  }
}

// transforms the plan.n points in `a` (the caller has filled it and synchronised), using `b` as the other buffer;
// returns the buffer that holds the result (after a barrier)
template <int SIGN>
__device__ __forceinline__ float2 *any_fft(float2 *a, float2 *b, AnyPlan const &plan, const float2 *__restrict__ tw, int tid,
                                           int nt) {
  int LEN = plan.n, S = 1;
  for (int k = 0; k < plan.npass; k++) {
    int const r = plan.radix[k];
    switch (r) {
    case 4: any_pass<4, SIGN>(a, b, plan.n, LEN, S, tw, tid, nt); break;
    case 2: any_pass<2, SIGN>(a, b, plan.n, LEN, S, tw, tid, nt); break;
    case 3: any_pass<3, SIGN>(a, b, plan.n, LEN, S, tw, tid, nt); break;
    case 5: any_pass<5, SIGN>(a, b, plan.n, LEN, S, tw, tid, nt); break;
    default: any_pass<7, SIGN>(a, b, plan.n, LEN, S, tw, tid, nt); break;
    }
    __syncthreads();
    LEN /= r;
    S *= r;
    float2 *t = a;
    a = b;
    b = t;
  }
  return a;
}

} // namespace kh
