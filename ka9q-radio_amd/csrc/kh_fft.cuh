// kh_fft.cuh -- LDS-resident Stockham FFT core for gfx950 (wave64, 160 KiB LDS per CU).
//
// One workgroup transforms W independent sequences of R complex points that sit in LDS.
// Element (i, c) (point i of sequence c) lives at buf[i*SI + c*SC].  Every pass reads its radix-4
// (or radix-2) operands into registers, barriers, and writes the butterflies back into the SAME
// buffer, so a transform needs R*W*8 bytes of LDS, not twice that.  R, W, the thread count and the
// strides are template parameters: the whole pass schedule is unrolled at compile time and the
// operands live in statically indexed registers (a runtime-indexed array would go to scratch).
//
// The DFT is unnormalised, SIGN=-1 forward / +1 backward: the contract the reference relies on from
// FFTW (filter.c:81,198: fftwf_plan_dft_1d FFTW_FORWARD / FFTW_BACKWARD).
#pragma once
#include <hip/hip_runtime.h>

namespace kh {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// tw[k] = exp(-2 pi i k / R) for k in [0, R): computed in double on the host.
// For the backward transform the conjugate is taken on the fly.
template <int SIGN>
__device__ __forceinline__ float2 twiddle(const float2 *__restrict__ tw, int k) {
  float2 w = tw[k];
  if (SIGN > 0) w.y = -w.y;
  return w;
}

// One Stockham pass over sub-length LEN with stride S (LEN*S == R).
//   reads  x[q + S*(p + k*M)]           k < RADIX, M = LEN/RADIX
//   writes y[q + S*(RADIX*p + j)] = (sum_k x_k W_RADIX^{jk}) * W_LEN^{p j}
// C_FAST: consecutive threads walk the sequence index c first (use when SC == 1).
template <int R, int W, int NT, int SI, int SC, bool C_FAST, int SIGN, int LEN, int S, int RADIX>
__device__ __forceinline__ void fft_pass(float2 *buf, const float2 *__restrict__ tw, int tid) {
  constexpr int NB = R / RADIX;       // butterflies per sequence
  constexpr int ITEMS = NB * W;       // butterflies per workgroup
  static_assert(ITEMS % NT == 0, "butterflies must divide evenly over the workgroup");
  constexpr int IPT = ITEMS / NT;
  constexpr int M = LEN / RADIX;
  float2 v[IPT][RADIX];
#pragma unroll
  for (int u = 0; u < IPT; u++) {
    int const idx = tid + u * NT;
    int const c = C_FAST ? (idx % W) : (idx / NB);
    int const rest = C_FAST ? (idx / W) : (idx % NB); // = p*S + q
#pragma unroll
    for (int k = 0; k < RADIX; k++) v[u][k] = buf[(rest + k * NB) * SI + c * SC];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < IPT; u++) {
    int const idx = tid + u * NT;
    int const c = C_FAST ? (idx % W) : (idx / NB);
    int const rest = C_FAST ? (idx / W) : (idx % NB);
    int const p = rest / S;
    int const q = rest % S;
    int const obase = (q + S * RADIX * p) * SI + c * SC;
    constexpr int TSTEP = R / LEN; // W_LEN^k = tw[k*TSTEP]
    if (RADIX == 4) {
      float2 const a = v[u][0], b = v[u][1], cc = v[u][2], d = v[u][3];
      float2 const apc = cadd(a, cc), amc = csub(a, cc), bpd = cadd(b, d), bmd = csub(b, d);
      // -i*bmd forward, +i*bmd backward
      float2 const jb = SIGN < 0 ? make_float2(bmd.y, -bmd.x) : make_float2(-bmd.y, bmd.x);
      float2 y0 = cadd(apc, bpd);
      float2 y1 = cadd(amc, jb);
      float2 y2 = csub(apc, bpd);
      float2 y3 = csub(amc, jb);
      if (M > 1) { // the last pass (LEN == RADIX) has p == 0: unit twiddles
        y1 = cmul(y1, twiddle<SIGN>(tw, p * TSTEP));
        y2 = cmul(y2, twiddle<SIGN>(tw, 2 * p * TSTEP));
        y3 = cmul(y3, twiddle<SIGN>(tw, 3 * p * TSTEP));
      }
      buf[obase] = y0;
      buf[obase + S * SI] = y1;
      buf[obase + 2 * S * SI] = y2;
      buf[obase + 3 * S * SI] = y3;
    } else {
      float2 const a = v[u][0], b = v[u][1];
      float2 y0 = cadd(a, b);
      float2 y1 = csub(a, b);
      if (M > 1) y1 = cmul(y1, twiddle<SIGN>(tw, p * TSTEP));
      buf[obase] = y0;
      buf[obase + S * SI] = y1;
    }
  }
  __syncthreads();
}

template <int R, int W, int NT, int SI, int SC, bool C_FAST, int SIGN, int LEN, int S>
struct FftPasses {
  static __device__ __forceinline__ void run(float2 *buf, const float2 *__restrict__ tw, int tid) {
    if constexpr (LEN > 1) {
      constexpr int RADIX = (LEN % 4 == 0) ? 4 : 2;
      fft_pass<R, W, NT, SI, SC, C_FAST, SIGN, LEN, S, RADIX>(buf, tw, tid);
      FftPasses<R, W, NT, SI, SC, C_FAST, SIGN, LEN / RADIX, S * RADIX>::run(buf, tw, tid);
    }
  }
};

// In-LDS transform; the caller has filled buf and synchronised.  On return (after a barrier)
// buf holds the transform in natural order.
template <int R, int W, int NT, int SI, int SC, bool C_FAST, int SIGN>
__device__ __forceinline__ void lds_fft(float2 *buf, const float2 *__restrict__ tw, int tid) {
  FftPasses<R, W, NT, SI, SC, C_FAST, SIGN, R, 1>::run(buf, tw, tid);
}

// Strided tile accesses go through buffer instructions: ONE per-lane byte offset register serves every row of the
// tile, the row offset (wave-uniform) travels in an SGPR.  With 64-bit flat addresses the compiler keeps one address
// pair per outstanding load alive: 64 VGPRs for a 32-row column, which costs a workgroup per CU.
typedef unsigned int kh_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) { // base: wave-uniform
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  kh_u32x2 const t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
}
// AUX: the instruction's cache policy bits (gfx942/gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void buf_store2(float2 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  kh_u32x2 t;
  t.x = __float_as_uint(v.x);
  t.y = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, AUX);
}

} // namespace kh
