// kh_regfft.cuh -- small FFTs (n = 2..32) that live entirely in one lane's registers.
//
// The large forward transform keeps a tile of sequences in registers (every lane owns 8..32 points of one
// sequence), does one small transform per lane, exchanges through LDS once, and does the second small transform.
// Every loop here is unrolled at compile time and every index is a constant after unrolling, so the arrays are
// plain VGPRs (a runtime index would send them to scratch memory).
//
// Unnormalised DFT, SIGN = -1 forward / +1 backward: the FFTW contract the reference relies on (filter.c:81,198).
#pragma once
#include <hip/hip_runtime.h>

namespace kh {

// cos(2 pi k / 32), k = 0..8
constexpr double KH_COS32[9] = {1.0,
                                0.98078528040323044912618223613424,
                                0.92387953251128675612818318939679,
                                0.83146961230254523707878837761791,
                                0.70710678118654752440084436210485,
                                0.55557023301960222474283081394853,
                                0.38268343236508977172845998403040,
                                0.19509032201612826784828486847702,
                                0.0};
__host__ __device__ constexpr double cos32(int k) {
  k &= 31;
  if (k > 16) k = 32 - k;
  return k <= 8 ? KH_COS32[k] : -KH_COS32[16 - k];
}
__host__ __device__ constexpr double sin32(int k) { return cos32(k - 8); }

// x * exp(SIGN * 2 pi i * k / 32) with k known at compile time after unrolling
template <int SIGN>
__host__ __device__ __forceinline__ float2 mul_w32(float2 x, int k) {
  k &= 31;
  if (k == 0) return x;
  if (k == 16) return make_float2(-x.x, -x.y);
  if (k == 8) return SIGN > 0 ? make_float2(-x.y, x.x) : make_float2(x.y, -x.x);   // * (+-i)
  if (k == 24) return SIGN > 0 ? make_float2(x.y, -x.x) : make_float2(-x.y, x.x);  // * (-+i)
  float const c = (float)cos32(k);
  float const s = (float)(SIGN * sin32(k));
  return make_float2(x.x * c - x.y * s, x.x * s + x.y * c);
}

// One Stockham pass (same indexing as kh_fft.cuh's fft_pass) from x to y, both in registers.
template <int N, int SIGN, int LEN, int S>
__host__ __device__ __forceinline__ void reg_pass(const float2 (&x)[N], float2 (&y)[N]) {
  constexpr int RADIX = (LEN % 4 == 0) ? 4 : 2;
  constexpr int NB = N / RADIX;
  constexpr int TSTEP = 32 / LEN; // W_LEN^k = W_32^(k*TSTEP)
#pragma unroll
  for (int rest = 0; rest < NB; rest++) {
    int const p = rest / S, q = rest % S;
    int const o = q + S * RADIX * p;
    if constexpr (RADIX == 4) {
      float2 const a = x[rest], b = x[rest + NB], c = x[rest + 2 * NB], d = x[rest + 3 * NB];
      float2 const apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
      float2 const bpd = make_float2(b.x + d.x, b.y + d.y), bmd = make_float2(b.x - d.x, b.y - d.y);
      float2 const jb = SIGN < 0 ? make_float2(bmd.y, -bmd.x) : make_float2(-bmd.y, bmd.x);
      y[o] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
      y[o + S] = mul_w32<SIGN>(make_float2(amc.x + jb.x, amc.y + jb.y), p * TSTEP);
      y[o + 2 * S] = mul_w32<SIGN>(make_float2(apc.x - bpd.x, apc.y - bpd.y), 2 * p * TSTEP);
      y[o + 3 * S] = mul_w32<SIGN>(make_float2(amc.x - jb.x, amc.y - jb.y), 3 * p * TSTEP);
    } else {
      float2 const a = x[rest], b = x[rest + NB];
      y[o] = make_float2(a.x + b.x, a.y + b.y);
      y[o + S] = mul_w32<SIGN>(make_float2(a.x - b.x, a.y - b.y), p * TSTEP);
    }
  }
}

// v <- DFT_N(v), natural order in and out.  N in {2, 4, 8, 16, 32}.
template <int N, int SIGN>
__host__ __device__ __forceinline__ void reg_fft(float2 (&v)[N]) {
  static_assert(N == 2 || N == 4 || N == 8 || N == 16 || N == 32, "register FFT sizes");
  float2 t[N];
  if constexpr (N == 2) {
    reg_pass<N, SIGN, 2, 1>(v, t);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
  } else if constexpr (N == 4) {
    reg_pass<N, SIGN, 4, 1>(v, t);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
  } else if constexpr (N == 8) {
    reg_pass<N, SIGN, 8, 1>(v, t);
    reg_pass<N, SIGN, 2, 4>(t, v);
  } else if constexpr (N == 16) {
    reg_pass<N, SIGN, 16, 1>(v, t);
    reg_pass<N, SIGN, 4, 4>(t, v);
  } else {
    reg_pass<N, SIGN, 32, 1>(v, t);
    reg_pass<N, SIGN, 8, 4>(t, v);
    reg_pass<N, SIGN, 2, 16>(v, t);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
  }
}

// ---- the same transforms on packed pairs: gfx950 issues v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (two fp32 operations
// per lane, one issue slot), and a complex value is such a pair.  For kernels bound by vector issue rather than by memory
// (kh_chan2: ~70 % of its SIMD time is instruction issue, profiles/r03 counters): a 32-point transform is 259 vector
// instructions instead of 420, a complex product 2-3 instead of 4.  Same operations per component, so the same roundings
// up to how products and sums are fused.
typedef float v2f __attribute__((ext_vector_type(2)));
namespace pk {
__device__ __forceinline__ v2f load(float2 t) { return (v2f){t.x, t.y}; }
__device__ __forceinline__ float2 store(v2f t) { return make_float2(t.x, t.y); }
// a * b
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
  v2f const asw = __builtin_shufflevector(a, a, 1, 0);
  return __builtin_elementwise_fma(asw, (v2f){-b.y, b.y}, a * (v2f){b.x, b.x});
}
template <int SIGN>
__device__ __forceinline__ v2f mul_w32(v2f x, int k) {
  k &= 31;
  if (k == 0) return x;
  if (k == 16) return -x;
  v2f const sw = __builtin_shufflevector(x, x, 1, 0);
  if (k == 8) return SIGN > 0 ? (v2f){-sw.x, sw.y} : (v2f){sw.x, -sw.y};  // * (+-i)
  if (k == 24) return SIGN > 0 ? (v2f){sw.x, -sw.y} : (v2f){-sw.x, sw.y}; // * (-+i)
  float const c = (float)cos32(k), s = (float)(SIGN * sin32(k));
  return __builtin_elementwise_fma(sw, (v2f){-s, s}, x * (v2f){c, c}); // (x.x c - x.y s, x.y c + x.x s)
}
template <int N, int SIGN, int LEN, int S>
__device__ __forceinline__ void reg_pass(const v2f (&x)[N], v2f (&y)[N]) {
  constexpr int RADIX = (LEN % 4 == 0) ? 4 : 2;
  constexpr int NB = N / RADIX;
  constexpr int TSTEP = 32 / LEN;
#pragma unroll
  for (int rest = 0; rest < NB; rest++) {
    int const p = rest / S, q = rest % S;
    int const o = q + S * RADIX * p;
    if constexpr (RADIX == 4) {
      v2f const a = x[rest], b = x[rest + NB], c = x[rest + 2 * NB], d = x[rest + 3 * NB];
      v2f const apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
      v2f const sw = __builtin_shufflevector(bmd, bmd, 1, 0);
      v2f const jb = SIGN < 0 ? (v2f){sw.x, -sw.y} : (v2f){-sw.x, sw.y};
      y[o] = apc + bpd;
      y[o + S] = mul_w32<SIGN>(amc + jb, p * TSTEP);
      y[o + 2 * S] = mul_w32<SIGN>(apc - bpd, 2 * p * TSTEP);
      y[o + 3 * S] = mul_w32<SIGN>(amc - jb, 3 * p * TSTEP);
    } else {
      v2f const a = x[rest], b = x[rest + NB];
      y[o] = a + b;
      y[o + S] = mul_w32<SIGN>(a - b, p * TSTEP);
    }
  }
}
template <int N, int SIGN>
__device__ __forceinline__ void reg_fft(v2f (&v)[N]) {
  static_assert(N == 16 || N == 32, "packed register FFT sizes");
  v2f t[N];
  if constexpr (N == 16) {
    reg_pass<N, SIGN, 16, 1>(v, t);
    reg_pass<N, SIGN, 4, 4>(t, v);
  } else {
    reg_pass<N, SIGN, 32, 1>(v, t);
    reg_pass<N, SIGN, 8, 4>(t, v);
    reg_pass<N, SIGN, 2, 16>(v, t);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
  }
}
} // namespace pk

} // namespace kh
