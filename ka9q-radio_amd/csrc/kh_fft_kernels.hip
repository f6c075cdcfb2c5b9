// kh_fft_kernels.hip -- forward overlap-save transform, per-channel filter output, response design.
//
// Reference behaviour restated here (citations into the reference tree):
//   execute_filter_input   filter.c:212-238   forward DFT of N = L+M-1 points per block
//   execute_filter_output  filter.c:240-384   bin select/rotate x response, CROSS_CONJ, backward DFT of N/D points
//   set_filter/window_filter/window_rfilter/make_kaiser/i0/noise_gain  filter.c:410-702
//
// Roofline: everything here streams; HBM bound.  No MFMA: nothing is a dense contraction.
#include "kh_internal.h"
#include "kh_n0.cuh"
#include "kh_fft.cuh"
#include "kh_regfft.cuh"
#include "kh_anyfft.cuh"
#include "kh_fm.cuh"
#include "../../include/ka9q_hip.h"
#include <cstdlib>
#include <type_traits>

namespace kh {

enum { FT_NONE = 0, FT_COMPLEX = 1, FT_CROSS_CONJ = 2, FT_REAL = 3 }; // filter.h:18-23

// two LDS buffers of n points; above 64 KiB the kernel has to be told once
template <typename K>
static bool allow_big_lds(K kernel, size_t bytes) {
  if (bytes <= 65536) return true;
  return KH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KH_MAX_SMALL_N * (int)sizeof(float2)));
}

// ================================================================ forward FFT, N <= 8192: one workgroup per block
// algorithmic bytes per launch: nblocks*ninputs*(8N read + 8N write)
// n0: when n0.part is set, the epilogue also leaves compute_n0's first-pass sum for this block (radio.c:450-461,
// iteration 0: every bin outside the passband) in n0.part[(input*nblocks + block)*N0_SLOTS].
__device__ __forceinline__ bool n0_bin_counts(int k, int N, N0Fuse const &n0) { // radio.c:440-447
  int d = k - n0.start;
  if (d < 0) d += N;
  return d < n0.bincount;
}

// radio.c:451-462 on a spectrum that sits in LDS: sum and count of the counted bins below twice the first mean
// (pw0 / bincount), left as slice 0 of kh_n0_final's 64 {sum, count} pairs for this (input, block); the other slices 0.
template <int NT>
__device__ __forceinline__ void n0_second_pass(const float2 *x, int N, N0Fuse const &n0, float pw0, float *p1, float *red) {
  float const th = (pw0 / n0.bincount) * 2;
  float s = 0.f, c = 0.f;
  for (int k = threadIdx.x; k < N; k += NT) {
    float2 const v = x[k];
    float const p = v.x * v.x + v.y * v.y;
    if (n0_bin_counts(k, N, n0) && p < th) {
      s += p;
      c += 1.f;
    }
  }
  s = block_sum<NT>(s, red);
  c = block_sum<NT>(c, red);
  for (int k = threadIdx.x; k < 2 * 64; k += NT) p1[k] = k == 0 ? s : k == 1 ? c : 0.f;
}

template <int R, int NT>
__global__ __launch_bounds__(NT) void kh_fft_fwd_lds(const void *__restrict__ time, size_t in_stride, int real_input,
                                                     float2 *__restrict__ spec, size_t spec_in_stride, int L,
                                                     const float2 *__restrict__ tw, N0Fuse n0) {
  __shared__ __attribute__((aligned(16))) float2 buf[R];
  __shared__ float red[NT / 64 > 0 ? NT / 64 : 1];
  int const tid = threadIdx.x;
  int const b = blockIdx.x, i = blockIdx.y;
  if (real_input) {
    const float2 *src = reinterpret_cast<const float2 *>(static_cast<const float *>(time) + i * in_stride + (size_t)b * L);
    for (int k = tid; k < R / 2; k += NT) { // two reals per lane
      float2 v = src[k];
      buf[2 * k] = make_float2(v.x, 0.f);
      buf[2 * k + 1] = make_float2(v.y, 0.f);
    }
  } else {
    const float4 *src = reinterpret_cast<const float4 *>(static_cast<const float2 *>(time) + i * in_stride + (size_t)b * L);
    for (int k = tid; k < R / 2; k += NT) reinterpret_cast<float4 *>(buf)[k] = src[k];
  }
  __syncthreads();
  lds_fft<R, 1, NT, 1, 1, false, -1>(buf, tw, tid);
  float4 *dst = reinterpret_cast<float4 *>(spec + i * spec_in_stride + (size_t)b * R);
  float pw = 0.f;
  for (int k = tid; k < R / 2; k += NT) {
    float4 const v = reinterpret_cast<float4 *>(buf)[k];
    dst[k] = v;
    if (n0.part != nullptr) {
      if (n0_bin_counts(2 * k, R, n0)) pw += v.x * v.x + v.y * v.y;
      if (n0_bin_counts(2 * k + 1, R, n0)) pw += v.z * v.z + v.w * v.w;
    }
  }
  if (n0.part != nullptr) { // uniform
    pw = block_sum<NT>(pw, red);
    size_t const pt = (size_t)i * gridDim.x + b;
    if (tid == 0) n0.part[pt * N0_SLOTS] = pw;
    if (n0.part1 != nullptr) n0_second_pass<NT>(buf, R, n0, pw, n0.part1 + pt * (2 * 64), red);
  }
}


// forward FFT of any 2,3,5,7-smooth N <= 8192 (the reference's nextfastfft sizes): one workgroup per block, two LDS buffers
__global__ __launch_bounds__(256) void kh_fft_fwd_any(const void *__restrict__ time, size_t in_stride, int real_input,
                                                      float2 *__restrict__ spec, size_t spec_in_stride, int L,
                                                      const float2 *__restrict__ tw, N0Fuse n0, AnyPlan plan) {
  extern __shared__ __attribute__((aligned(16))) float2 dynf[];
  __shared__ float red[4];
  int const tid = threadIdx.x, R = plan.n;
  int const b = blockIdx.x, i = blockIdx.y;
  if (real_input) {
    const float *src = static_cast<const float *>(time) + i * in_stride + (size_t)b * L;
    for (int k = tid; k < R; k += 256) dynf[k] = make_float2(src[k], 0.f);
  } else {
    const float2 *src = static_cast<const float2 *>(time) + i * in_stride + (size_t)b * L;
    for (int k = tid; k < R; k += 256) dynf[k] = src[k];
  }
  __syncthreads();
  const float2 *res = any_fft<-1>(dynf, dynf + R, plan, tw, tid, 256);
  float2 *dst = spec + i * spec_in_stride + (size_t)b * R;
  float pw = 0.f;
  for (int k = tid; k < R; k += 256) {
    float2 const v = res[k];
    dst[k] = v;
    if (n0.part != nullptr && n0_bin_counts(k, R, n0)) pw += v.x * v.x + v.y * v.y;
  }
  if (n0.part != nullptr) { // uniform
    pw = block_sum<256>(pw, red);
    size_t const pt = (size_t)i * gridDim.x + b;
    if (tid == 0) n0.part[pt * N0_SLOTS] = pw;
    if (n0.part1 != nullptr) n0_second_pass<256>(res, R, n0, pw, n0.part1 + pt * (2 * 64), red);
  }
}

// ================================================================ forward FFT, N = N1*N2 > 8192: four-step, two launches
// X[k1 + N1*k2] = sum_{n2} W_N^{n2 k1} [ sum_{n1} x[n1*N2 + n2] W_N1^{n1 k1} ] W_N2^{n2 k2}
// kh_fft_cols: 16 adjacent columns n2 per workgroup (128-byte row segments), N1-point transforms down the columns,
//         four-step twiddle, result A[k1][n2] written in the input's own layout.
// kh_fft_rows: 16 adjacent rows k1 per workgroup, N2-point transforms along the rows, transposed store (128-byte
//         segments of X), optional compute_n0 first-pass partial sum.
// Both keep the tile in REGISTERS: an R = RA*RB point sequence is RB interleaved RA-point transforms (one per lane,
// kh_regfft.cuh), a twiddle, ONE exchange through LDS (real parts, then imaginary parts, so the buffer is 4 bytes
// per point: R = 512 needs 34 KiB and four workgroups share a CU), and RA interleaved RB-point transforms:
//   X[c + RA*d] = sum_b W_RB^{b d} ( W_R^{b c} sum_a x[RB*a + b] W_RA^{a c} ).
// algorithmic bytes per launch, each pass: nblocks*ninputs*(8N read + 8N write)
// v[c] *= W_R^{b c}, c < RA, with W_R^{b(4q+j)} = W_R^{4bq} W_R^{bj}: RA/4 + 2 table reads per lane instead of RA - 1
template <int RA>
__device__ __forceinline__ void inner_twiddle(float2 (&v)[RA], const float2 *__restrict__ tw, int b) {
  float2 const u1 = tw[b], u2 = tw[2 * b], u3 = tw[3 * b];
  v[1] = cmul(v[1], u1);
  v[2] = cmul(v[2], u2);
  v[3] = cmul(v[3], u3);
#pragma unroll
  for (int q = 1; q < RA / 4; q++) {
    float2 const t = tw[4 * q * b];
    v[4 * q] = cmul(v[4 * q], t);
    v[4 * q + 1] = cmul(v[4 * q + 1], cmul(t, u1));
    v[4 * q + 2] = cmul(v[4 * q + 2], cmul(t, u2));
    v[4 * q + 3] = cmul(v[4 * q + 3], cmul(t, u3));
  }
}
__device__ __forceinline__ float2 big_tw(BigTwiddle const &bt, unsigned idx) { // W_N^idx, idx < N
  return cmul(bt.coarse[idx >> 10], bt.fine[idx & 1023]);
}


// ---- fused ingest (radio.c:136-178,188,197 + filter.c:230): the column pass can take its input straight from the
// raw ring: wire format -> float x gain (no NCO: a launch with an active oscillator goes through kh_ingest), the IF
// power partial sums of the new samples, and the overlap-save history of the next launch.
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }
template <int PT>
struct RawFmt;
template <>
struct RawFmt<KH_PT_CF32> {
  static constexpr int BPS = 8;
  static __device__ __forceinline__ kh_u32x2 fetch(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  }
  static __device__ __forceinline__ float2 decode(kh_u32x2 w, float g) { return make_float2(g * u2f(w.x), g * u2f(w.y)); }
};
template <>
struct RawFmt<KH_PT_IQ16LE> { // radio.c:159-167
  static constexpr int BPS = 4;
  static __device__ __forceinline__ kh_u32x2 fetch(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    kh_u32x2 w;
    w.x = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    w.y = 0;
    return w;
  }
  static __device__ __forceinline__ float2 decode(kh_u32x2 w, float g) {
    float const gain = (1.f / 32767.f) * g;
    return make_float2(gain * (short)(w.x & 0xffff), gain * (short)(w.x >> 16));
  }
};
template <>
struct RawFmt<KH_PT_PCM16BE> { // radio.c:148-157
  static constexpr int BPS = 4;
  static __device__ __forceinline__ kh_u32x2 fetch(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return RawFmt<KH_PT_IQ16LE>::fetch(r, voff, soff);
  }
  static __device__ __forceinline__ float2 decode(kh_u32x2 w, float g) {
    float const gain = (1.f / 32767.f) * g;
    unsigned const b0 = w.x & 0xff, b1 = (w.x >> 8) & 0xff, b2 = (w.x >> 16) & 0xff, b3 = w.x >> 24;
    return make_float2(gain * (short)((b0 << 8) | b1), gain * (short)((b2 << 8) | b3));
  }
};
template <>
struct RawFmt<KH_PT_IQ8> { // radio.c:169-176
  static constexpr int BPS = 2;
  static __device__ __forceinline__ kh_u32x2 fetch(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    kh_u32x2 w;
    w.x = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    w.y = 0;
    return w;
  }
  static __device__ __forceinline__ float2 decode(kh_u32x2 w, float g) {
    float const gain = (1.f / 127.f) * g;
    return make_float2(gain * (signed char)(w.x & 0xff), gain * (signed char)((w.x >> 8) & 0xff));
  }
};
template <>
struct RawFmt<KH_PT_IQ12> { // radio.c:138-146
  static constexpr int BPS = 3;
  static __device__ __forceinline__ kh_u32x2 fetch(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    kh_u32x2 w;
    unsigned const d0 = __builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0);
    unsigned const d1 = __builtin_amdgcn_raw_buffer_load_b8(r, voff + 1, soff, 0);
    unsigned const d2 = __builtin_amdgcn_raw_buffer_load_b8(r, voff + 2, soff, 0);
    w.x = d0 | (d1 << 8) | (d2 << 16);
    w.y = 0;
    return w;
  }
  static __device__ __forceinline__ float2 decode(kh_u32x2 w, float g) {
    float const gain = (1.f / 32767.f) * g;
    int const d0 = w.x & 0xff, d1 = (w.x >> 8) & 0xff, d2 = (w.x >> 16) & 0xff;
    return make_float2(gain * (short)(((d0 << 8) | d1) & 0xfff0), gain * (short)(((d1 << 8) | d2) << 4));
  }
};

struct ColsIngest {
  const unsigned char *raw;   // raw ring, input i at raw + i*raw_stride
  size_t raw_stride;
  unsigned raw_bytes;         // bytes of one input's ring
  int payload;                // enum kh_payload
  const float2 *hist_in;      // M-1 mixed samples that precede this launch, input i at hist_in + i*hist_stride
  float2 *hist_out;           // the last M-1 samples of this launch go here (filter.c:230)
  size_t hist_stride;
  const IngestDesc *desc;     // gain_factor (the oscillators are inactive on this path)
  float *energy_part;         // IF power partial sums: [(i*nblocks + b) * ntiles + tile]
  int M1;                     // M-1
};

// element (row0 + k*rstep)*N2 + col of block b's window, k < CNT, as samples: window element n is launch-relative
// sample s = b*L - M1 + n; s < 0 lies in the history.
template <int PT, int CNT>
__device__ __forceinline__ void ingest_tile(float2 (&v)[CNT], ColsIngest const &ing, int i, int s_lane, int rstep_elems,
                                            float gain) {
  using F = RawFmt<PT>;
  __amdgpu_buffer_rsrc_t const rr = make_rsrc(ing.raw + (size_t)i * ing.raw_stride, ing.raw_bytes);
  // New samples: one running per-lane byte offset (wave-uniform row offsets would cost CNT scalar registers).  A
  // sample before the launch (s < 0) makes the offset negative, i.e. huge as the unsigned number the buffer unit
  // compares with the ring's size: the load returns zero without touching memory.
  int const step = rstep_elems * F::BPS;
  int voff = s_lane * F::BPS;
#pragma unroll
  for (int k = 0; k < CNT; k++) {
    v[k] = F::decode(F::fetch(rr, voff, 0), gain);
    voff += step;
  }
  // The window reaches back into the history only in the first block(s) of a launch: those lanes then replace their
  // zeros by the history samples.  Buffer loads again (one running offset; 64-bit addresses would keep a register pair
  // per outstanding load alive): sample s < 0 is element M1 + s of the history, and s >= 0 falls beyond its M1
  // elements, so the load returns zero and the select keeps the new sample.
  if (__builtin_amdgcn_ballot_w64(s_lane < 0) != 0) { // wave-uniform
    __amdgpu_buffer_rsrc_t const rh = make_rsrc(ing.hist_in + (size_t)i * ing.hist_stride, (unsigned)ing.M1 * 8u);
    int hoff = (ing.M1 + s_lane) * 8;
#pragma unroll
    for (int k = 0; k < CNT; k++) {
      float2 const h = buf_load2(rh, hoff, 0);
      if (s_lane + k * rstep_elems < 0) v[k] = h;
      hoff += rstep_elems * 8;
    }
  }
}

constexpr int TILE_W = 16;

template <int RA, int RB, int NT, bool INGEST>
__global__ __launch_bounds__(NT, 4) void kh_fft_cols(const float2 *__restrict__ time, size_t in_stride,
                                                  float2 *__restrict__ tmp, size_t spec_in_stride, int N2, int L,
                                                  const float2 *__restrict__ tw, ColsTwiddle ct, ColsIngest ing, int b0,
                                                  int nblocks, unsigned *__restrict__ n0sync, float *__restrict__ ewin) {
  constexpr int W = TILE_W, R = RA * RB;
  constexpr int I1 = W * RB / NT; // RA-point transforms per lane
  constexpr int I2 = W * RA / NT; // RB-point transforms per lane
  static_assert(I1 * NT == W * RB && I2 * NT == W * RA, "tile must divide evenly over the workgroup");
  constexpr int CS = RB * W + 16; // pitch of c: an odd multiple of 16 floats, so two c's in a 32-lane group differ in bank
  __shared__ float ex[RA * CS];
  __shared__ float red[2 * (NT / 64)];
  int const tid = threadIdx.x;
  // this launch handles blocks b0 .. b0 + gridDim.y - 1 of the nblocks of the call (kh_bank_execute); its scratch slot
  // in tmp is the block's index within the launch: the groups of a call reuse the same scratch
  int const tile = blockIdx.x, bl = blockIdx.y, b = b0 + bl, i = blockIdx.z;
  unsigned const tile_bytes = (unsigned)R * (unsigned)N2 * 8u; // the block's N points (8 MiB at most)
  int const row_bytes = N2 * 8;
  // compute_n0, direct mode (N0Fuse): the row pass of this block counts its tiles in and may raise a flag; both words
  // start at zero (plain stores: the row pass is a later kernel on the same stream)
  if (n0sync != nullptr && tile == 0 && tid == 0)
    *reinterpret_cast<uint2 *>(n0sync + N0_SYNC_STRIDE * ((size_t)i * nblocks + b)) = make_uint2(0u, 0u);
  float2 v[I1][RA];
  if constexpr (!INGEST) {
    __amdgpu_buffer_rsrc_t const rin = make_rsrc(time + i * in_stride + (size_t)b * L + tile * W, tile_bytes);
    int const voff = (tid / W) * row_bytes + (tid % W) * 8; // item u: NT/W rows further down
#pragma unroll
    for (int u = 0; u < I1; u++)
#pragma unroll
      for (int a = 0; a < RA; a++) v[u][a] = buf_load2(rin, voff, (RB * a + u * (NT / W)) * row_bytes);
  } else {
    // window element n = row*N2 + col is launch-relative sample b*L - M1 + n; item (u, a) is row (tid/W) + u*(NT/W) + RB*a
    float const gain = ing.desc[i].gain_factor;
    int const n_lane = (tid / W) * N2 + tile * W + (tid % W);
    int const s_lane = b * L - ing.M1 + n_lane;
#pragma unroll
    for (int u = 0; u < I1; u++) {
      int const su = s_lane + u * (NT / W) * N2;
      switch (ing.payload) {
      case KH_PT_IQ16LE: ingest_tile<KH_PT_IQ16LE, RA>(v[u], ing, i, su, RB * N2, gain); break;
      case KH_PT_PCM16BE: ingest_tile<KH_PT_PCM16BE, RA>(v[u], ing, i, su, RB * N2, gain); break;
      case KH_PT_IQ8: ingest_tile<KH_PT_IQ8, RA>(v[u], ing, i, su, RB * N2, gain); break;
      case KH_PT_IQ12: ingest_tile<KH_PT_IQ12, RA>(v[u], ing, i, su, RB * N2, gain); break;
      default: ingest_tile<KH_PT_CF32, RA>(v[u], ing, i, su, RB * N2, gain); break;
      }
    }
    // IF power of the block's new samples (radio.c:188,197: n >= M-1) and the next launch's history (filter.c:230:
    // the last M-1 points of the launch = window elements n >= L of its last block)
    float e = 0.f, eh = 0.f; // new samples; the history part of the window (compute_n0's bracket wants both, kh_n0.cuh)
#pragma unroll
    for (int u = 0; u < I1; u++)
#pragma unroll
      for (int a = 0; a < RA; a++) {
        int const n = n_lane + (u * (NT / W) + RB * a) * N2;
        float2 const x = v[u][a];
        float const p = x.x * x.x + x.y * x.y;
        if (n >= ing.M1)
          e += p;
        else
          eh += p;
      }
    if (b == nblocks - 1) { // uniform: window elements n >= L of the call's last block are the next history
      // element n - L of hist_out; n < L makes the offset negative, i.e. out of range: the buffer unit drops the store
      __amdgpu_buffer_rsrc_t const rho = make_rsrc(ing.hist_out + (size_t)i * ing.hist_stride, (unsigned)ing.M1 * 8u);
#pragma unroll
      for (int u = 0; u < I1; u++) {
        int hoff = (n_lane + u * (NT / W) * N2 - L) * 8;
#pragma unroll
        for (int a = 0; a < RA; a++) {
          buf_store2(v[u][a], rho, hoff, 0);
          hoff += RB * N2 * 8;
        }
      }
    }
    { // both sums behind one pair of barriers
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o, 64);
        eh += __shfl_xor(eh, o, 64);
      }
      if ((tid & 63) == 0) red[tid >> 6] = e, red[NT / 64 + (tid >> 6)] = eh;
      __syncthreads();
      if (tid == 0) {
        float te = 0.f, th = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; wv++) te += red[wv], th += red[NT / 64 + wv];
        ing.energy_part[((size_t)i * nblocks + b) * gridDim.x + tile] = te;
        // compute_n0, bracket mode (kh_n0.cuh): the energy of the block's N window samples
        if (ewin != nullptr) ewin[N0_SYNC_STRIDE * ((size_t)i * nblocks + b) + tile] = te + th;
      }
      __syncthreads();
    }
  }
  if (!INGEST && ewin != nullptr) { // uniform
    float e = 0.f;
#pragma unroll
    for (int u = 0; u < I1; u++)
#pragma unroll
      for (int a = 0; a < RA; a++) e += v[u][a].x * v[u][a].x + v[u][a].y * v[u][a].y;
    e = block_sum<NT>(e, red);
    if (tid == 0) ewin[N0_SYNC_STRIDE * ((size_t)i * nblocks + b) + tile] = e;
  }
#pragma unroll
  for (int u = 0; u < I1; u++) {
    int const bb = (tid + u * NT) / W;
    reg_fft<RA, -1>(v[u]);
    inner_twiddle<RA>(v[u], tw, bb);
  }
  // exchange: real parts, then imaginary parts, through the same 4-byte-per-point buffer
  float zr[I2][RB], zi[I2][RB];
#pragma unroll
  for (int u = 0; u < I1; u++) {
    int const item = tid + u * NT, w = item % W, bb = item / W;
#pragma unroll
    for (int c = 0; c < RA; c++) ex[c * CS + bb * W + w] = v[u][c].x;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < I2; u++) {
    int const item = tid + u * NT, w = item % W, c = item / W;
#pragma unroll
    for (int bb = 0; bb < RB; bb++) zr[u][bb] = ex[c * CS + bb * W + w];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < I1; u++) {
    int const item = tid + u * NT, w = item % W, bb = item / W;
#pragma unroll
    for (int c = 0; c < RA; c++) ex[c * CS + bb * W + w] = v[u][c].y;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < I2; u++) {
    int const item = tid + u * NT, w = item % W, c = item / W;
#pragma unroll
    for (int bb = 0; bb < RB; bb++) zi[u][bb] = ex[c * CS + bb * W + w];
  }
  size_t const N = (size_t)R * N2;
  __amdgpu_buffer_rsrc_t const rout = make_rsrc(tmp + i * spec_in_stride + (size_t)bl * N + tile * W, tile_bytes);
#pragma unroll
  for (int u = 0; u < I2; u++) {
    int const item = tid + u * NT, w = item % W, c = item / W;
    float2 z[RB];
#pragma unroll
    for (int bb = 0; bb < RB; bb++) z[bb] = make_float2(zr[u][bb], zi[u][bb]);
    reg_fft<RB, -1>(z);
    // four-step twiddle W_N^{n2 k1}, k1 = c + RA*(8q + j):  W_N^{n2 (c + 8 RA q)} W_N^{n2 RA j}, both factors from tables
    // laid out for this kernel (16-byte loads; the generic coarse x fine product cost 22 8-byte loads and 11 products
    // per lane, 6 us of the pass at N = 2^18).  Issuing the reads ahead of the transform, or three waves per SIMD
    // without the spills, measured the same.
    // (scheduling fences: the compiler otherwise hoists all of the table reads above the transform and spills more)
    __builtin_amdgcn_sched_barrier(0);
    int const n2 = tile * W + w;
    float2 uj[8], tq[RB / 8];
    {
      const float4 *up = reinterpret_cast<const float4 *>(ct.uj) + (size_t)n2 * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float4 const t = up[j];
        uj[2 * j] = make_float2(t.x, t.y);
        uj[2 * j + 1] = make_float2(t.z, t.w);
      }
      const float4 *tp = reinterpret_cast<const float4 *>(ct.tq) + ((size_t)n2 * RA + c) * (RB / 16);
#pragma unroll
      for (int q = 0; q < RB / 16; q++) {
        float4 const t = tp[q];
        tq[2 * q] = make_float2(t.x, t.y);
        tq[2 * q + 1] = make_float2(t.z, t.w);
      }
    }
#pragma unroll
    for (int q = 0; q < RB / 8; q++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int const d = 8 * q + j;
        float2 const wf = j ? cmul(tq[q], uj[j]) : tq[q];
        buf_store2(cmul(z[d], wf), rout, (c * N2 + w) * 8, RA * d * row_bytes);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---- a bank that keeps only its channels' rows of the spectrum (SpecMask): compute_n0's second iteration must never need
// the spectrum again.  Every row-pass workgroup counts itself in per block (n0.arrive) once its first-iteration sum is
// visible; the one that arrives last -- nobody waits for anybody -- evaluates the bracket's condition exactly as the second
// step will (kh_n0.cuh, n0_bracket_finish: same operands, same operations).  If it fails (more than a quarter of the power
// inside the excluded range, or a wavefront's candidates beyond their slots) that workgroup takes the block's second
// iteration itself: the row transforms once more from the scratch it has just read (the next column pass has not
// started), the exact threshold applied as the outputs appear, the {sum, count} pairs where n0_pass1_reread would leave
// them.  One workgroup for a whole block: right, slow and rare; the host sees the count of such blocks and keeps whole
// spectra for a while instead (kh_bank.cpp).
// Eight rows at a time through the LDS transform of kh_fft.cuh, 32 registers: the register transform of the pass itself,
// inlined a second time, made the compiler spill 40-120 registers of the pass (as a called function: 900 bytes of stack per
// lane); as a kernel of its own it cannot be launched on a condition only the device knows, and launched regardless it
// waits for workgroup slots beside the next launch's column pass (measured: 5-10 us per step, on either stream).
template <int R, int NT>
__device__ __forceinline__ void n0_repair_block(const float2 *__restrict__ blk /*the block's N points of scratch*/, int N1, int ntiles,
                                                const float2 *__restrict__ tw, float2 *buf /*8 R points of LDS*/, float *red,
                                                N0Fuse const &n0, float th, float *p1) {
  constexpr int W = TILE_W, G = 8; // rows per group
  static_assert((G * R) % (2 * NT) == 0, "a group of rows divides evenly over the workgroup");
  int const tid = threadIdx.x;
  int const N = R * N1;
  for (int tile = 0; tile < ntiles; tile++) {
    float s = 0.f, cnt = 0.f;
    for (int g = 0; g < W / G; g++) {
      int const row0 = tile * W + g * G; // k1 of the group's first row
      const float4 *src = reinterpret_cast<const float4 *>(blk + (size_t)row0 * R);
      for (int k = tid; k < G * R / 2; k += NT) reinterpret_cast<float4 *>(buf)[k] = src[k];
      __syncthreads();
      lds_fft<R, G, NT, 1, R, false, -1>(buf, tw, tid); // row r of the group at buf + r*R, natural order on return
      for (int k = tid; k < G * R; k += NT) {
        float const p = n0_power(buf[k]);
        int const bin = row0 + k / R + N1 * (k % R); // X[k1 + N1*k2]
        if (n0_bin_counts(bin, N, n0) && p < th) { // radio.c:457-460
          s += p;
          cnt += 1.f;
        }
      }
      __syncthreads();
    }
    s = block_sum<NT>(s, red);
    cnt = block_sum<NT>(cnt, red);
    if (tid == 0) p1[2 * tile] = s, p1[2 * tile + 1] = cnt;
  }
  if (tid >= ntiles && tid < N0_GX) reinterpret_cast<float2 *>(p1)[tid] = make_float2(0.f, 0.f); // kh_n0_final adds all N0_GX pairs
}

// SPEC_AUX: cache policy of the spectrum stores.  A launch whose spectrum alone is beyond ~100 MB (N >= 2^19 at the bench
// shapes) stores it non-temporal (2): the spectrum is read once, much later, by the channel kernels, while the raw ring and
// the intermediate of the transform are re-read at once and fit the 256 MiB Infinity Cache only if the spectrum does not
// push them out.  Measured: config 3's share 176 -> 157 us/step, config 4's 164 -> 152; nothing at configs[1], where
// everything fits (nt on the raw loads, or sc0 / sc1 on these stores, measured worse).
// MODE: compute_n0's second iteration: 0 not here, 1 direct (rendezvous of the block's tiles), 2 bracket (kh_n0.cuh)
// MASKED: only the rows of X that SpecMask names are stored (kh_internal.h).  A wavefront's rows k2 = c + RA*d, c = tid/16,
// are 4*wave .. 4*wave + 3 for every d, so whether store d goes out is a scalar bit: no register, no per-lane predicate.
template <int RA, int RB, int NT, int MODE, int SPEC_AUX, bool MASKED = false>
__global__ __launch_bounds__(NT, 4) void kh_fft_rows(const float2 *__restrict__ tmp, float2 *__restrict__ spec,
                                                  size_t spec_in_stride, int N1, const float2 *__restrict__ tw,
                                                  N0Fuse n0, int b0, int nblocks, SpecMask sm) {
  constexpr int W = TILE_W, R = RA * RB;
  constexpr int I1 = W * RB / NT, I2 = W * RA / NT;
  static_assert(I1 * NT == W * RB && I2 * NT == W * RA, "tile must divide evenly over the workgroup");
  // (bracket mode only: in the plain store loop of MODE 0 every form of the conditional store spilled 11-19 registers
  // around the last transform, and a spill there waits for every store in flight)
  static_assert(!MASKED || (I2 == 1 && MODE == 2 && RB <= 32 && NT / 64 <= 8), "the mask is one bit per store of a wavefront");
  unsigned const need = MASKED ? sm.need[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)] : ~0u;
  constexpr int RP = RB + 1;  // pitch of a row: odd, so the 16 rows a 32-lane group reads differ in bank
  constexpr int CS = W * RP;  // pitch of c (an odd multiple of 16 floats)
  __shared__ __attribute__((aligned(16))) float ex[RA * CS];
  __shared__ float red[NT / 64];
  __shared__ float th_sh;
  __shared__ int ok_sh, arrived_sh;
  __shared__ int last_sh; // MASKED: this workgroup is the last of its block to arrive
  constexpr bool DIRECT = MODE == 1;
  if (DIRECT && threadIdx.x == 0) arrived_sh = 0; // (behind the exchange's barriers when it is used)
  int const tid = threadIdx.x;
  int const tile = blockIdx.x, bl = blockIdx.y, b = b0 + bl, i = blockIdx.z; // (b0, nblocks: as in kh_fft_cols)
  size_t const N = (size_t)R * N1;
  const float2 *src = tmp + i * spec_in_stride + (size_t)bl * N + (size_t)tile * W * R;
  if constexpr (DIRECT) { // (experiment, KA9Q_HIP_N0_STAGGER: see the rendezvous below)
    for (int k = (bl & 7) * n0.stagger; k > 0; k--) __builtin_amdgcn_s_sleep(8);
  }
  if constexpr (MODE == 2) { // the bracket's centre (read after the exchange's barriers)
    if (tid < 64) {
      float const t = n0_parseval_th(n0.ewin + N0_SYNC_STRIDE * ((size_t)i * nblocks + b), R / W, (int)N, n0.bincount);
      if (tid == 0) th_sh = t;
    }
  }
  float2 v[I1][RA];
#pragma unroll
  for (int u = 0; u < I1; u++) { // lanes run along the row: RB consecutive points, then the next row
    int const item = tid + u * NT, bb = item % RB, w = item / RB;
#pragma unroll
    for (int a = 0; a < RA; a++) v[u][a] = src[(size_t)w * R + RB * a + bb];
  }
#pragma unroll
  for (int u = 0; u < I1; u++) {
    int const bb = (tid + u * NT) % RB;
    reg_fft<RA, -1>(v[u]);
    inner_twiddle<RA>(v[u], tw, bb);
  }
  float zr[I2][RB], zi[I2][RB];
#pragma unroll
  for (int u = 0; u < I1; u++) {
    int const item = tid + u * NT, bb = item % RB, w = item / RB;
#pragma unroll
    for (int c = 0; c < RA; c++) ex[c * CS + w * RP + bb] = v[u][c].x;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < I2; u++) { // lanes run across the 16 rows: the transposed store below is 128 bytes wide
    int const item = tid + u * NT, w = item % W, c = item / W;
#pragma unroll
    for (int bb = 0; bb < RB; bb++) zr[u][bb] = ex[c * CS + w * RP + bb];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < I1; u++) {
    int const item = tid + u * NT, bb = item % RB, w = item / RB;
#pragma unroll
    for (int c = 0; c < RA; c++) ex[c * CS + w * RP + bb] = v[u][c].y;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < I2; u++) {
    int const item = tid + u * NT, w = item % W, c = item / W;
#pragma unroll
    for (int bb = 0; bb < RB; bb++) zi[u][bb] = ex[c * CS + w * RP + bb];
  }
  __amdgpu_buffer_rsrc_t const rout = make_rsrc(spec + i * spec_in_stride + (size_t)b * N + tile * W, (unsigned)N * 8u);
  if constexpr (MODE == 0) {
    float pw = 0.f;
#pragma unroll
    for (int u = 0; u < I2; u++) {
      int const item = tid + u * NT, w = item % W, c = item / W;
      float2 z[RB];
#pragma unroll
      for (int bb = 0; bb < RB; bb++) z[bb] = make_float2(zr[u][bb], zi[u][bb]);
      reg_fft<RB, -1>(z);
#pragma unroll
      for (int d = 0; d < RB; d++) {
        int const k2 = c + RA * d;
        float2 const x = z[d];
        buf_store2<SPEC_AUX>(x, rout, (c * N1 + w) * 8, RA * d * N1 * 8); // X[k1 + N1*k2], k1 = tile*W + w
        if (n0.part != nullptr && n0_bin_counts(tile * W + w + N1 * k2, (int)N, n0)) pw += x.x * x.x + x.y * x.y;
      }
    }
    if (n0.part != nullptr) { // uniform: radio.c:451-461, first iteration (every counted bin)
      pw = block_sum<NT>(pw, red);
      if (tid == 0) n0.part[((size_t)i * nblocks + b) * N0_SLOTS + tile] = pw;
    }
  } else if constexpr (MODE == 2) {
    // ---- compute_n0, bracket mode (kh_n0.cuh): first-iteration sum per tile as above; below the bracket: sum and count
    // per wavefront; inside it: the values themselves, per wavefront in lane order.  Nothing here waits for another tile.
    static_assert(I2 == 1, "bracket mode keeps one row of outputs per lane");
    int const w = tid % W, c = tid / W;
    float2 z[RB];
#pragma unroll
    for (int bb = 0; bb < RB; bb++) z[bb] = make_float2(zr[0][bb], zi[0][bb]);
    reg_fft<RB, -1>(z);
    float const thP = th_sh, lo = thP * N0_BR_LO, hi = thP * N0_BR_HI;
    float pw = 0.f, slo = 0.f, clo = 0.f;
    int ncand = 0;
    unsigned inside = 0; // bit d: output d is a candidate
    static_assert(RB <= 32, "one bit per output");
    auto classify = [&](auto narrow) {
#pragma unroll
      for (int d = 0; d < RB; d++) {
        float const p = n0_power(z[d]);
        // (narrow: the excluded range around DC lies in the first and the last row of the N1 x N2 spectrum, k2 = 0 or N2 - 1)
        bool const counts = (!decltype(narrow)::value || d == 0 || d == RB - 1) ? n0_bin_counts(tile * W + w + N1 * (c + RA * d), (int)N, n0) : true;
        bool const below = counts && p < lo, cand = counts && !below && p < hi;
        pw += counts ? p : 0.f;
        slo += below ? p : 0.f;
        clo += below ? 1.f : 0.f;
        inside |= cand ? 1u << d : 0u;
      }
    };
    if (n0.narrow) // uniform
      classify(std::true_type{});
    else
      classify(std::false_type{});
    ncand = __popc(inside);
    // where this lane's candidates go: after those of the lanes before it
    int off = ncand;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int const t = __shfl_up(off, o, 64);
      if ((tid & 63) >= o) off += t;
    }
    int const total = __shfl(off, 63, 64);
    off -= ncand;
    size_t const pt = (size_t)i * nblocks + b;
    // candidates: staged in LDS, then written out 256 bytes at a time -- as 32 stores with two or three lanes each they cost
    // more than the spectrum's own stores (N = 2^20: row pass 62 -> 79 us).  The staging area is the part of the exchange
    // buffer that only this wavefront read in the last phase of the exchange (rows c = tid/16 = 4 wave .. 4 wave + 3): its
    // own reads are behind it (a wavefront's LDS operations complete in order), nobody else touches it: no barrier.
    constexpr int NWV = NT / 64;
    static_assert(W == 16 && 4 * CS >= N0_CAPW, "a wavefront's four rows of the exchange buffer hold its candidates");
    int const K = gridDim.x * NWV, idx = tile * NWV + (tid >> 6);
    float *hdr = n0.cand + pt * ((size_t)K * (N0_CAPW + 3)); // [3][K]: candidates found, sum and count below the bracket
    float *region = hdr + 3 * (size_t)K + (size_t)idx * N0_CAPW;
    float *stage = ex + (tid >> 6) * 4 * CS;
    if constexpr (MASKED) {
      // arrival (see n0_repair_block), ahead of the stores so that lane 0's round trips overlap with the other wavefronts'
      // stores: a wavefront whose candidates outgrow their slots says so first, then the tile's first-iteration sum goes
      // out with agent scope and, once it has landed, the block's counter goes up
      unsigned *arrive = n0.arrive + N0_SYNC_STRIDE * pt;
      if ((tid & 63) == 0 && total > N0_CAPW) {
        (void)__hip_atomic_fetch_add(arrive + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      pw = block_sum<NT>(pw, red);
      if (tid == 0) {
        __hip_atomic_store(n0.part + pt * N0_SLOTS + tile, pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last_sh = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      }
    }
#pragma unroll
    for (int d = 0; d < RB; d++)
      if (!MASKED || (need >> d & 1u)) buf_store2<SPEC_AUX>(z[d], rout, (c * N1 + w) * 8, RA * d * N1 * 8); // X[k1 + N1*k2], k2 = c + RA*d
#pragma unroll
    for (int d = 0; d < RB; d++)
      if (inside >> d & 1u) {
        if (off < N0_CAPW) stage[off] = n0_power(z[d]);
        off++;
      }
    int const nout = total < N0_CAPW ? total : N0_CAPW;
    for (int k = tid & 63; k < nout; k += 64) region[k] = stage[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      slo += __shfl_xor(slo, o, 64);
      clo += __shfl_xor(clo, o, 64);
    }
    if ((tid & 63) == 0) {
      hdr[idx] = (float)total; // (above N0_CAPW: the block's second iteration re-reads the spectrum)
      hdr[K + idx] = slo;
      hdr[2 * K + idx] = clo;
    }
    if constexpr (!MASKED) {
      pw = block_sum<NT>(pw, red);
      if (tid == 0) n0.part[pt * N0_SLOTS + tile] = pw;
    } else {
      __syncthreads();
      if (!last_sh) return; // uniform
      // every tile of the block has arrived: the exact threshold and the bracket's condition, as n0_bracket_finish has them
      int const ntiles = gridDim.x;
      if (tid < 64) {
        float t = tid < ntiles ? __hip_atomic_load(n0.part + pt * N0_SLOTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        t = n0_wave_sum(t);
        if (tid == 0) {
          float const th = (t / n0.bincount) * 2; // radio.c:456,461
          bool const bad = !(th >= thP * N0_BR_LO && th < thP * N0_BR_HI) ||
                           __hip_atomic_load(n0.arrive + N0_SYNC_STRIDE * pt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
          th_sh = th;
          ok_sh = !bad;
          if (bad && n0.host_fail != nullptr) (void)__hip_atomic_fetch_add(n0.host_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      __syncthreads();
      if (ok_sh) return; // uniform: the second step (next kernel on the other stream) finds the same and takes the candidates
      static_assert(sizeof(ex) >= 8 * R * sizeof(float2), "eight rows fit the exchange buffer");
      n0_repair_block<R, NT>(tmp + i * spec_in_stride + (size_t)bl * N, N1, ntiles, tw, reinterpret_cast<float2 *>(ex), red, n0, th_sh,
                             n0.part1 + pt * (2 * N0_GX));
    }
  } else {
  // ---- compute_n0, direct mode (N0Fuse)
  static_assert(I2 == 1, "direct mode keeps one row of outputs per lane");
  int const w = tid % W, c = tid / W;
  float2 z[RB];
#pragma unroll
  for (int bb = 0; bb < RB; bb++) z[bb] = make_float2(zr[0][bb], zi[0][bb]);
  reg_fft<RB, -1>(z);
  // The power of every counted output (radio.c:440-447) stays behind in registers, and the tile's first-iteration sum
  // is published BEFORE the stores go out, so that it travels while they drain -- without a workgroup barrier in
  // front of the stores: every wave leaves its sum in LDS and counts itself in there; the wave that comes last adds
  // the sums in wave order and signals for the workgroup: agent-scope store of the sum, wait for it, agent-scope add
  // (MI355X_MICROARCH.md: hand-off by one lane per storing workgroup, write-through stores and loads on both sides).
  float pv[RB]; // -1 for the outputs that do not count
  float pw = 0.f;
#pragma unroll
  for (int d = 0; d < RB; d++) {
    bool const counts = n0_bin_counts(tile * W + w + N1 * (c + RA * d), (int)N, n0);
    float const p = z[d].x * z[d].x + z[d].y * z[d].y;
    pv[d] = counts ? p : -1.f;
    if (counts) pw += p;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pw += __shfl_xor(pw, o, 64);
  size_t const pt = (size_t)i * nblocks + b;
  constexpr int NTILES_MAX = N0_SLOTS;
  int const ntiles = gridDim.x; // row tiles of the block, <= 64
  if ((tid & 63) == 0) {
    red[tid >> 6] = pw;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the sum is in LDS before the count says so
    if (__hip_atomic_fetch_add(&arrived_sh, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == NT / 64 - 1) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < NT / 64; wv++) t += red[wv];
      __hip_atomic_store(n0.part + pt * N0_SLOTS + tile, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(n0.sync + N0_SYNC_STRIDE * pt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int d = 0; d < RB; d++) buf_store2<SPEC_AUX>(z[d], rout, (c * N1 + w) * 8, RA * d * N1 * 8); // X[k1 + N1*k2], k2 = c + RA*d
  __syncthreads(); // (red is read again below)
  // The block's other tiles: dispatched next to this one, so their sums arrive within microseconds.  The wait is
  // bounded all the same (a workgroup that is not resident cannot be waited for): when it runs out the block's flag
  // goes up and a later kernel takes the second iteration from the stored spectrum.
  if (tid == 0) {
    unsigned seen = 0;
    for (int it = 0; it < n0.spin; it++) { // x ~0.1 us
      seen = __hip_atomic_load(n0.sync + N0_SYNC_STRIDE * pt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= (unsigned)ntiles) break;
      __builtin_amdgcn_s_sleep(2);
    }
    ok_sh = seen >= (unsigned)ntiles;
    if (!ok_sh) n0.sync[N0_SYNC_STRIDE * pt + 1] = 1u; // read by a later kernel
  }
  __syncthreads();
  if (!ok_sh) return; // uniform
  if (tid < 64) { // the wave whose lane polled: radio.c:456,461, twice the mean of the first iteration
    float t = tid < ntiles ? __hip_atomic_load(n0.part + pt * N0_SLOTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (tid == 0) th_sh = (t / n0.bincount) * 2;
  }
  __syncthreads();
  float const th = th_sh;
  float s1 = 0.f, c1 = 0.f;
#pragma unroll
  for (int d = 0; d < RB; d++)
    if (pv[d] >= 0.f && pv[d] < th) { // radio.c:457-460
      s1 += pv[d];
      c1 += 1.f;
    }
  s1 = block_sum<NT>(s1, red);
  c1 = block_sum<NT>(c1, red);
  float2 *pairs = reinterpret_cast<float2 *>(n0.part1 + pt * (2 * NTILES_MAX));
  if (tid == 0) pairs[tile] = make_float2(s1, c1);
  if (tile == 0 && tid >= ntiles && tid < NTILES_MAX) pairs[tid] = make_float2(0.f, 0.f); // kh_n0_final adds all 64
  }
}

template <int R>
static int launch_small(hipStream_t st, const void *time, size_t in_stride, int real_input, float2 *spec,
                        size_t spec_in_stride, int L, int nblocks, int ninputs, N0Fuse n0) {
  // two radix-4 butterflies per thread and pass at the large sizes: the block's 8 R bytes of LDS allow two workgroups
  // per CU at R = 8192, and 256-thread workgroups left the CU at 8 wavefronts
  constexpr int NT = (R / 4 < 256) ? R / 4 : (R >= 4096 ? R / 8 : 256);
  const float2 *tw = twiddle_table(R);
  if (!tw) return -1;
  hipLaunchKernelGGL((kh_fft_fwd_lds<R, NT>), dim3(nblocks, ninputs), dim3(NT), 0, st, time, in_stride, real_input, spec,
                     spec_in_stride, L, tw, n0);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

template <int RA, int RB, int NT>
static int launch_cols(hipStream_t st, const float2 *time, size_t in_stride, float2 *tmp, size_t spec_in_stride, int N2,
                       int L, int nblocks, int ninputs, const FftIngest *fi, int b0, int gblocks, unsigned *n0sync, float *ewin) {
  const float2 *tw = twiddle_table(RA * RB);
  static_assert(RB % 16 == 0, "tq rows are read as float4");
  ColsTwiddle const ct = cols_twiddle(N2, RA, RB);
  if (!tw || !ct.uj || !ct.tq) return -1;
  dim3 const grid(N2 / TILE_W, gblocks, ninputs);
  if (fi) {
    ColsIngest const ing{static_cast<const unsigned char *>(fi->raw), fi->raw_stride_bytes, fi->raw_bytes, fi->payload,
                         fi->hist_in, fi->hist_out, fi->hist_stride, fi->desc, fi->energy_part, fi->M - 1};
    hipLaunchKernelGGL((kh_fft_cols<RA, RB, NT, true>), grid, dim3(NT), 0, st, time, in_stride, tmp, spec_in_stride, N2, L,
                       tw, ct, ing, b0, nblocks, n0sync, ewin);
  } else {
    hipLaunchKernelGGL((kh_fft_cols<RA, RB, NT, false>), grid, dim3(NT), 0, st, time, in_stride, tmp, spec_in_stride, N2,
                       L, tw, ct, ColsIngest{}, b0, nblocks, n0sync, ewin);
  }
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}
template <int RA, int RB, int NT>
static int launch_rows(hipStream_t st, const float2 *tmp, float2 *spec, size_t spec_in_stride, int N1, int nblocks,
                       int ninputs, N0Fuse n0, int b0, int gblocks, const SpecMask *mask) {
  const float2 *tw = twiddle_table(RA * RB);
  if (!tw) return -1;
  static int const nt_env = getenv("KA9Q_HIP_SPEC_NT") ? atoi(getenv("KA9Q_HIP_SPEC_NT")) : -1; // -1: by size
  size_t const spec_bytes = (size_t)RA * RB * N1 * 8 * (size_t)nblocks * ninputs;            // of the whole call
  bool const nt = nt_env < 0 ? spec_bytes >= ((size_t)100 << 20) : nt_env != 0;
  dim3 const grid(N1 / TILE_W, gblocks, ninputs);
  {
    int const N = RA * RB * N1, E = N - n0.bincount; // excluded bins: E of them, cyclically before n0.start
    int const first = ((n0.start - E) % N + N) % N, last = ((n0.start - 1) % N + N) % N;
    auto edge = [&](int k) { return k < N1 ? -1 : k >= N - N1 ? 1 : 0; };
    n0.narrow = E <= 2 * N1 && edge(first) != 0 && edge(last) != 0 &&
                (edge(first) == edge(last) ? first <= last : edge(first) == 1);
  }
  SpecMask const all{{0, 0, 0, 0, 0, 0, 0, 0}, 0};
  if constexpr (TILE_W * RA == NT) {
    if (mask != nullptr && mask->on) { // (launch_forward_fft has checked: bracket mode, with the arrival counters)
      if (nt)
        hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 2, 2, true>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, *mask);
      else
        hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 2, 0, true>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, *mask);
      return KH_HIP(hipGetLastError()) ? 0 : -1;
    }
  }
  if constexpr (TILE_W * RA == NT) { // (one row of outputs per lane: the second iteration can stay in registers)
    if (n0.direct == 2 && nt)
      hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 2, 2>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, all);
    else if (n0.direct == 2)
      hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 2, 0>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, all);
    else if (n0.direct && nt)
      hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 1, 2>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, all);
    else if (n0.direct)
      hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 1, 0>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, all);
    if (n0.direct) return KH_HIP(hipGetLastError()) ? 0 : -1;
  }
  if (nt)
    hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 0, 2>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, all);
  else
    hipLaunchKernelGGL((kh_fft_rows<RA, RB, NT, 0, 0>), grid, dim3(NT), 0, st, tmp, spec, spec_in_stride, N1, tw, n0, b0, nblocks, all);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

static int ilog2(unsigned v) {
  int e = 0;
  while ((1u << e) < v) e++;
  return e;
}

// N = N1*N2 with N1 <= N2: for odd log2 the column pass gets the smaller transform -- kh_fft_cols<16,32,256> (N1 = 512)
// runs at 5.2 TB/s, the 512-thread <32,32,512> (N1 = 1024) at 4.3
static int four_step_n1(int N) { return 1 << (ilog2((unsigned)N) / 2); }

int n0_partials(int N) { // first-pass partial sums the forward transform leaves per (input, block)
  if (N <= KH_MAX_SMALL_N) return 1;
  return four_step_n1(N) / TILE_W; // one per row tile
}

int n0_cols_tiles(int N) { return N <= KH_MAX_SMALL_N ? 0 : (N / four_step_n1(N)) / TILE_W; }
int n0_rows_waves(int N) { // (launch_rows' kernels: TILE_W x RA threads)
  if (N <= KH_MAX_SMALL_N) return 0;
  int const N2 = N / four_step_n1(N);
  return N2 >= 1024 ? 8 : N2 >= 256 ? 4 : 2;
}
size_t n0_cand_floats(int N) { return N <= KH_MAX_SMALL_N ? 0 : (size_t)n0_partials(N) * n0_rows_waves(N) * (N0_CAPW + 3); }

int fused_ingest_tiles(int N) { // IF power partial sums the fused column pass leaves per (input, block); 0: no fused pass
  if (N <= KH_MAX_SMALL_N) return 0;
  return (N / four_step_n1(N)) / TILE_W; // one per column tile
}

// blocks per (column pass, row pass) pair of a four-step transform: KA9Q_HIP_FFT_GROUP_MB (default 128) MiB of scratch
// per group, but never fewer blocks than fill the chip once (1024 workgroups of 256 threads, 512 of 512)
int fft_group_blocks(int N, int nblocks, int ninputs) {
  static int const mb = getenv("KA9Q_HIP_FFT_GROUP_MB") ? atoi(getenv("KA9Q_HIP_FFT_GROUP_MB")) : 128;
  if (mb <= 0) return nblocks;
  size_t const per_block = (size_t)N * 8 * (size_t)ninputs;
  int g = (int)(((size_t)mb << 20) / per_block);
  int const N1 = four_step_n1(N), N2 = N / N1;
  auto slots = [](int n) { return n >= 1024 ? 512 : n >= 256 ? 1024 : 2048; }; // workgroups of that pass the chip holds
  int const fill_cols = (slots(N1) + N2 / TILE_W * ninputs - 1) / (N2 / TILE_W * ninputs);
  int const fill_rows = (slots(N2) + N1 / TILE_W * ninputs - 1) / (N1 / TILE_W * ninputs);
  int const fill = fill_cols > fill_rows ? fill_cols : fill_rows;
  if (g < fill) g = fill;
  if (g < 1) g = 1;
  return g < nblocks ? g : nblocks;
}

void spec_mask_clear(SpecMask *m) { *m = SpecMask{{0, 0, 0, 0, 0, 0, 0, 0}, 0}; }
static int rows_ra(int N2) { return N2 >= 1024 ? 32 : N2 >= 256 ? 16 : 8; } // launch_rows' instances
void spec_mask_add(SpecMask *m, int N, long long lo, long long hi) {
  if (N <= KH_MAX_SMALL_N || hi < lo) return;
  int const N1 = four_step_n1(N), N2 = N / N1, RA = rows_ra(N2);
  if (hi - lo >= N) lo = 0, hi = N - 1;
  long long const r0 = (lo >= 0 ? lo : lo - (N1 - 1)) / N1, r1 = (hi >= 0 ? hi : hi - (N1 - 1)) / N1; // floor
  for (long long r = r0; r <= r1; r++) {
    int const k2 = (int)(((r % N2) + N2) % N2);
    m->need[(k2 % RA) / 4] |= 1u << (k2 / RA);
  }
}
bool spec_mask_has_row(SpecMask const *m, int N, int row) {
  if (!m->on || N <= KH_MAX_SMALL_N) return true;
  int const N1 = four_step_n1(N), N2 = N / N1, RA = rows_ra(N2);
  return row >= 0 && row < N2 && (m->need[(row % RA) / 4] >> (row / RA) & 1u);
}

int launch_forward_fft(hipStream_t st, const void *time, size_t in_stride, int real_input, float2 *spec, float2 *tmp,
                       size_t spec_in_stride, int N, int L, int nblocks, int ninputs, hipEvent_t mid_event, N0Fuse n0,
                       const FftIngest *fi, const SpecMask *mask) {
  bool const pow2 = N >= 256 && !(N & (N - 1)) && N <= (1 << 20);
  if (!pow2 && !make_any_plan(N).npass) {
    set_error("forward FFT size %d unsupported (a power of two in [256, 2^20], or factors 2, 3, 5, 7 and at most %d)", N,
              KH_MAX_SMALL_N);
    return -1;
  }
  if (nblocks <= 0 || ninputs <= 0) return 0;
  if (N <= KH_MAX_SMALL_N) {
    if (fi) {
      set_error("fused ingest needs N > %d", KH_MAX_SMALL_N);
      return -1;
    }
    int rc = -1;
    if (!pow2) { // the reference's nextfastfft sizes (main.c:326-330)
      AnyPlan const plan = make_any_plan(N);
      const float2 *tw = twiddle_table(N);
      size_t const lds = 2 * (size_t)N * sizeof(float2);
      if (!tw || !allow_big_lds(kh_fft_fwd_any, lds)) return -1;
      hipLaunchKernelGGL(kh_fft_fwd_any, dim3(nblocks, ninputs), dim3(256), lds, st, time, in_stride, real_input, spec,
                         spec_in_stride, L, tw, n0, plan);
      rc = KH_HIP(hipGetLastError()) ? 0 : -1;
      if (mid_event) hipEventRecord(mid_event, st);
      return rc;
    }
    switch (N) {
    case 256: rc = launch_small<256>(st, time, in_stride, real_input, spec, spec_in_stride, L, nblocks, ninputs, n0); break;
    case 512: rc = launch_small<512>(st, time, in_stride, real_input, spec, spec_in_stride, L, nblocks, ninputs, n0); break;
    case 1024: rc = launch_small<1024>(st, time, in_stride, real_input, spec, spec_in_stride, L, nblocks, ninputs, n0); break;
    case 2048: rc = launch_small<2048>(st, time, in_stride, real_input, spec, spec_in_stride, L, nblocks, ninputs, n0); break;
    case 4096: rc = launch_small<4096>(st, time, in_stride, real_input, spec, spec_in_stride, L, nblocks, ninputs, n0); break;
    case 8192: rc = launch_small<8192>(st, time, in_stride, real_input, spec, spec_in_stride, L, nblocks, ninputs, n0); break;
    }
    if (mid_event) hipEventRecord(mid_event, st);
    return rc;
  }
  if (real_input) {
    set_error("REAL input is limited to N <= %d", KH_MAX_SMALL_N);
    return -1;
  }
  int const e = ilog2((unsigned)N);
  int const N1 = four_step_n1(N), N2 = N / N1;
  (void)e;
  const float2 *t = static_cast<const float2 *>(time);
  // Blocks per pair of launches.  The column pass writes the block's N points to scratch and the row pass reads them
  // back: a group's scratch is sized to stay in the 256 MiB Infinity Cache between the two, and every group of a call
  // reuses the same scratch, so most of that round trip never reaches HBM.
  int const group = fft_group_blocks(N, nblocks, ninputs);
  if (mask != nullptr && mask->on && (n0.part == nullptr || n0.part1 == nullptr || n0.direct != 2 || n0.arrive == nullptr)) {
    set_error("a masked spectrum needs compute_n0 in bracket mode");
    return -1;
  }
  // (the column pass zeroes its blocks' words: the rendezvous' counter and flag, or a masked spectrum's arrival counters)
  unsigned *const n0sync = n0.direct == 1 ? n0.sync : (mask != nullptr && mask->on) ? n0.arrive : nullptr;
  float *const ewin = n0.direct == 2 ? n0.ewin : nullptr;
  int rc = 0;
  for (int b0 = 0; b0 < nblocks && !rc; b0 += group) {
    int const gb = nblocks - b0 < group ? nblocks - b0 : group;
    rc = -1;
    switch (N1) {
    case 128: rc = launch_cols<8, 16, 128>(st, t, in_stride, tmp, spec_in_stride, N2, L, nblocks, ninputs, fi, b0, gb, n0sync, ewin); break;
    case 256: rc = launch_cols<16, 16, 256>(st, t, in_stride, tmp, spec_in_stride, N2, L, nblocks, ninputs, fi, b0, gb, n0sync, ewin); break;
    case 512: rc = launch_cols<16, 32, 256>(st, t, in_stride, tmp, spec_in_stride, N2, L, nblocks, ninputs, fi, b0, gb, n0sync, ewin); break;
    case 1024: rc = launch_cols<32, 32, 512>(st, t, in_stride, tmp, spec_in_stride, N2, L, nblocks, ninputs, fi, b0, gb, n0sync, ewin); break;
    }
    if (rc) return rc;
    if (mid_event && b0 == 0) hipEventRecord(mid_event, st);
    switch (N2) {
    case 128: rc = launch_rows<8, 16, 128>(st, tmp, spec, spec_in_stride, N1, nblocks, ninputs, n0, b0, gb, mask); break;
    case 256: rc = launch_rows<16, 16, 256>(st, tmp, spec, spec_in_stride, N1, nblocks, ninputs, n0, b0, gb, mask); break;
    case 512: rc = launch_rows<16, 32, 256>(st, tmp, spec, spec_in_stride, N1, nblocks, ninputs, n0, b0, gb, mask); break;
    case 1024: rc = launch_rows<32, 32, 512>(st, tmp, spec, spec_in_stride, N1, nblocks, ninputs, n0, b0, gb, mask); break;
    default: rc = -1;
    }
  }
  return rc;
}

// ================================================================ filter output: one workgroup per (channel, block)
// algorithmic bytes per workgroup: 8*ND spectrum bins + 8*ND response + 8*olen (4*olen REAL) output
template <int ND, int NT>
__global__ __launch_bounds__(NT) void kh_filter_out(const float2 *__restrict__ spec, size_t spec_in_stride, int N,
                                                    int in_real, const ChanDesc *__restrict__ chans,
                                                    const float2 *__restrict__ response, int olen, void *out,
                                                    size_t out_stride, const float2 *__restrict__ tw, FmFront fm,
                                                    AnyPlan plan) {
  // ND == 0: any 2,3,5,7-smooth size (plan.n), transform by any_fft in two LDS buffers of the launch's dynamic LDS
  constexpr bool GEN = ND == 0;
  extern __shared__ __attribute__((aligned(16))) float2 dynf[];
  __shared__ __attribute__((aligned(16))) float2 statf[GEN ? 1 : ND];
  float2 *f = GEN ? dynf : statf;
  int const nd = GEN ? plan.n : ND;
  __shared__ float red[NT / 64 > 0 ? NT / 64 : 1];
  __shared__ double w_A[NT / 64 > 0 ? NT / 64 : 1], w_B[NT / 64 > 0 ? NT / 64 : 1];
  int const tid = threadIdx.x;
  int const ch = blockIdx.x, b = blockIdx.y;
  ChanDesc const cd = chans[ch];
  const float2 *fd = spec + (size_t)cd.input * spec_in_stride + (size_t)b * N;
  const float2 *resp = response + (size_t)ch * nd;
  bool const out_real = cd.out_type == FT_REAL;
  int const rotate = cd.rotate;

  if (!in_real && !out_real) { // complex -> complex, filter.c:274-289
    int m0 = N - nd / 2 + 1 + rotate;
    if (m0 >= N) m0 -= N;
    if (m0 < 0) m0 += N;
    for (int i = tid; i < nd; i += NT) {
      int p = nd / 2 + 1 + i;
      if (p >= nd) p -= nd;
      int m = m0 + i;
      if (m >= N) m -= N;
      f[p] = cmul(resp[p], fd[m]);
    }
  } else {
    int m0 = rotate;
    if (m0 < 0) m0 += N;
    if (in_real && out_real) { // real -> real, filter.c:290-299 (bins above nd/2 are rebuilt below)
      for (int p = tid; p <= nd / 2; p += NT) {
        int m = m0 + p;
        if (m >= N) m -= N;
        f[p] = cmul(resp[p], fd[m]);
      }
    } else if (!in_real && out_real) { // complex -> real, filter.c:300-317
      for (int p = tid; p <= nd / 2; p += NT) {
        int m = m0 + p;
        if (m >= N) m -= N;
        float2 v = cmul(resp[p], fd[m]);
        if (p > 0 && p < nd / 2) v = cadd(v, cconj(cmul(resp[nd - p], fd[m])));
        f[p] = v;
      }
    } else { // real -> complex, filter.c:318-335
      for (int p = tid; p <= nd / 2; p += NT) {
        int m = m0 + p;
        if (m >= N) m -= N;
        float2 const x = fd[m];
        f[p] = cmul(resp[p], x);
        if (p > 0 && p < nd / 2) f[nd - p] = cmul(resp[nd - p], cconj(x));
      }
    }
  }
  __syncthreads();
  if (cd.out_type == FT_CROSS_CONJ) { // filter.c:371-381
    for (int p = 1 + tid; p < nd / 2; p += NT) {
      int const dn = nd - p;
      float2 const pos = f[p], neg = f[dn];
      f[p] = cadd(pos, cconj(neg));
      f[dn] = csub(neg, cconj(pos));
    }
    __syncthreads();
  }
  if (out_real) { // c2r (filter.c:207,382): Hermitian extension, imaginary parts of DC and Nyquist ignored
    for (int p = tid; p <= nd / 2; p += NT) {
      float2 v = f[p];
      if (p == 0 || p == nd / 2)
        f[p] = make_float2(v.x, 0.f);
      else
        f[nd - p] = cconj(v);
    }
    __syncthreads();
  }
  if constexpr (GEN)
    f = any_fft<+1>(f, f + nd, plan, tw, tid, NT);
  else
    lds_fft<ND, 1, NT, 1, 1, false, +1>(f, tw, tid);
  // the valid part of the circular convolution is the last olen points (filter.c:197,206)
  if (out_real) {
    float *o = static_cast<float *>(out) + (size_t)ch * out_stride + (size_t)b * olen;
    for (int n = tid; n < olen; n += NT) o[n] = f[nd - olen + n].x;
  } else {
    bool const is_fm = fm.stat != nullptr && cd.fm_stats; // uniform
    if (!is_fm || fm.write_baseband) {
      float2 *o = static_cast<float2 *>(out) + (size_t)ch * out_stride + (size_t)b * olen;
      for (int n = tid; n < olen; n += NT) o[n] = f[nd - olen + n];
    }
    if (is_fm) { // fm.c:53-70 and the block-local part of fm.c:85-112 while the block is still in LDS
      float2 *y = f + (nd - olen);
      float bb = 0.f, sa = 0.f;
      for (int n = tid; n < olen; n += NT) {
        float2 const v = y[n];
        float const t = v.x * v.x + v.y * v.y;
        bb += t;
        sa += sqrtf(t);
      }
      bb = block_sum<NT>(bb, red);
      sa = block_sum<NT>(sa, red);
      float const avg = sa / olen;
      float var = 0.f;
      for (int n = tid; n < olen; n += NT) {
        float2 const v = y[n];
        float const a = sqrtf(v.x * v.x + v.y * v.y);
        var += (a - avg) * (a - avg);
      }
      var = block_sum<NT>(var, red) / (olen - 1);
      float2 const first = y[0], last = y[olen - 1];
      float snr = 0.f;
      if (tid < 64) snr = (float)fm_snr((double)(avg * avg / var)); // wave 0 (NT >= 64)
      __syncthreads(); // first/last are read before fm_front overwrites the block
      const DemodDesc *dd = fm.descs + ch;
      size_t const rec = (size_t)ch * fm.stat_stride + b;
      float const angsum = fm_front<NT>(y, olen, dd->fm_gain, dd->flat, var, fm.work + (size_t)ch * fm.work_stride + (size_t)b * olen,
                                        fm.carry + rec, red, w_A, w_B);
      if (tid == 0) fm.stat[rec] = FmBlockStat{bb / olen, var, fmaxf(0.0f, snr), angsum, first, last};
    }
  }
}

template <int ND>
static int launch_fo(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int in_real, const ChanDesc *chans,
                     const float2 *response, int olen, void *out, size_t out_stride, int nblocks, int nchan,
                     FmFront const &fm) {
  constexpr int NT = (ND / 4 < 256) ? ND / 4 : 256;
  const float2 *tw = twiddle_table(ND);
  if (!tw) return -1;
  hipLaunchKernelGGL((kh_filter_out<ND, NT>), dim3(nchan, nblocks), dim3(NT), 0, st, spec, spec_in_stride, N, in_real,
                     chans, response, olen, out, out_stride, tw, fm, AnyPlan{});
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}
static int launch_fo_any(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int in_real, const ChanDesc *chans,
                         const float2 *response, int ND, int olen, void *out, size_t out_stride, int nblocks, int nchan,
                         FmFront const &fm) {
  AnyPlan const plan = make_any_plan(ND);
  const float2 *tw = twiddle_table(ND);
  if (!plan.npass || !tw) {
    set_error("decimated FFT size %d unsupported (factors 2, 3, 5, 7; at most %d)", ND, KH_MAX_SMALL_N);
    return -1;
  }
  size_t const lds = 2 * (size_t)ND * sizeof(float2);
  if (!allow_big_lds(kh_filter_out<0, 256>, lds)) return -1;
  hipLaunchKernelGGL((kh_filter_out<0, 256>), dim3(nchan, nblocks), dim3(256), lds, st, spec, spec_in_stride, N, in_real,
                     chans, response, olen, out, out_stride, tw, fm, plan);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

int launch_filter_output(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int in_real,
                         const ChanDesc *chans, const float2 *response, int ND, int olen, void *out, size_t out_stride,
                         int nblocks, int nchan, FmFront const &fm, int complex_out, const N0Pass1Args *n0, int *n0_taken) {
  if (n0_taken) *n0_taken = 0;
  if (nblocks <= 0 || nchan <= 0) return 0;
  if (olen > ND || ND > N) {
    set_error("filter output: olen %d > N_dec %d or N_dec > N %d", olen, ND, N);
    return -1;
  }
  static int const wave_env = getenv("KA9Q_HIP_CHAN_WAVE") ? atoi(getenv("KA9Q_HIP_CHAN_WAVE")) : 1; // A/B switch
  if (wave_env && complex_out && !in_real && chan_wave_supported(ND, olen)) {
    if (n0_taken) *n0_taken = n0 != nullptr;
    if (fm.stat == nullptr && fm2_usable(ND, olen, nblocks, 0)) // no FM epilogue wanted: the 32-points-per-lane form
      return launch_chan2(st, spec, spec_in_stride, N, chans, response, ND, olen, static_cast<float2 *>(out), out_stride, nblocks,
                          nchan, nullptr, 0, 1, n0);
    return launch_chan_wave(st, spec, spec_in_stride, N, chans, response, ND, olen, static_cast<float2 *>(out), out_stride,
                            nblocks, nchan, fm, n0);
  }
#define FO(n) \
  case n: return launch_fo<n>(st, spec, spec_in_stride, N, in_real, chans, response, olen, out, out_stride, nblocks, nchan, fm)
  switch (ND) {
    FO(256);
    FO(512);
    FO(1024);
    FO(2048);
    FO(4096);
    FO(8192);
  }
#undef FO
  return launch_fo_any(st, spec, spec_in_stride, N, in_real, chans, response, ND, olen, out, out_stride, nblocks, nchan, fm);
}

// ================================================================ response design (control path, but on the device)
__device__ float dev_i0f(float z) { // filter.c:410-421
  float const t = (z * z) / 4;
  float sum = 1 + t;
  float term = t;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * k);
    sum += term;
    if ((double)term < 1e-12 * (double)sum) break;
  }
  return sum;
}
// one element of make_kaiser, filter.c:481-501
__device__ float dev_kaiser(int n, int M, float numc, float inv_denom, float pc) {
  if ((M & 1) && n == (M - 1) / 2) return 1.f;
  int const nn = n < M / 2 ? n : M - 1 - n;
  float const p = pc * nn - 1;
  return dev_i0f(numc * sqrtf(1 - p * p)) * inv_denom;
}

__global__ void kh_make_kaiser(float *window, int M, float beta) {
  float const numc = (float)(M_PI * (double)beta);
  float const inv_denom = (float)(1. / (double)dev_i0f(numc));
  float const pc = (float)(2.0 / (double)(M - 1));
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < M; n += gridDim.x * blockDim.x)
    window[n] = dev_kaiser(n, M, numc, inv_denom, pc);
}
int launch_make_kaiser(hipStream_t st, float *window_dev, int M, float beta) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(kh_make_kaiser, dim3((M + 255) / 256), dim3(256), 0, st, window_dev, M, beta);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// set_filter's brick wall, filter.c:674-691
__global__ void kh_brickwall(float2 *resp, int ND, float low, float high, float gain) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < ND; n += gridDim.x * blockDim.x) {
    float f;
    if (n <= ND / 2)
      f = (float)n / ND;
    else
      f = (float)(n - ND) / ND;
    resp[n] = (f >= low && f <= high) ? make_float2(gain, 0.f) : make_float2(0.f, 0.f);
  }
}
int launch_brickwall(hipStream_t st, float2 *resp_dev, int ND, float low, float high, float gain) {
  hipLaunchKernelGGL(kh_brickwall, dim3((ND + 255) / 256), dim3(256), 0, st, resp_dev, ND, low, high, gain);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// window_filter (filter.c:509-568) and window_rfilter (573-628): to the time domain, rotate by M/2, Kaiser
// window, gain 1/N, zero pad, back to the frequency domain.  One workgroup.
template <int ND, int NT>
__global__ __launch_bounds__(NT) void kh_window_filter(float2 *resp, int M, float beta, int real_filter,
                                                       const float2 *__restrict__ tw, AnyPlan plan) {
  constexpr bool GEN = ND == 0; // as in kh_filter_out
  extern __shared__ __attribute__((aligned(16))) float2 dynf[];
  __shared__ __attribute__((aligned(16))) float2 statf[GEN ? 1 : ND];
  float2 *buf = GEN ? dynf : statf;
  int const nd = GEN ? plan.n : ND;
  constexpr int PT = GEN ? (KH_MAX_SMALL_N / NT) : (ND / NT > 0 ? ND / NT : 1);
  int const tid = threadIdx.x;
  if (real_filter) { // c2r input: nd/2+1 bins, Hermitian extension (filter.c:589-590)
    for (int p = tid; p <= nd / 2; p += NT) {
      float2 v = resp[p];
      if (p == 0 || p == nd / 2)
        buf[p] = make_float2(v.x, 0.f);
      else {
        buf[p] = v;
        buf[nd - p] = cconj(v);
      }
    }
  } else {
    for (int n = tid; n < nd; n += NT) buf[n] = resp[n];
  }
  __syncthreads();
  if constexpr (GEN)
    buf = any_fft<+1>(buf, buf == dynf ? dynf + nd : dynf, plan, tw, tid, NT);
  else
    lds_fft<ND, 1, NT, 1, 1, false, +1>(buf, tw, tid);
  float const numc = (float)(M_PI * (double)beta);
  float const inv_denom = (float)(1. / (double)dev_i0f(numc));
  float const pc = (float)(2.0 / (double)(M - 1));
  float const gain = (float)(1. / (double)nd);
  float2 v[PT];
#pragma unroll
  for (int u = 0; u < PT; u++) {
    int const n = tid + u * NT;
    float2 r = make_float2(0.f, 0.f); // zero padding, filter.c:546
    if (n < M) {
      // filter.c:543-544 runs n = M-1 .. 0 in place: a source index above n and below M has already
      // been replaced by its own windowed value when it is read.
      int const src = (n - M / 2 + nd) % nd;
      float2 s;
      if (src > n && src < M) {
        int const src2 = (src - M / 2 + nd) % nd;
        float const ws = dev_kaiser(src, M, numc, inv_denom, pc);
        float2 const o = buf[src2];
        s = make_float2(o.x * ws * gain, o.y * ws * gain);
      } else {
        s = buf[src];
      }
      if (real_filter) s.y = 0.f;
      float const w = dev_kaiser(n, M, numc, inv_denom, pc);
      r = make_float2(s.x * w * gain, s.y * w * gain);
    }
    v[u] = r;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PT; u++)
    if (tid + u * NT < nd) buf[tid + u * NT] = v[u];
  __syncthreads();
  if constexpr (GEN)
    buf = any_fft<-1>(buf, buf == dynf ? dynf + nd : dynf, plan, tw, tid, NT);
  else
    lds_fft<ND, 1, NT, 1, 1, false, -1>(buf, tw, tid);
  int const nout = real_filter ? nd / 2 + 1 : nd;
  for (int n = tid; n < nout; n += NT) resp[n] = buf[n];
}

template <int ND>
static int launch_wf(hipStream_t st, float2 *resp, int M, float beta, int real_filter) {
  constexpr int NT = (ND / 4 < 256) ? ND / 4 : 256;
  const float2 *tw = twiddle_table(ND);
  if (!tw) return -1;
  hipLaunchKernelGGL((kh_window_filter<ND, NT>), dim3(1), dim3(NT), 0, st, resp, M, beta, real_filter, tw, AnyPlan{});
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}
int launch_window_filter(hipStream_t st, float2 *resp_dev, int L, int M, float beta, int real_filter) {
  int const ND = L + M - 1;
  if (M < 1 || M > ND) {
    set_error("window_filter: bad M %d for N %d", M, ND);
    return -1;
  }
  switch (ND) {
  case 256: return launch_wf<256>(st, resp_dev, M, beta, real_filter);
  case 512: return launch_wf<512>(st, resp_dev, M, beta, real_filter);
  case 1024: return launch_wf<1024>(st, resp_dev, M, beta, real_filter);
  case 2048: return launch_wf<2048>(st, resp_dev, M, beta, real_filter);
  case 4096: return launch_wf<4096>(st, resp_dev, M, beta, real_filter);
  case 8192: return launch_wf<8192>(st, resp_dev, M, beta, real_filter);
  }
  AnyPlan const plan = make_any_plan(ND);
  const float2 *tw = twiddle_table(ND);
  if (!plan.npass || !tw) {
    set_error("window_filter: FFT size %d unsupported (factors 2, 3, 5, 7; at most %d)", ND, KH_MAX_SMALL_N);
    return -1;
  }
  size_t const lds = 2 * (size_t)ND * sizeof(float2);
  if (!allow_big_lds(kh_window_filter<0, 256>, lds)) return -1;
  hipLaunchKernelGGL((kh_window_filter<0, 256>), dim3(1), dim3(256), lds, st, resp_dev, M, beta, real_filter, tw, plan);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// noise_gain, filter.c:631-656: scale * sum |response|^2
__global__ __launch_bounds__(256) void kh_noise_gain(const float2 *__restrict__ resp, int count, float scale, float *out) {
  __shared__ float part[256];
  float s = 0.f;
  for (int n = threadIdx.x; n < count; n += 256) {
    float2 const r = resp[n];
    s += r.x * r.x + r.y * r.y;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = scale * part[0];
}
int launch_noise_gain(hipStream_t st, const float2 *resp_dev, int count, float scale, float *out_dev) {
  hipLaunchKernelGGL(kh_noise_gain, dim3(1), dim3(256), 0, st, resp_dev, count, scale, out_dev);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ overlap-save history carry (filter.c:230)
__global__ void kh_carry(const char *__restrict__ src, char *__restrict__ dst, size_t row_stride, size_t bytes) {
  const char *s = src + (size_t)blockIdx.y * row_stride;
  char *d = dst + (size_t)blockIdx.y * row_stride;
  size_t const n4 = bytes / 4;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<float *>(d)[k] = reinterpret_cast<const float *>(s)[k];
}
int launch_carry(hipStream_t st, const void *src, void *dst, size_t row_stride_bytes, size_t bytes, int nrows) {
  if (bytes == 0 || nrows <= 0) return 0;
  size_t const n4 = bytes / 4;
  int gx = (int)((n4 + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(kh_carry, dim3(gx, nrows), dim3(256), 0, st, static_cast<const char *>(src),
                     static_cast<char *>(dst), row_stride_bytes, bytes);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ streaming ingest: pinned host ring -> device raw ring
// The transfer of kh_bank_stream_submit as a kernel (KA9Q_HIP_STREAM_SDMA=0; the default is hipMemcpyAsync on the copy
// engines, 34 GB/s against this kernel's 28 GB/s on the test box): the lanes read the pinned slot over PCIe (host memory
// mapped into the device's address space) and write the device ring.  A few workgroups hold enough 16-byte reads in
// flight to keep the link busy and leave the other CUs to the launch running beside it.
// Rows start 16-byte aligned (raw_stride is a multiple of 256); algorithmic bytes: `bytes` read + `bytes` written per row.
__global__ __launch_bounds__(256) void kh_host_pull(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst,
                                                    size_t row_stride, size_t bytes) {
  const unsigned char *s = src + (size_t)blockIdx.y * row_stride;
  unsigned char *d = dst + (size_t)blockIdx.y * row_stride;
  size_t const n16 = bytes / 16, step = (size_t)gridDim.x * blockDim.x;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *s4 = reinterpret_cast<const u32x4 *>(s);
  u32x4 *d4 = reinterpret_cast<u32x4 *>(d);
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; k + 3 * step < n16; k += 4 * step) {
    u32x4 const a = __builtin_nontemporal_load(s4 + k), b = __builtin_nontemporal_load(s4 + k + step),
                c = __builtin_nontemporal_load(s4 + k + 2 * step), e = __builtin_nontemporal_load(s4 + k + 3 * step);
    d4[k] = a, d4[k + step] = b, d4[k + 2 * step] = c, d4[k + 3 * step] = e;
  }
  for (; k < n16; k += step) d4[k] = __builtin_nontemporal_load(s4 + k);
  for (size_t t = n16 * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < bytes; t += step) d[t] = s[t];
}
int launch_host_pull(hipStream_t st, const void *src_host_mapped, void *dst, size_t row_stride_bytes, size_t bytes, int nrows) {
  if (bytes == 0 || nrows <= 0) return 0;
  int gx = (int)((bytes / 16 + 4 * 256 - 1) / (4 * 256));
  gx = gx < 1 ? 1 : gx > 32 ? 32 : gx;
  hipLaunchKernelGGL(kh_host_pull, dim3(gx, nrows), dim3(256), 0, st, static_cast<const unsigned char *>(src_host_mapped),
                     static_cast<unsigned char *>(dst), row_stride_bytes, bytes);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ notch: filter.c:719-727
// One-pole notch at the mixer frequency: spin down, subtract and track the DC term, spin back up.  The recurrence runs
// through every sample (dcstate and the fp64 phasor), in the reference's order of operations and precisions: one lane.
__global__ __launch_bounds__(64) void kh_notch(NotchState *__restrict__ st, float2 *__restrict__ buf, int n) {
  if (threadIdx.x != 0) return;
  double pr = st->pr, pi = st->pi;
  double const sr = st->sr, si = st->si;
  float dr = st->dr, di = st->di;
  float const bw = st->bw;
  for (int k = 0; k < n; k++) {
    float2 const x = buf[k];
    // s = s * conj(osc_phase) - dcstate, in double, rounded to float
    float const ur = (float)(((double)x.x * pr + (double)x.y * pi) - (double)dr);
    float const ui = (float)(((double)x.y * pr - (double)x.x * pi) - (double)di);
    dr += bw * ur; // dcstate += bw * s
    di += bw * ui;
    buf[k] = make_float2((float)((double)ur * pr - (double)ui * pi), (float)((double)ur * pi + (double)ui * pr)); // s *= osc_phase
    double const npr = pr * sr - pi * si; // osc_phase *= osc_step
    pi = pr * si + pi * sr;
    pr = npr;
  }
  st->pr = pr;
  st->pi = pi;
  st->dr = dr;
  st->di = di;
}
int launch_notch(hipStream_t st, NotchState *state, float2 *buf, int n) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kh_notch, dim3(1), dim3(64), 0, st, state, buf, n);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
