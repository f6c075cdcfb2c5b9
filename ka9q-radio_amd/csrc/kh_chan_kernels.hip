// kh_chan_kernels.hip -- filter output for complex channels with N_dec = 1024 or 512 and olen = N_dec/2, wave-resident.
//
// Reference behaviour restated here (citations into the reference tree):
//   execute_filter_output  filter.c:240-289 (complex -> complex bin select/rotate x response), 371-382 (CROSS_CONJ
//                          fold, backward DFT of N/D points), filter.c:197,206 (the last olen points are the output)
//   demod_fm               fm.c:53-70 (block statistics), 85-112 (discriminator, threshold extension, de-emphasis)
//
// Mapping: one 64-lane wavefront owns one (channel, block) and never waits for another wave (layout: see the comment
// above ChanArgs).  Of the N_dec outputs only the upper half is wanted (overlap-save); it ends up as 8 samples per
// lane, consecutive lanes holding consecutive samples.  For FM channels the block statistics, the conjugate-product
// discriminator (neighbour = lane-1: a rotation of the wave) and, after a second transpose that gives every lane 8
// consecutive samples, the de-emphasis integrator from a zero carry-in follow without the block ever leaving the wave.
// kh_fm_finish_w (kh_demod_kernels.hip) then folds in what depends on earlier blocks.
//
// Roofline: HBM.  algorithmic bytes per (channel, block): 8*ND spectrum bins + 8*ND response (L2 after the first
// block) + 8*olen baseband (if kept) + FM: 8*olen work + 32 statistics.
#include "kh_internal.h"
#include "kh_fft.cuh"
#include "kh_regfft.cuh"
#include "kh_fm.cuh"
#include "kh_n0.cuh"
#include <cstdlib>

namespace kh {

enum { CW_COMPLEX = 1, CW_CROSS_CONJ = 2 }; // filter.h:18-23

// v[c] *= W^{SIGN b c}, c < RA, from the table tw[k] = exp(-2 pi i k / (RA*RB)); RA/4 + 2 table reads per lane
template <int RA, int SIGN>
__device__ __forceinline__ void chan_twiddle(float2 (&v)[RA], const float2 *__restrict__ tw, int b) {
  float2 const u1 = twiddle<SIGN>(tw, b), u2 = twiddle<SIGN>(tw, 2 * b), u3 = twiddle<SIGN>(tw, 3 * b);
  v[1] = cmul(v[1], u1);
  v[2] = cmul(v[2], u2);
  v[3] = cmul(v[3], u3);
#pragma unroll
  for (int q = 1; q < RA / 4; q++) {
    float2 const t = twiddle<SIGN>(tw, 4 * q * b);
    v[4 * q] = cmul(v[4 * q], t);
    v[4 * q + 1] = cmul(v[4 * q + 1], cmul(t, u1));
    v[4 * q + 2] = cmul(v[4 * q + 2], cmul(t, u2));
    v[4 * q + 3] = cmul(v[4 * q + 3], cmul(t, u3));
  }
}

// v[j] *= W^{SIGN STRIDE h j}, j < RA, from the same table (STRIDE = table length / transform length)
template <int RA, int SIGN, int STRIDE>
__device__ __forceinline__ void chan_twiddle_strided(float2 (&v)[RA], const float2 *__restrict__ tw, int h) {
  float2 const u1 = twiddle<SIGN>(tw, STRIDE * h), u2 = twiddle<SIGN>(tw, STRIDE * 2 * h), u3 = twiddle<SIGN>(tw, STRIDE * 3 * h);
  v[1] = cmul(v[1], u1);
  v[2] = cmul(v[2], u2);
  v[3] = cmul(v[3], u3);
#pragma unroll
  for (int q = 1; q < RA / 4; q++) {
    float2 const t = twiddle<SIGN>(tw, STRIDE * 4 * q * h);
    v[4 * q] = cmul(v[4 * q], t);
    v[4 * q + 1] = cmul(v[4 * q + 1], cmul(t, u1));
    v[4 * q + 2] = cmul(v[4 * q + 2], cmul(t, u2));
    v[4 * q + 3] = cmul(v[4 * q + 3], cmul(t, u3));
  }
}

template <int R>
__device__ __forceinline__ float group_sum(float v) { // over the R lanes of a group; every lane gets the total
#pragma unroll
  for (int o = R / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, R);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// One whole wavefront per (channel, block): ND = RA * 64, RA points per lane.
//   stage A  lane b < 64: RA-point transform over a of f[64 a + b], twiddle W_ND^{b c}
//   exchange ex[c][b] through LDS (RA rows of 64 complex points)
//   stage B  the 64-point transform of row c is split four ways (decimation in frequency): lane (c, h), h < 4, forms
//            u[j] = (sum_k z[j + 16 k] i^{k h}) W_64^{j h}, j < 16, and transforms it: X[c + RA (4 d' + h)], d' < 16.
//            Every lane of a row reads the whole row (LDS broadcast), nothing is exchanged afterwards.
// Lane l = c + RA h ends up with outputs n = l + 64 d'; the block's output (upper half) is d' >= 8: sample
// n' = l + 64 e, e < 8: consecutive lanes hold consecutive samples.  Half the register count of the R*R form and
// twice the waves: that is what counts while a launch has fewer (channel, block) pairs than the chip has SIMDs x 2.
struct ChanArgs {
  const float2 *spec;
  size_t spec_in_stride;
  int N;
  const ChanDesc *chans;
  const float2 *response;
  float2 *out;
  size_t out_stride;
  const float2 *tw;
  FmFront fm;
  int nchan, npairs;
};
constexpr int CW_RA = 16, CW_PITCH = 64 + 1;    // rows of 64 complex points, pitch in float2
constexpr int CW_LDS = 2 * CW_RA * (32 + 1);      // float2 per wave: two 32-lane pairs (1056) > one 64-lane pair (1040)
static_assert(CW_LDS >= CW_RA * CW_PITCH, "both layouts fit");

// the wave's barrier: LDS operations of one wave execute in order, so all that is needed between a transposing write
// and its read is that the compiler keeps them apart (several of these waves share a workgroup with compute_n0's
// second pass in kh_back_a: a workgroup barrier would tie them together for nothing)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// RB lanes own one (channel, block): RB = 64 (N_dec = 1024, one pair per wavefront) or RB = 32 (N_dec = 512, two pairs
// per wavefront: the same channel's blocks b and b + 1, so that everything per channel stays wave-uniform).  `lane` is
// the lane within the pair's group; `valid` is false for the padding pair of an odd block count, which computes on
// block b (clamped by the caller) and stores nothing.
template <int RA, int RB>
__device__ __forceinline__ void chan_wave64_body(ChanArgs const &A, float2 *ex /*RA*(RB+1) float2 of this pair*/, int ch, int b,
                                                 bool valid, int lane) {
  static_assert(RA == 16 && (RB == 64 || RB == 32), "RB lanes = RA rows x RB/16 splits");
  constexpr int ND = RA * RB, OLEN = ND / 2, K = OLEN / RB, SPLIT = RB / RA; // K = 8 samples per lane
  static_assert(K == 8, "eight samples per lane");
  constexpr int PITCH = RB + 1;   // in float2: rows start in different banks
  constexpr int PITCH2 = K + 1;   // second transpose, in float2 (angle, weight)
  const float2 *__restrict__ spec = A.spec;
  const float2 *__restrict__ response = A.response;
  const float2 *__restrict__ tw = A.tw;
  float2 *__restrict__ out = A.out;
  FmFront const &fm = A.fm;
  int const N = A.N, nchan = A.nchan;
  size_t const spec_in_stride = A.spec_in_stride, out_stride = A.out_stride;
  (void)nchan;
  const ChanDesc *chans = A.chans;
  ChanDesc const cd = chans[ch];
  const float2 *fd = spec + (size_t)cd.input * spec_in_stride + (size_t)b * N;
  const float2 *resp = response + (size_t)ch * ND;

  // ---- filter.c:274-289: f[p] = response[p] * fdomain[m], p = (ND/2+1+i) mod ND, m = (N-ND/2+1+rotate+i) mod N
  int m0 = N - ND / 2 + 1 + cd.rotate;
  if (m0 >= N) m0 -= N;
  if (m0 < 0) m0 += N;
  float2 v[RA];
#pragma unroll
  for (int a = 0; a < RA; a++) {
    int const p = RB * a + lane;
    int const i = (p - (ND / 2 + 1)) & (ND - 1);
    int m = m0 + i;
    if (m >= N) m -= N;
    v[a] = cmul(resp[p], fd[m]);
  }
  if (cd.out_type == CW_CROSS_CONJ) { // filter.c:371-381, on the fly: the mirrored bin is loaded as well
#pragma unroll
    for (int a = 0; a < RA; a++) {
      int const p = RB * a + lane;
      if (p != 0 && p != ND / 2) {
        int const dn = ND - p;
        int const i = (dn - (ND / 2 + 1)) & (ND - 1);
        int m = m0 + i;
        if (m >= N) m -= N;
        float2 const o = cmul(resp[dn], fd[m]);
        v[a] = p < ND / 2 ? cadd(v[a], cconj(o)) : csub(v[a], cconj(o));
      }
    }
  }

  // ---- stage A
  reg_fft<RA, +1>(v);
  chan_twiddle<RA, +1>(v, tw, lane);
#pragma unroll
  for (int c = 0; c < RA; c++) ex[c * PITCH + lane] = v[c];
  wave_sync();

  // ---- stage B
  int const c = lane & (RA - 1), h = lane / RA;
  float2 u[16];
  if constexpr (SPLIT == 4) {
    float const sgn = (h & 1) ? -1.f : 1.f;                                  // i^{2h}
    float const cr = h == 0 ? 1.f : h == 2 ? -1.f : 0.f, ci = h == 1 ? 1.f : h == 3 ? -1.f : 0.f; // i^h
    const float2 *row = ex + c * PITCH;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      float2 const z0 = row[j], z1 = row[j + 16], z2 = row[j + 32], z3 = row[j + 48];
      float2 const s = make_float2(fmaf(sgn, z2.x, z0.x), fmaf(sgn, z2.y, z0.y)); // z0 + i^{2h} z2
      float2 const t = make_float2(fmaf(sgn, z3.x, z1.x), fmaf(sgn, z3.y, z1.y)); // z1 + i^{2h} z3
      u[j] = make_float2(s.x + cr * t.x - ci * t.y, s.y + cr * t.y + ci * t.x);   // + i^h (...)
    }
  } else { // two-way split of the 32-point row: u[j] = z[j] + (-1)^h z[j + 16]
    float const sgn = h ? -1.f : 1.f;
    const float2 *row = ex + c * PITCH;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      float2 const z0 = row[j], z1 = row[j + 16];
      u[j] = make_float2(fmaf(sgn, z1.x, z0.x), fmaf(sgn, z1.y, z0.y));
    }
  }
  chan_twiddle_strided<16, +1, ND / RB>(u, tw, h); // u[j] *= W_RB^{j h}
  reg_fft<16, +1>(u);
  float2 y[K]; // sample n' = lane + RB e of the block (filter.c:197,206: the last OLEN points)
#pragma unroll
  for (int e = 0; e < K; e++) y[e] = u[8 + e];

  bool const is_fm = fm.stat != nullptr && cd.fm_stats; // uniform over the wave
  if ((!is_fm || fm.write_baseband) && valid) {
    float2 *o = out + (size_t)ch * out_stride + (size_t)b * OLEN;
#pragma unroll
    for (int e = 0; e < K; e++) o[lane + RB * e] = y[e];
  }
  if (!is_fm) return;

  // ---- fm.c:53-70: baseband power, amplitude mean and variance, Rice-corrected SNR
  float amp[K];
  float bb = 0.f, sa = 0.f;
#pragma unroll
  for (int e = 0; e < K; e++) {
    float const t = y[e].x * y[e].x + y[e].y * y[e].y;
    bb += t;
    amp[e] = __builtin_amdgcn_sqrtf(t);
    sa += amp[e];
  }
  bb = group_sum<RB>(bb);
  sa = group_sum<RB>(sa);
  float const avg = sa / OLEN;
  float var = 0.f;
#pragma unroll
  for (int e = 0; e < K; e++) var += (amp[e] - avg) * (amp[e] - avg);
  var = group_sum<RB>(var) / (OLEN - 1);
  float snr;
  if constexpr (RB == 64) {
    snr = fmaxf(0.0f, (float)fm_snr((double)(avg * avg / var)));
  } else { // fm_snr wants a wave-uniform argument: one call per half
    float const r = avg * avg / var;
    float const r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 0));
    float const r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 32));
    float const s0_ = fmaxf(0.0f, (float)fm_snr((double)r0)), s1_ = fmaxf(0.0f, (float)fm_snr((double)r1));
    snr = (threadIdx.x & 32) ? s1_ : s0_;
  }
  float2 const first = make_float2(__shfl(y[0].x, 0, RB), __shfl(y[0].y, 0, RB));
  float2 const last = make_float2(__shfl(y[K - 1].x, RB - 1, RB), __shfl(y[K - 1].y, RB - 1, RB));

  // ---- fm.c:85-93 for samples 1 .. OLEN-1: p = y[n] conj(y[n-1]); n-1 sits in lane-1 (lane 0: lane 63, one row up)
  float ang[K], wgt[K];
  float angsum = 0.f;
  {
    float const inv_var = 1.f / var;
    int const src = (lane + RB - 1) & (RB - 1);
    float2 up = make_float2(0.f, 0.f);
#pragma unroll
    for (int e = 0; e < K; e++) {
      float2 const rot = make_float2(__shfl(y[e].x, src, RB), __shfl(y[e].y, src, RB));
      float2 const prev = lane ? rot : up;
      up = rot;
      float2 const p = cmulc(y[e], prev);
      float a = fast_atan2f(p.y, p.x);                                  // cargf, fm.c:89
      float w = __builtin_amdgcn_sqrtf(p.x * p.x + p.y * p.y) * inv_var; // cabsf(p) / fm_variance, fm.c:91
      if (w > 1) w = 1;
      if (e == 0 && lane == 0) a = 0.f, w = 0.f; // sample 0 needs the block before: kh_fm_finish
      ang[e] = a;
      wgt[e] = w;
      angsum += a;
    }
  }
  angsum = group_sum<RB>(angsum);

  // ---- second transpose: lane l gets samples K l .. K l + K - 1
  float angc[K], wgtc[K];
  wave_sync();
#pragma unroll
  for (int e = 0; e < K; e++) {
    int const n = lane + RB * e;
    ex[n + n / K] = make_float2(ang[e], wgt[e]);
  }
  wave_sync();
#pragma unroll
  for (int j = 0; j < K; j++) {
    float2 const aw = ex[lane * PITCH2 + j];
    angc[j] = aw.x;
    wgtc[j] = aw.y;
  }

  // ---- fm.c:105-111 from a ZERO carry-in: s_n = (s_{n-1} * decay) + x_n; the block's carry-out goes to kh_fm_finish
  const DemodDesc *dd = fm.descs + ch;
  float const gain = dd->fm_gain;
  int const flat = dd->flat;
  float s0[K];
  double carry_out = 0.0;
  if (flat) {
#pragma unroll
    for (int j = 0; j < K; j++) s0[j] = angc[j] * gain; // fm.c:105
  } else {
    double x[K];
    double B = 0.0; // this lane's chunk as the map carry_out = decay^K * carry_in + B
#pragma unroll
    for (int j = 0; j < K; j++) {
      x[j] = (double)angc[j] * .114 * (double)gain * (double)wgtc[j]; // fm.c:110
      B = (B + x[j]) * FM_DECAY;
    }
    // inclusive scan of the maps over the lanes.  Every lane's A is the same constant decay^K, and a lane that takes part
    // in the step at distance o has accumulated exactly o chunks, so its multiplier is decay^(K o): no A travels
    double Bi = B;
    double Apow = FM_DECAY_K8; // decay^(K o), squared from step to step
    static_assert(K == 8, "FM_DECAY_K8 is decay^8");
#pragma unroll
    for (int o = 1; o < RB; o <<= 1) {
      double const Bp = __shfl_up(Bi, o, RB);
      if (lane >= o) Bi = fma(Apow, Bp, Bi);
      Apow *= Apow;
    }
    double cin = __shfl_up(Bi, 1, RB); // carry into this lane's chunk for a zero carry into the block
    if (lane == 0) cin = 0.0;
    carry_out = __shfl(Bi, RB - 1, RB);
#pragma unroll
    for (int j = 0; j < K; j++) {
      cin += x[j];     // samples[n] = lastaudio += ...
      s0[j] = (float)cin;
      cin *= FM_DECAY; // lastaudio *= 0.99376949
    }
  }
  if (!valid) return;
  size_t const rec = (size_t)ch * fm.stat_stride + b;
  float4 *wo = reinterpret_cast<float4 *>(fm.work + (size_t)ch * fm.work_stride + (size_t)b * OLEN + (size_t)lane * K);
#pragma unroll
  for (int j = 0; j < K; j += 2) wo[j / 2] = make_float4(angc[j], s0[j], angc[j + 1], s0[j + 1]);
  if (lane == 0) {
    fm.stat[rec] = FmBlockStat{bb / OLEN, var, snr, angsum, first, last};
    fm.carry[rec] = carry_out;
  }
}


// pairs of one wavefront: RB = 64: pair w; RB = 32: the two halves take blocks 2 j and 2 j + 1 of the same channel
template <int RB>
__device__ __forceinline__ void chan_wave_dispatch(ChanArgs const &A, float2 *ex_wave, int w, int nblocks, int lane64) {
  constexpr int PAIR_LDS = CW_RA * (RB + 1);
  if constexpr (RB == 64) {
    chan_wave64_body<CW_RA, 64>(A, ex_wave, w % A.nchan, w / A.nchan, true, lane64);
  } else {
    int const half = lane64 >> 5;
    int const ch = w % A.nchan, b = 2 * (w / A.nchan) + half;
    chan_wave64_body<CW_RA, 32>(A, ex_wave + half * PAIR_LDS, ch, b < nblocks ? b : nblocks - 1, b < nblocks, lane64 & 31);
  }
}
template <int RB>
__host__ __device__ constexpr int chan_waves(int nchan, int nblocks) { // wavefronts of a launch
  return RB == 64 ? nchan * nblocks : nchan * ((nblocks + 1) / 2);
}

// one wavefront per workgroup (filter.h API path, and banks without compute_n0)
template <int RB>
__global__ __launch_bounds__(64) void kh_chan_wave64(ChanArgs A, int nblocks) {
  __shared__ float2 ex[CW_LDS];
  chan_wave_dispatch<RB>(A, ex, blockIdx.x, nblocks, threadIdx.x);
}

// Horizontal fusion for the bank: workgroups [0, nchan_wg) carry BACK_WAVES channel wavefronts each, the rest are
// compute_n0's second iteration (radio.c:451-462, kh_n0.cuh) over N0_GX slices per (input, block).  The channel
// waves wait on memory latency and use little bandwidth, the n0 workgroups stream the spectrum: in one launch they
// share the chip instead of queueing behind one another.
constexpr int BACK_WAVES = 2; // channel wavefronts per workgroup: their LDS sets how many workgroups of either kind a CU holds
template <int RB>
__global__ __launch_bounds__(64 * BACK_WAVES) void kh_back_a(ChanArgs A, N0Pass1Args Z, int nchan_wg, int nblocks, int nwaves) {
  __shared__ float2 ex[BACK_WAVES * CW_LDS];
  if ((int)blockIdx.x < nchan_wg) {
    int const wave = threadIdx.x >> 6;
    int const w = blockIdx.x * BACK_WAVES + wave;
    if (w < nwaves) chan_wave_dispatch<RB>(A, ex + wave * CW_LDS, w, nblocks, threadIdx.x & 63);
  } else {
    n0_pass1_body(Z, blockIdx.x - nchan_wg, reinterpret_cast<float *>(ex));
  }
}

// 0 if this shape has no wave-resident kernel (the caller falls back to kh_filter_out)
int chan_wave_supported(int ND, int olen) { return (ND == 1024 || ND == 512) && olen * 2 == ND; }

template <int RB>
static void launch_chan_wave_rb(hipStream_t st, ChanArgs const &A, int nblocks, const N0Pass1Args *n0) {
  int const nwaves = chan_waves<RB>(A.nchan, nblocks);
  if (n0) {
    int const nchan_wg = (nwaves + BACK_WAVES - 1) / BACK_WAVES;
    hipLaunchKernelGGL(kh_back_a<RB>, dim3(nchan_wg + n0->gx * n0->nblocks * n0->ninputs), dim3(64 * BACK_WAVES), 0, st, A, *n0,
                       nchan_wg, nblocks, nwaves);
  } else {
    hipLaunchKernelGGL(kh_chan_wave64<RB>, dim3(nwaves), dim3(64), 0, st, A, nblocks);
  }
}

int launch_chan_wave(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, const ChanDesc *chans,
                     const float2 *response, int ND, int olen, float2 *out, size_t out_stride, int nblocks, int nchan,
                     FmFront const &fm, const N0Pass1Args *n0) {
  if (!chan_wave_supported(ND, olen)) {
    set_error("wave-resident filter output: N_dec %d / olen %d unsupported", ND, olen);
    return -1;
  }
  if (nblocks <= 0 || nchan <= 0) return 0;
  const float2 *tw = twiddle_table(ND);
  if (!tw) return -1;
  int const npairs = nchan * nblocks;
  ChanArgs const A{spec, spec_in_stride, N, chans, response, out, out_stride, tw, fm, nchan, npairs};
  if (ND == 1024)
    launch_chan_wave_rb<64>(st, A, nblocks, n0);
  else
    launch_chan_wave_rb<32>(st, A, nblocks, n0);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
