// kh_demod_kernels.hip -- ingest (convert + NCO mix), compute_n0, FM / linear / AM demodulators, scaleclip,
// half-band decimators.
//
// Reference behaviour restated here (citations into the reference tree):
//   proc_samples convert+mix  radio.c:132-204      osc.c:22-58 (set_osc/step_osc/renorm_osc)
//   compute_n0                radio.c:422-465
//   demod_fm                  fm.c:43-139, fm_snr fm.c:152-205
//   demod_linear              linear.c:57-156, run_pll osc.c:91-107
//   demod_am                  am.c:42-91
//   scaleclip                 audio.c:22-28
//   hb15_block / hb3_block    decimate.c:113-163
//
// Mapping: work that is parallel inside a block (discriminator, statistics, PCM) runs one workgroup per
// channel; the per-sample recurrences of linear.c / am.c (PLL, hang AGC, DC tracker) are serial by
// definition and run one LANE per channel, 64 channels in lock-step per wavefront.
#include "kh_internal.h"
#include "kh_fft.cuh"
#include "../../include/ka9q_hip.h"

namespace kh {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ================================================================ ingest: radio.c:136-191
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

// grid (chunks per block, nblocks, ninputs), 256 threads, 4 samples per thread
// algorithmic bytes per sample: wire bytes read + 8 written
__global__ __launch_bounds__(256) void kh_ingest(const unsigned char *__restrict__ raw, size_t raw_stride_bytes,
                                                 int payload, float2 *__restrict__ time, size_t time_stride,
                                                 const IngestDesc *__restrict__ descs, float *__restrict__ block_energy,
                                                 int L, int nblocks) {
  __shared__ float part[4];
  int const b = blockIdx.y, i = blockIdx.z;
  IngestDesc const d = descs[i];
  const unsigned char *src = raw + (size_t)i * raw_stride_bytes;
  float2 *dst = time + (size_t)i * time_stride;
  float energy = 0.f;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    int const k = blockIdx.x * 1024 + u * 256 + threadIdx.x;
    if (k < L) {
      size_t const n = (size_t)b * L + k;
      float2 x = load_raw(src, n, payload, d.gain_factor);
      if (d.doppler.active) { // radio.c:180-184: complex float *= complex double
        double c, s;
        nco_phasor(d.doppler, (unsigned)n, c, s);
        double const re = (double)x.x * c - (double)x.y * s;
        double const im = (double)x.x * s + (double)x.y * c;
        x = make_float2((float)re, (float)im);
      }
      if (d.lo2.active) { // radio.c:187
        double c, s;
        nco_phasor(d.lo2, (unsigned)n, c, s);
        double const re = (double)x.x * c - (double)x.y * s;
        double const im = (double)x.x * s + (double)x.y * c;
        x = make_float2((float)re, (float)im);
      }
      energy += x.x * x.x + x.y * x.y; // radio.c:188
      dst[n] = x;
    }
  }
  energy = wave_sum(energy);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = energy;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&block_energy[i * nblocks + b], part[0] + part[1] + part[2] + part[3]);
}

int launch_ingest(hipStream_t st, const void *raw, size_t raw_stride_bytes, int payload, float2 *time,
                  size_t time_stride, const IngestDesc *desc, float *block_energy, int L, int nblocks, int ninputs) {
  if (nblocks <= 0 || ninputs <= 0) return 0;
  if (!KH_HIP(hipMemsetAsync(block_energy, 0, sizeof(float) * (size_t)nblocks * ninputs, st))) return -1;
  hipLaunchKernelGGL(kh_ingest, dim3((L + 1023) / 1024, nblocks, ninputs), dim3(256), 0, st,
                     static_cast<const unsigned char *>(raw), raw_stride_bytes, payload, time, time_stride, desc,
                     block_energy, L, nblocks);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ compute_n0: radio.c:422-465
// pass 0: sum of |X|^2 over the bins outside the passband; pass 1: sum and count of those below twice that
// mean.  acc holds 3 floats per (input, block): {sum0, sum1, count1}.
// algorithmic bytes: 8*bincount per pass
template <int PASS>
__global__ __launch_bounds__(256) void kh_n0_pass(const float2 *__restrict__ spec, size_t spec_in_stride, int N,
                                                  int start, int bincount, float *__restrict__ acc, int nblocks) {
  __shared__ float ps[4], pc[4];
  int const b = blockIdx.y, i = blockIdx.z;
  const float2 *fd = spec + (size_t)i * spec_in_stride + (size_t)b * N;
  float *a = acc + 3 * ((size_t)i * nblocks + b);
  float thresh = INFINITY;
  if (PASS == 1) thresh = (a[0] / bincount) * 2; // radio.c:456,461
  float s = 0.f, c = 0.f;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < bincount; n += gridDim.x * 256) {
    int in = start + n; // radio.c:440-446
    if (in >= N) in -= N;
    float2 const x = fd[in];
    float const p = x.x * x.x + x.y * x.y;
    if (p < thresh) {
      s += p;
      c += 1.f;
    }
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    ps[threadIdx.x >> 6] = s;
    pc[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (PASS == 0)
      atomicAdd(&a[0], ps[0] + ps[1] + ps[2] + ps[3]);
    else {
      atomicAdd(&a[1], ps[0] + ps[1] + ps[2] + ps[3]);
      atomicAdd(&a[2], pc[0] + pc[1] + pc[2] + pc[3]);
    }
  }
}

// one thread per input: finish compute_n0 and apply radio.c:197-202 block by block
__global__ void kh_n0_final(const float *__restrict__ acc, const float *__restrict__ block_energy, int L, int N,
                            int samprate, int bincount, int valid, float *__restrict__ n0_state,
                            kh_input_status *__restrict__ status, int nblocks, int ninputs) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ninputs) return;
  float n0 = n0_state[i];
  for (int b = 0; b < nblocks; b++) {
    const float *a = acc + 3 * ((size_t)i * nblocks + b);
    float v = 0.f; // "Can this happen?" radio.c:433-435
    if (valid) v = (a[1] / a[2]) / ((float)N * samprate);
    if (!isnan(n0))
      n0 = (float)((double)n0 + .005 * (double)(v - n0));
    else
      n0 = v;
    status[(size_t)i * nblocks + b].if_power = block_energy[i * nblocks + b] / L;
    status[(size_t)i * nblocks + b].n0 = n0;
  }
  n0_state[i] = n0;
}
__global__ void kh_ifpower_only(const float *__restrict__ block_energy, int L, kh_input_status *__restrict__ status,
                                int total) {
  int const k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= total) return;
  status[k].if_power = block_energy[k] / L;
  status[k].n0 = NAN;
}

int launch_n0(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int samprate, float low, float high,
              const float *block_energy, int L, float *n0_state, float *acc, void *input_status, int nblocks,
              int ninputs) {
  if (nblocks <= 0 || ninputs <= 0) return 0;
  kh_input_status *status = static_cast<kh_input_status *>(input_status);
  if (spec == nullptr) { // n0 disabled: only if_power
    int const total = nblocks * ninputs;
    hipLaunchKernelGGL(kh_ifpower_only, dim3((total + 63) / 64), dim3(64), 0, st, block_energy, L, status, total);
    return KH_HIP(hipGetLastError()) ? 0 : -1;
  }
  int const low_n = (int)((N * low) / samprate); // radio.c:430-431: float arithmetic, truncated
  int const high_n = (int)((N * high) / samprate);
  int const bincount = N - (high_n - low_n);
  int const valid = !(bincount < 0 || bincount > N);
  int start = high_n + 1;
  if (start >= N)
    start -= N;
  else if (start < 0)
    start += N;
  if (!KH_HIP(hipMemsetAsync(acc, 0, sizeof(float) * 3 * (size_t)nblocks * ninputs, st))) return -1;
  if (valid && bincount > 0) {
    int gx = (bincount + 4095) / 4096;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(kh_n0_pass<0>, dim3(gx, nblocks, ninputs), dim3(256), 0, st, spec, spec_in_stride, N, start,
                       bincount, acc, nblocks);
    hipLaunchKernelGGL(kh_n0_pass<1>, dim3(gx, nblocks, ninputs), dim3(256), 0, st, spec, spec_in_stride, N, start,
                       bincount, acc, nblocks);
  }
  hipLaunchKernelGGL(kh_n0_final, dim3((ninputs + 63) / 64), dim3(64), 0, st, acc, block_energy, L, N, samprate,
                     bincount, valid, n0_state, status, nblocks, ninputs);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ scaleclip: audio.c:22-28
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

// ================================================================ FM: fm.c
__device__ double fm_i0(double z) { // fm.c:152-163
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
__device__ double fm_i1(double z) { // fm.c:166-179
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
__device__ double fm_xi(double thetasq) { // fm.c:180-186
  double t = (2 + thetasq) * fm_i0(0.25 * thetasq) + thetasq * fm_i1(0.25 * thetasq);
  t *= t;
  return 2 + thetasq - (0.125 * M_PI) * exp(-0.5 * thetasq) * t;
}
__device__ double fm_snr(double r) { // fm.c:190-205
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

constexpr int FM_NT = 512;
constexpr int FM_MAXS = 8192; // samples per group held in LDS
constexpr int FM_MAXG = 64;   // blocks per group
#define FM_SK(n) ((n) + ((n) >> 4))

// one workgroup per FM channel; blocks are processed in groups of G = min(FM_MAXG, FM_MAXS/olen)
__global__ __launch_bounds__(FM_NT) void kh_demod_fm(const float2 *__restrict__ baseband, size_t bb_stride, int olen,
                                                     int nblocks, const DemodDesc *__restrict__ descs,
                                                     DemodState *__restrict__ states, float *__restrict__ audio,
                                                     int16_t *__restrict__ pcm, size_t pcm_stride, int big_endian,
                                                     kh_status *__restrict__ status, int status_stride,
                                                     const int *__restrict__ list) {
  __shared__ float s_ang[FM_SK(FM_MAXS) + 1];
  __shared__ float s_aud[FM_SK(FM_MAXS) + 1]; // amplitude, then |p|/var weight, then audio
  __shared__ float b_bb[FM_MAXG], b_avg[FM_MAXG], b_var[FM_MAXG], b_snr[FM_MAXG], b_lvl[FM_MAXG];
  __shared__ float b_avgf[FM_MAXG], b_pdev[FM_MAXG], b_carry[FM_MAXG];
  __shared__ int b_open[FM_MAXG], b_sq[FM_MAXG];
  __shared__ float2 b_prev[FM_MAXG];
  __shared__ double t_A[FM_NT], t_B[FM_NT], t_cin[FM_NT];
  __shared__ float sh_last;

  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int const ch = list[blockIdx.x];
  DemodDesc const d = descs[ch];
  DemodState *st = &states[ch];
  const float2 *y = baseband + (size_t)ch * bb_stride;
  float const gain = d.fm_gain;
  double const decay = 0.99376949; // fm.c:111,130

  int G = FM_MAXS / olen;
  if (G > FM_MAXG) G = FM_MAXG;
  if (G < 1) G = 1; // guarded by the launcher: olen <= FM_MAXS

  float2 state = st->fm_state;
  float lastaudio = st->fm_lastaudio;
  int squelch_open = st->fm_squelch_open;

  for (int g0 = 0; g0 < nblocks; g0 += G) {
    int const gb = (nblocks - g0 < G) ? nblocks - g0 : G;
    int const ns = gb * olen;
    const float2 *yg = y + (size_t)g0 * olen;

    // ---- amplitude statistics, fm.c:53-67: one wave per block
    for (int b = wave; b < gb; b += FM_NT / 64) {
      float bb = 0.f, sa = 0.f;
      for (int n = lane; n < olen; n += 64) {
        float2 const v = yg[b * olen + n];
        float const t = v.x * v.x + v.y * v.y;
        float const a = sqrtf(t);
        s_aud[FM_SK(b * olen + n)] = a;
        bb += t;
        sa += a;
      }
      bb = wave_sum(bb);
      sa = wave_sum(sa);
      float const avg = sa / olen;
      float var = 0.f;
      for (int n = lane; n < olen; n += 64) {
        float const a = s_aud[FM_SK(b * olen + n)];
        var += (a - avg) * (a - avg);
      }
      var = wave_sum(var) / (olen - 1);
      if (lane == 0) {
        b_bb[b] = bb / olen;
        b_avg[b] = avg;
        b_var[b] = var;
      }
    }
    __syncthreads();
    // ---- Rice-corrected SNR, fm.c:69-70: one lane per block
    if (tid < gb) {
      float const avg = b_avg[tid];
      float snr = (float)fm_snr((double)(avg * avg / b_var[tid]));
      b_snr[tid] = fmaxf(0.0f, snr);
    }
    __syncthreads();
    // ---- squelch, fm.c:73-78,123: serial over blocks
    if (tid == 0) {
      float2 prev = state;
      for (int b = 0; b < gb; b++) {
        if (b_snr[b] > 4.f) squelch_open = 2;
        int open = 0;
        if (squelch_open > 0) {
          squelch_open--;
          open = 1;
        }
        b_open[b] = open;
        b_sq[b] = squelch_open;
        b_prev[b] = prev; // `state` on entry to the block
        prev = open ? yg[b * olen + olen - 1] : make_float2(0.f, 0.f);
      }
      state = prev;
    }
    __syncthreads();
    // ---- discriminator, fm.c:85-93
    for (int n = tid; n < ns; n += FM_NT) {
      int const b = n / olen, k = n - b * olen;
      float ang = 0.f, amp = 0.f;
      if (b_open[b]) {
        float2 const cur = yg[n];
        float2 const prev = (k == 0) ? b_prev[b] : yg[n - 1];
        float2 const p = cmulc(cur, prev);
        ang = atan2f(p.y, p.x);     // cargf
        amp = hypotf(p.x, p.y) / b_var[b]; // cabsf(p) / fm_variance
        if (amp > 1) amp = 1;
      }
      s_ang[FM_SK(n)] = ang;
      s_aud[FM_SK(n)] = amp;
    }
    __syncthreads();
    // ---- de-emphasis integrator, fm.c:104-112,124-132.  s_n = c_{n-1} + x_n ; c_n = s_n * decay, a linear
    // recurrence: per-thread chunk maps composed serially by one lane, then each chunk replayed from its
    // carry-in with the reference's float roundings.
    int const CH = (ns + FM_NT - 1) / FM_NT;
    int const n0 = tid * CH;
    int const n1 = (n0 + CH < ns) ? n0 + CH : ns;
    if (!d.flat) {
      double A = 1.0, B = 0.0;
      for (int n = n0; n < n1; n++) {
        int const b = n / olen;
        double x = 0.0;
        if (b_open[b]) x = (double)s_ang[FM_SK(n)] * .114 * (double)gain * (double)s_aud[FM_SK(n)];
        B = (B + x) * decay;
        A *= decay;
      }
      t_A[tid] = A;
      t_B[tid] = B;
      __syncthreads();
      if (tid == 0) {
        double c = (double)lastaudio;
        for (int t = 0; t < FM_NT; t++) {
          t_cin[t] = c;
          c = t_A[t] * c + t_B[t];
        }
      }
      __syncthreads();
      float c = (float)t_cin[tid];
      for (int n = n0; n < n1; n++) {
        int const b = n / olen;
        if (n - b * olen == 0) b_carry[b] = c; // lastaudio on entry to the block
        float s;
        if (b_open[b]) {
          double const x = (double)s_ang[FM_SK(n)] * .114 * (double)gain * (double)s_aud[FM_SK(n)];
          s = (float)((double)c + x); // samples[n] = lastaudio += ...
        } else {
          s = c; // samples[n] = lastaudio
        }
        c = (float)((double)s * decay);
        s_aud[FM_SK(n)] = s;
      }
      if (n1 == ns && n0 < ns) sh_last = c;
    } else {
      for (int n = n0; n < n1; n++) {
        int const b = n / olen;
        if (n - b * olen == 0) b_carry[b] = lastaudio;
        s_aud[FM_SK(n)] = b_open[b] ? s_ang[FM_SK(n)] * gain : 0.f; // fm.c:105,126
      }
      if (tid == 0) sh_last = lastaudio;
    }
    __syncthreads();
    lastaudio = sh_last;
    // ---- frequency offset and peak deviation, fm.c:81-83,96-102,114-121: serial per block, one lane per block
    if (tid < gb && b_open[tid]) {
      int const b = tid;
      float pdev_pos = 0, pdev_neg = 0, avg_f = 0;
      float la = b_carry[b];
      for (int k = 0; k < olen; k++) {
        float const ang = s_ang[FM_SK(b * olen + k)];
        avg_f += ang;
        if (k == 0)
          pdev_pos = pdev_neg = ang;
        else if (ang > pdev_pos)
          pdev_pos = ang;
        else if (la < pdev_neg)
          pdev_neg = ang;
        if (!d.flat) la = (float)((double)s_aud[FM_SK(b * olen + k)] * decay);
      }
      avg_f /= olen;
      b_avgf[b] = d.dsamprate * avg_f * (float)(0.5 * M_1_PI);
      pdev_pos -= avg_f;
      pdev_neg -= avg_f;
      b_pdev[b] = d.dsamprate * fmaxf(pdev_pos, -pdev_neg) * (float)(0.5 * M_1_PI);
    }
    // ---- output level (fm.c:135-138), audio and PCM out
    for (int b = wave; b < gb; b += FM_NT / 64) {
      float lv = 0.f;
      for (int n = lane; n < olen; n += 64) {
        float const s = s_aud[FM_SK(b * olen + n)];
        lv += s * s;
      }
      lv = wave_sum(lv);
      if (lane == 0) b_lvl[b] = lv / olen;
    }
    size_t const obase = (size_t)g0 * olen;
    if (audio) {
      float *ao = audio + (size_t)ch * pcm_stride + obase;
      for (int n = tid; n < ns; n += FM_NT) ao[n] = s_aud[FM_SK(n)];
    }
    int16_t *po = pcm + (size_t)ch * pcm_stride + obase;
    for (int n = tid; n < ns; n += FM_NT) po[n] = pcm_word(s_aud[FM_SK(n)], big_endian);
    __syncthreads();
    if (tid < gb) {
      kh_status *so = status + (size_t)ch * status_stride + g0 + tid;
      kh_status r;
      r.bb_power = b_bb[tid];
      r.snr = b_snr[tid];
      // foffset / pdeviation keep their previous values while the squelch is closed (fm.c:116,121)
      if (b_open[tid]) {
        r.foffset = b_avgf[tid];
        r.pdeviation = b_pdev[tid];
      } else {
        r.foffset = NAN;
        r.pdeviation = NAN;
      }
      r.cphase = 0.f;
      r.lock_timer = 0.f;
      r.pll_lock = 0;
      r.output_level = b_lvl[tid];
      r.agc_gain = NAN;
      r.squelch_open = b_sq[tid];
      r.reserved[0] = r.reserved[1] = 0.f;
      *so = r;
    }
    __syncthreads();
  }
  if (tid == 0) {
    st->fm_state = state;
    st->fm_lastaudio = lastaudio;
    st->fm_squelch_open = squelch_open;
  }
}

// ================================================================ oscillator recurrence, osc.c:39-58 (serial users)
struct DOsc {
  double re, im, sre, sim;
  int steps;
};
__device__ __forceinline__ void dosc_step(DOsc &o) { // step_osc with rate == 0
  double const nre = o.re * o.sre - o.im * o.sim;
  double const nim = o.re * o.sim + o.im * o.sre;
  o.re = nre;
  o.im = nim;
  if (++o.steps == 16384) { // Renorm_rate, osc.c:11,47-48,52-55
    o.steps = 0;
    double const m = hypot(o.re, o.im);
    o.re /= m;
    o.im /= m;
  }
}

// ================================================================ linear: linear.c:57-156, one lane per channel
__global__ __launch_bounds__(64) void kh_demod_linear(float2 *__restrict__ baseband, size_t bb_stride, int olen,
                                                      int nblocks, const DemodDesc *__restrict__ descs,
                                                      DemodState *__restrict__ states, float *__restrict__ audio,
                                                      int16_t *__restrict__ pcm, size_t pcm_stride, int big_endian,
                                                      kh_status *__restrict__ status, int status_stride,
                                                      const int *__restrict__ list, int count) {
  int const k = blockIdx.x * 64 + threadIdx.x;
  if (k >= count) return;
  int const ch = list[k];
  DemodDesc const d = descs[ch];
  DemodState s = states[ch];
  float2 *y = baseband + (size_t)ch * bb_stride;
  float *ao = audio ? audio + (size_t)ch * pcm_stride : nullptr;
  int16_t *po = pcm + (size_t)ch * pcm_stride;

  DOsc shift;
  shift.re = s.shift_re;
  shift.im = s.shift_im;
  shift.steps = s.shift_steps;
  sincos(2 * d.shift_freq * M_PI, &shift.sim, &shift.sre); // set_osc, osc.c:30 via radio.c:359
  DOsc vco;
  vco.re = s.vco_re;
  vco.im = s.vco_im;
  vco.steps = s.vco_steps;
  vco.sre = 1;
  vco.sim = 0;
  float const snrthresh = 2; // linear.c:32

  for (int b = 0; b < nblocks; b++) {
    float2 *yb = y + (size_t)b * olen;
    kh_status r;
    r.pdeviation = 0.f;
    r.cphase = 0.f;
    r.foffset = 0.f;
    r.lock_timer = 0.f;
    r.squelch_open = 0;
    r.reserved[0] = r.reserved[1] = 0.f;
    if (d.pll) {
      if (s.snr < snrthresh) // linear.c:64-68
        s.lock_count -= olen;
      else
        s.lock_count += olen;
      if (s.lock_count >= d.lock_limit) {
        s.lock_count = d.lock_limit;
        s.pll_lock = 1;
      }
      if (s.lock_count <= -d.lock_limit) {
        s.lock_count = -d.lock_limit;
        s.pll_lock = 0;
      }
      r.lock_timer = (float)s.lock_count;
      float signal = 0, noise = 0;
      for (int n = 0; n < olen; n++) { // linear.c:81-97
        float2 v = yb[n];
        float2 const ph = make_float2((float)vco.re, (float)vco.im);
        v = cmulc(v, ph); // s *= conjf(pll.vco.phasor)
        float phase;
        if (d.square)
          phase = atan2f(2 * v.x * v.y, v.x * v.x - v.y * v.y); // cargf(s*s)
        else
          phase = atan2f(v.y, v.x);
        // run_pll, osc.c:91-107
        float feedback = d.pll_integrator_gain * s.pll_integrator + d.pll_prop_gain * phase;
        s.pll_integrator += phase;
        feedback = (double)feedback > 0.49 ? (float)0.49 : (double)feedback < -0.49 ? (float)-0.49 : feedback;
        // set_osc(&pll->vco, feedback, 0), osc.c:22-36
        if (isnan(vco.re) || isnan(vco.im) || vco.re * vco.re + vco.im * vco.im < 0.9) {
          vco.re = 1;
          vco.im = 0;
          vco.steps = 0;
        }
        s.vco_freq = (double)feedback;
        sincos(2 * s.vco_freq * M_PI, &vco.sim, &vco.sre);
        dosc_step(vco);
        yb[n] = v; // linear.c:91
        signal += v.x * v.x;
        noise += v.y * v.y;
      }
      float cphase = (float)atan2(vco.im, vco.re); // linear.c:98
      if (d.square) cphase /= 2;
      r.cphase = cphase;
      r.foffset = (float)(s.vco_freq * (double)d.out_samprate);
      if (noise != 0) {
        s.snr = (signal / noise) - 1;
        if (s.snr < 0) s.snr = 0;
      } else
        s.snr = NAN;
    }
    float energy = 0, output_level = 0;
    for (int n = 0; n < olen; n++) { // linear.c:114-148
      float2 const v = yb[n];
      // complex float * complex double -> double product, rounded to float
      float2 const sm = make_float2((float)((double)v.x * shift.re - (double)v.y * shift.im),
                                    (float)((double)v.x * shift.im + (double)v.y * shift.re));
      dosc_step(shift);
      float const norm = sm.x * sm.x + sm.y * sm.y;
      energy += norm;
      float const amplitude = sqrtf(norm);
      if (d.agc) { // linear.c:127-136
        if (isnan(s.agc_gain) || amplitude * s.agc_gain > d.headroom) {
          s.agc_gain = d.headroom / amplitude;
          s.hangcount = (int)d.hangtime;
        } else if (s.hangcount > 0) {
          s.hangcount--;
        } else {
          s.agc_gain *= d.recovery_rate;
        }
      }
      size_t const o = (size_t)b * olen + n;
      if (d.env) {
        float const smp = amplitude * s.agc_gain;
        output_level += smp * smp;
        if (ao) ao[o] = smp;
        po[o] = pcm_word(smp, big_endian);
      } else if (d.channels == 1) {
        float const smp = sm.x * s.agc_gain;
        output_level += smp * smp;
        if (ao) ao[o] = smp;
        po[o] = pcm_word(smp, big_endian);
      } else {
        float2 const w = make_float2(sm.x * s.agc_gain, sm.y * s.agc_gain);
        yb[n] = w; // linear.c:145
        output_level += w.x * w.x + w.y * w.y;
        if (ao) {
          ao[2 * o] = w.x;
          ao[2 * o + 1] = w.y;
        }
        po[2 * o] = pcm_word(w.x, big_endian);
        po[2 * o + 1] = pcm_word(w.y, big_endian);
      }
    }
    r.output_level = output_level / (olen * d.channels);
    r.bb_power = energy / olen;
    r.snr = s.snr;
    r.pll_lock = s.pll_lock;
    r.agc_gain = s.agc_gain;
    status[(size_t)ch * status_stride + b] = r;
  }
  s.shift_re = shift.re;
  s.shift_im = shift.im;
  s.shift_steps = shift.steps;
  s.vco_re = vco.re;
  s.vco_im = vco.im;
  s.vco_steps = vco.steps;
  states[ch] = s;
}

// ================================================================ AM: am.c:42-91, one lane per channel
__global__ __launch_bounds__(64) void kh_demod_am(const float2 *__restrict__ baseband, size_t bb_stride, int olen,
                                                  int nblocks, const DemodDesc *__restrict__ descs,
                                                  DemodState *__restrict__ states, float *__restrict__ audio,
                                                  int16_t *__restrict__ pcm, size_t pcm_stride, int big_endian,
                                                  kh_status *__restrict__ status, int status_stride,
                                                  const int *__restrict__ list, int count) {
  int const k = blockIdx.x * 64 + threadIdx.x;
  if (k >= count) return;
  int const ch = list[k];
  DemodDesc const d = descs[ch];
  DemodState s = states[ch];
  const float2 *y = baseband + (size_t)ch * bb_stride;
  float *ao = audio ? audio + (size_t)ch * pcm_stride : nullptr;
  int16_t *po = pcm + (size_t)ch * pcm_stride;
  float const DC_filter_coeff = .0001f; // am.c:36
  for (int b = 0; b < nblocks; b++) {
    float signal = 0, level = 0;
    for (int n = 0; n < olen; n++) {
      size_t const o = (size_t)b * olen + n;
      float2 const v = y[o];
      float const sampsq = v.x * v.x + v.y * v.y;
      signal += sampsq;
      float const samp = sqrtf(sampsq);
      s.am_dc += DC_filter_coeff * (samp - s.am_dc); // am.c:74
      if (isnan(s.agc_gain)) {                       // am.c:76-85
        s.agc_gain = d.headroom / s.am_dc;
      } else if (s.agc_gain * s.am_dc > d.headroom) {
        s.agc_gain = d.headroom / s.am_dc;
        s.am_hangcount = (int)d.am_hang_samples;
      } else if (s.am_hangcount != 0) {
        s.am_hangcount--;
      } else {
        s.agc_gain *= s.am_recovery_per_sample;
      }
      float const smp = (samp - s.am_dc) * s.agc_gain;
      level += smp * smp;
      if (ao) ao[o] = smp;
      po[o] = pcm_word(smp, big_endian);
    }
    kh_status r;
    r.bb_power = signal / olen; // am.c:90
    r.snr = NAN;
    r.foffset = 0.f;
    r.pdeviation = 0.f;
    r.cphase = 0.f;
    r.lock_timer = 0.f;
    r.pll_lock = 0;
    r.output_level = level / olen;
    r.agc_gain = s.agc_gain;
    r.squelch_open = 0;
    r.reserved[0] = r.reserved[1] = 0.f;
    status[(size_t)ch * status_stride + b] = r;
  }
  states[ch] = s;
}

int launch_demod(hipStream_t st, float2 *baseband, size_t bb_stride, int olen, int nblocks, int nchan,
                 const DemodDesc *desc, DemodState *state, float *audio, int16_t *pcm, size_t pcm_stride,
                 int pcm_big_endian, void *status, int status_stride, const int *fm_list, int n_fm, const int *lin_list,
                 int n_lin, const int *am_list, int n_am) {
  (void)nchan;
  if (nblocks <= 0) return 0;
  kh_status *stt = static_cast<kh_status *>(status);
  if (n_fm > 0) {
    if (olen > FM_MAXS || olen < 2) {
      set_error("FM demodulator: olen %d outside [2, %d]", olen, FM_MAXS);
      return -1;
    }
    hipLaunchKernelGGL(kh_demod_fm, dim3(n_fm), dim3(FM_NT), 0, st, baseband, bb_stride, olen, nblocks, desc, state,
                       audio, pcm, pcm_stride, pcm_big_endian, stt, status_stride, fm_list);
  }
  if (n_lin > 0)
    hipLaunchKernelGGL(kh_demod_linear, dim3((n_lin + 63) / 64), dim3(64), 0, st, baseband, bb_stride, olen, nblocks,
                       desc, state, audio, pcm, pcm_stride, pcm_big_endian, stt, status_stride, lin_list, n_lin);
  if (n_am > 0)
    hipLaunchKernelGGL(kh_demod_am, dim3((n_am + 63) / 64), dim3(64), 0, st, baseband, bb_stride, olen, nblocks, desc,
                       state, audio, pcm, pcm_stride, pcm_big_endian, stt, status_stride, am_list, n_am);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ decimate.c
// y[k] = x[2k-6] + c0(x[2k+1]+x[2k-13]) + c1(x[2k-1]+x[2k-11]) + c2(x[2k-3]+x[2k-9]) + c3(x[2k-5]+x[2k-7])
// (decimate.c:124-144 unrolled); x[-1..-14] come from the caller's struct hb15_state.
struct Hb15Dev {
  float coeffs[4], even_samples[4], odd_samples[4], old_odd_samples[4];
};
__device__ __forceinline__ float hb15_hist(const Hb15Dev &s, int idx) { // idx in [-14, -1]
  int const j = -idx; // 1..14
  if (j & 1) {        // odd sample o_{-m}, x[-2m+1]: j = 2m-1
    int const m = (j + 1) / 2; // 1..7
    return m <= 3 ? s.odd_samples[m] : s.old_odd_samples[7 - m];
  }
  int const m = j / 2; // even sample e_{-m}, m = 1..7; only m <= 3 is ever needed
  return m <= 3 ? s.even_samples[m] : 0.f;
}
// grid (ceil(cnt/256), nrails); in: 2*cnt floats per rail, out: cnt floats per rail
__global__ __launch_bounds__(256) void kh_hb15(const float *__restrict__ in, float *__restrict__ out, size_t rail_stride,
                                               int cnt, const Hb15Dev *__restrict__ states, int state_stride,
                                               int state_index) {
  int const r = blockIdx.y;
  int const k = blockIdx.x * 256 + threadIdx.x;
  if (k >= cnt) return;
  const float *x = in + (size_t)r * rail_stride;
  Hb15Dev const s = states[r * state_stride + state_index];
  auto X = [&](int idx) -> float { return idx >= 0 ? x[idx] : hb15_hist(s, idx); };
  float result = X(2 * k - 6);
  result += (X(2 * k + 1) + X(2 * k - 13)) * s.coeffs[0];
  result += (X(2 * k - 1) + X(2 * k - 11)) * s.coeffs[1];
  result += (X(2 * k - 3) + X(2 * k - 9)) * s.coeffs[2];
  result += (X(2 * k - 5) + X(2 * k - 7)) * s.coeffs[3];
  out[(size_t)r * rail_stride + k] = result;
}
// state after cnt iterations (decimate.c:145-147), one thread per rail
__global__ void kh_hb15_state(const float *__restrict__ in, size_t rail_stride, int cnt, Hb15Dev *__restrict__ states,
                              int state_stride, int state_index, int nrails) {
  int const r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrails || cnt <= 0) return;
  const float *x = in + (size_t)r * rail_stride;
  Hb15Dev const s = states[r * state_stride + state_index];
  auto X = [&](int idx) -> float { return idx >= 0 ? x[idx] : hb15_hist(s, idx); };
  auto E = [&](int j) -> float { return X(2 * j); };     // e_j
  auto O = [&](int j) -> float { return X(2 * j + 1); }; // o_j
  Hb15Dev n = s;
  n.even_samples[0] = E(cnt - 1);
  n.even_samples[1] = E(cnt - 1);
  n.even_samples[2] = E(cnt - 2);
  n.even_samples[3] = E(cnt - 3);
  n.odd_samples[0] = O(cnt - 1);
  n.odd_samples[1] = O(cnt - 1);
  n.odd_samples[2] = O(cnt - 2);
  n.odd_samples[3] = O(cnt - 3);
  n.old_odd_samples[3] = O(cnt - 4);
  n.old_odd_samples[2] = O(cnt - 5);
  n.old_odd_samples[1] = O(cnt - 6);
  n.old_odd_samples[0] = O(cnt - 7);
  states[r * state_stride + state_index] = n;
}
// y[k] = 2 x[2k] + x[2k+1] + x[2k-1] (decimate.c:150-163)
__global__ __launch_bounds__(256) void kh_hb3(const float *__restrict__ in, float *__restrict__ out, size_t rail_stride,
                                              int cnt, const float *__restrict__ states, int state_stride,
                                              int state_index) {
  int const r = blockIdx.y;
  int const k = blockIdx.x * 256 + threadIdx.x;
  if (k >= cnt) return;
  const float *x = in + (size_t)r * rail_stride;
  float const old = k > 0 ? x[2 * k - 1] : states[r * state_stride + state_index];
  out[(size_t)r * rail_stride + k] = 2 * x[2 * k] + x[2 * k + 1] + old;
}
__global__ void kh_hb3_state(const float *__restrict__ in, size_t rail_stride, int cnt, float *__restrict__ states,
                             int state_stride, int state_index, int nrails) {
  int const r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrails || cnt <= 0) return;
  states[r * state_stride + state_index] = in[(size_t)r * rail_stride + 2 * cnt - 1];
}

// rails: two device buffers of nrails*rail_stride floats are needed (ping-pong); `rails` holds the input in its
// first half [0, nrails*rail_stride) and scratch in the second.  The result ends in the first half.
int launch_decimate(hipStream_t st, float *rails, size_t rail_stride, int nrails, int chunk, int log_decimate,
                    int stage_threshold, void *hb15_states, float *hb3_states) {
  if (nrails <= 0 || chunk <= 0 || log_decimate <= 0) return 0;
  float *a = rails, *b = rails + (size_t)nrails * rail_stride;
  for (int j = log_decimate - 1; j >= 0; j--) { // hackrf.c:497-501
    int const cnt = chunk << j;
    dim3 const grid((cnt + 255) / 256, nrails);
    if (j >= stage_threshold) {
      hipLaunchKernelGGL(kh_hb3, grid, dim3(256), 0, st, a, b, rail_stride, cnt, hb3_states, log_decimate, j);
      hipLaunchKernelGGL(kh_hb3_state, dim3((nrails + 63) / 64), dim3(64), 0, st, a, rail_stride, cnt, hb3_states,
                         log_decimate, j, nrails);
    } else {
      hipLaunchKernelGGL(kh_hb15, grid, dim3(256), 0, st, a, b, rail_stride, cnt,
                         static_cast<const Hb15Dev *>(hb15_states), log_decimate, j);
      hipLaunchKernelGGL(kh_hb15_state, dim3((nrails + 63) / 64), dim3(64), 0, st, a, rail_stride, cnt,
                         static_cast<Hb15Dev *>(hb15_states), log_decimate, j, nrails);
    }
    float *t = a;
    a = b;
    b = t;
  }
  if (a != rails) { // odd number of stages: bring the result home
    for (int r = 0; r < nrails; r++)
      if (!KH_HIP(hipMemcpyAsync(rails + (size_t)r * rail_stride, a + (size_t)r * rail_stride, sizeof(float) * chunk,
                                 hipMemcpyDeviceToDevice, st)))
        return -1;
  }
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
