// kh_packet_kernels.hip -- packet.c's receive side for many PCM sessions at once:
//   kh_afsk_ingest   packet.c:340-348   network-order int16 -> float (with ntohs's unsigned result), into the REAL
//                                       overlap-save time buffer of every session
//   kh_afsk_decode   packet.c:437-545   mark/space correlators, Gardner timing, NRZI, HDLC unstuffing, crc_good
//                                       (ax25.c:140-156); one lane per session
//   kh_tones_corr    pl.c:248-269       55 CTCSS + 8 DTMF tone correlators over one 50 ms interval per workgroup
//   kh_tones_decide  pl.c:360-439       process_pl / process_dtmf per session
// The REAL -> COMPLEX analytic filter between the first two (packet.c:325,407-408: L = 1000, M = 1049, N = 2048, decimate 1,
// 100 Hz .. 4 kHz, beta 3) is the library's own forward transform and filter-output kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kh_internal.h"
#include "../../include/ka9q_hip.h"

namespace kh {

// algorithmic bytes: 2 in + 4 out per sample
__global__ __launch_bounds__(256) void kh_afsk_ingest(const uint16_t *__restrict__ pcm, size_t pcm_stride,
                                                      float *__restrict__ time, size_t time_stride, int hist, int count) {
  int const s = blockIdx.y;
  int const n = blockIdx.x * 256 + threadIdx.x;
  if (n >= count) return;
  uint16_t const w = pcm[(size_t)s * pcm_stride + n];
  uint16_t const host = (uint16_t)((w << 8) | (w >> 8)); // ntohs: UNSIGNED, so a negative sample arrives as s + 65536
  time[(size_t)s * time_stride + hist + n] = (float)host * (1.f / 32768); // SCALE, packet.c:43
}

__device__ __forceinline__ int crc_good_dev(const unsigned char *frame, int length) { // ax25.c:140-156
  unsigned crc = 0xffff;
  for (int k = 0; k < length; k++) {
    unsigned byte = frame[k];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      unsigned const feedback = ((crc ^ byte) & 1) ? 0x8408u : 0u;
      crc = (crc >> 1) ^ feedback;
      byte >>= 1;
    }
  }
  return crc == 0xf0b8;
}

struct DOscP { // step_osc with rate == 0 (osc.c:39-50), phasor and step in fp64
  double re, im, sre, sim;
  int steps;
};
__device__ __forceinline__ void oscp_step(DOscP &o) {
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
__device__ __forceinline__ float2 mul_cf_cd(float2 y, DOscP const &o) { // complex float x complex double -> complex float
  double const yr = y.x, yi = y.y;
  return make_float2((float)(yr * o.re - yi * o.im), (float)(yr * o.im + yi * o.re));
}
__device__ __forceinline__ float nrm(float2 v) { return v.x * v.x + v.y * v.y; } // cnrmf, dsp.h:65

// One lane per session.  y: [nsess][y_stride] analytic signal, nsamp samples of this launch.  frames/bits are
// per-session arrays of max_frames / max_bits records; their fill counts start at zero in every launch.
__global__ __launch_bounds__(64) void kh_afsk_decode(const float2 *__restrict__ y, size_t y_stride, int nsamp, int nsess,
                                                     AfskState *__restrict__ states, kh_ax25_frame *__restrict__ frames,
                                                     int max_frames, int *__restrict__ nframes,
                                                     kh_afsk_bit *__restrict__ bits, int max_bits, int *__restrict__ nbits,
                                                     double mark_sre, double mark_sim, double space_sre,
                                                     double space_sim, unsigned long long sample_base) {
  int const s = blockIdx.x * 64 + threadIdx.x;
  if (s >= nsess) return;
  constexpr int Samppbit = 40; // packet.c:52
  AfskState *st = states + s;
  unsigned char *hdlc = st->hdlc_frame;
  DOscP mark{st->mark_re, st->mark_im, mark_sre, mark_sim, st->mark_steps};
  DOscP space{st->space_re, st->space_im, space_sre, space_sim, st->space_steps};
  int symphase = st->symphase;
  float2 mark_acc = st->mark_acc, space_acc = st->space_acc, mark_off = st->mark_off, space_off = st->space_off;
  float last_val = st->last_val, mid_val = st->mid_val;
  int frame_bit = st->frame_bit, flagsync = st->flagsync, ones = st->ones;
  unsigned out_ts = st->out_timestamp, out_seq = st->out_seq;
  unsigned long long out_packets = st->out_packets, out_bytes = st->out_bytes;
  const float2 *ys = y + (size_t)s * y_stride;
  kh_ax25_frame *fo = frames + (size_t)s * max_frames;
  kh_afsk_bit *bo = bits ? bits + (size_t)s * max_bits : nullptr;
  int nf = 0, nb = 0;
  // samples are fetched eight at a time, one group ahead: each lane walks its own row, and a load per sample would
  // put a memory round trip into every step of the recurrence
  constexpr int G = 8;
  float2 nxt[G];
#pragma unroll
  for (int u = 0; u < G; u++) nxt[u] = u < nsamp ? ys[u] : make_float2(0.f, 0.f);
  for (int n0 = 0; n0 < nsamp; n0 += G) {
  float2 cur[G];
#pragma unroll
  for (int u = 0; u < G; u++) cur[u] = nxt[u];
#pragma unroll
  for (int u = 0; u < G; u++) nxt[u] = n0 + G + u < nsamp ? ys[n0 + G + u] : make_float2(0.f, 0.f);
#pragma unroll
  for (int u = 0; u < G; u++) { // packet.c:440-544
    int const n = n0 + u;
    if (n >= nsamp) break;
    float2 const v = cur[u];
    float2 m = mul_cf_cd(v, mark); // filter->output.c[n] * step_osc(&mark): the phasor before the step
    oscp_step(mark);
    mark_acc.x += m.x;
    mark_acc.y += m.y;
    mark_off.x += m.x;
    mark_off.y += m.y;
    m = mul_cf_cd(v, space);
    oscp_step(space);
    space_acc.x += m.x;
    space_acc.y += m.y;
    space_off.x += m.x;
    space_off.y += m.y;
    if (++symphase == Samppbit / 2) {
      mid_val = nrm(mark_off) - nrm(space_off);
      mark_off = space_off = make_float2(0.f, 0.f);
    }
    if (symphase < Samppbit) continue;
    symphase = 0;
    float const cur_val = nrm(mark_acc) - nrm(space_acc);
    mark_acc = space_acc = make_float2(0.f, 0.f);
    if (bo && nb < max_bits) {
      bo[nb].sample = sample_base + n;
      bo[nb].cur_val = cur_val;
      bo[nb].reserved = 0;
    }
    nb++;
    if (cur_val * last_val < 0) {
      symphase += ((cur_val - last_val) * mid_val) > 0 ? +1 : -1; // packet.c:469
      if (ones == 6) { // flag
        if (flagsync) {
          frame_bit -= 7;
          int const bytes = frame_bit / 8;
          // (the reference indexes hdlc_frame[1024] without a bound; longer frames are dropped here)
          if (bytes > 0 && bytes <= KH_AX25_MAX && crc_good_dev(hdlc, bytes)) {
            if (nf < max_frames) {
              kh_ax25_frame *f = fo + nf;
              f->sample = sample_base + n;
              f->timestamp = out_ts;
              f->seq = (uint16_t)out_seq;
              f->bytes = (uint16_t)bytes;
              for (int k = 0; k < KH_AX25_MAX; k++) f->data[k] = k < bytes ? hdlc[k] : 0;
              nf++;
            }
            out_seq = (out_seq + 1) & 0xffff; // packet.c:498
            out_ts += bytes;                  // packet.c:501
            out_packets++;                    // packet.c:510-511
            out_bytes += bytes;
          }
        }
        for (int k = 0; k < KH_AX25_MAX / 4; k++) reinterpret_cast<unsigned *>(hdlc)[k] = 0; // packet.c:514-517
        frame_bit = 0;
        flagsync = 1;
      } else if (ones == 5) {
        // stuffed zero
      } else if (ones < 5) {
        if (flagsync) frame_bit++;
      }
      ones = 0;
    } else {
      if (++ones == 7) { // abort
        for (int k = 0; k < KH_AX25_MAX / 4; k++) reinterpret_cast<unsigned *>(hdlc)[k] = 0;
        frame_bit = 0;
        flagsync = 0;
      } else if (flagsync) {
        if (frame_bit >= 0 && frame_bit / 8 < KH_AX25_MAX) hdlc[frame_bit / 8] |= (unsigned char)(1 << (frame_bit % 8));
        frame_bit++;
      }
    }
    last_val = cur_val;
  }
  }
  st->mark_re = mark.re;
  st->mark_im = mark.im;
  st->mark_steps = mark.steps;
  st->space_re = space.re;
  st->space_im = space.im;
  st->space_steps = space.steps;
  st->symphase = symphase;
  st->mark_acc = mark_acc;
  st->space_acc = space_acc;
  st->mark_off = mark_off;
  st->space_off = space_off;
  st->last_val = last_val;
  st->mid_val = mid_val;
  st->frame_bit = frame_bit;
  st->flagsync = flagsync;
  st->ones = ones;
  st->out_timestamp = out_ts;
  st->out_seq = out_seq;
  st->out_packets = out_packets;
  st->out_bytes = out_bytes;
  nframes[s] = nf;
  if (nbits) nbits[s] = nb;
}

int launch_afsk_ingest(hipStream_t st, const void *pcm, size_t pcm_stride, float *time, size_t time_stride, int hist,
                       int count, int nsess) {
  if (count <= 0 || nsess <= 0) return 0;
  hipLaunchKernelGGL(kh_afsk_ingest, dim3((count + 255) / 256, nsess), dim3(256), 0, st,
                     static_cast<const uint16_t *>(pcm), pcm_stride, time, time_stride, hist, count);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}
int launch_afsk_decode(hipStream_t st, const float2 *y, size_t y_stride, int nsamp, int nsess, AfskState *states,
                       void *frames, int max_frames, int *nframes, void *bits, int max_bits, int *nbits,
                       unsigned long long sample_base) {
  if (nsamp <= 0 || nsess <= 0) return 0;
  double msre, msim, ssre, ssim; // set_osc, osc.c:30: cispi(2 f), f = -1200/48000 and -2200/48000 (packet.c:414,419)
  sincos(2 * (-1200. / 48000.f) * M_PI, &msim, &msre);
  sincos(2 * (-2200. / 48000.f) * M_PI, &ssim, &ssre);
  hipLaunchKernelGGL(kh_afsk_decode, dim3((nsess + 63) / 64), dim3(64), 0, st, y, y_stride, nsamp, nsess, states,
                     static_cast<kh_ax25_frame *>(frames), max_frames, nframes, static_cast<kh_afsk_bit *>(bits), max_bits,
                     nbits, msre, msim, ssre, ssim, sample_base);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ pl.c: CTCSS ("PL") and DTMF correlator banks
// pl.c:249-269 runs 63 oscillators per audio sample and adds samp * phasor to 63 integrators; the integration
// intervals are fixed (12000 samples for PL, 2400 for DTMF, pl.c:26-27), so every interval is an independent sum.
// One workgroup per (DTMF interval, session): the 2400 samples sit in LDS, thread (tone, quarter) runs 600 of them
// with the oscillator started in closed form (step_osc returns prod_{k<=n} step_k with step_k = step_0 ss^k:
// phase(n) = n f + r n(n+1)/2 cycles; r is what is left of pl.c:320's sweep rate 9 after cispi(18) ~ 1 - 2.2e-15 i)
// and stepped in fp64, and the four quarters are added in a fixed order.  A PL interval is five such sums, added by
// kh_tones_decide.  algorithmic bytes: 2 per sample.
constexpr int TONES_W = 2400, TONES_Q = 4, TONES_N = KH_PL_NTONES + 8;
__global__ __launch_bounds__(256) void kh_tones_corr(const uint16_t *__restrict__ pcm, size_t pcm_stride,
                                                     const ToneTable *__restrict__ tab, double2 *__restrict__ part,
                                                     int max_windows, unsigned long long base) {
  __shared__ float x[TONES_W];
  __shared__ double2 acc[TONES_N * TONES_Q];
  int const w = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const uint16_t *src = pcm + (size_t)s * pcm_stride + (size_t)w * TONES_W;
  for (int n = tid; n < TONES_W; n += 256) {
    uint16_t const v = src[n];
    short const host = (short)(uint16_t)((v << 8) | (v >> 8)); // (short)ntohs(), pl.c:251
    x[n] = (1.f / 32767) * (float)host;                        // SCALE16 = 1./SHRT_MAX, pl.c:29
  }
  __syncthreads();
  if (tid < TONES_N * TONES_Q) {
    int const t = tid / TONES_Q, q = tid % TONES_Q;
    constexpr int CH = TONES_W / TONES_Q;
    double const f = tab->freq[t], r = tab->rate[t];
    unsigned long long const n0 = base + (unsigned long long)w * TONES_W + (unsigned)(q * CH);
    double const dn = (double)n0;
    double ph = dn * f;
    ph -= floor(ph);
    ph += r * (0.5 * dn * (dn + 1.0)); // |r| ~ 3.5e-16: far below one cycle for any run length of interest
    double pre, pim, sre, sim;
    sincospi(2.0 * ph, &pim, &pre);
    sincospi(2.0 * (f + r * (dn + 0.5 * CH)), &sim, &sre);
    double are = 0.0, aim = 0.0;
    const float *xs = x + q * CH;
    for (int n = 0; n < CH; n++) {
      double const v = (double)xs[n];
      are = fma(v, pre, are);
      aim = fma(v, pim, aim);
      double const nre = pre * sre - pim * sim;
      pim = pre * sim + pim * sre;
      pre = nre;
    }
    acc[tid] = make_double2(are, aim);
  }
  __syncthreads();
  if (tid < TONES_N) {
    double2 a = acc[tid * TONES_Q];
#pragma unroll
    for (int q = 1; q < TONES_Q; q++) {
      a.x += acc[tid * TONES_Q + q].x;
      a.y += acc[tid * TONES_Q + q].y;
    }
    part[((size_t)s * max_windows + w) * TONES_N + tid] = a;
  }
}

__device__ __forceinline__ int dtmf_group_dev(const double2 *p, float *snr_out) { // pl.c:395-410
  int index = -1;
  float max_energy = 0.f, total_energy = 0.f;
  for (int n = 0; n < 4; n++) {
    float const re = (float)p[n].x, im = (float)p[n].y; // the integrators are complex float
    float const energy = re * re + im * im;
    total_energy += energy;
    if (energy >= max_energy) {
      max_energy = energy;
      index = n;
    }
  }
  float const snr = max_energy / (total_energy - max_energy);
  if (snr < 10) index = -1; // (NaN from digital silence does not take this branch: quirk kept)
  *snr_out = snr;
  return index;
}

// one lane per session; the launch's intervals in order
__global__ __launch_bounds__(64) void kh_tones_decide(const double2 *__restrict__ part, int max_windows, int nwindows,
                                                      int nsess, const ToneTable *__restrict__ tab,
                                                      TonesState *__restrict__ states, kh_pl_result *__restrict__ pl,
                                                      int max_pl, int *__restrict__ npl, kh_dtmf_result *__restrict__ dt) {
  int const s = blockIdx.x * 64 + threadIdx.x;
  if (s >= nsess) return;
  TonesState *st = states + s;
  int done = st->pl_windows;
  int np = 0;
  for (int w = 0; w < nwindows; w++) {
    const double2 *p = part + ((size_t)s * max_windows + w) * TONES_N;
    for (int t = 0; t < KH_PL_NTONES; t++) {
      st->pl_part[t].x += p[t].x;
      st->pl_part[t].y += p[t].y;
    }
    if (++done == KH_PL_BLOCK / TONES_W) { // process_pl, pl.c:360-387
      done = 0;
      float e0 = -1.f, e1 = -1.f;
      int i0 = 0;
      for (int t = 0; t < KH_PL_NTONES; t++) {
        float const re = (float)st->pl_part[t].x, im = (float)st->pl_part[t].y;
        float const e = re * re + im * im;
        st->pl_part[t] = make_double2(0.0, 0.0);
        if (e > e0) {
          e1 = e0;
          e0 = e;
          i0 = t;
        } else if (e > e1)
          e1 = e;
      }
      float const snr = 10.f * log10f(e0) - 10.f * log10f(e1); // power2dB, dsp.h:74
      if (np < max_pl) {
        kh_pl_result r;
        r.tone = snr >= 6 ? tab->tone_hz[i0] : 0.f;
        r.snr = snr;
        r.e0 = e0;
        r.e1 = e1;
        r.index = i0;
        pl[(size_t)s * max_pl + np] = r;
      }
      np++;
    }
    kh_dtmf_result d; // process_dtmf, pl.c:390-439
    d.low_index = dtmf_group_dev(p + KH_PL_NTONES, &d.low_snr);
    d.high_index = dtmf_group_dev(p + KH_PL_NTONES + 4, &d.high_snr);
    const char matrix[17] = "123A456B789C*0#D";
    d.digit = (d.low_index != -1 && d.high_index != -1) ? matrix[d.low_index * 4 + d.high_index] : 0;
    dt[(size_t)s * max_windows + w] = d;
  }
  st->pl_windows = done;
  npl[s] = np;
}

int launch_tones(hipStream_t st, const void *pcm, size_t pcm_stride, const ToneTable *tab, void *part, int max_windows,
                 int nwindows, int nsess, TonesState *states, void *pl, int max_pl, int *npl, void *dt,
                 unsigned long long base) {
  if (nwindows <= 0 || nsess <= 0) return 0;
  hipLaunchKernelGGL(kh_tones_corr, dim3(nwindows, nsess), dim3(256), 0, st, static_cast<const uint16_t *>(pcm), pcm_stride,
                     tab, static_cast<double2 *>(part), max_windows, base);
  hipLaunchKernelGGL(kh_tones_decide, dim3((nsess + 63) / 64), dim3(64), 0, st, static_cast<const double2 *>(part),
                     max_windows, nwindows, nsess, tab, states, static_cast<kh_pl_result *>(pl), max_pl, npl,
                     static_cast<kh_dtmf_result *>(dt));
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
