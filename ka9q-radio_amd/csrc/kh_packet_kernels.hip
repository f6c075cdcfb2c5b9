// kh_packet_kernels.hip -- packet.c's receive side for many PCM sessions at once:
//   kh_afsk_ingest   packet.c:340-348   network-order int16 -> float (with ntohs's unsigned result), into the REAL
//                                       overlap-save time buffer of every session
//   kh_afsk_mix      packet.c:445,449   output x mark/space tone phasors, every sample in parallel
//   kh_afsk_decode   packet.c:437-545   boxcar correlators, Gardner timing, NRZI, HDLC unstuffing, crc_good
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

// cnrmf, dsp.h:65.  Two products and a sum, each rounded (no fused multiply-add): the bit clock steps on the SIGN of a
// difference of two of these (packet.c:469), and with every operation a plain IEEE one the loop can be replayed
// operation for operation on a host (tools/fuzz_afsk.py's referee over the device's own products)
__device__ __forceinline__ float nrm(float2 v) {
#pragma clang fp contract(off)
  float const a = v.x * v.x, b = v.y * v.y;
  return a + b;
}

// packet.c:445,449: s = filter->output.c[n] * step_osc(&mark|&space) for every sample of the launch, all of them at
// once.  step_osc returns exp(2 pi i f n) for the n-th sample of the session (packet.c:410-419: both oscillators start
// at 1 with the session); the phase is taken in closed form, frac(f n) in fp64, instead of by the recurrence.  The
// product is complex float x complex double, rounded to float on assignment, as in the reference.
// algorithmic bytes: 8 in + 16 out per sample.
__global__ __launch_bounds__(256) void kh_afsk_mix(const float2 *__restrict__ y, size_t y_stride, float4 *__restrict__ ms,
                                                   size_t ms_stride, int nsamp, double f_mark, double f_space,
                                                   unsigned long long sample_base) {
  int const s = blockIdx.y;
  int const n = blockIdx.x * 256 + threadIdx.x;
  if (n >= nsamp) return;
  float2 const v = y[(size_t)s * y_stride + n];
  double const dn = (double)(sample_base + (unsigned long long)n);
  double pm = dn * f_mark, ps = dn * f_space;
  pm -= floor(pm);
  ps -= floor(ps);
  double mre, mim, sre, sim;
  sincospi(2.0 * pm, &mim, &mre);
  sincospi(2.0 * ps, &sim, &sre);
  double const yr = v.x, yi = v.y;
  ms[(size_t)s * ms_stride + n] = make_float4((float)(yr * mre - yi * mim), (float)(yr * mim + yi * mre),
                                              (float)(yr * sre - yi * sim), (float)(yr * sim + yi * sre));
}

// HDLC for one finished bit (packet.c:466-543 after the clock adjustment): NRZI, flags, bit unstuffing, CRC, frame out
struct HdlcRegs {
  int frame_bit, flagsync, ones;
  unsigned out_ts, out_seq;
  unsigned long long out_packets, out_bytes;
  int nf;
};
__device__ __forceinline__ void hdlc_bit(HdlcRegs &h, bool transition, unsigned long long sample, unsigned char *hdlc,
                                         kh_ax25_frame *fo, int max_frames) {
  if (transition) { // NRZI zero
    if (h.ones == 6) { // flag
      if (h.flagsync) {
        h.frame_bit -= 7;
        int const bytes = h.frame_bit / 8;
        // (the reference indexes hdlc_frame[1024] without a bound; longer frames are dropped here)
        if (bytes > 0 && bytes <= KH_AX25_MAX && crc_good_dev(hdlc, bytes)) {
          if (h.nf < max_frames) {
            kh_ax25_frame *f = fo + h.nf;
            f->sample = sample;
            f->timestamp = h.out_ts;
            f->seq = (uint16_t)h.out_seq;
            f->bytes = (uint16_t)bytes;
            for (int k = 0; k < KH_AX25_MAX; k++) f->data[k] = k < bytes ? hdlc[k] : 0;
          }
          h.nf++;
          h.out_seq = (h.out_seq + 1) & 0xffff; // packet.c:498
          h.out_ts += bytes;                    // packet.c:501
          h.out_packets++;                      // packet.c:510-511
          h.out_bytes += bytes;
        }
      }
      for (int k = 0; k < KH_AX25_MAX / 4; k++) reinterpret_cast<unsigned *>(hdlc)[k] = 0; // packet.c:514-517
      h.frame_bit = 0;
      h.flagsync = 1;
    } else if (h.ones == 5) {
      // stuffed zero
    } else if (h.ones < 5) {
      if (h.flagsync) h.frame_bit++;
    }
    h.ones = 0;
  } else { // NRZI one
    if (++h.ones == 7) { // abort
      for (int k = 0; k < KH_AX25_MAX / 4; k++) reinterpret_cast<unsigned *>(hdlc)[k] = 0;
      h.frame_bit = 0;
      h.flagsync = 0;
    } else if (h.flagsync) {
      if (h.frame_bit >= 0 && h.frame_bit / 8 < KH_AX25_MAX) hdlc[h.frame_bit / 8] |= (unsigned char)(1 << (h.frame_bit % 8));
      h.frame_bit++;
    }
  }
}

// One lane per session; ms: the products of kh_afsk_mix.  What is serial per session is the bit clock: four complex
// accumulators per sample, and at the end of a bit the Gardner step that moves the next bit's boundary
// (packet.c:453-469).  Sessions are not aligned with each other, so in a wavefront of 64 some lane ends a bit at almost
// every sample: that path is kept to the clock itself, and the bit (did the tone change, at which sample) goes into a
// per-lane queue in LDS.  The HDLC state machine (packet.c:470-543) drains the queues once per 960 samples, all lanes
// in step over their k-th queued bit -- the same bits in the same order as the reference sees them.
constexpr int AFSK_DRAIN = 960, AFSK_QMAX = 32; // at most 960/39 + 1 = 25 bits end within 960 samples
__global__ __launch_bounds__(64) void kh_afsk_decode(const float4 *__restrict__ ms, size_t ms_stride, int nsamp, int nsess,
                                                     AfskState *__restrict__ states, kh_ax25_frame *__restrict__ frames,
                                                     int max_frames, int *__restrict__ nframes,
                                                     kh_afsk_bit *__restrict__ bits, int max_bits, int *__restrict__ nbits,
                                                     unsigned long long sample_base, int spw) {
  __shared__ unsigned queue[AFSK_QMAX][64];
  int const lane = threadIdx.x;
  // spw sessions per wavefront: what bounds this kernel is how many bytes its few wavefronts keep in flight, so with
  // fewer sessions than the chip has lanes the sessions are spread over more wavefronts (the other lanes idle)
  int const s0 = blockIdx.x * spw + lane;
  bool const mine = lane < spw && s0 < nsess;
  int const s = mine ? s0 : nsess - 1;
  constexpr int Samppbit = 40; // packet.c:52
  AfskState *st = states + s;
  unsigned char *hdlc = st->hdlc_frame;
  int symphase = st->symphase;
  float2 mark_acc = st->mark_acc, space_acc = st->space_acc, mark_off = st->mark_off, space_off = st->space_off;
  float last_val = st->last_val, mid_val = st->mid_val;
  HdlcRegs h{st->frame_bit, st->flagsync, st->ones, st->out_timestamp, st->out_seq, st->out_packets, st->out_bytes, 0};
  const float4 *src = ms + (size_t)s * ms_stride;
  kh_ax25_frame *fo = frames + (size_t)s * max_frames;
  kh_afsk_bit *bo = bits ? bits + (size_t)s * max_bits : nullptr;
  int nb = 0, qn = 0;
  auto drain = [&]() {
    int qmax = qn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qmax = max(qmax, __shfl_xor(qmax, o, 64));
    for (int k = 0; k < qmax; k++)
      if (k < qn && mine) {
        unsigned const e = queue[k][lane];
        hdlc_bit(h, e & 1u, sample_base + (e >> 1), hdlc, fo, max_frames);
      }
    qn = 0;
  };
  // Samples are fetched G at a time, one group ahead.  Each lane walks its own row, so a group costs one memory round
  // trip (~1.6 us) however little work it holds: the group is as long as the register file allows (2 x 32 float4).
  constexpr int G = 32;
  static_assert(AFSK_DRAIN % G == 0, "queues are drained between groups");
  if (!__builtin_amdgcn_ballot_w64(mine)) return;
  float4 nxt[G];
#pragma unroll
  for (int u = 0; u < G; u++) nxt[u] = u < nsamp ? src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n0 = 0; n0 < nsamp; n0 += G) {
    if (n0 && n0 % AFSK_DRAIN == 0) drain();
    float4 cur[G];
#pragma unroll
    for (int u = 0; u < G; u++) cur[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < G; u++) nxt[u] = n0 + G + u < nsamp ? src[n0 + G + u] : make_float4(0.f, 0.f, 0.f, 0.f);
    // (without this fence the scheduler sinks each of those loads down to the sample that precedes its use and then
    // waits for it: one memory round trip per sample)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; u++) { // packet.c:440-469
      int const n = n0 + u;
      if (n >= nsamp) break;
      float4 const v = cur[u];
      mark_acc.x += v.x;
      mark_acc.y += v.y;
      mark_off.x += v.x;
      mark_off.y += v.y;
      space_acc.x += v.z;
      space_acc.y += v.w;
      space_off.x += v.z;
      space_off.y += v.w;
      if (++symphase == Samppbit / 2) {
        mid_val = nrm(mark_off) - nrm(space_off);
        mark_off = space_off = make_float2(0.f, 0.f);
      }
      if (symphase < Samppbit) continue;
      symphase = 0;
      float const cur_val = nrm(mark_acc) - nrm(space_acc);
      mark_acc = space_acc = make_float2(0.f, 0.f);
      if (bo && nb < max_bits && mine) {
        bo[nb].sample = sample_base + n;
        bo[nb].cur_val = cur_val;
        bo[nb].reserved = 0;
      }
      nb++;
      bool const transition = cur_val * last_val < 0;
      if (transition) symphase += ((cur_val - last_val) * mid_val) > 0 ? +1 : -1; // packet.c:469
      if (qn < AFSK_QMAX) queue[qn][lane] = ((unsigned)n << 1) | (transition ? 1u : 0u);
      qn++;
      last_val = cur_val;
    }
  }
  drain();
  if (!mine) return;
  st->symphase = symphase;
  st->mark_acc = mark_acc;
  st->space_acc = space_acc;
  st->mark_off = mark_off;
  st->space_off = space_off;
  st->last_val = last_val;
  st->mid_val = mid_val;
  st->frame_bit = h.frame_bit;
  st->flagsync = h.flagsync;
  st->ones = h.ones;
  st->out_timestamp = h.out_ts;
  st->out_seq = h.out_seq;
  st->out_packets = h.out_packets;
  st->out_bytes = h.out_bytes;
  nframes[s] = h.nf;
  if (nbits) nbits[s] = nb;
}

int launch_afsk_ingest(hipStream_t st, const void *pcm, size_t pcm_stride, float *time, size_t time_stride, int hist,
                       int count, int nsess) {
  if (count <= 0 || nsess <= 0) return 0;
  hipLaunchKernelGGL(kh_afsk_ingest, dim3((count + 255) / 256, nsess), dim3(256), 0, st,
                     static_cast<const uint16_t *>(pcm), pcm_stride, time, time_stride, hist, count);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}
int launch_afsk_decode(hipStream_t st, const float2 *y, size_t y_stride, float4 *ms, size_t ms_stride, int nsamp, int nsess,
                       AfskState *states, void *frames, int max_frames, int *nframes, void *bits, int max_bits,
                       int *nbits, unsigned long long sample_base) {
  if (nsamp <= 0 || nsess <= 0) return 0;
  if (nsamp >= (1 << 30)) {
    set_error("kh_afsk: more than 2^30 samples per session in one launch");
    return -1;
  }
  // set_osc(&mark, -1200./Samprate, 0), set_osc(&space, -2200./Samprate, 0), packet.c:414,419 (Samprate is a float)
  double const f_mark = -1200. / 48000.f, f_space = -2200. / 48000.f;
  hipLaunchKernelGGL(kh_afsk_mix, dim3((nsamp + 255) / 256, nsess), dim3(256), 0, st, y, y_stride, ms, ms_stride, nsamp,
                     f_mark, f_space, sample_base);
  int spw = 64;
  while (spw > 4 && (nsess + spw - 1) / spw < 2048) spw >>= 1; // aim at two wavefronts per SIMD (1024 SIMDs)
  hipLaunchKernelGGL(kh_afsk_decode, dim3((nsess + spw - 1) / spw), dim3(64), 0, st, ms, ms_stride, nsamp, nsess, states,
                     static_cast<kh_ax25_frame *>(frames), max_frames, nframes, static_cast<kh_afsk_bit *>(bits), max_bits,
                     nbits, sample_base, spw);
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
