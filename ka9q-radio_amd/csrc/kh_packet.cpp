// kh_packet.cpp -- packet.c's receive side as one device-resident object.
//
// Per kh_afsk_execute call, for T blocks of KH_AFSK_L samples of every session:
//   ntohs + scale into the REAL filter input        packet.c:340-348       kh_afsk_ingest
//   execute_filter_input (REAL, L 1000, M 1049)     packet.c:325,345       kh_fft_fwd_lds<2048>
//   execute_filter_output (COMPLEX, decimate 1)     packet.c:407-408,438   kh_filter_out
//   AFSK correlators, Gardner timing, HDLC, CRC     packet.c:440-544       kh_afsk_mix, kh_afsk_decode
// All state (overlap history, tone oscillators, bit clock, partial frame, RTP counters) stays on the device.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ka9q_hip.h"
#include "kh_internal.h"

using namespace kh;

struct kh_afsk {
  int S = 0, T = 0, max_frames = 0, max_bits = 0;
  static constexpr int L = KH_AFSK_L, M = KH_AFSK_M, N = KH_AFSK_L + KH_AFSK_M - 1;
  hipStream_t stream = nullptr;
  uint16_t *d_pcm = nullptr;   // [S][T*L] pending input, network order
  float *d_time[2] = {nullptr, nullptr}; // [S][time_stride]: M-1 history + T*L new samples
  float2 *d_spec = nullptr;    // [S][T*N] (N/2+1 bins used per block)
  float2 *d_y = nullptr;       // [S][T*L] analytic signal
  float4 *d_ms = nullptr;      // [S][T*L] analytic signal x mark, x space phasors
  float2 *d_resp = nullptr;    // [S][N]
  ChanDesc *d_chans = nullptr; // [S]
  AfskState *d_state = nullptr;
  kh_ax25_frame *d_frames = nullptr;
  kh_afsk_bit *d_bits = nullptr;
  int *d_nframes = nullptr, *d_nbits = nullptr;
  size_t pcm_stride = 0, time_stride = 0, spec_stride = 0, y_stride = 0;
  int cur = 0;
  unsigned last_T = 0;
  unsigned long long consumed = 0;
  std::vector<size_t> fill;
};

extern "C" {

int kh_afsk_destroy(kh_afsk *a) {
  if (!a) return 0;
  if (a->stream) hipStreamSynchronize(a->stream);
  void *ptrs[] = {a->d_pcm, a->d_time[0], a->d_time[1], a->d_spec, a->d_y, a->d_ms, a->d_resp, a->d_chans, a->d_state,
                  a->d_frames, a->d_bits, a->d_nframes, a->d_nbits};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (a->stream) hipStreamDestroy(a->stream);
  delete a;
  return 0;
}

kh_afsk *kh_afsk_create(unsigned int sessions, unsigned int max_blocks, unsigned int max_frames, unsigned int max_bits) {
  if (ensure_init()) return nullptr;
  if (sessions == 0 || max_blocks == 0 || max_frames == 0 || sessions > (1u << 20) || max_blocks > 4096) {
    set_error("kh_afsk_create: sessions in [1, 2^20], max_blocks in [1, 4096] and max_frames >= 1 are required");
    return nullptr;
  }
  kh_afsk *a = new kh_afsk();
  a->S = (int)sessions;
  a->T = (int)max_blocks;
  a->max_frames = (int)max_frames;
  a->max_bits = (int)max_bits;
  size_t const S = sessions, T = max_blocks;
  constexpr int L = kh_afsk::L, M = kh_afsk::M, N = kh_afsk::N;
  a->pcm_stride = (T * L + 7) & ~(size_t)7;
  a->time_stride = ((M - 1) + T * L + 3) & ~(size_t)3;
  a->spec_stride = T * N;
  a->y_stride = T * L;
  bool ok = KH_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
  auto A = [&](auto **p, size_t bytes) {
    if (!ok) return;
    ok = KH_HIP(hipMalloc((void **)p, bytes ? bytes : 16));
    if (ok) ok = KH_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
  };
  A(&a->d_pcm, S * a->pcm_stride * sizeof(uint16_t));
  A(&a->d_time[0], S * a->time_stride * sizeof(float));
  A(&a->d_time[1], S * a->time_stride * sizeof(float));
  A(&a->d_spec, S * a->spec_stride * sizeof(float2));
  A(&a->d_y, S * a->y_stride * sizeof(float2));
  A(&a->d_ms, S * a->y_stride * sizeof(float4));
  A(&a->d_resp, S * N * sizeof(float2));
  A(&a->d_chans, S * sizeof(ChanDesc));
  A(&a->d_state, S * sizeof(AfskState));
  A(&a->d_frames, S * (size_t)max_frames * sizeof(kh_ax25_frame));
  A(&a->d_nframes, S * sizeof(int));
  if (max_bits) {
    A(&a->d_bits, S * (size_t)max_bits * sizeof(kh_afsk_bit));
    A(&a->d_nbits, S * sizeof(int));
  }
  if (ok) { // set_filter(filter, +100/Samprate, +4000/Samprate, 3.0), packet.c:408; filter.c:660-702 with decimate 1
    float const low = (float)(+100. / 48000.f), high = (float)(+4000. / 48000.f);
    float const gain = 1.f / (float)N; // COMPLEX output: no sqrt(1/2), filter.c:674-677
    ok = launch_brickwall(a->stream, a->d_resp, N, low, high, gain) == 0;
    ok = ok && launch_window_filter(a->stream, a->d_resp, L, M, 3.0f, 0) == 0;
    for (size_t s = 1; ok && s < S; s++)
      ok = KH_HIP(hipMemcpyAsync(a->d_resp + s * N, a->d_resp, N * sizeof(float2), hipMemcpyDeviceToDevice, a->stream));
  }
  if (ok) {
    std::vector<ChanDesc> ch(S);
    for (size_t s = 0; s < S; s++) ch[s] = ChanDesc{(int)s, 0, COMPLEX, 0};
    ok = KH_HIP(hipMemcpyAsync(a->d_chans, ch.data(), S * sizeof(ChanDesc), hipMemcpyHostToDevice, a->stream));
    // d_state is all zeros, as decode_task's locals start (packet.c:422-435); the tone oscillators start at phasor 1
    // with the session (packet.c:411-419; osc.c:24-27), i.e. they are exp(2 pi i f n) of the session's sample count
    ok = ok && KH_HIP(hipStreamSynchronize(a->stream));
  }
  if (!ok) {
    kh_afsk_destroy(a);
    return nullptr;
  }
  a->fill.assign(S, 0);
  return a;
}

static bool session_ok(kh_afsk *a, unsigned int session, const char *who) {
  if (!a || session >= (unsigned)a->S) {
    set_error("%s: bad handle or session", who);
    return false;
  }
  return true;
}

int kh_afsk_write_pcm(kh_afsk *a, unsigned int session, const int16_t *pcm_be, size_t nwords) {
  if (!session_ok(a, session, "kh_afsk_write_pcm") || (!pcm_be && nwords)) return -1;
  size_t &f = a->fill[session];
  if (f + nwords > (size_t)a->T * kh_afsk::L) {
    set_error("kh_afsk_write_pcm: %zu pending + %zu new words exceed max_blocks * %d", f, nwords, kh_afsk::L);
    return -1;
  }
  if (!KH_HIP(hipMemcpyAsync(a->d_pcm + session * a->pcm_stride + f, pcm_be, nwords * sizeof(int16_t),
                             hipMemcpyHostToDevice, a->stream)))
    return -1;
  f += nwords;
  return 0;
}

int kh_afsk_pcm_device_ptr(kh_afsk *a, void **dptr, size_t *stride_words) {
  if (!a || !dptr) return -1;
  *dptr = a->d_pcm;
  if (stride_words) *stride_words = a->pcm_stride;
  return 0;
}

int kh_afsk_commit_pcm(kh_afsk *a, unsigned int session, size_t nwords) {
  if (!session_ok(a, session, "kh_afsk_commit_pcm")) return -1;
  if (a->fill[session] + nwords > (size_t)a->T * kh_afsk::L) {
    set_error("kh_afsk_commit_pcm: more than max_blocks * %d words pending", kh_afsk::L);
    return -1;
  }
  a->fill[session] += nwords;
  return 0;
}

int kh_afsk_execute(kh_afsk *a, unsigned int nblocks) {
  if (!a || nblocks == 0 || nblocks > (unsigned)a->T) {
    set_error("kh_afsk_execute: nblocks must be in [1, max_blocks]");
    return -1;
  }
  constexpr int L = kh_afsk::L, M = kh_afsk::M, N = kh_afsk::N;
  size_t const need = (size_t)nblocks * L;
  for (int s = 0; s < a->S; s++)
    if (a->fill[s] != need) {
      set_error("kh_afsk_execute: session %d holds %zu words, the launch takes exactly %zu from every session", s,
                a->fill[s], need);
      return -1;
    }
  hipStream_t st = a->stream;
  float *time = a->d_time[a->cur];
  int rc = launch_afsk_ingest(st, a->d_pcm, a->pcm_stride, time, a->time_stride, M - 1, (int)need, a->S);
  if (!rc)
    rc = launch_forward_fft(st, time, a->time_stride, 1, a->d_spec, nullptr, a->spec_stride, N, L, (int)nblocks, a->S,
                            nullptr, N0Fuse{nullptr, 0, 0});
  // overlap-save history for the next launch (filter.c:230): the last M-1 samples move to the head of the other buffer
  if (!rc)
    rc = launch_carry(st, time + need, a->d_time[a->cur ^ 1], a->time_stride * sizeof(float), (size_t)(M - 1) * sizeof(float),
                      a->S);
  if (!rc)
    rc = launch_filter_output(st, a->d_spec, a->spec_stride, N, 1, a->d_chans, a->d_resp, N, L, a->d_y, a->y_stride,
                              (int)nblocks, a->S, FmFront{}, 1);
  if (!rc)
    rc = launch_afsk_decode(st, a->d_y, a->y_stride, a->d_ms, a->y_stride, (int)need, a->S, a->d_state, a->d_frames, a->max_frames,
                            a->d_nframes, a->d_bits, a->max_bits, a->d_nbits, a->consumed);
  if (rc) return -1;
  a->cur ^= 1;
  a->consumed += need;
  a->last_T = nblocks;
  for (auto &f : a->fill) f = 0;
  return 0;
}

int kh_afsk_sync(kh_afsk *a) {
  if (!a) return -1;
  return KH_HIP(hipStreamSynchronize(a->stream)) ? 0 : -1;
}

int kh_afsk_read_frames(kh_afsk *a, unsigned int session, struct kh_ax25_frame *out, size_t max, size_t *n) {
  if (!session_ok(a, session, "kh_afsk_read_frames") || !n) return -1;
  int cnt = 0;
  if (!KH_HIP(hipMemcpyAsync(&cnt, a->d_nframes + session, sizeof(int), hipMemcpyDeviceToHost, a->stream)) ||
      !KH_HIP(hipStreamSynchronize(a->stream)))
    return -1;
  size_t const take = std::min(std::min((size_t)cnt, (size_t)a->max_frames), max);
  if (take && out &&
      (!KH_HIP(hipMemcpyAsync(out, a->d_frames + (size_t)session * a->max_frames, take * sizeof(kh_ax25_frame),
                              hipMemcpyDeviceToHost, a->stream)) ||
       !KH_HIP(hipStreamSynchronize(a->stream))))
    return -1;
  *n = take;
  return 0;
}

int kh_afsk_read_bits(kh_afsk *a, unsigned int session, struct kh_afsk_bit *out, size_t max, size_t *n) {
  if (!session_ok(a, session, "kh_afsk_read_bits") || !n) return -1;
  if (!a->d_bits) {
    set_error("kh_afsk_read_bits: created with max_bits = 0");
    return -1;
  }
  int cnt = 0;
  if (!KH_HIP(hipMemcpyAsync(&cnt, a->d_nbits + session, sizeof(int), hipMemcpyDeviceToHost, a->stream)) ||
      !KH_HIP(hipStreamSynchronize(a->stream)))
    return -1;
  size_t const take = std::min(std::min((size_t)cnt, (size_t)a->max_bits), max);
  if (take && out &&
      (!KH_HIP(hipMemcpyAsync(out, a->d_bits + (size_t)session * a->max_bits, take * sizeof(kh_afsk_bit),
                              hipMemcpyDeviceToHost, a->stream)) ||
       !KH_HIP(hipStreamSynchronize(a->stream))))
    return -1;
  *n = take;
  return 0;
}

int kh_afsk_read_status(kh_afsk *a, unsigned int session, struct kh_afsk_status *out) {
  if (!session_ok(a, session, "kh_afsk_read_status") || !out) return -1;
  AfskState s;
  int nf = 0, nb = 0;
  bool ok = KH_HIP(hipMemcpyAsync(&s, a->d_state + session, sizeof s, hipMemcpyDeviceToHost, a->stream));
  ok = ok && KH_HIP(hipMemcpyAsync(&nf, a->d_nframes + session, sizeof(int), hipMemcpyDeviceToHost, a->stream));
  if (a->d_nbits) ok = ok && KH_HIP(hipMemcpyAsync(&nb, a->d_nbits + session, sizeof(int), hipMemcpyDeviceToHost, a->stream));
  ok = ok && KH_HIP(hipStreamSynchronize(a->stream));
  if (!ok) return -1;
  out->timestamp = s.out_timestamp;
  out->seq = (uint16_t)s.out_seq;
  out->flagsync = (uint16_t)s.flagsync;
  out->packets = s.out_packets;
  out->bytes = s.out_bytes;
  out->frames_last_launch = (uint32_t)nf;
  out->bits_last_launch = (uint32_t)nb;
  return 0;
}

int kh_afsk_read_baseband(kh_afsk *a, unsigned int session, kh_cf32 *out, size_t nsamples) {
  if (!session_ok(a, session, "kh_afsk_read_baseband") || !out) return -1;
  if (nsamples > (size_t)a->last_T * kh_afsk::L) {
    set_error("kh_afsk_read_baseband: the last launch produced %zu samples", (size_t)a->last_T * kh_afsk::L);
    return -1;
  }
  if (!KH_HIP(hipMemcpyAsync(out, a->d_y + session * a->y_stride, nsamples * sizeof(float2), hipMemcpyDeviceToHost,
                             a->stream)) ||
      !KH_HIP(hipStreamSynchronize(a->stream)))
    return -1;
  return 0;
}

// diagnostic: the tone products of the last launch, 4 floats per sample {Re, Im of y x mark, Re, Im of y x space}
// (kh_afsk_mix): what kh_afsk_decode's accumulators add up.  The sweeps replay packet.c:440-545 over them.
int kh_afsk_read_products(kh_afsk *a, unsigned int session, float *out, size_t nsamples) {
  if (!session_ok(a, session, "kh_afsk_read_products") || !out) return -1;
  if (nsamples > (size_t)a->last_T * kh_afsk::L) {
    set_error("kh_afsk_read_products: the last launch produced %zu samples", (size_t)a->last_T * kh_afsk::L);
    return -1;
  }
  if (!KH_HIP(hipMemcpyAsync(out, a->d_ms + session * a->y_stride, nsamples * sizeof(float4), hipMemcpyDeviceToHost,
                             a->stream)) ||
      !KH_HIP(hipStreamSynchronize(a->stream)))
    return -1;
  return 0;
}

} // extern "C"

// ================================================================ pl.c: CTCSS and DTMF decoders
struct kh_tones {
  int S = 0, W = 0, max_pl = 0;
  hipStream_t stream = nullptr;
  uint16_t *d_pcm = nullptr; // [S][W*2400] pending input, network order
  double2 *d_part = nullptr; // [S][W][63]
  ToneTable *d_tab = nullptr;
  TonesState *d_state = nullptr;
  kh_pl_result *d_pl = nullptr;     // [S][max_pl]
  kh_dtmf_result *d_dtmf = nullptr; // [S][W]
  int *d_npl = nullptr;
  size_t pcm_stride = 0;
  unsigned last_windows = 0;
  unsigned long long consumed = 0;
  std::vector<size_t> fill;
};

extern "C" {

int kh_tones_destroy(kh_tones *a) {
  if (!a) return 0;
  if (a->stream) hipStreamSynchronize(a->stream);
  void *ptrs[] = {a->d_pcm, a->d_part, a->d_tab, a->d_state, a->d_pl, a->d_dtmf, a->d_npl};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (a->stream) hipStreamDestroy(a->stream);
  delete a;
  return 0;
}

kh_tones *kh_tones_create(unsigned int sessions, unsigned int max_windows) {
  if (ensure_init()) return nullptr;
  if (sessions == 0 || max_windows == 0 || sessions > (1u << 20) || max_windows > 4096) {
    set_error("kh_tones_create: sessions in [1, 2^20] and max_windows in [1, 4096] are required");
    return nullptr;
  }
  static const float PL_tones[KH_PL_NTONES] = { // pl.c:36-44
      67.0,  69.3,  71.9,  74.4,  77.0,  79.7,  82.5,  85.4,  88.5,  91.5,  94.8,  97.4,
      100.0, 103.5, 107.2, 110.9, 114.8, 118.8, 123.0, 127.3, 131.8, 136.5, 141.3, 146.2,
      150.0, 151.4, 156.7, 162.2, 167.9, 173.8, 179.9, 186.2, 192.8, 199.5, 206.5, 213.8,
      221.3, 229.1, 237.1, 245.5, 254.1, 159.8, 165.5, 171.3, 177.3, 183.5, 189.9, 196.6,
      203.5, 210.7, 218.1, 225.7, 233.6, 241.8, 250.3,
  };
  static const float DTMF_low[4] = {697, 770, 852, 941}, DTMF_high[4] = {1209, 1336, 1477, 1633}; // pl.c:48-49
  kh_tones *a = new kh_tones();
  a->S = (int)sessions;
  a->W = (int)max_windows;
  a->max_pl = (int)max_windows / (KH_PL_BLOCK / KH_DTMF_BLOCK) + 1;
  size_t const S = sessions, W = max_windows;
  a->pcm_stride = W * KH_DTMF_BLOCK;
  bool ok = KH_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
  auto A = [&](auto **p, size_t bytes) {
    if (!ok) return;
    ok = KH_HIP(hipMalloc((void **)p, bytes ? bytes : 16));
    if (ok) ok = KH_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
  };
  A(&a->d_pcm, S * a->pcm_stride * sizeof(uint16_t));
  A(&a->d_part, S * W * (KH_PL_NTONES + 8) * sizeof(double2));
  A(&a->d_tab, sizeof(ToneTable));
  A(&a->d_state, S * sizeof(TonesState));
  A(&a->d_pl, S * (size_t)a->max_pl * sizeof(kh_pl_result));
  A(&a->d_dtmf, S * W * sizeof(kh_dtmf_result));
  A(&a->d_npl, S * sizeof(int));
  if (ok) {
    ToneTable t;
    memset(&t, 0, sizeof t);
    int const SAMPRATE = 48000; // pl.c:25
    // set_osc(&pl_osc[n], PL_tones[n]/SAMPRATE, 9), pl.c:320: a float quotient; the "rate" of 9 cycles/sample^2 leaves
    // phasor_step_step = cispi(18) (osc.c:32), i.e. a sweep of arg(cispi(18)) / 2 pi cycles/sample^2
    double const sweep = atan2(sin(18 * M_PI), cos(18 * M_PI)) / (2 * M_PI);
    for (int n = 0; n < KH_PL_NTONES; n++) {
      t.freq[n] = (double)(PL_tones[n] / SAMPRATE);
      t.rate[n] = sweep;
      t.tone_hz[n] = PL_tones[n];
    }
    for (int n = 0; n < 4; n++) { // pl.c:327-328
      t.freq[KH_PL_NTONES + n] = (double)(DTMF_low[n] / SAMPRATE);
      t.freq[KH_PL_NTONES + 4 + n] = (double)(DTMF_high[n] / SAMPRATE);
    }
    ok = KH_HIP(hipMemcpy(a->d_tab, &t, sizeof t, hipMemcpyHostToDevice));
  }
  if (!ok) {
    kh_tones_destroy(a);
    return nullptr;
  }
  a->fill.assign(S, 0);
  return a;
}

static bool tsession_ok(kh_tones *a, unsigned int session, const char *who) {
  if (!a || session >= (unsigned)a->S) {
    set_error("%s: bad handle or session", who);
    return false;
  }
  return true;
}

int kh_tones_write_pcm(kh_tones *a, unsigned int session, const int16_t *pcm_be, size_t nwords) {
  if (!tsession_ok(a, session, "kh_tones_write_pcm") || (!pcm_be && nwords)) return -1;
  size_t &f = a->fill[session];
  if (f + nwords > a->pcm_stride) {
    set_error("kh_tones_write_pcm: %zu pending + %zu new words exceed max_windows * %d", f, nwords, KH_DTMF_BLOCK);
    return -1;
  }
  if (!KH_HIP(hipMemcpyAsync(a->d_pcm + session * a->pcm_stride + f, pcm_be, nwords * sizeof(int16_t),
                             hipMemcpyHostToDevice, a->stream)))
    return -1;
  f += nwords;
  return 0;
}

int kh_tones_pcm_device_ptr(kh_tones *a, void **dptr, size_t *stride_words) {
  if (!a || !dptr) return -1;
  *dptr = a->d_pcm;
  if (stride_words) *stride_words = a->pcm_stride;
  return 0;
}

int kh_tones_commit_pcm(kh_tones *a, unsigned int session, size_t nwords) {
  if (!tsession_ok(a, session, "kh_tones_commit_pcm")) return -1;
  if (a->fill[session] + nwords > a->pcm_stride) {
    set_error("kh_tones_commit_pcm: more than max_windows * %d words pending", KH_DTMF_BLOCK);
    return -1;
  }
  a->fill[session] += nwords;
  return 0;
}

int kh_tones_execute(kh_tones *a, unsigned int nwindows) {
  if (!a || nwindows == 0 || nwindows > (unsigned)a->W) {
    set_error("kh_tones_execute: nwindows must be in [1, max_windows]");
    return -1;
  }
  size_t const need = (size_t)nwindows * KH_DTMF_BLOCK;
  for (int s = 0; s < a->S; s++)
    if (a->fill[s] != need) {
      set_error("kh_tones_execute: session %d holds %zu words, the launch takes exactly %zu from every session", s,
                a->fill[s], need);
      return -1;
    }
  if (launch_tones(a->stream, a->d_pcm, a->pcm_stride, a->d_tab, a->d_part, a->W, (int)nwindows, a->S, a->d_state, a->d_pl,
                   a->max_pl, a->d_npl, a->d_dtmf, a->consumed))
    return -1;
  a->consumed += need;
  a->last_windows = nwindows;
  for (auto &f : a->fill) f = 0;
  return 0;
}

int kh_tones_sync(kh_tones *a) {
  if (!a) return -1;
  return KH_HIP(hipStreamSynchronize(a->stream)) ? 0 : -1;
}

int kh_tones_read_pl(kh_tones *a, unsigned int session, struct kh_pl_result *out, size_t max, size_t *n) {
  if (!tsession_ok(a, session, "kh_tones_read_pl") || !n) return -1;
  int cnt = 0;
  if (!KH_HIP(hipMemcpyAsync(&cnt, a->d_npl + session, sizeof(int), hipMemcpyDeviceToHost, a->stream)) ||
      !KH_HIP(hipStreamSynchronize(a->stream)))
    return -1;
  size_t const take = std::min(std::min((size_t)cnt, (size_t)a->max_pl), max);
  if (take && out &&
      (!KH_HIP(hipMemcpyAsync(out, a->d_pl + (size_t)session * a->max_pl, take * sizeof(kh_pl_result), hipMemcpyDeviceToHost,
                              a->stream)) ||
       !KH_HIP(hipStreamSynchronize(a->stream))))
    return -1;
  *n = take;
  return 0;
}

int kh_tones_read_dtmf(kh_tones *a, unsigned int session, struct kh_dtmf_result *out, size_t max, size_t *n) {
  if (!tsession_ok(a, session, "kh_tones_read_dtmf") || !n) return -1;
  size_t const take = std::min((size_t)a->last_windows, max);
  if (take && out &&
      (!KH_HIP(hipMemcpyAsync(out, a->d_dtmf + (size_t)session * a->W, take * sizeof(kh_dtmf_result), hipMemcpyDeviceToHost,
                              a->stream)) ||
       !KH_HIP(hipStreamSynchronize(a->stream))))
    return -1;
  *n = take;
  return 0;
}

} // extern "C"
