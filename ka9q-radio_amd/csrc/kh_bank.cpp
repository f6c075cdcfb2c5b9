// kh_bank.cpp -- the channel bank: one shared forward FFT per block, C channels, many blocks per launch.
//
// This is the multi-channel receiver the reference only sketches (execute_filter_output's `rotate`,
// filter.c:270-289, "for eventual use in a demodulator channel bank").  Per execute call, for T blocks:
//   proc_samples convert + Doppler/LO2 mix + IF power     radio.c:132-204   kh_ingest
//   execute_filter_input                                  filter.c:212-238  kh_fft_fwd_lds | kh_fft_pass1+2
//   compute_n0 + smoothing                                radio.c:197-202,422-465   kh_n0_pass<0,1>, kh_n0_final
//   execute_filter_output per channel                     filter.c:240-384  kh_filter_out
//   demod_fm / demod_linear / demod_am + scaleclip        fm.c, linear.c, am.c, audio.c:22-28
// All state (overlap history, NCO phase, demodulator state) stays on the device between calls.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sched.h>

#include "../../include/ka9q_hip.h"
#include "kh_internal.h"
#include "kh_n0.cuh"

using namespace kh;

namespace {

struct HostOsc {     // host mirror of struct osc (osc.h:9-17) in phase/frequency form
  double phase = 0;  // cycles, of the phasor the next step_osc call returns
  double freq_set = 0, rate = 0; // as given to set_osc
  double freq_now = 0;           // phase of phasor_step (advances when sweeping)
  void set(double f, double r) { // osc.c:22-36: phase is kept
    freq_set = f;
    rate = r;
    freq_now = f;
  }
  void advance(size_t n) { // n step_osc calls, osc.c:39-50
    long double const dn = (long double)n;
    long double ph = (long double)phase + dn * (long double)freq_now + (long double)rate * 0.5L * dn * (dn + 1.0L);
    ph -= floorl(ph);
    phase = (double)ph;
    freq_now += rate * (double)n;
  }
};

size_t payload_bytes(int payload, size_t n) {
  switch (payload) {
  case KH_PT_IQ16LE:
  case KH_PT_PCM16BE: return 4 * n;
  case KH_PT_IQ8: return 2 * n;
  case KH_PT_IQ12: return 3 * n;
  default: return 8 * n;
  }
}

} // namespace

struct kh_bank {
  kh_bank_config cfg;
  int N, ND, olen, T, C, P;
  hipStream_t stream = nullptr;
  // The launch is cut in two: FRONT (convert + mix, forward FFT: streams the whole block through HBM) on `front`,
  // BACK (compute_n0, channels, demodulators: little memory traffic, latency bound) on `stream`.  Everything the
  // front hands to the back is double buffered (sp = slot of the current launch), so the front of launch k+1 runs
  // beside the back of launch k.
  hipStream_t front = nullptr;
  // mixed banks: the linear / AM demodulators run beside the FM demodulator (enqueue_back), on this stream
  hipStream_t side = nullptr; // created with the first launch that forks (a bank that never does has no third stream)
  int side_prio = 0;
  bool side_use_prio = false;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int demod_fork = 1; // KA9Q_HIP_DEMOD_FORK=0: one after the other on the back stream
  static constexpr int MAX_SLOTS = 4;
  hipEvent_t ev_front_done[MAX_SLOTS] = {}, ev_back_done[MAX_SLOTS] = {};
  int sp = 0, last_sp = 0, pipeline = 1;
  std::vector<IngestDesc> idesc_dev; // what d_idesc holds
  bool slot_used[MAX_SLOTS] = {};    // ev_back_done[slot] has been recorded
  int nslots = 3; // ring depth of what the front half hands to the back half (KA9Q_HIP_SLOTS, 2..4)
  int fused_ingest = 1;   // policy (KA9Q_HIP_FUSED_INGEST)
  int ride_n0 = 1;        // policy (KA9Q_HIP_RIDE_N0)
  int n0_direct_mode = -1; // policy (KA9Q_HIP_N0_DIRECT): 0 never, 1 rendezvous, 2 or -1 (default) bracket
  int n0_stagger = 0;      // KA9Q_HIP_N0_STAGGER (N0Fuse::stagger)
  int n0_spin = 1000;      // KA9Q_HIP_N0_SPIN: polls before a row-pass workgroup gives up (0: every block is re-read)
  bool fused_now = false; // this launch: large transform and no oscillator active
  // device
  unsigned char *d_raw = nullptr;
  size_t raw_stride = 0; // bytes per input
  float2 *d_time[2] = {nullptr, nullptr};
  size_t time_stride = 0; // samples per input: (M-1) + T*L (+pad)
  int cur = 0;
  float2 *d_spec[MAX_SLOTS] = {}, *d_tmp = nullptr;
  size_t spec_stride = 0; // T*N
  ChanDesc *d_chans = nullptr;
  float2 *d_resp = nullptr;
  DemodDesc *d_ddesc = nullptr;
  DemodState *d_dstate = nullptr;
  IngestDesc *d_idesc = nullptr;
  static constexpr int IDESC_SLOTS = 4;   // pinned staging ring for the per-launch oscillator snapshot
  IngestDesc *h_idesc = nullptr;
  hipEvent_t idesc_ev[IDESC_SLOTS] = {};
  int idesc_slot = 0;
  float2 *d_bb = nullptr;
  size_t bb_stride = 0; // T*olen
  float *d_audio = nullptr;
  int16_t *d_pcm = nullptr;
  size_t pcm_stride = 0; // 2*T*olen
  kh_status *d_status = nullptr;
  kh_input_status *d_istatus = nullptr;
  float *d_energy[MAX_SLOTS] = {}, *d_n0acc[MAX_SLOTS] = {}, *d_n0state = nullptr, *d_scalar = nullptr;
  int *d_lists = nullptr; // fm | linear | am, C entries each
  FmBlockStat *d_fmstat = nullptr; // FM path scratch (kh_internal.h)
  FmWork *d_fmwork = nullptr;
  double *d_fmcarry = nullptr, *d_decaypow = nullptr;
  Fm2Carry *d_fm2x = nullptr; // kh_fm_demod2 with split channels (small banks): [C][T], marked with fm2_epoch
  unsigned fm2_epoch = 0;
  int fm_parity = 0;
  // host
  std::vector<ChanDesc> chans;
  std::vector<DemodDesc> ddesc;
  std::vector<kh_channel_config> ccfg;
  std::vector<float> noise_gains;
  std::vector<HostOsc> lo2, doppler;
  std::vector<float> gain;
  std::vector<size_t> fill; // samples present in each input's raw ring, from sample `head` on
  // The unread samples of every input start `head` samples into its ring: a launch moves the head instead of the
  // samples (a latency-mode caller writes T*L samples and consumes them with T calls of execute(1): moving what is left
  // after every launch was T-1 copy-engine operations between the front stream's kernels per launch).  The ring is
  // compacted only when a write would run past its end (ring_compact), and the head returns to 0 whenever a launch empties it.
  size_t head = 0;
  std::vector<int> lists;
  int n_fm = 0, n_lin = 0, n_am = 0;
  int n_lin_nopll = 0; // the linear list holds the channels without PLL first
  bool chans_dirty = true;
  unsigned last_blocks = 0;
  int keep_audio = 1, keep_baseband = 1;
  // kh_bank_keep_spectrum(bank, 0): a four-step row pass stores only the rows of the spectrum that the channels' filter
  // outputs read (SpecMask, kh_internal.h); mask is rebuilt by upload_channels, masked_now is this launch's decision
  int keep_spectrum = 1;
  SpecMask mask{};
  bool masked_now = false;
  unsigned *d_arrive = nullptr;           // N0Fuse::arrive: N0_SYNC_STRIDE words per (input, block)
  unsigned *h_fail = nullptr, *h_fail_dev = nullptr; // N0Fuse::host_fail: pinned, and its address as the device sees it
  // A block whose bracket fails is finished by ONE row-pass workgroup (n0_repair_block): right, but slow.  When the count of
  // such blocks has moved, the bank stores whole spectra for sparse_hold launches (the second step's ordinary re-read
  // serves them), twice as long every time in a row, and tries again afterwards.
  unsigned fail_seen = 0;
  unsigned long long launch_no = 0, sparse_resume_at = 0, sparse_last_fail = 0;
  unsigned sparse_hold = 0, sparse_hold_min = 64; // KA9Q_HIP_SPARSE_HOLD (0: never fall back to whole spectra)
  int rtp = 0;                       // PCM packetisation plan (audio.c:32-135) after every launch
  kh_rtp_chunk *d_rtp = nullptr;     // [C][rtp_stride]
  kh_rtp_state *d_rtpstate = nullptr;
  int rtp_stride = 0;
  // What a back half needs to know of the launch it belongs to (the caller moves on to the next launch's front half
  // while the back half is being queued: BackEnq)
  struct BackJob {
    unsigned nblocks;
    int sp, fm_parity;
    bool fused_now, masked_now;
  };
  // KA9Q_HIP_ENQ_THREAD=1 (default 0): the back half's launches go through a thread of the bank's own.  The caller queues
  // the front half (its kernels, the hand-over event), posts the job and returns; the thread queues the wait for that
  // event, the back half and the slot's "done" event.  Measured at BASELINE configs[1] (tools/hostbound.py): kh_bank_execute
  // takes the caller 18-22 us instead of 35-40 (bursts), 30 instead of 52 us per step in a long run -- and the step stays
  // at 59 us, because the host was not what bound it: with no kernels at all the step's events and hand-overs take the
  // queues 12.5 us, the front half's kernels alone make it 50.4, the back half's alone 34.3, both 59.2
  // (KA9Q_HIP_DIAG_SKIP).  So it is off by default: it buys a caller that also moves packets 20 us per launch, not throughput.
  // One job at a time: a job is posted only when the one before is queued, so everything the caller reads later
  // (ev_back_done of a slot nslots launches ago, the streams it synchronises after quiet()) has been queued.
  // Every other entry point that touches the back stream or the bank's channel state calls quiet() first.
  struct BackEnq {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<int> state{0}; // 0: idle; 1: a job is posted or being queued; 2: quit
    BackJob job{};
    int rc = 0;                // sticky until quiet() reports it
    char err[512] = "";
  };
  BackEnq *enq = nullptr;
  int enq_thread = 0;
  unsigned evflags = hipEventDisableTiming; // of the events that order the bank's own streams
  // hipGraph cache: the kernel sequence of one execute call, captured the second time a shape is seen
  struct GraphEntry {
    unsigned nblocks;
    int cur, fmpar, sp;
    const void *raw;       // the raw ring the front half reads (kh_bank_stream_submit alternates between two)
    bool pipelined, fused, masked;
    int seen;              // launches seen with this key; -1: capture failed, stay eager
    hipGraphExec_t front, back;
  };
  std::vector<GraphEntry> graphs;
  // measured on MI355X (tools/hostbound.py): replaying two small graphs costs the host more than their handful of
  // launches, so graphs are off unless asked for
  int use_graph = 0;
  // streaming ingest (kh_bank_stream_*): a pinned host ring of `stream_depth` slots per input, fed to two device raw
  // rings by a transfer kernel (kh_host_pull) on a copy stream of its own, so that the transfer of launch k+1 runs beside launch k
  int16_t *h_pcm = nullptr;       // pinned staging of kh_bank_read_pcm_all (C x pcm_stride words), allocated on first use
  unsigned char *h_ring = nullptr, *h_ring_dev = nullptr; // the pinned ring and its address as the device sees it
  size_t h_slot_bytes = 0;        // bytes of one input's slot (max_blocks*L samples of the wire format)
  unsigned stream_depth = 0;
  unsigned long long stream_k = 0; // submits so far
  hipStream_t copy = nullptr;
  int stream_sdma = 0, stream_hostwait = 0; // KA9Q_HIP_STREAM_SDMA, KA9Q_HIP_STREAM_HOSTWAIT as read by kh_bank_stream_open
  // Events of submit k live at index k % (depth + 2): an event is recorded again only after every wait that was queued
  // on its earlier record has been passed.
  std::vector<hipEvent_t> ev_h2d;      // the transfer of submit k has completed
  std::vector<hipEvent_t> ev_raw_free; // the front half of submit k has read its device ring
  unsigned char *d_raw_buf[2] = {nullptr, nullptr}; // [0] is d_raw as created
  // profiling: 1 = everything on one stream with an event pair around every stage (each kernel alone on the chip);
  // 2 = the normal two-stream pipeline with the same event pairs, each on the stream its stage runs on (what a kernel
  // takes beside the other half's kernels); the events of a launch are read when its slot comes round again
  int profiling = 0;
  hipEvent_t evp[MAX_SLOTS][KH_STAGE_COUNT + 2] = {};
  bool evp_pending[MAX_SLOTS] = {};
  hipEvent_t *cur_ev = nullptr; // the event set enqueue_front / enqueue_back record into
  hipEvent_t ev[KH_STAGE_COUNT + 2] = {};
  float stage_ms[KH_STAGE_COUNT] = {};
  unsigned stage_n[KH_STAGE_COUNT] = {};
};

static bool is_pow2(unsigned v) { return v && !(v & (v - 1)); }

static void drop_graphs(kh_bank *b) { // anything a captured launch baked in has changed
  for (auto &g : b->graphs) {
    if (g.front) hipGraphExecDestroy(g.front);
    if (g.back) hipGraphExecDestroy(g.back);
  }
  b->graphs.clear();
}

// ---- the thread that queues back halves (kh_bank::BackEnq)
static int enqueue_back_job(kh_bank *b, hipStream_t st, kh_bank::BackJob const &j, bool prof);
static void enq_main(kh_bank *b) {
  kh_bank::BackEnq &w = *b->enq;
  if (ensure_init()) { // (sets this thread's current device)
    w.rc = -1;
    snprintf(w.err, sizeof(w.err), "%s", ka9q_hip_last_error());
  }
  for (;;) {
    int s = 0;
    for (int i = 0; i < 40000 && (s = w.state.load(std::memory_order_acquire)) == 0; i++) __builtin_ia32_pause(); // ~0.2 ms
    if (s == 0) {
      std::unique_lock<std::mutex> lk(w.m);
      w.cv.wait(lk, [&] { return w.state.load(std::memory_order_acquire) != 0; });
      s = w.state.load(std::memory_order_acquire);
    }
    if (s == 2) return;
    kh_bank::BackJob const j = w.job;
    int rc = w.rc;
    if (!rc) {
      hipStream_t const sb = b->stream;
      // (waiting for the front half on this thread instead, with no barrier in the back stream's queue, was tried: +-0.5 us)
      if (!KH_HIP(hipStreamWaitEvent(sb, b->ev_front_done[j.sp], 0))) rc = -1;
      if (!rc) rc = enqueue_back_job(b, sb, j, false);
      if (!rc && !KH_HIP(hipEventRecord(b->ev_back_done[j.sp], sb))) rc = -1;
      if (rc) {
        w.rc = rc;
        snprintf(w.err, sizeof(w.err), "%s", ka9q_hip_last_error());
      }
    }
    w.state.store(0, std::memory_order_release);
  }
}
// waits until the thread has queued what it was given; -1 (once) if a back half failed there
static int quiet(kh_bank *b) {
  if (!b->enq) return 0;
  kh_bank::BackEnq &w = *b->enq;
  for (unsigned i = 0; w.state.load(std::memory_order_acquire) == 1; i++) {
    if (i < 20000)
      __builtin_ia32_pause();
    else
      sched_yield();
  }
  if (w.rc) {
    set_error("%s", w.err);
    w.rc = 0;
    return -1;
  }
  return 0;
}
static int enq_post(kh_bank *b, kh_bank::BackJob const &j) {
  if (!b->enq) {
    b->enq = new kh_bank::BackEnq;
    b->enq->th = std::thread(enq_main, b);
  }
  if (quiet(b)) return -1;
  kh_bank::BackEnq &w = *b->enq;
  {
    std::lock_guard<std::mutex> lk(w.m);
    w.job = j;
    w.state.store(1, std::memory_order_release);
  }
  w.cv.notify_one();
  return 0;
}
static void enq_stop(kh_bank *b) {
  if (!b->enq) return;
  (void)quiet(b);
  {
    std::lock_guard<std::mutex> lk(b->enq->m);
    b->enq->state.store(2, std::memory_order_release);
  }
  b->enq->cv.notify_one();
  b->enq->th.join();
  delete b->enq;
  b->enq = nullptr;
}

extern "C" {

int kh_bank_stream_close(kh_bank *b);
int kh_bank_keep_spectrum(kh_bank *b, int enable);
int kh_bank_destroy(kh_bank *b) {
  if (!b) return 0;
  enq_stop(b);
  if (b->stream) hipStreamSynchronize(b->stream);
  if (b->front) hipStreamSynchronize(b->front);
  kh_bank_stream_close(b);
  drop_graphs(b);
  for (int k = 0; k < kh_bank::MAX_SLOTS; k++) {
    if (b->d_spec[k]) hipFree(b->d_spec[k]);
    if (b->d_energy[k]) hipFree(b->d_energy[k]);
    if (b->d_n0acc[k]) hipFree(b->d_n0acc[k]);
  }
  void *ptrs[] = {b->d_raw,   b->d_time[0], b->d_time[1], b->d_tmp, b->d_chans, b->d_resp,
                  b->d_ddesc, b->d_dstate,  b->d_idesc,   b->d_bb,     b->d_audio,  b->d_pcm,    b->d_status,
                  b->d_istatus, b->d_n0state, b->d_scalar, b->d_arrive,
                  b->d_lists,
                  b->d_fmstat, b->d_fmwork, b->d_fmcarry, b->d_decaypow, b->d_rtp, b->d_rtpstate, b->d_fm2x};
  for (void *p : ptrs)
    if (p) hipFree(p);
  for (auto &e : b->ev)
    if (e) hipEventDestroy(e);
  for (auto &set : b->evp)
    for (auto &e : set)
      if (e) hipEventDestroy(e);
  for (auto &e : b->idesc_ev)
    if (e) hipEventDestroy(e);
  if (b->h_idesc) hipHostFree(b->h_idesc);
  if (b->h_fail) hipHostFree(b->h_fail);
  if (b->h_pcm) hipHostFree(b->h_pcm);
  for (auto &e : b->ev_front_done)
    if (e) hipEventDestroy(e);
  for (auto &e : b->ev_back_done)
    if (e) hipEventDestroy(e);
  if (b->ev_fork) hipEventDestroy(b->ev_fork);
  if (b->ev_join) hipEventDestroy(b->ev_join);
  if (b->side) hipStreamDestroy(b->side);
  if (b->front) hipStreamDestroy(b->front);
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
  return 0;
}

kh_bank *kh_bank_create(const struct kh_bank_config *cfg) {
  if (!cfg) return nullptr;
  if (ensure_init()) return nullptr;
  unsigned const N = cfg->L + cfg->M - 1;
  if (cfg->L == 0 || cfg->M == 0 || !((is_pow2(N) && N >= 256 && N <= (1u << 20)) || make_any_plan((int)N).npass)) {
    set_error("kh_bank_create: N = L+M-1 = %u must be a power of two in [256, 2^20] or have only the factors 2, 3, 5, 7 and be "
              "at most %d", N, KH_MAX_SMALL_N);
    return nullptr;
  }
  if (cfg->decimate == 0 || N % cfg->decimate || cfg->L % cfg->decimate) {
    set_error("kh_bank_create: decimate %u must divide N %u and L %u", cfg->decimate, N, cfg->L);
    return nullptr;
  }
  unsigned const ND = N / cfg->decimate;
  if (!((is_pow2(ND) && ND >= 256 && ND <= 8192) || make_any_plan((int)ND).npass)) {
    set_error("kh_bank_create: N/decimate = %u must be a power of two in [256, 8192] or have only the factors 2, 3, 5, 7", ND);
    return nullptr;
  }
  if (cfg->n_inputs == 0 || cfg->n_channels == 0 || cfg->max_blocks == 0 || cfg->samprate <= 0) {
    set_error("kh_bank_create: n_inputs, n_channels, max_blocks and samprate must be positive");
    return nullptr;
  }
  if (N > KH_MAX_SMALL_N && (cfg->L % 2)) {
    set_error("kh_bank_create: L must be even when N > %d", KH_MAX_SMALL_N);
    return nullptr;
  }
  switch (cfg->payload) {
  case KH_PT_CF32: case KH_PT_IQ12: case KH_PT_IQ16LE: case KH_PT_IQ8: case KH_PT_PCM16BE: break;
  default: set_error("kh_bank_create: unknown payload type %d", cfg->payload); return nullptr;
  }
  // the fused column pass addresses the raw ring with 32-bit byte offsets (buffer descriptors): a ring of 2 GiB or
  // more would wrap them silently
  if (N > KH_MAX_SMALL_N && payload_bytes(cfg->payload, (size_t)cfg->max_blocks * cfg->L) >= ((size_t)1 << 31)) {
    set_error("kh_bank_create: max_blocks*L = %zu samples of this payload exceed the 2 GiB raw ring the kernels can address",
              (size_t)cfg->max_blocks * cfg->L);
    return nullptr;
  }
  if (cfg->max_blocks > (unsigned)N0_MAXB) { // kh_n0_final keeps one launch's per-block means in LDS
    set_error("kh_bank_create: max_blocks %u above %d", cfg->max_blocks, N0_MAXB);
    return nullptr;
  }
  kh_bank *b = new kh_bank();
  b->cfg = *cfg;
  b->N = N;
  b->ND = ND;
  b->olen = cfg->L / cfg->decimate;
  b->T = cfg->max_blocks;
  b->C = cfg->n_channels;
  b->P = cfg->n_inputs;
  size_t const T = b->T, C = b->C, P = b->P, L = cfg->L, M = cfg->M;
  bool ok = true;
  // The back half is short, latency-bound kernels; the front half fills the chip with workgroups.  With equal
  // priorities the dispatcher drains the front kernel before it looks at the other queue (measured: 14 us of overlap
  // out of 44); the back stream therefore gets the higher priority.  KA9Q_HIP_PRIO=0 turns that off.
  int prio_lo = 0, prio_hi = 0;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // lowest, greatest (numerically smaller = higher)
  bool use_prio = prio_lo != prio_hi;
  if (const char *e = getenv("KA9Q_HIP_PRIO")) use_prio = use_prio && atoi(e) != 0;
  if (use_prio) {
    ok = ok && KH_HIP(hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, prio_hi));
    ok = ok && KH_HIP(hipStreamCreateWithPriority(&b->front, hipStreamNonBlocking, prio_lo));
  } else {
    ok = ok && KH_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    ok = ok && KH_HIP(hipStreamCreateWithFlags(&b->front, hipStreamNonBlocking));
  }
  b->side_prio = use_prio ? prio_hi : 0;
  b->side_use_prio = use_prio;
  if (const char *e = getenv("KA9Q_HIP_DEMOD_FORK")) b->demod_fork = atoi(e);
  // The hand-over events between the bank's streams (front half done, back half done) are recorded without the
  // system-scope fence that hipEventRecord performs by default (hipEventDisableSystemFence; KA9Q_HIP_EVENT_FENCE=1: with
  // it).  They order kernels of ONE device -- every kernel ends with a release at agent scope, which is what the other
  // stream's kernels need -- and the host reads nothing on their account: what it reads, it reads through copies queued on
  // the back stream and a stream synchronisation, which fence for the host by themselves.  The fence is a cache write-back
  // and invalidation in the queue after each half: 2 us of a 54.5 us step at configs[1], 2 of 113 at config 4's share
  // (2, a device-scope release, hipEventReleaseToDevice: no gain).
  static int const ev_fence = getenv("KA9Q_HIP_EVENT_FENCE") ? atoi(getenv("KA9Q_HIP_EVENT_FENCE")) : 0;
  unsigned const evflags = hipEventDisableTiming | (ev_fence == 0 ? (unsigned)hipEventDisableSystemFence : ev_fence == 2 ? (unsigned)hipEventReleaseToDevice : 0u);
  b->evflags = evflags; // (the fork and join events of a mixed bank's demodulators as well)
  for (auto &e : b->ev_front_done) ok = ok && KH_HIP(hipEventCreateWithFlags(&e, evflags));
  for (auto &e : b->ev_back_done) ok = ok && KH_HIP(hipEventCreateWithFlags(&e, evflags));
  ok = ok && KH_HIP(hipHostMalloc((void **)&b->h_idesc, kh_bank::IDESC_SLOTS * P * sizeof(IngestDesc), hipHostMallocDefault));
  for (auto &e : b->idesc_ev) ok = ok && KH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  b->raw_stride = (payload_bytes(cfg->payload, T * L) + 15) & ~(size_t)15;
  b->time_stride = ((M - 1) + T * L + 1) & ~(size_t)1;
  b->spec_stride = T * N;
  b->bb_stride = T * b->olen;
  b->pcm_stride = 2 * T * b->olen;
  auto A = [&](auto **p, size_t bytes, bool zero) {
    if (!ok) return;
    ok = KH_HIP(hipMalloc((void **)p, bytes ? bytes : 16));
    if (ok && zero) ok = KH_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
  };
  A(&b->d_raw, P * b->raw_stride, true);
  A(&b->d_time[0], P * b->time_stride * sizeof(float2), true);
  A(&b->d_time[1], P * b->time_stride * sizeof(float2), true);
  if (const char *e = getenv("KA9Q_HIP_SLOTS")) b->nslots = std::min(std::max(atoi(e), 2), (int)kh_bank::MAX_SLOTS);
  for (int k = 0; k < b->nslots; k++) A(&b->d_spec[k], P * b->spec_stride * sizeof(float2), false);
  if (N > KH_MAX_SMALL_N) A(&b->d_tmp, P * b->spec_stride * sizeof(float2), false);
  A(&b->d_chans, C * sizeof(ChanDesc), true);
  A(&b->d_resp, C * ND * sizeof(float2), true);
  A(&b->d_ddesc, C * sizeof(DemodDesc), true);
  A(&b->d_dstate, C * sizeof(DemodState), true);
  A(&b->d_idesc, P * sizeof(IngestDesc), true);
  A(&b->d_bb, C * b->bb_stride * sizeof(float2), true);
  A(&b->d_audio, C * b->pcm_stride * sizeof(float), true);
  A(&b->d_pcm, C * b->pcm_stride * sizeof(int16_t), true);
  A(&b->d_status, C * T * sizeof(kh_status), true);
  A(&b->d_istatus, P * T * sizeof(kh_input_status), true);
  for (int k = 0; k < b->nslots; k++) {
    A(&b->d_energy[k], P * T * std::max(ingest_chunks(cfg->L), fused_ingest_tiles(N)) * sizeof(float), true);
    A(&b->d_n0acc[k], (n0_part_floats() + (cfg->compute_n0 ? n0_cand_floats(N) : 0)) * P * T * sizeof(float), true);
  }
  A(&b->d_n0state, P * sizeof(float), false);
  if (N > KH_MAX_SMALL_N && cfg->compute_n0) { // (kh_bank_keep_spectrum(bank, 0): arrival counters and the host's failure count)
    A(&b->d_arrive, (size_t)P * T * N0_SYNC_STRIDE * sizeof(unsigned), true);
    if (ok) ok = KH_HIP(hipHostMalloc((void **)&b->h_fail, 64, hipHostMallocMapped));
    if (ok) {
      *b->h_fail = 0;
      ok = KH_HIP(hipHostGetDevicePointer((void **)&b->h_fail_dev, b->h_fail, 0));
    }
  }
  A(&b->d_scalar, sizeof(float), true);
  A(&b->d_lists, 3 * C * sizeof(int), true);
  A(&b->d_fmstat, C * T * sizeof(FmBlockStat), true);
  A(&b->d_fmwork, C * b->bb_stride * sizeof(FmWork), true);
  A(&b->d_fmcarry, C * T * sizeof(double), true);
  A(&b->d_fm2x, C * T * sizeof(Fm2Carry), true);
  A(&b->d_decaypow, (b->olen + 1) * sizeof(double), false);
  b->rtp_stride = (int)(T * ((2 * b->olen + 479) / 480)); // stereo: 2 words per output sample
  A(&b->d_rtp, C * b->rtp_stride * sizeof(kh_rtp_chunk), true);
  A(&b->d_rtpstate, C * sizeof(kh_rtp_state), true);
  for (auto &e : b->ev)
    if (ok) ok = KH_HIP(hipEventCreate(&e));
  if (ok) { // 0.99376949^k (fm.c:111), by repeated multiplication in long double
    std::vector<double> dp(b->olen + 1);
    long double v = 1.0L;
    for (int k = 0; k <= b->olen; k++) {
      dp[k] = (double)v;
      v *= 0.99376949L;
    }
    ok = KH_HIP(hipMemcpy(b->d_decaypow, dp.data(), dp.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (ok) { // n0 starts as NaN ("Happens at startup", radio.c:199-202)
    std::vector<float> nan(P, NAN);
    ok = KH_HIP(hipMemcpy(b->d_n0state, nan.data(), P * sizeof(float), hipMemcpyHostToDevice));
  }
  if (!ok) {
    kh_bank_destroy(b);
    return nullptr;
  }
  if (const char *e = getenv("KA9Q_HIP_GRAPH")) b->use_graph = atoi(e);
  if (const char *e = getenv("KA9Q_HIP_PIPELINE")) b->pipeline = atoi(e);
  if (const char *e = getenv("KA9Q_HIP_ENQ_THREAD")) b->enq_thread = atoi(e);
  if (const char *e = getenv("KA9Q_HIP_FUSED_INGEST")) b->fused_ingest = atoi(e);
  if (const char *e = getenv("KA9Q_HIP_RIDE_N0")) b->ride_n0 = atoi(e);
  if (const char *e = getenv("KA9Q_HIP_N0_DIRECT")) b->n0_direct_mode = atoi(e);
  if (const char *e = getenv("KA9Q_HIP_KEEP_SPECTRUM")) kh_bank_keep_spectrum(b, atoi(e) != 0); // (A/B switch: the default)
  if (const char *e = getenv("KA9Q_HIP_SPARSE_HOLD")) b->sparse_hold_min = (unsigned)atoi(e);
  if (const char *e = getenv("KA9Q_HIP_N0_SPIN")) b->n0_spin = std::min(std::max(atoi(e), 0), 100000);
  if (const char *e = getenv("KA9Q_HIP_N0_STAGGER")) b->n0_stagger = std::min(std::max(atoi(e), 0), 64);
  b->chans.assign(C, ChanDesc{0, 0, COMPLEX, 0});
  b->ddesc.assign(C, DemodDesc{});
  b->ccfg.assign(C, kh_channel_config{});
  b->noise_gains.assign(C, NAN);
  b->lo2.assign(P, HostOsc());
  b->doppler.assign(P, HostOsc());
  b->gain.assign(P, 1.0f);
  b->fill.assign(P, 0);
  b->lists.assign(3 * C, 0);
  return b;
}

int kh_bank_set_channel(kh_bank *b, unsigned int ch, const struct kh_channel_config *c) {
  if (!b || !c || ch >= (unsigned)b->C) {
    set_error("kh_bank_set_channel: bad argument");
    return -1;
  }
  if (c->input >= (unsigned)b->P) {
    set_error("kh_bank_set_channel: input %u out of range", c->input);
    return -1;
  }
  if (c->rotate <= -b->N || c->rotate >= b->N) {
    set_error("kh_bank_set_channel: |rotate| must be < N");
    return -1;
  }
  if (c->out_type != COMPLEX && c->out_type != CROSS_CONJ) {
    set_error("kh_bank_set_channel: out_type must be COMPLEX or CROSS_CONJ");
    return -1;
  }
  if (c->demod < KH_DEMOD_NONE || c->demod > KH_DEMOD_AM) {
    set_error("kh_bank_set_channel: unknown demodulator %d", c->demod);
    return -1;
  }
  float const out_rate = (float)b->cfg.samprate / b->cfg.decimate;
  // set_filter(out, low/out_rate, high/out_rate, beta): radio.c:408-411, filter.c:660-702
  float const flow = c->low / out_rate, fhigh = c->high / out_rate;
  if (std::isnan(flow) || std::isnan(fhigh) || std::isnan(c->kaiser_beta) || fabsf(flow) > 0.5f || fabsf(fhigh) > 0.5f) {
    set_error("kh_bank_set_channel: filter edges outside +-0.5 of the output rate");
    return -1;
  }
  if (quiet(b)) return -1; // (the thread that queues back halves reads the channel lists)
  hipStream_t st = b->stream;
  float gain = 1.f / (float)b->N;
  if (c->out_type == CROSS_CONJ) gain *= (float)M_SQRT1_2;
  float2 *resp = b->d_resp + (size_t)ch * b->ND;
  int const M_dec = (b->cfg.M - 1) / b->cfg.decimate + 1;
  if (b->olen + M_dec - 1 != b->ND) {
    set_error("kh_bank_set_channel: L/D + (M-1)/D != N/D");
    return -1;
  }
  if (launch_brickwall(st, resp, b->ND, flow, fhigh, gain)) return -1;
  if (launch_window_filter(st, resp, b->olen, M_dec, c->kaiser_beta, 0)) return -1;
  float const scale = c->out_type == CROSS_CONJ ? 2.f * b->N : (float)b->N;
  if (launch_noise_gain(st, resp, b->ND, scale, b->d_scalar)) return -1;
  float ng = NAN;
  if (!KH_HIP(hipMemcpyAsync(&ng, b->d_scalar, sizeof(float), hipMemcpyDeviceToHost, st))) return -1;
  if (!KH_HIP(hipStreamSynchronize(st))) return -1;
  b->noise_gains[ch] = ng;
  b->ccfg[ch] = *c;
  b->chans[ch] = ChanDesc{(int)c->input, c->rotate, c->out_type, c->demod == KH_DEMOD_FM};
  DemodDesc d{};
  d.demod = c->demod;
  d.flat = c->flat;
  d.headroom = c->headroom;
  d.dsamprate = out_rate;                                                         // fm.c:27
  d.fm_gain = (float)((c->headroom * M_1_PI * out_rate) / fabsf(c->low - c->high)); // fm.c:47
  d.pll = c->pll;
  d.square = c->square;
  d.agc = c->agc;
  d.env = c->env;
  d.channels = c->channels == 2 ? 2 : 1;
  d.hangtime = c->hangtime;
  d.recovery_rate = c->recovery_rate;
  int const out_samprate_i = (int)out_rate; // demod->output.samprate is an int (radio.h:144)
  d.out_samprate = (float)out_samprate_i;
  d.am_hang_samples = c->hangtime * out_samprate_i; // am.c:80
  {                                                 // init_pll(&pll, loop_bw, 1/sqrt2, 0, samprate), osc.c:64-85
    float const samptime = (float)(1. / (float)out_samprate_i);
    float nf = c->loop_bw;
    nf *= samptime;
    float const vcogain = (float)(2 * M_PI);
    float const natfreq = (float)(nf * 2 * M_PI);
    float const tau1 = vcogain * 1.f / (natfreq * natfreq);
    float const tau2 = (float)(2 * (float)M_SQRT1_2 / natfreq);
    d.pll_prop_gain = tau2 / tau1;
    d.pll_integrator_gain = 1 / tau1;
  }
  d.lock_limit = (int)(1.f * out_samprate_i); // linear.c:34-36
  d.shift_freq = c->shift;
  b->ddesc[ch] = d;
  // fresh demodulator state, as at thread start (fm.c:26-33, linear.c:29-44, am.c:26-35)
  DemodState s{};
  s.vco_re = 1.0; // linear.c:40 leaves struct pll uninitialised; set_osc's is_phasor_init makes it 1 (osc.c:24-27)
  s.shift_re = 1.0;
  s.agc_gain = c->agc_gain;
  s.snr = 0.f; // linear.c:44
  {            // am.c:21,31: dB2voltage(recovery_rate * samptime)
    float const samptime = b->cfg.decimate / (float)b->cfg.samprate;
    s.am_recovery_per_sample = powf(10.f, (c->recovery_rate * samptime) / 20.f);
  }
  // (both FM parity slots start zeroed: fm.c:26,32,33)
  if (!KH_HIP(hipMemcpyAsync(b->d_dstate + ch, &s, sizeof(s), hipMemcpyHostToDevice, st))) return -1;
  if (!KH_HIP(hipStreamSynchronize(st))) return -1;
  b->chans_dirty = true;
  drop_graphs(b);
  return 0;
}

// ---- demodulator state of a channel as an opaque record (moving a channel to another bank / another GPU's process)
size_t kh_bank_channel_state_size(void) { return sizeof(DemodState); }
int kh_bank_get_channel_state(kh_bank *b, unsigned int ch, void *out, size_t bytes) {
  if (!b || ch >= (unsigned)b->C || !out || bytes != sizeof(DemodState)) {
    set_error("kh_bank_get_channel_state: bad channel, or the buffer is not kh_bank_channel_state_size() bytes");
    return -1;
  }
  DemodState s;
  if (kh_bank_sync(b)) return -1; // (the launches in flight)
  if (!KH_HIP(hipMemcpyAsync(&s, b->d_dstate + ch, sizeof(s), hipMemcpyDeviceToHost, b->stream)) ||
      !KH_HIP(hipStreamSynchronize(b->stream)))
    return -1;
  if (b->fm_parity) { // the record holds the FM state the NEXT launch reads in slot 0, whatever this bank's parity is
    std::swap(s.fm_state[0], s.fm_state[1]);
    std::swap(s.fm_lastaudio[0], s.fm_lastaudio[1]);
    std::swap(s.fm_squelch_open[0], s.fm_squelch_open[1]);
  }
  memcpy(out, &s, sizeof(s));
  return 0;
}
int kh_bank_set_channel_state(kh_bank *b, unsigned int ch, const void *in, size_t bytes) {
  if (!b || ch >= (unsigned)b->C || !in || bytes != sizeof(DemodState)) {
    set_error("kh_bank_set_channel_state: bad channel, or the buffer is not kh_bank_channel_state_size() bytes");
    return -1;
  }
  if (kh_bank_sync(b)) return -1;
  DemodState s;
  memcpy(&s, in, sizeof(s));
  // both slots: the next launch reads slot fm_parity, and nothing reads the other before that launch has rewritten it
  s.fm_state[1] = s.fm_state[0];
  s.fm_lastaudio[1] = s.fm_lastaudio[0];
  s.fm_squelch_open[1] = s.fm_squelch_open[0];
  hipStream_t st = b->stream;
  if (!KH_HIP(hipMemcpyAsync(b->d_dstate + ch, &s, sizeof(s), hipMemcpyHostToDevice, st))) return -1;
  return KH_HIP(hipStreamSynchronize(st)) ? 0 : -1;
}

int kh_bank_set_lo(kh_bank *b, unsigned int input, double lo2_freq, double doppler_freq, double doppler_rate) {
  if (!b || input >= (unsigned)b->P) return -1;
  b->lo2[input].set(lo2_freq, 0.0);                    // set_second_LO, radio.c:350
  b->doppler[input].set(doppler_freq, doppler_rate);   // set_doppler, radio.c:238
  return 0;
}
int kh_bank_set_second_lo(kh_bank *b, unsigned int input, double lo2_freq) {
  if (!b || input >= (unsigned)b->P) return -1;
  b->lo2[input].set(lo2_freq, 0.0); // radio.c:350
  return 0;
}
int kh_bank_set_doppler(kh_bank *b, unsigned int input, double doppler_freq, double doppler_rate) {
  if (!b || input >= (unsigned)b->P) return -1;
  b->doppler[input].set(doppler_freq, doppler_rate); // radio.c:238
  return 0;
}
int kh_bank_set_gain(kh_bank *b, unsigned int input, float gain_factor) {
  if (!b || input >= (unsigned)b->P) return -1;
  b->gain[input] = gain_factor;
  return 0;
}

// the stream the front half (ingest, ring compaction) is issued on outside profiling passes: raw-ring writes go
// there too, so that they are ordered behind a launch that still reads the ring (KA9Q_HIP_PIPELINE=0: one stream)
static hipStream_t raw_stream(kh_bank *b) { return b->pipeline ? b->front : b->stream; }
// Moves every input's unread samples to the start of its ring, ordered behind the front halves that still read them (same
// stream).  The move is by `head` samples, so pieces of at most `head` samples never overlap their own destination, and
// they go in ascending order on the one stream: each lands where the piece before it has already been read.
static int ring_compact(kh_bank *b) {
  if (b->head == 0) return 0;
  hipStream_t const ws = raw_stream(b);
  for (int i = 0; i < b->P; i++) {
    unsigned char *base = b->d_raw + i * b->raw_stride;
    for (size_t off = 0; off < b->fill[i]; off += b->head) {
      size_t const n = std::min(b->head, b->fill[i] - off);
      if (!KH_HIP(hipMemcpyAsync(base + payload_bytes(b->cfg.payload, off), base + payload_bytes(b->cfg.payload, b->head + off),
                                 payload_bytes(b->cfg.payload, n), hipMemcpyDeviceToDevice, ws)))
        return -1;
    }
  }
  b->head = 0;
  return 0;
}

int kh_bank_write_input(kh_bank *b, unsigned int input, const void *payload, size_t count) {
  if (!b || input >= (unsigned)b->P || (!payload && count)) return -1;
  size_t const cap = (size_t)b->T * b->cfg.L;
  if (b->fill[input] + count > cap) {
    set_error("kh_bank_write_input: ring overflow (%zu + %zu > %zu samples)", b->fill[input], count, cap);
    return -1;
  }
  if (b->head + b->fill[input] + count > cap && ring_compact(b)) return -1;
  unsigned char *dst = b->d_raw + input * b->raw_stride + payload_bytes(b->cfg.payload, b->head + b->fill[input]);
  // on the stream the front half runs on: behind the ingest of any launch still reading the ring
  hipStream_t const ws = raw_stream(b);
  if (!KH_HIP(hipMemcpyAsync(dst, payload, payload_bytes(b->cfg.payload, count), hipMemcpyDefault, ws)))
    return -1;
  if (!KH_HIP(hipStreamSynchronize(ws))) return -1; // the caller may reuse its buffer
  b->fill[input] += count;
  return 0;
}
int kh_bank_write_zeros(kh_bank *b, unsigned int input, size_t count) {
  if (!b || input >= (unsigned)b->P) return -1;
  size_t const cap = (size_t)b->T * b->cfg.L;
  if (b->fill[input] + count > cap) {
    set_error("kh_bank_write_zeros: ring overflow");
    return -1;
  }
  if (b->head + b->fill[input] + count > cap && ring_compact(b)) return -1;
  unsigned char *dst = b->d_raw + input * b->raw_stride + payload_bytes(b->cfg.payload, b->head + b->fill[input]);
  if (!KH_HIP(hipMemsetAsync(dst, 0, payload_bytes(b->cfg.payload, count), raw_stream(b)))) return -1;
  b->fill[input] += count;
  return 0;
}
int kh_bank_raw_device_ptr(kh_bank *b, unsigned int input, void **dptr, size_t *capacity_samples) {
  if (!b || input >= (unsigned)b->P || !dptr) return -1;
  *dptr = b->d_raw + input * b->raw_stride;
  if (capacity_samples) *capacity_samples = (size_t)b->T * b->cfg.L;
  return 0;
}
int kh_bank_commit_input(kh_bank *b, unsigned int input, size_t count) {
  if (!b || input >= (unsigned)b->P) return -1;
  if (count > (size_t)b->T * b->cfg.L) return -1;
  // (kh_bank_raw_device_ptr is the START of the ring: whatever was left unread is dropped with the head)
  if (b->head) {
    for (auto &f : b->fill) f = 0;
    b->head = 0;
  }
  b->fill[input] = count;
  return 0;
}

// ---- streaming ingest: what replaces proc_samples' per-packet loop (radio.c:132-191) when the samples arrive on the
// host.  The producer (the thread that owns the socket) writes wire-format samples straight into a pinned slot, the
// library moves the slot to the device on a copy stream and runs the launch behind it.
int kh_bank_stream_open(kh_bank *b, unsigned int depth) {
  if (!b || depth < 2 || depth > 64) {
    set_error("kh_bank_stream_open: depth must be in [2, 64]");
    return -1;
  }
  if (b->stream_depth) {
    set_error("kh_bank_stream_open: already open");
    return -1;
  }
  if (quiet(b)) return -1;
  size_t const P = b->P;
  b->h_slot_bytes = b->raw_stride;
  bool ok = KH_HIP(hipHostMalloc((void **)&b->h_ring, (size_t)depth * P * b->h_slot_bytes, hipHostMallocMapped));
  // (the staging of kh_bank_read_pcm_all as well: nothing is allocated once samples flow)
  if (ok && !b->h_pcm) ok = KH_HIP(hipHostMalloc((void **)&b->h_pcm, (size_t)b->C * b->pcm_stride * sizeof(int16_t), hipHostMallocDefault));
  ok = ok && KH_HIP(hipHostGetDevicePointer((void **)&b->h_ring_dev, b->h_ring, 0));
  {
    // at a priority of its own: the runtime maps streams to its few hardware queues by priority class, and a barrier
    // packet of one stream holds up every stream that shares its queue.  The back stream is created at the greatest
    // priority and the front stream at the least (kh_bank_create), so the copy stream takes the one between them and
    // shares a queue with neither of the streams whose launches depend on its transfers.
    int least = 0, greatest = 0;
    ok = ok && KH_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    int const own = (least - greatest >= 2) ? (least + greatest) / 2 : greatest;
    int const prio = getenv("KA9Q_HIP_STREAM_PRIO") ? atoi(getenv("KA9Q_HIP_STREAM_PRIO")) : 1;
    b->stream_sdma = getenv("KA9Q_HIP_STREAM_SDMA") ? atoi(getenv("KA9Q_HIP_STREAM_SDMA")) : 1;
    // 1 (default): the HOST waits for the transfer before it queues the launch, so no stream ever waits on another
    // stream's event on account of the ring; the transfer of submit k+1 still runs beside the launch of submit k.
    // 0: the front stream waits on the transfer's event.
    b->stream_hostwait = getenv("KA9Q_HIP_STREAM_HOSTWAIT") ? atoi(getenv("KA9Q_HIP_STREAM_HOSTWAIT")) : 1;
    ok = ok && (prio ? KH_HIP(hipStreamCreateWithPriority(&b->copy, hipStreamNonBlocking, own))
                     : KH_HIP(hipStreamCreateWithFlags(&b->copy, hipStreamNonBlocking)));
  }
  b->d_raw_buf[0] = b->d_raw;
  ok = ok && KH_HIP(hipMalloc((void **)&b->d_raw_buf[1], P * b->raw_stride));
  ok = ok && KH_HIP(hipMemset(b->d_raw_buf[1], 0, P * b->raw_stride));
  b->ev_h2d.assign(depth + 2, nullptr);
  b->ev_raw_free.assign(depth + 2, nullptr);
  for (auto &e : b->ev_h2d) ok = ok && KH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : b->ev_raw_free) ok = ok && KH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  b->stream_k = 0;
  b->stream_depth = depth;
  drop_graphs(b); // (a captured front half has the raw ring's address baked in)
  if (!ok) {
    kh_bank_stream_close(b);
    return -1;
  }
  return 0;
}
int kh_bank_stream_close(kh_bank *b) {
  if (!b) return 0;
  (void)quiet(b);
  if (b->copy) hipStreamSynchronize(b->copy);
  if (b->front) hipStreamSynchronize(b->front);
  if (b->stream) hipStreamSynchronize(b->stream);
  for (auto &e : b->ev_h2d)
    if (e) hipEventDestroy(e);
  b->ev_h2d.clear();
  for (auto &e : b->ev_raw_free)
    if (e) hipEventDestroy(e);
  b->ev_raw_free.clear();
  drop_graphs(b);
  if (b->d_raw_buf[0]) b->d_raw = b->d_raw_buf[0]; // what kh_bank_destroy frees
  if (b->d_raw_buf[1]) hipFree(b->d_raw_buf[1]);
  b->d_raw_buf[0] = b->d_raw_buf[1] = nullptr;
  if (b->h_ring) hipHostFree(b->h_ring);
  b->h_ring = b->h_ring_dev = nullptr;
  if (b->copy) hipStreamDestroy(b->copy);
  b->copy = nullptr;
  b->stream_depth = 0;
  return 0;
}
void *kh_bank_stream_slot(kh_bank *b, unsigned int input, size_t *capacity_samples) {
  if (!b || !b->stream_depth || input >= (unsigned)b->P) {
    set_error("kh_bank_stream_slot: stream not open or input out of range");
    return nullptr;
  }
  unsigned const slot = (unsigned)(b->stream_k % b->stream_depth);
  size_t const R = b->ev_h2d.size();
  // the transfer that last left this slot (depth submits ago) must have completed before the producer overwrites it
  if (b->stream_k >= b->stream_depth && !KH_HIP(hipEventSynchronize(b->ev_h2d[(b->stream_k - b->stream_depth) % R]))) return nullptr;
  if (capacity_samples) *capacity_samples = (size_t)b->T * b->cfg.L;
  return b->h_ring + ((size_t)slot * b->P + input) * b->h_slot_bytes;
}
int kh_bank_stream_submit(kh_bank *b, unsigned int nblocks) {
  if (!b || !b->stream_depth) {
    set_error("kh_bank_stream_submit: stream not open");
    return -1;
  }
  if (nblocks == 0 || nblocks > (unsigned)b->T) {
    set_error("kh_bank_stream_submit: nblocks %u outside [1, %d]", nblocks, b->T);
    return -1;
  }
  unsigned long long const k = b->stream_k;
  unsigned const slot = (unsigned)(k % b->stream_depth);
  size_t const R = b->ev_h2d.size(), e = k % R;
  int const dev = (int)(k & 1);
  size_t const bytes = payload_bytes(b->cfg.payload, (size_t)nblocks * b->cfg.L);
  // The front half that last read this device ring (two submits ago) must have run before the transfer overwrites it.
  // Waited for on the HOST: that front half started when the transfer before last ended, so it is normally over, and
  // the transfer of submit k-1 keeps the link busy meanwhile; a producer that runs ahead of the GPU is held here.
  if (k >= 2 && !KH_HIP(hipEventSynchronize(b->ev_raw_free[(k - 2) % R]))) return -1;
  if (b->stream_sdma) { // the copy engines (measured 34 GB/s against the transfer kernel's 28 GB/s; KA9Q_HIP_STREAM_SDMA=0 takes the kernel)
    for (int i = 0; i < b->P; i++)
      if (!KH_HIP(hipMemcpyAsync(b->d_raw_buf[dev] + (size_t)i * b->raw_stride, b->h_ring + ((size_t)slot * b->P + i) * b->h_slot_bytes,
                                 bytes, hipMemcpyHostToDevice, b->copy)))
        return -1;
  } else if (launch_host_pull(b->copy, b->h_ring_dev + (size_t)slot * b->P * b->h_slot_bytes, b->d_raw_buf[dev], b->raw_stride, bytes,
                              b->P))
    return -1;
  if (!KH_HIP(hipEventRecord(b->ev_h2d[e], b->copy))) return -1;
  // the stream the front half takes (profiling passes and one-stream mode put it on the back stream) waits for the samples
  hipStream_t const fs = (b->profiling == 1 || !b->pipeline) ? b->stream : b->front;
  // (KA9Q_HIP_STREAM_HOSTWAIT=1: the host waits for the transfer instead, and no stream waits for another)
  if (b->stream_hostwait ? !KH_HIP(hipEventSynchronize(b->ev_h2d[e])) : !KH_HIP(hipStreamWaitEvent(fs, b->ev_h2d[e], 0))) return -1;
  b->d_raw = b->d_raw_buf[dev];
  b->head = 0;
  for (int i = 0; i < b->P; i++) b->fill[i] = (size_t)nblocks * b->cfg.L;
  int const rc = kh_bank_execute(b, nblocks);
  if (rc) return rc;
  if (!KH_HIP(hipEventRecord(b->ev_raw_free[e], fs))) return -1;
  b->stream_k++;
  return 0;
}
// diagnostic: which of the bank's streams and events still have work outstanding, as text
int kh_bank_debug_pending(kh_bank *b, char *buf, size_t n) {
  if (!b || !buf || n == 0) return -1;
  auto q = [](hipError_t e) { (void)hipGetLastError(); return e == hipSuccess ? '.' : e == hipErrorNotReady ? 'P' : 'E'; };
  std::string o = "streams back/front/copy: ";
  o += q(hipStreamQuery(b->stream));
  o += q(hipStreamQuery(b->front));
  o += b->copy ? q(hipStreamQuery(b->copy)) : '-';
  o += " front_done: ";
  for (auto &e : b->ev_front_done) o += q(hipEventQuery(e));
  o += " back_done: ";
  for (auto &e : b->ev_back_done) o += q(hipEventQuery(e));
  o += " h2d: ";
  for (auto &e : b->ev_h2d) o += q(hipEventQuery(e));
  o += " raw_free: ";
  for (auto &e : b->ev_raw_free) o += q(hipEventQuery(e));
  snprintf(buf, n, "%s", o.c_str());
  return 0;
}
// PCM of every channel of the last launch in one transfer: out[ch*nwords .. ] (host memory); synchronises
int kh_bank_read_pcm_all(kh_bank *b, int16_t *out, size_t nwords) {
  if (!b || !out || nwords > b->pcm_stride || quiet(b)) return -1;
  // through pinned memory: a strided copy into pageable memory goes row by row (measured: 4.4 ms for 640 rows of 1 KiB)
  if (!b->h_pcm) { // first use on a bank without a stream (kh_bank_stream_open allocates it): with the queues drained
    if (kh_bank_sync(b)) return -1;
    if (!KH_HIP(hipHostMalloc((void **)&b->h_pcm, (size_t)b->C * b->pcm_stride * sizeof(int16_t), hipHostMallocDefault))) return -1;
  }
  // the launches first, then the copy: a strided device-to-host copy queued behind launches that themselves wait for
  // host-to-device transfers of the streaming ring did not return on ROCm 7.2 (three submits in flight; the copy
  // engines' queues); with the stream drained the copy depends on nothing
  if (!KH_HIP(hipStreamSynchronize(b->stream))) return -1;
  if (!KH_HIP(hipMemcpy2DAsync(b->h_pcm, nwords * sizeof(int16_t), b->d_pcm, b->pcm_stride * sizeof(int16_t),
                               nwords * sizeof(int16_t), b->C, hipMemcpyDeviceToHost, b->stream)))
    return -1;
  if (!KH_HIP(hipStreamSynchronize(b->stream))) return -1;
  memcpy(out, b->h_pcm, (size_t)b->C * nwords * sizeof(int16_t));
  return 0;
}

static int upload_channels(kh_bank *b) {
  b->n_fm = b->n_lin = b->n_am = 0;
  int const C = b->C;
  for (int ch = 0; ch < C; ch++) {
    switch (b->ddesc[ch].demod) {
    case KH_DEMOD_FM: b->lists[b->n_fm++] = ch; break;
    case KH_DEMOD_LINEAR:
      if (!b->ddesc[ch].pll) b->lists[C + b->n_lin++] = ch;
      break;
    case KH_DEMOD_AM: b->lists[2 * C + b->n_am++] = ch; break;
    default: break;
    }
  }
  b->n_lin_nopll = b->n_lin;
  for (int ch = 0; ch < C; ch++)
    if (b->ddesc[ch].demod == KH_DEMOD_LINEAR && b->ddesc[ch].pll) b->lists[C + b->n_lin++] = ch;
  // the bins a channel's filter output reads (filter.c:274-335: every in/out type combination stays within N_dec/2 of rotate)
  spec_mask_clear(&b->mask);
  // (a channel that was never set sits at rotate 0 and is computed like any other)
  for (int ch = 0; ch < C; ch++)
    spec_mask_add(&b->mask, b->N, (long long)b->chans[ch].rotate - b->ND / 2, (long long)b->chans[ch].rotate + b->ND / 2);
  b->mask.on = 1;
  hipStream_t st = b->stream;
  bool ok = KH_HIP(hipMemcpyAsync(b->d_chans, b->chans.data(), C * sizeof(ChanDesc), hipMemcpyHostToDevice, st));
  ok = ok && KH_HIP(hipMemcpyAsync(b->d_ddesc, b->ddesc.data(), C * sizeof(DemodDesc), hipMemcpyHostToDevice, st));
  ok = ok && KH_HIP(hipMemcpyAsync(b->d_lists, b->lists.data(), 3 * C * sizeof(int), hipMemcpyHostToDevice, st));
  ok = ok && KH_HIP(hipStreamSynchronize(st));
  if (ok) b->chans_dirty = false;
  return ok ? 0 : -1;
}

// The kernel sequence of one execute call, in two halves.  Both run eagerly, or under stream capture when
// kh_bank_execute builds the hipGraphs of this shape.
// FRONT: convert + mix (radio.c:136-191) and the forward transform (filter.c:212-238) into slot b->sp.
static bool n0_in_fft(const kh_bank *b);
static int n0_direct(const kh_bank *b);
// KA9Q_HIP_DIAG_SKIP (measurement only, results are garbage): 1 queues no back-half kernels, 2 no front-half kernels --
// what a step costs with one half's kernels alone, events and hand-overs as ever; 4: see kh_bank_execute
static int diag_skip() {
  static int const v = getenv("KA9Q_HIP_DIAG_SKIP") ? atoi(getenv("KA9Q_HIP_DIAG_SKIP")) : 0;
  return v;
}
static int enqueue_front(kh_bank *b, hipStream_t st, unsigned int nblocks, bool prof) {
  int const L = b->cfg.L, M = b->cfg.M, N = b->N, P = b->P, sp = b->sp;
  if (diag_skip() & 2) return 0;
  size_t const need = (size_t)nblocks * L;
  float2 *time = b->d_time[b->cur];
  if (prof) hipEventRecord(b->cur_ev[0], st);
  // the last M-1 samples of the launch are the next launch's overlap-save history (filter.c:230): ingest stores
  // them at the head of the other buffer as it goes, unless the history reaches back beyond this launch
  bool const fused_carry = (size_t)(M - 1) <= need;
  // large transforms read the raw ring themselves (fused convert + IF power + history) unless an oscillator is active
  bool const fused = b->fused_now;
  const unsigned char *const raw = b->d_raw + payload_bytes(b->cfg.payload, b->head); // the first unread sample of input 0
  FftIngest const fi{raw, b->raw_stride, (unsigned)payload_bytes(b->cfg.payload, (size_t)b->T * L - b->head), b->cfg.payload,
                     time, b->d_time[b->cur ^ 1], b->time_stride, b->d_idesc, b->d_energy[sp], M};
  if (!fused && launch_ingest(st, raw, b->raw_stride, b->cfg.payload, time + (M - 1), b->time_stride, b->d_idesc,
                              b->d_energy[sp], L, nblocks, P, fused_carry ? b->d_time[b->cur ^ 1] : nullptr,
                              need - (fused_carry ? (M - 1) : 0)))
    return -1;
  if (prof) hipEventRecord(b->cur_ev[1], st);
  // launch_forward_fft records the mid event (ev[2]) between the two passes itself
  N0Plan const plan = n0_plan(N, b->cfg.samprate, b->cfg.n0_low, b->cfg.n0_high);
  bool const n0_on = b->cfg.compute_n0 && plan.valid;
  N0Fuse nf{n0_on ? b->d_n0acc[sp] : nullptr, plan.start, plan.bincount, nullptr};
  size_t const PT = (size_t)P * nblocks; // (the regions of d_n0acc are laid out for this call's block count)
  if (n0_on && (n0_in_fft(b) || n0_direct(b))) nf.part1 = b->d_n0acc[sp] + PT * N0_SLOTS;
  if (n0_on && n0_direct(b)) {
    nf.sync = reinterpret_cast<unsigned *>(b->d_n0acc[sp] + PT * (N0_SLOTS + 2 * N0_GX));
    nf.direct = n0_direct(b);
    nf.spin = b->n0_spin;
    nf.stagger = b->n0_stagger;
    nf.ewin = b->d_n0acc[sp] + PT * (N0_SLOTS + 2 * N0_GX); // (bracket mode: the same words, as floats)
    nf.cand = b->d_n0acc[sp] + (size_t)n0_part_floats() * P * b->T;
  }
  // a bank that keeps only its channels' rows of the spectrum: the row pass finishes a block whose bracket fails itself
  bool const masked = b->masked_now && n0_on;
  if (masked) {
    nf.arrive = b->d_arrive;
    nf.host_fail = b->h_fail_dev;
  }
  if (launch_forward_fft(st, time, b->time_stride, 0, b->d_spec[sp], b->d_tmp, b->spec_stride, N, L, nblocks, P,
                         prof ? b->cur_ev[2] : nullptr, nf, fused ? &fi : nullptr, masked ? &b->mask : nullptr))
    return -1;
  // overlap-save history for the next call (filter.c:230), into the other buffer
  if (!fused && !fused_carry && launch_carry(st, time + need, b->d_time[b->cur ^ 1], b->time_stride * sizeof(float2),
                                   (size_t)(M - 1) * sizeof(float2), P))
    return -1;
  if (prof) hipEventRecord(b->cur_ev[3], st);
  return 0;
}
static bool n0_in_fft(const kh_bank *b) {
  static int const env = getenv("KA9Q_HIP_N0_IN_FFT") ? atoi(getenv("KA9Q_HIP_N0_IN_FFT")) : 1; // A/B switch
  return env && b->cfg.compute_n0 && b->N <= KH_MAX_SMALL_N;
}
// Four-step transforms: the row pass can take compute_n0's second iteration itself (N0Fuse, kh_internal.h), which
// saves reading the spectrum a second time (8 N bytes per block) and costs the row pass the rendezvous of a block's tiles
// (N = 2^20, T = 16: 43.8 -> 58.2 us alone, profiles/r03_f1_*).  Round 2 took it from N = 2^19 on: at N = 2^18 the re-read
// hid behind that round's latency-bound channel wavefronts.  With round 3's back half it does not any more (the re-read
// is most of the channel stage, 24.5 -> 13.8 us at configs[1], and the step 76-80 -> 63 us), so: every four-step size,
// unless KA9Q_HIP_N0_DIRECT says 0 (never) or 1 (whenever possible).
// Since then there is bracket mode as well (2, kh_n0.cuh), which needs no rendezvous: the row pass loses 3 us (N = 2^18) to
// 10 us (N = 2^20) less than with the rendezvous, the column pass adds the window's energy to its IF-power sum, and the
// channel stage carries the second step, 1 to 4 us.  Measured per step, rendezvous / bracket (two runs each, the bracket
// with its staging area private to a wavefront and the exclusion test only where it can bite): bench.py (N = 2^18)
// 63.8-64.1 / 61.9-62.4 us; N = 2^19, 256 NBFM channels 128.8-128.9 / 129.3-130.1; config 3's mixed share 191.6 / 191.6;
// N = 2^20 135-136 / 128.7-128.8 (512 channels), 196-198 / 190-195 (4096).  The bracket is the default at every four-step
// size: no workgroup ever waits for another, and which path a block takes depends on its data alone (the rendezvous'
// time-out does not).  KA9Q_HIP_N0_DIRECT = 1 selects the rendezvous, 0 neither.
static int n0_direct(const kh_bank *b) {
  if (b->n0_direct_mode == 0 || !b->cfg.compute_n0 || b->N <= KH_MAX_SMALL_N) return 0;
  if (!n0_plan(b->N, b->cfg.samprate, b->cfg.n0_low, b->cfg.n0_high).valid) return 0;
  return b->n0_direct_mode == 1 ? 1 : 2;
}
// BACK: compute_n0 (radio.c:422-465), per channel filter output (filter.c:240-384), demodulators, scaleclip.
static int enqueue_back(kh_bank *b, hipStream_t st, unsigned int nblocks, bool prof) {
  return enqueue_back_job(b, st, kh_bank::BackJob{nblocks, b->sp, b->fm_parity, b->fused_now, b->masked_now}, prof);
}
static int enqueue_back_job(kh_bank *b, hipStream_t st, kh_bank::BackJob const &job, bool prof) {
  unsigned int const nblocks = job.nblocks;
  int const L = b->cfg.L, N = b->N, P = b->P, C = b->C, sp = job.sp;
  if (diag_skip() & 1) return 0;
  N0Plan const plan = n0_plan(N, b->cfg.samprate, b->cfg.n0_low, b->cfg.n0_high);
  // compute_n0's second iteration (radio.c:451-462) only reads the spectrum: it rides along in the channel kernel's
  // launch when that kernel is the wave-resident one; otherwise it is a kernel of its own, in front.
  int const echunks = job.fused_now ? fused_ingest_tiles(N) : ingest_chunks(L);
  const float2 *n0spec = b->cfg.compute_n0 ? b->d_spec[sp] : nullptr;
  N0Pass1Args z;
  // (transforms that fit one workgroup have taken the second iteration themselves, from LDS: enqueue_front)
  bool const in_fft = n0_in_fft(b);
  // (four-step transforms in direct mode: the row pass has left the second iteration's partial sums as well, except for
  // blocks whose flag it raised; the iteration below then only runs for those)
  const unsigned *dsync = n0_direct(b) ? reinterpret_cast<const unsigned *>(b->d_n0acc[sp] + (size_t)P * nblocks * (N0_SLOTS + 2 * N0_GX))
                                       : nullptr;
  const float *dcand = n0_direct(b) == 2 ? b->d_n0acc[sp] + (size_t)n0_part_floats() * P * b->T : nullptr;
  bool const ride = !in_fft && b->ride_n0 && chan_wave_supported(b->ND, b->olen) &&
                    n0_pass1_args(&z, n0spec, b->spec_stride, N, plan, b->d_n0acc[sp], nblocks, P, dsync, dcand);
  // (a masked spectrum cannot be re-read: the row pass has finished the blocks whose bracket fails)
  int const no_reread = job.masked_now && n0spec != nullptr && plan.valid;
  if (ride) z.no_reread = no_reread;
  if (!ride && launch_n0(st, n0spec, b->spec_stride, N, b->cfg.samprate, plan, b->d_energy[sp], echunks, L, b->d_n0state,
                         b->d_n0acc[sp], b->d_istatus, nblocks, P, in_fft ? 1 : 0, nullptr, dsync, dcand, no_reread))
    return -1;
  if (prof) hipEventRecord(b->cur_ev[4], st);
  int taken = 0;
  // round 3's back half (kh_chan2 + kh_fm_snr_lanes + kh_fm_demod2) for N_dec = 512 / 1024; KA9Q_HIP_BACK2=0 and every
  // other shape take round 2's kernels
  bool const back2 = fm2_usable(b->ND, b->olen, (int)nblocks, b->n_fm) != 0;
  if (back2) {
    if (launch_chan2(st, b->d_spec[sp], b->spec_stride, N, b->d_chans, b->d_resp, b->ND, b->olen, b->d_bb, b->bb_stride, nblocks, C,
                     b->n_fm ? b->d_fmstat : nullptr, b->T, b->keep_baseband, ride ? &z : nullptr))
      return -1;
    taken = ride;
  } else if (launch_filter_output(st, b->d_spec[sp], b->spec_stride, N, 0, b->d_chans, b->d_resp, b->ND, b->olen, b->d_bb,
                                  b->bb_stride, nblocks, C,
                                  FmFront{b->n_fm ? b->d_fmstat : nullptr, b->d_fmwork, b->d_fmcarry, b->d_ddesc, b->T, b->bb_stride,
                                          b->keep_baseband},
                                  1, ride ? &z : nullptr, &taken))
    return -1;
  // kh_n0_final's work (one workgroup per input) rides in the FM finish launch when there is one of the wave form
  N0FinalArgs nf;
  bool const ride_final = ride && taken && b->n_fm > 0 && (back2 || fm_finish_wave_usable(b->olen, nblocks));
  if (ride && launch_n0(st, n0spec, b->spec_stride, N, b->cfg.samprate, plan, b->d_energy[sp], echunks, L, b->d_n0state,
                        b->d_n0acc[sp], b->d_istatus, nblocks, P, taken, ride_final ? &nf : nullptr, dsync, dcand, no_reread))
    return -1;
  if (prof) hipEventRecord(b->cur_ev[5], st);
  // A mixed bank's demodulators are independent kernels, each far from filling the chip (a workgroup per channel): the
  // linear / AM ones go to a stream of their own beside the FM demodulator and join before the half ends (config 3's
  // share, 128 NBFM + 64 AM + 64 USB: demod stage 94 us one after the other).
  // Not under hipGraph capture: the runtime picks the streams of a replayed graph's branches itself, and two branches that
  // land on one hardware queue with the join ahead of the side branch's kernels never finish (one full-suite run of the
  // graph tests with the fork inside the captured half timed out in kh_bank_sync).  Plain launches reach the queues in the
  // order below -- everything the join waits for is queued ahead of it -- whatever the mapping of streams to queues.
  bool fork = back2 && b->demod_fork && !b->use_graph && b->n_fm > 0 && b->n_lin + b->n_am > 0;
  if (fork && b->side == nullptr) {
    bool ok = b->side_use_prio ? KH_HIP(hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, b->side_prio))
                               : KH_HIP(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    ok = ok && KH_HIP(hipEventCreateWithFlags(&b->ev_fork, b->evflags)) && KH_HIP(hipEventCreateWithFlags(&b->ev_join, b->evflags));
    if (!ok) fork = false;
  }
  hipStream_t const sl = fork ? b->side : st;
  if (fork) {
    if (!KH_HIP(hipEventRecord(b->ev_fork, st)) || !KH_HIP(hipStreamWaitEvent(b->side, b->ev_fork, 0))) return -1;
  }
  if (back2 && b->n_fm > 0 &&
      launch_fm2(st, b->d_bb, b->bb_stride, b->olen, nblocks, b->d_ddesc, b->d_dstate, b->keep_audio ? b->d_audio : nullptr, b->d_pcm,
                 b->pcm_stride, b->cfg.pcm_big_endian, b->d_status, b->T,
                 b->n_fm == C ? nullptr : b->d_lists /* all FM: the list is the identity, and a load the kernel's first step can do without */,
                 b->n_fm, b->d_fmstat, b->d_decaypow,
                 job.fm_parity, ride_final ? &nf : nullptr,
                 b->use_graph ? nullptr : b->d_fm2x /* a captured launch would replay its epoch */,
                 (++b->fm2_epoch ? b->fm2_epoch : ++b->fm2_epoch)))
    return -1;
  if (launch_demod(sl, b->d_bb, b->bb_stride, b->olen, nblocks, C, b->d_ddesc, b->d_dstate,
                   b->keep_audio ? b->d_audio : nullptr, b->d_pcm, b->pcm_stride, b->cfg.pcm_big_endian, b->d_status,
                   b->T, b->d_lists, back2 ? 0 : b->n_fm, b->d_lists + C, b->n_lin, b->d_lists + 2 * C, b->n_am,
                   FmScratch{b->d_fmstat, b->d_fmwork, b->d_fmcarry, b->d_decaypow, job.fm_parity},
                   (ride_final && !back2) ? &nf : nullptr, b->n_lin_nopll))
    return -1;
  if (fork) {
    if (!KH_HIP(hipEventRecord(b->ev_join, b->side)) || !KH_HIP(hipStreamWaitEvent(st, b->ev_join, 0))) return -1;
  }
  if (b->rtp && launch_rtp_plan(st, b->d_pcm, b->pcm_stride, b->olen, nblocks, C, b->d_ddesc, b->d_rtp, b->rtp_stride,
                                b->d_rtpstate))
    return -1;
  if (prof) hipEventRecord(b->cur_ev[6], st);
  return 0;
}

// run one half eagerly, or capture it into a hipGraph the second time its shape is seen and replay it from then on
static int run_half(kh_bank *b, hipStream_t st, unsigned int nblocks, int (*half)(kh_bank *, hipStream_t, unsigned int, bool),
                    hipGraphExec_t *exec, int seen) {
  if (!b->use_graph || seen < 1) return half(b, st, nblocks, false);
  if (!*exec) {
    hipGraph_t graph = nullptr;
    bool ok = KH_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    if (ok) {
      int const erc = half(b, st, nblocks, false);
      ok = KH_HIP(hipStreamEndCapture(st, &graph)) && erc == 0 && graph != nullptr;
    }
    if (ok) ok = KH_HIP(hipGraphInstantiate(exec, graph, nullptr, nullptr, 0));
    if (graph) hipGraphDestroy(graph);
    if (!ok) {
      *exec = nullptr;
      return -2; // capture failed: the caller falls back to eager launches for this shape
    }
  }
  return KH_HIP(hipGraphLaunch(*exec, st)) ? 0 : -1;
}

// stage s = E[first[s]] .. E[first[s] + 1]; in the two-stream pipeline the n0 stage starts at the back half's own first
// event (E[KH_STAGE_COUNT]): E[3] belongs to the other stream
static void accumulate_stage_events(kh_bank *b, hipEvent_t *E, bool two_streams) {
  static int const first[KH_STAGE_COUNT] = {0, 1, 2, 3, 4, 5, -1};
  for (int s = 0; s < KH_STAGE_COUNT; s++) {
    if (first[s] < 0) continue;
    float ms = 0.f;
    hipEvent_t const e0 = (two_streams && s == KH_STAGE_N0) ? E[KH_STAGE_COUNT] : E[first[s]];
    if (hipEventElapsedTime(&ms, e0, E[first[s] + 1]) == hipSuccess) {
      b->stage_ms[s] += ms;
      b->stage_n[s]++;
    }
  }
  (void)hipGetLastError();
}
static int collect_stage_events(kh_bank *b, int slot) {
  if (!KH_HIP(hipEventSynchronize(b->evp[slot][6]))) return -1;
  accumulate_stage_events(b, b->evp[slot], true);
  b->evp_pending[slot] = false;
  return 0;
}

int kh_bank_execute(kh_bank *b, unsigned int nblocks) {
  if (!b) return -1;
  if (nblocks == 0 || nblocks > (unsigned)b->T) {
    set_error("kh_bank_execute: nblocks %u outside [1, %d]", nblocks, b->T);
    return -1;
  }
  if (ensure_init()) return -1;
  int const L = b->cfg.L, P = b->P;
  size_t const need = (size_t)nblocks * L;
  for (int i = 0; i < P; i++)
    if (b->fill[i] < need) {
      set_error("kh_bank_execute: input %d holds %zu samples, %zu needed", i, b->fill[i], need);
      return -1;
    }
  if (b->chans_dirty && (quiet(b) || upload_channels(b))) return -1;
  bool const prof = b->profiling == 1;                  // every stage alone, one stream
  bool const prof2 = b->profiling == 2 && b->pipeline;  // the two-stream pipeline, with events
  hipStream_t const sb = b->stream;                                // back half
  hipStream_t const sf = (prof || !b->pipeline) ? sb : b->front;   // front half
  int const sp = b->sp;
  // oscillator snapshot for this launch, staged through a pinned ring so that the host never waits for the device
  int const slot = b->idesc_slot;
  b->idesc_slot = (slot + 1) % kh_bank::IDESC_SLOTS;
  if (!KH_HIP(hipEventSynchronize(b->idesc_ev[slot]))) return -1; // the copy that last used this slot (long done)
  IngestDesc *id = b->h_idesc + (size_t)slot * P;
  for (int i = 0; i < P; i++) {
    HostOsc &lo = b->lo2[i], &dp = b->doppler[i];
    id[i].gain_factor = b->gain[i];
    id[i].pad = 0;
    id[i].lo2 = NcoArgs{lo.phase, lo.freq_now, lo.rate, !(lo.freq_now == 0 && lo.rate == 0 && lo.phase == 0), 0};
    id[i].doppler = NcoArgs{dp.phase, dp.freq_now, dp.rate, dp.freq_set != 0, 0}; // radio.c:180
    lo.advance(need);
    if (dp.freq_set != 0) dp.advance(need);
  }
  b->fused_now = b->fused_ingest && b->N > KH_MAX_SMALL_N;
  for (int i = 0; i < P; i++)
    if (id[i].lo2.active || id[i].doppler.active) b->fused_now = false;
  // rows of the spectrum no channel reads are not stored when nobody else wants them; built into the bracket form of the
  // row pass only (its second iteration never re-reads the spectrum: kh_n0_fix), so a bank without compute_n0 keeps storing all
  b->masked_now = !b->keep_spectrum && b->N > KH_MAX_SMALL_N && n0_direct(b) == 2 && b->d_arrive != nullptr;
  if (b->masked_now) {
    unsigned const f = __atomic_load_n(b->h_fail, __ATOMIC_RELAXED); // blocks the row pass had to finish alone, so far
    if (f != b->fail_seen) {
      b->fail_seen = f;
      if (b->sparse_hold_min) {
        bool const again = b->sparse_hold && b->launch_no - b->sparse_last_fail < 4ull * b->sparse_hold;
        b->sparse_hold = again ? std::min(2 * b->sparse_hold, 4096u) : b->sparse_hold_min;
        b->sparse_resume_at = b->launch_no + b->sparse_hold;
      }
      b->sparse_last_fail = b->launch_no;
    }
    if (b->launch_no < b->sparse_resume_at) b->masked_now = false;
  }
  b->launch_no++;
  if (sf != b->front && !KH_HIP(hipStreamSynchronize(b->front))) return -1; // an earlier pipelined front half
  // The back half that last read slot sp (nslots launches ago) must be done before the front half overwrites it.  The
  // HOST waits for it (KA9Q_HIP_SLOT_HOSTWAIT=0: the front stream does).  A caller that queues launches faster than the
  // GPU runs them -- any throughput loop -- finds that back half still pending, and a wait left to the front stream is a
  // barrier packet across hardware queues in front of every column pass: 3 us of a 58 us step at configs[1], 7 of 120 at
  // config 4's share, although the barrier is long satisfied when the queue reaches it (profiles/r03_step_anatomy.txt).
  // Waiting here costs nothing the GPU notices: the caller stays at most nslots launches ahead, i.e. two whole launches
  // are queued behind the running one.  (In the other mode the host asks first: a wait that is already satisfied would
  // still put its barrier into the queue.)
  static int const slot_hostwait = getenv("KA9Q_HIP_SLOT_HOSTWAIT") ? atoi(getenv("KA9Q_HIP_SLOT_HOSTWAIT")) : 1;
  if (slot_hostwait && sf != sb && b->slot_used[sp]) {
    if (!KH_HIP(hipEventSynchronize(b->ev_back_done[sp]))) return -1;
  } else if (sf != sb && b->slot_used[sp] && hipEventQuery(b->ev_back_done[sp]) != hipSuccess &&
             !KH_HIP(hipStreamWaitEvent(sf, b->ev_back_done[sp], 0)))
    return -1;
  (void)hipGetLastError(); // hipEventQuery's hipErrorNotReady is not an error
  // (a copy-engine operation between two kernels of a stream costs a hand-over in each direction: the descriptors
  // are uploaded only when they differ from what the device already holds, i.e. while an oscillator runs or after a
  // gain change)
  if (b->idesc_dev.size() != (size_t)P || memcmp(b->idesc_dev.data(), id, P * sizeof(IngestDesc)) != 0) {
    if (!KH_HIP(hipMemcpyAsync(b->d_idesc, id, P * sizeof(IngestDesc), hipMemcpyHostToDevice, sf))) return -1;
    if (!KH_HIP(hipEventRecord(b->idesc_ev[slot], sf))) return -1;
    b->idesc_dev.assign(id, id + P);
  }

  int rc = 0;
  bool const threaded = !prof && !prof2 && sf != sb && !b->use_graph && b->enq_thread;
  if (!threaded && quiet(b)) return -1; // (a launch after the mode has changed)
  if (prof) {
    b->cur_ev = b->ev;
    rc = enqueue_front(b, sf, nblocks, true);
    if (!rc) rc = enqueue_back(b, sb, nblocks, true);
  } else if (prof2) {
    if (b->evp_pending[sp] && collect_stage_events(b, sp)) return -1; // (this slot's launch of nslots launches ago)
    b->cur_ev = b->evp[sp];
    rc = enqueue_front(b, sf, nblocks, true);
    if (!rc && (!KH_HIP(hipEventRecord(b->ev_front_done[sp], sf)) || !KH_HIP(hipStreamWaitEvent(sb, b->ev_front_done[sp], 0)) ||
                !KH_HIP(hipEventRecord(b->evp[sp][KH_STAGE_COUNT], sb)))) // back half may start: the n0 stage's first event
      rc = -1;
    if (!rc) rc = enqueue_back(b, sb, nblocks, true);
    if (!rc) b->evp_pending[sp] = true;
  } else if (threaded) {
    rc = enqueue_front(b, sf, nblocks, false);
    if (!rc && !KH_HIP(hipEventRecord(b->ev_front_done[sp], sf))) rc = -1;
    // (the wait for that event, the back half and ev_back_done[sp]: enq_main)
    if (!rc) rc = enq_post(b, kh_bank::BackJob{nblocks, sp, b->fm_parity, b->fused_now, b->masked_now});
  } else {
    kh_bank::GraphEntry *g = nullptr;
    for (auto &x : b->graphs)
      if (x.nblocks == nblocks && x.cur == b->cur && x.fmpar == b->fm_parity && x.sp == sp && x.raw == b->d_raw + payload_bytes(b->cfg.payload, b->head) &&
          x.pipelined == (sf != sb) && x.fused == b->fused_now && x.masked == b->masked_now)
        g = &x;
    if (!g) { // first sight: run eagerly (this also warms the twiddle tables, which allocate)
      b->graphs.push_back(kh_bank::GraphEntry{nblocks, b->cur, b->fm_parity, sp, b->d_raw + payload_bytes(b->cfg.payload, b->head), sf != sb, b->fused_now, b->masked_now, 0, nullptr, nullptr});
      g = &b->graphs.back();
    }
    rc = run_half(b, sf, nblocks, enqueue_front, &g->front, g->seen);
    if (rc == -2) {
      g->seen = -1;
      rc = enqueue_front(b, sf, nblocks, false);
    }
    // (KA9Q_HIP_DIAG_SKIP & 4: the back stream does NOT wait for the front half -- a planted hazard that
    // tools/stress_handover.py has to find)
    if (!rc && sf != sb &&
        (!KH_HIP(hipEventRecord(b->ev_front_done[sp], sf)) ||
         (!(diag_skip() & 4) && !KH_HIP(hipStreamWaitEvent(sb, b->ev_front_done[sp], 0)))))
      rc = -1;
    if (!rc) {
      rc = run_half(b, sb, nblocks, enqueue_back, &g->back, g->seen);
      if (rc == -2) {
        g->seen = -1;
        rc = enqueue_back(b, sb, nblocks, false);
      }
    }
    if (!rc && g->seen >= 0) g->seen++;
  }
  if (rc) return rc;
  if (sf != sb) {
    if (!threaded && !KH_HIP(hipEventRecord(b->ev_back_done[sp], sb))) return -1;
    b->slot_used[sp] = true;
  }
  if (b->n_fm) b->fm_parity ^= 1;
  b->cur ^= 1;
  b->last_sp = sp;
  b->sp = (sp + 1) % b->nslots;
  // consume the raw ring: the head moves, the samples stay (ring_compact moves them when a write needs the room)
  bool empty = true;
  for (int i = 0; i < P; i++) {
    b->fill[i] -= need;
    if (b->fill[i]) empty = false;
  }
  b->head = empty ? 0 : b->head + need;
  b->last_blocks = nblocks;
  if (prof) {
    if (!KH_HIP(hipStreamSynchronize(sb))) return -1;
    accumulate_stage_events(b, b->ev, false);
  }
  return 0;
}

int kh_bank_sync(kh_bank *b) {
  if (!b || quiet(b)) return -1;
  if (!KH_HIP(hipStreamSynchronize(b->front))) return -1;
  return KH_HIP(hipStreamSynchronize(b->stream)) ? 0 : -1;
}

static int read_back(kh_bank *b, void *dst, const void *src, size_t bytes) {
  if (quiet(b)) return -1;
  if (!KH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b->stream))) return -1;
  return KH_HIP(hipStreamSynchronize(b->stream)) ? 0 : -1;
}
int kh_bank_read_baseband(kh_bank *b, unsigned int ch, kh_cf32 *out, size_t nsamples) {
  if (!b || ch >= (unsigned)b->C || !out || nsamples > b->bb_stride) return -1;
  return read_back(b, out, b->d_bb + ch * b->bb_stride, nsamples * sizeof(float2));
}
int kh_bank_read_pcm(kh_bank *b, unsigned int ch, int16_t *out, size_t nwords) {
  if (!b || ch >= (unsigned)b->C || !out || nwords > b->pcm_stride) return -1;
  return read_back(b, out, b->d_pcm + ch * b->pcm_stride, nwords * sizeof(int16_t));
}
int kh_bank_read_audio(kh_bank *b, unsigned int ch, float *out, size_t nwords) {
  if (!b || ch >= (unsigned)b->C || !out || nwords > b->pcm_stride) return -1;
  return read_back(b, out, b->d_audio + ch * b->pcm_stride, nwords * sizeof(float));
}
int kh_bank_read_status(kh_bank *b, unsigned int ch, struct kh_status *out, size_t nblocks) {
  if (!b || ch >= (unsigned)b->C || !out || nblocks > (size_t)b->T) return -1;
  return read_back(b, out, b->d_status + (size_t)ch * b->T, nblocks * sizeof(kh_status));
}
int kh_bank_read_input_status(kh_bank *b, unsigned int input, struct kh_input_status *out, size_t nblocks) {
  if (!b || input >= (unsigned)b->P || !out || nblocks > b->last_blocks) return -1;
  // records of one launch are packed [input][last_blocks]
  return read_back(b, out, b->d_istatus + (size_t)input * b->last_blocks, nblocks * sizeof(kh_input_status));
}
int kh_bank_read_spectrum(kh_bank *b, unsigned int input, unsigned int block, kh_cf32 *out) {
  if (!b || input >= (unsigned)b->P || block >= (unsigned)b->T || !out) return -1;
  if (b->masked_now) {
    set_error("kh_bank_read_spectrum: the last launch kept only its channels' rows of the spectrum (kh_bank_keep_spectrum(bank, 0))");
    return -1;
  }
  return read_back(b, out, b->d_spec[b->last_sp] + input * b->spec_stride + (size_t)block * b->N, b->N * sizeof(float2));
}
int kh_bank_read_response(kh_bank *b, unsigned int ch, kh_cf32 *out, float *noise_gain_out) {
  if (!b || ch >= (unsigned)b->C) return -1;
  if (noise_gain_out) *noise_gain_out = b->noise_gains[ch];
  if (!out) return 0;
  return read_back(b, out, b->d_resp + (size_t)ch * b->ND, b->ND * sizeof(float2));
}
int kh_bank_pcm_device_ptr(kh_bank *b, void **dptr, size_t *words_per_channel) {
  if (!b || !dptr) return -1;
  *dptr = b->d_pcm;
  if (words_per_channel) *words_per_channel = b->pcm_stride;
  return 0;
}
void *kh_bank_stream(kh_bank *b) { return b ? (void *)b->stream : nullptr; }

int kh_bank_keep_audio(kh_bank *b, int enable) {
  if (!b || quiet(b)) return -1;
  b->keep_audio = enable != 0;
  drop_graphs(b);
  return 0;
}
int kh_bank_rtp_enable(kh_bank *b, int enable) {
  if (!b || quiet(b)) return -1;
  b->rtp = enable != 0;
  drop_graphs(b);
  return 0;
}
int kh_bank_rtp_set_state(kh_bank *b, unsigned int ch, const struct kh_rtp_state *s) {
  if (!b || !s || ch >= (unsigned)b->C || quiet(b)) return -1;
  if (!KH_HIP(hipMemcpyAsync(b->d_rtpstate + ch, s, sizeof(*s), hipMemcpyHostToDevice, b->stream))) return -1;
  return KH_HIP(hipStreamSynchronize(b->stream)) ? 0 : -1;
}
int kh_bank_read_rtp(kh_bank *b, unsigned int ch, struct kh_rtp_chunk *out, size_t max_chunks, size_t *nchunks,
                     struct kh_rtp_state *after) {
  if (!b || ch >= (unsigned)b->C) return -1;
  if (!b->rtp) {
    set_error("kh_bank_read_rtp: the plan is not enabled (kh_bank_rtp_enable)");
    return -1;
  }
  DemodDesc const &d = b->ddesc[ch];
  int const chans = d.demod == KH_DEMOD_LINEAR ? d.channels : 1;
  size_t const cpb = ((size_t)b->olen * chans + 479) / 480;
  size_t const n = d.demod == KH_DEMOD_NONE ? 0 : b->last_blocks * cpb;
  if (nchunks) *nchunks = n;
  if (out) {
    if (max_chunks < n) {
      set_error("kh_bank_read_rtp: %zu chunks, room for %zu", n, max_chunks);
      return -1;
    }
    if (n && read_back(b, out, b->d_rtp + (size_t)ch * b->rtp_stride, n * sizeof(kh_rtp_chunk))) return -1;
  }
  if (after && read_back(b, after, b->d_rtpstate + ch, sizeof(*after))) return -1;
  return 0;
}
int kh_bank_keep_spectrum(kh_bank *b, int enable) {
  if (!b || quiet(b)) return -1;
  b->keep_spectrum = enable != 0;
  drop_graphs(b);
  return 0;
}
int kh_bank_keep_baseband(kh_bank *b, int enable) {
  if (!b || quiet(b)) return -1;
  b->keep_baseband = enable != 0;
  drop_graphs(b);
  return 0;
}
int kh_bank_use_graph(kh_bank *b, int enable) {
  if (!b || quiet(b)) return -1;
  b->use_graph = enable != 0;
  if (!enable) drop_graphs(b);
  return 0;
}
int kh_bank_profile(kh_bank *b, int enable) {
  if (!b) return -1;
  if (kh_bank_sync(b)) return -1; // the modes order their work differently
  if (enable < 0 || enable > 2) return -1;
  for (int k = 0; k < kh_bank::MAX_SLOTS; k++) b->evp_pending[k] = false;
  if (enable == 2)
    for (auto &set : b->evp)
      for (auto &e : set)
        if (!e && !KH_HIP(hipEventCreate(&e))) return -1;
  b->profiling = enable;
  memset(b->stage_ms, 0, sizeof(b->stage_ms));
  memset(b->stage_n, 0, sizeof(b->stage_n));
  return 0;
}
int kh_bank_stage_times(kh_bank *b, float ms[KH_STAGE_COUNT], unsigned int launches[KH_STAGE_COUNT]) {
  if (!b) return -1;
  for (int k = 0; k < kh_bank::MAX_SLOTS; k++)
    if (b->evp_pending[k] && collect_stage_events(b, k)) return -1;
  for (int s = 0; s < KH_STAGE_COUNT; s++) {
    if (ms) ms[s] = b->stage_ms[s];
    if (launches) launches[s] = b->stage_n[s];
  }
  return 0;
}
const char *kh_stage_kernel_name(int stage) {
  switch (stage) {
  case KH_STAGE_INGEST: return "kh_ingest";                   // zero when the column pass reads the raw ring itself
  case KH_STAGE_FFT1: return "kh_fft_cols|kh_fft_fwd_lds";    // first (or only) pass of the forward transform
  case KH_STAGE_FFT2: return "kh_fft_rows";
  case KH_STAGE_N0: return "kh_n0_pass1+kh_n0_final";         // zero when the second iteration rides in kh_back_a
  case KH_STAGE_CHAN: return "kh_chan2|kh_back_a|kh_chan_wave64|kh_filter_out";
  case KH_STAGE_DEMOD: return "kh_fm_demod2|kh_fm_finish_w|kh_demod_linear|kh_demod_am";
  case KH_STAGE_CARRY: return "kh_carry";
  }
  return "";
}

} // extern "C"
