// kh_internal.h -- private declarations shared by the kernels and the host-side C ABI of libka9q_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define KH_MAX_SMALL_N 8192 // largest transform done by one workgroup in LDS

namespace kh {

struct N0Pass1Args; // kh_n0.cuh
struct N0FinalArgs; // kh_n0.cuh

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
bool check_hip(hipError_t e, const char *what);
#define KH_HIP(call) ::kh::check_hip((call), #call)
int ensure_init();
int initialised_device(); // the device the library runs on, or -1 before ka9q_hip_init / the first lazy init
hipStream_t default_stream();

// ---------------------------------------------------------------- twiddle tables (device resident, cached per size)
// exp(-2 pi i k / n), k < n, computed in double on the host
const float2 *twiddle_table(int n);
// four-step twiddle W_N^idx = coarse[idx >> 10] * fine[idx & 1023]
struct BigTwiddle {
  const float2 *coarse; // exp(-2 pi i (j<<10) / N), j < max(1, N >> 10)
  const float2 *fine;   // exp(-2 pi i k / N), k < 1024
};
BigTwiddle big_twiddle(int n);
// The column pass's four-step twiddles W_N^{n2 k1}, k1 = c + RA*(8q + j), split as W^{n2 (c + 8 RA q)} W^{n2 RA j} and laid
// out so that a lane reads its 8 + RB/8 factors with 16-byte loads: uj[n2][8], tq[n2][RA][RB/8]
struct ColsTwiddle {
  const float2 *uj, *tq;
};
ColsTwiddle cols_twiddle(int N2, int RA, int RB);
// kh_chan2's stage-A twiddles exp(+2 pi i l c / ND), laid out [c][l], c < 32, l < ND/32
const float2 *chan2_twiddle(int ND);

// ---------------------------------------------------------------- transforms of any 2,3,5,7-smooth size (kh_anyfft.cuh)
struct AnyPlan {
  int n, npass;      // npass == 0: n is not 2,3,5,7-smooth, or outside [16, KH_MAX_SMALL_N]
  int radix[14];     // product n; 4s first, then 2, 3, 5, 7
};
AnyPlan make_any_plan(int n);
inline bool is_pow2_int(int v) { return v > 0 && !(v & (v - 1)); }

// ---------------------------------------------------------------- per-channel device records
struct ChanDesc {     // one per channel, read by the filter-output kernel
  int input;          // index of the input spectrum
  int rotate;         // filter.c:240
  int out_type;       // enum filtertype
  int fm_stats;       // 1: the epilogue leaves fm.c:53-70's block statistics for the FM kernels
};

// FM path scratch.  The filter-output epilogue (kh_filter_out) leaves, per FM channel and block, the block
// statistics and the block-local part of the demodulator; kh_fm_finish folds in what depends on earlier blocks.
struct FmBlockStat {
  float bb_power, variance, snr; // fm.c:53-70
  float angsum;                  // sum of the discriminator angles of samples 1 .. olen-1 (fm.c:96)
  float2 first, last;            // y[0], y[olen-1]: sample 0's angle needs the block before (fm.c:86-87)
};
// kh_fm_demod2 with a channel's blocks split over several workgroups: a block's integrator carry-out for a zero carry-in
// (bits of a double), and the epoch (launch number, never 0) of the launch that left it; [nchan][stat_stride], written and
// read with agent-scope atomics only
struct Fm2Carry {
  unsigned long long B;
  unsigned epoch, pad;
};
struct FmWork { // per sample: discriminator angle and the de-emphasis integrator run from a zero carry-in
  float ang, s0;
};
struct DemodDesc;
struct FmFront {      // handed to launch_filter_output; stat == nullptr: no FM channels
  FmBlockStat *stat;  // [nchan][stat_stride]
  FmWork *work;       // [nchan][work_stride]
  double *carry;      // [nchan][stat_stride]: carry-out of the integrator for carry-in zero
  const DemodDesc *descs;
  int stat_stride;
  size_t work_stride;
  int write_baseband; // 0: FM channels do not store their complex baseband
};
struct FmScratch {    // handed to launch_demod
  FmBlockStat *stat;
  FmWork *work;
  double *carry;
  const double *decay_pow; // 0.99376949^k, k <= olen
  int parity;         // which DemodState FM slot this launch reads
};

struct NcoArgs {      // one oscillator over one launch; phase(i) = phase0 + i*freq + rate*i*(i+1)/2 (cycles)
  double phase0, freq, rate;
  int active;         // 0: multiply skipped (phasor identically 1, or osc.freq == 0 for Doppler: radio.c:180)
  int pad;
};
struct IngestDesc {   // one per input
  NcoArgs doppler, lo2;
  float gain_factor;
  int pad;
};

struct DemodDesc {    // one per channel, read by the demodulator kernels
  int demod;          // enum kh_demod
  int flat;
  float headroom;
  float fm_gain;      // fm.c:47
  float dsamprate;    // fm.c:27
  int pll, square, agc, env, channels;
  float hangtime, recovery_rate;
  float out_samprate;
  float am_hang_samples;    // am.c:80
  float pll_integrator_gain, pll_prop_gain; // osc.c:77-78
  int lock_limit;     // linear.c:36
  double shift_freq;  // cycles per output sample
};

struct DemodState {   // one per channel, device resident, carried from launch to launch
  // fm.c:26,32,33; two slots: a launch reads slot `parity` and leaves the next launch's state in the other
  float2 fm_state[2];
  float fm_lastaudio[2];
  int fm_squelch_open[2];
  // linear.c
  int hangcount, lock_count, pll_lock, vco_steps;
  float pll_integrator, snr, agc_gain;
  double vco_re, vco_im;       // pll.vco.phasor
  double vco_freq;             // pll.vco.freq
  double shift_re, shift_im;   // demod->shift.phasor
  int shift_steps;
  // am.c
  int am_hangcount;
  float am_dc, am_recovery_per_sample;
};

// ---------------------------------------------------------------- launchers (kh_kernels.hip)
// forward transform of `nblocks` overlapping blocks per input.  time: cf32 (or f32 when real_input) stream,
// block b of input i starts at time + i*in_stride + b*L (elements).  spec: i*spec_in_stride + b*N.
// tmp: scratch of the same size as spec (only used when N > KH_MAX_SMALL_N).
// n0: the epilogue of the last pass also leaves compute_n0's first-iteration partial sums (radio.c:450-461):
// n0_partials(N) floats per (input, block) at n0.part + (input*nblocks + block)*N0_SLOTS.  n0.part == nullptr: none.
constexpr int N0_SLOTS = 64;
struct N0Fuse {
  float *part;
  int start, bincount; // radio.c:440-447: bins (start + n) mod N, n < bincount
  // N <= KH_MAX_SMALL_N only: the block's whole spectrum is in the workgroup's LDS, so the second iteration
  // (radio.c:451-462: bins below twice the first mean) is taken there as well and the spectrum is never re-read.
  // part1 = part + ninputs*nblocks*N0_SLOTS (kh_n0.cuh's layout: 2*64 floats per (input, block)); nullptr: not wanted.
  float *part1 = nullptr;
  // N > KH_MAX_SMALL_N, direct mode: the row pass takes the second iteration itself, from the registers that still
  // hold its part of the spectrum, so the spectrum is not read a second time.  Its threshold is twice the mean of ALL
  // row tiles' first-iteration sums of the block: every workgroup publishes its sum (part, agent-scope store) and
  // counts itself in (sync[N0_SYNC_STRIDE pt]), then waits -- for a bounded time -- until the block's other tiles have done the
  // same.  A workgroup whose wait runs out raises sync[N0_SYNC_STRIDE pt + 1]; the second iteration of that block is then taken by
  // re-reading the spectrum (kh_n0_pass1 / kh_back_a), as without direct mode.  part1 receives the {sum, count} pair
  // of every row tile (kh_n0_final's layout).  The column pass clears both words of its blocks.
  unsigned *sync = nullptr;
  int direct = 0;
  int spin = 1000; // polls of ~0.1 us before a workgroup gives up (KA9Q_HIP_N0_SPIN; 0 sends every block to the re-read)
  int stagger = 0; // direct mode: a workgroup of block b starts (b mod 8) * stagger * ~0.25 us late (KA9Q_HIP_N0_STAGGER; see kh_fft_rows)
  // N > KH_MAX_SMALL_N, bracket mode (direct == 2; kh_n0.cuh "bracket"): no workgroup waits for another.  The column pass
  // leaves the energy of the block's N window samples (ewin: one partial per column tile in the block's N0_SYNC_STRIDE
  // words), which bounds the threshold from above by Parseval; the row pass sums what lies below the bracket's lower
  // edge and stores the few bins inside the bracket (cand); the exact threshold is applied to those when it is known.
  float *ewin = nullptr, *cand = nullptr;
  // a masked spectrum (SpecMask): per (input, block) N0_SYNC_STRIDE words -- [0] row tiles that have arrived, [1] spare,
  // [2] wavefronts whose candidates outgrew their slots; zeroed by the column pass.  host_fail (may be null): a word in
  // pinned host memory that counts the blocks whose bracket failed (kh_fft_rows, n0_repair_block)
  unsigned *arrive = nullptr, *host_fail = nullptr;
  int narrow = 0; // (set by launch_rows) the excluded bins all lie in the spectrum's first or last row of N1 bins: only two of a lane's outputs can be excluded
};
constexpr int N0_SYNC_STRIDE = 64; // 32-bit words per (input, block) in N0Fuse::sync: lines of its own
int n0_partials(int N);
// Which rows of the four-step spectrum X[k1 + N1*k2] (row = k2, N1 bins) the row pass stores.  A bank whose spectrum
// nobody else asks for (kh_bank_keep_spectrum(bank, 0)) needs only the rows its channels' filter outputs read
// (filter.c:274-335: N_dec bins around `rotate`); compute_n0 (radio.c:422-465) then has to be finished from the row
// pass's registers (bracket mode) and, for a block the bracket fails, by kh_n0_fix from the transform's scratch.
// A row-pass wavefront holds rows 4*wave .. 4*wave + 3 (+ RA*d, d < RB): need[wave] bit d says whether it stores them.
struct SpecMask {
  unsigned need[8];
  int on; // 0: every row is stored
};
void spec_mask_clear(SpecMask *m);
// the rows that hold bins lo .. hi (inclusive; any integers, taken modulo N) of a four-step transform of N points
void spec_mask_add(SpecMask *m, int N, long long lo, long long hi);
bool spec_mask_has_row(SpecMask const *m, int N, int row);
// fi != nullptr (N > KH_MAX_SMALL_N only): the first pass reads the raw ring itself instead of `time` -- wire format
// -> float x gain_factor (radio.c:136-178; no oscillator: the caller uses launch_ingest when one is active), leaves
// fused_ingest_tiles(N) IF-power partial sums per (input, block) and stores the last M-1 samples of the launch in
// hist_out (filter.c:230).  hist_in holds the M-1 samples that precede the launch.
struct FftIngest {
  const void *raw;
  size_t raw_stride_bytes;
  unsigned raw_bytes;
  int payload;
  const float2 *hist_in;
  float2 *hist_out;
  size_t hist_stride;
  const IngestDesc *desc;
  float *energy_part;
  int M;
};
int fused_ingest_tiles(int N);
// mask (may be null; four-step sizes only): only these rows of the spectrum are stored.  Needs compute_n0 in bracket mode
// (n0.direct == 2) with n0.arrive and n0.part1: a block whose bracket fails is finished by the row pass itself
// (n0_repair_block), and the second step must be told not to re-read the spectrum (N0Pass1Args::no_reread).
int launch_forward_fft(hipStream_t st, const void *time, size_t in_stride, int real_input, float2 *spec, float2 *tmp,
                       size_t spec_in_stride, int N, int L, int nblocks, int ninputs, hipEvent_t mid_event, N0Fuse n0,
                       const FftIngest *fi = nullptr, const SpecMask *mask = nullptr);
// bin-select x response x inverse transform.  out: cf32 (or f32 when out_type REAL) at
// out + ch*out_stride + b*olen.
// complex_out: 1 if no channel has out_type REAL (known on the host): shapes that have one then take the
// wave-resident kernel of kh_chan_kernels.hip.
// n0 (may be null): compute_n0's second iteration rides along in the same launch when the wave-resident kernel is
// taken; *n0_taken says whether it did.
int launch_filter_output(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int in_real,
                         const ChanDesc *chans, const float2 *response, int ND, int olen, void *out, size_t out_stride,
                         int nblocks, int nchan, FmFront const &fm, int complex_out, const N0Pass1Args *n0 = nullptr,
                         int *n0_taken = nullptr);
int chan_wave_supported(int ND, int olen);
// the 32-points-per-lane form (kh_chan2_kernels.hip): baseband of every channel; stat (may be null): FM channels' block
// statistics with fm_snr's ARGUMENT in .snr (launch_fm2 turns it into the SNR)
int launch_chan2(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, const ChanDesc *chans,
                 const float2 *response, int ND, int olen, float2 *out, size_t out_stride, int nblocks, int nchan,
                 FmBlockStat *stat, int stat_stride, int store_all, const N0Pass1Args *n0);
// (store_all == 0: FM channels' blocks whose squelch is certainly closed are not written)
int launch_chan_wave(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, const ChanDesc *chans,
                     const float2 *response, int ND, int olen, float2 *out, size_t out_stride, int nblocks, int nchan,
                     FmFront const &fm, const N0Pass1Args *n0);
// wire format -> cf32 * gain * Doppler * LO2 and per-block energy (radio.c:136-191)
// energy_part: ingest_chunks(L) floats per (input, block), summed by launch_n0
int ingest_chunks(int L);
int n0_part_floats();
// hist_dst (may be null): the other time buffer; samples n >= hist_from of the launch are also stored at its head
int launch_ingest(hipStream_t st, const void *raw, size_t raw_stride_bytes, int payload, float2 *time, size_t time_stride,
                  const IngestDesc *desc, float *energy_part, int L, int nblocks, int ninputs, float2 *hist_dst,
                  size_t hist_from);
// radio.c:422-465 + 197-202.  n0_plan: the bin range of radio.c:430-447; launch_n0: second iteration, smoothing and
// IF power (the first iteration's partial sums come from launch_forward_fft)
struct N0Plan {
  int valid, start, bincount;
};
N0Plan n0_plan(int N, int samprate, float low, float high);
// fills *z with the second iteration's arguments; 0 if there is nothing to do (compute_n0 off, empty bin range)
// sync (may be null): direct mode's words (N0Fuse); the iteration then only runs for blocks whose flag is raised
// cand (may be null): bracket mode's candidates; sync then holds the column pass's window energies
int n0_pass1_args(N0Pass1Args *z, const float2 *spec, size_t spec_in_stride, int N, N0Plan plan, float *part, int nblocks,
                  int ninputs, const unsigned *sync = nullptr, const float *cand = nullptr);
// pass1_done: the second iteration has already run (inside launch_chan_wave's kernel)
int launch_n0(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int samprate, N0Plan plan,
              const float *energy_part, int echunks /*partial sums per (input, block)*/, int L, float *n0_state,
              float *part /*n0_part_floats() per (input,block)*/, void *input_status, int nblocks, int ninputs,
              int pass1_done = 0, N0FinalArgs *final_out = nullptr, const unsigned *sync = nullptr, const float *cand = nullptr,
              int no_reread = 0);
int n0_cols_tiles(int N);     // column tiles (= window energy partials) per block of a four-step transform
int n0_rows_waves(int N);     // wavefronts per row tile
size_t n0_cand_floats(int N); // bracket mode: candidate floats per (input, block); 0 for single-workgroup transforms
// final_out != nullptr: kh_n0_final is not launched; *final_out receives its arguments for launch_demod
// demodulators + scaleclip
int launch_demod(hipStream_t st, float2 *baseband, size_t bb_stride, int olen, int nblocks, int nchan,
                 const DemodDesc *desc, DemodState *state, float *audio, int16_t *pcm, size_t pcm_stride,
                 int pcm_big_endian, void *status /*kh_status[nchan][max_blocks]*/, int status_stride,
                 const int *fm_list, int n_fm, const int *lin_list, int n_lin, const int *am_list, int n_am,
                 const FmScratch &fm, const N0FinalArgs *n0final = nullptr, int n_lin_nopll = 0);
// (lin_list: the channels without PLL first; n_lin_nopll of them)
// n0final (may be null): kh_n0_final's work rides in the FM finish launch; only valid if fm_finish_wave_usable() and n_fm > 0
int fm_finish_wave_usable(int olen, int nblocks);
// round 3's FM path behind launch_chan2 (kh_fm2_kernels.hip): kh_fm_snr_lanes + kh_fm_demod2.  stat: what launch_chan2
// left, [channel][status_stride]; n0final (may be null) rides in the first of the two launches
int fm2_usable(int ND, int olen, int nblocks, int n_fm);
int launch_fm2(hipStream_t st, const float2 *baseband, size_t bb_stride, int olen, int nblocks, const DemodDesc *desc,
               DemodState *state, float *audio, int16_t *pcm, size_t pcm_stride, int pcm_big_endian, void *status,
               int status_stride, const int *fm_list, int n_fm, FmBlockStat *stat, const double *decay_pow, int parity,
               const N0FinalArgs *n0final, Fm2Carry *xch = nullptr, unsigned epoch = 0);

// audio.c:32-135: one kh_rtp_chunk per PCM chunk of <= 480 words (plan: [nchan][plan_stride]), kh_rtp_state per channel
int launch_rtp_plan(hipStream_t st, const int16_t *pcm, size_t pcm_stride, int olen, int nblocks, int nchan,
                    const DemodDesc *descs, void *plan, int plan_stride, void *states);
// response design on the device (filter.c:481-568, 573-628, 631-702)
int launch_make_kaiser(hipStream_t st, float *window_dev, int M, float beta);
int launch_window_filter(hipStream_t st, float2 *resp_dev, int L, int M, float beta, int real_filter);
int launch_brickwall(hipStream_t st, float2 *resp_dev, int ND, float low, float high, float gain);
int launch_noise_gain(hipStream_t st, const float2 *resp_dev, int count, float scale, float *out_dev);
// filter.c:719-727 over n samples in place; st: {phase, step, dcstate, bw} on the device, updated
struct NotchState {
  double pr, pi, sr, si;
  float dr, di, bw;
  int pad;
};
int launch_notch(hipStream_t st, NotchState *state, float2 *buf, int n);
// decimate.c
int launch_decimate(hipStream_t st, float *rails, size_t rail_stride, int nrails, int chunk, int log_decimate,
                    int stage_threshold, void *hb15_states, float *hb3_states);
bool decimate_can_fuse(int log_decimate, int stage_threshold);
int launch_decimate_fused(hipStream_t st, const float *in, size_t in_stride, float *out, size_t out_stride, int nrails, int chunk,
                          int log_decimate, int stage_threshold, const void *hb15_in, void *hb15_out, const float *hb3_in,
                          float *hb3_out);
// copy count elements of elem_bytes from src to dst for each of n rows (overlap-save history carry, filter.c:230)
int launch_carry(hipStream_t st, const void *src, void *dst, size_t row_stride_bytes, size_t bytes, int nrows);
int launch_host_pull(hipStream_t st, const void *src_host_mapped, void *dst, size_t row_stride_bytes, size_t bytes, int nrows);

// ---------------------------------------------------------------- packet.c (kh_packet_kernels.hip, kh_packet.cpp)
struct AfskState { // the locals of decode_task (packet.c:410-435) and sp->rtp_state_out, one per session
  int symphase; // (the tone oscillators are a function of the session's sample count: kh_afsk_mix)
  int pad;
  float2 mark_acc, space_acc, mark_off, space_off;
  float last_val, mid_val;
  int frame_bit, flagsync, ones;
  unsigned out_timestamp, out_seq;
  unsigned long long out_packets, out_bytes;
  unsigned char hdlc_frame[1024];
};
int launch_afsk_ingest(hipStream_t st, const void *pcm_be, size_t pcm_stride, float *time, size_t time_stride, int hist,
                       int count, int nsess);
int launch_afsk_decode(hipStream_t st, const float2 *y, size_t y_stride, float4 *ms, size_t ms_stride, int nsamp, int nsess,
                       AfskState *states, void *frames, int max_frames, int *nframes, void *bits, int max_bits,
                       int *nbits, unsigned long long sample_base);

// pl.c: tone tables and per-session carry between launches
struct ToneTable {
  double freq[64];   // cycles/sample as set_osc receives them (pl.c:320,327-328): 55 PL, 4 DTMF low, 4 DTMF high
  double rate[64];   // effective sweep in cycles/sample^2 (pl.c:320 passes 9: arg(cispi(18)) / 2 pi)
  float tone_hz[64]; // PL_tones, pl.c:36-44
};
struct TonesState {
  double2 pl_part[55]; // PL integrators of the 50 ms intervals seen so far in the running 250 ms interval
  int pl_windows;      // how many of them (0..4)
  int pad;
};
int launch_tones(hipStream_t st, const void *pcm_be, size_t pcm_stride, const ToneTable *tab, void *part, int max_windows,
                 int nwindows, int nsess, TonesState *states, void *pl, int max_pl, int *npl, void *dt,
                 unsigned long long base);

// front-end corrections (kh_frontend_kernels.hip, kh_frontend.cpp)
struct FeParams { // what one block is corrected with (the state before that block's update)
  float dc_re, dc_im, gain_i, gain_q, secphi, tanphi;
  int rotate_phase, pad;
};
} // namespace kh
struct kh_fe_state;
namespace kh {
int fe_chunks(int samples);
int fe_pack_groups(int chunk);
int launch_frontend(hipStream_t st, int funcube, const void *raw, int samples, int nblocks, int offset, float rate_factor,
                    kh_fe_state *state, long long *part, FeParams *params, float *out_i, float *out_q, void *out_be);
int launch_fe_pack(hipStream_t st, int pt, const float *ip, const float *qp, int chunk, float atten, void *out,
                   float *energy_part);
// kh_hackrf_pack_noalloc: address of its static device scratch on `device` (current when called), once per device
int pack_noalloc_setup(int device);

} // namespace kh
