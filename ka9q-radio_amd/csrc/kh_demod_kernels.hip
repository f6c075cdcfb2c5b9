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
#include <type_traits>
#include "kh_internal.h"
#include <cstdlib>
#include "kh_fft.cuh"
#include "kh_fm.cuh"
#include "kh_ingest.cuh"
#include "kh_n0.cuh"
#include "../../include/ka9q_hip.h"

namespace kh {

__device__ __forceinline__ float wave_sum(float v) { return n0_wave_sum(v); }

// ================================================================ ingest: radio.c:136-191
// grid (chunks per block, nblocks, ninputs), 256 threads, INGEST_CHUNK/256 samples per thread
// algorithmic bytes per sample: wire bytes read + 8 written
// hist_dst: the other time buffer.  The last hist_len samples of the launch (n >= hist_from) are stored there as
// well, at its head: they are the overlap-save history of the next launch (filter.c:230's memmove).
constexpr int INGEST_CHUNK = 4096;
__global__ __launch_bounds__(256) void kh_ingest(const unsigned char *__restrict__ raw, size_t raw_stride_bytes,
                                                 int payload, float2 *__restrict__ time, size_t time_stride,
                                                 const IngestDesc *__restrict__ descs, float *__restrict__ energy_part,
                                                 int L, int nblocks, float2 *__restrict__ hist_dst, size_t hist_from) {
  __shared__ float part[4];
  int const b = blockIdx.y, i = blockIdx.z;
  const IngestDesc *dp = descs + i;
  float const gain_factor = dp->gain_factor;
  int const dop_active = dp->doppler.active, lo_active = dp->lo2.active;
  const unsigned char *src = raw + (size_t)i * raw_stride_bytes;
  float2 *dst = time + (size_t)i * time_stride;
  float2 *hdst = hist_dst ? hist_dst + (size_t)i * time_stride : nullptr;
  float energy = 0.f;
#pragma unroll 4
  for (int u = 0; u < INGEST_CHUNK / 256; u++) {
    int const k = blockIdx.x * INGEST_CHUNK + u * 256 + threadIdx.x;
    if (k < L) {
      size_t const n = (size_t)b * L + k;
      float2 x = load_raw(src, n, payload, gain_factor);
      if (dop_active) { // radio.c:180-184: complex float *= complex double
        double c, s;
        nco_phasor(dp->doppler, (unsigned)n, c, s);
        double const re = (double)x.x * c - (double)x.y * s;
        double const im = (double)x.x * s + (double)x.y * c;
        x = make_float2((float)re, (float)im);
      }
      if (lo_active) { // radio.c:187
        double c, s;
        nco_phasor(dp->lo2, (unsigned)n, c, s);
        double const re = (double)x.x * c - (double)x.y * s;
        double const im = (double)x.x * s + (double)x.y * c;
        x = make_float2((float)re, (float)im);
      }
      energy += x.x * x.x + x.y * x.y; // radio.c:188
      dst[n] = x;
      if (hdst != nullptr && n >= hist_from) hdst[n - hist_from] = x;
    }
  }
  energy = wave_sum(energy);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = energy;
  __syncthreads();
  if (threadIdx.x == 0)
    energy_part[((size_t)i * nblocks + b) * gridDim.x + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

int ingest_chunks(int L) { return (L + INGEST_CHUNK - 1) / INGEST_CHUNK; }

int launch_ingest(hipStream_t st, const void *raw, size_t raw_stride_bytes, int payload, float2 *time,
                  size_t time_stride, const IngestDesc *desc, float *energy_part, int L, int nblocks, int ninputs,
                  float2 *hist_dst, size_t hist_from) {
  if (nblocks <= 0 || ninputs <= 0) return 0;
  hipLaunchKernelGGL(kh_ingest, dim3(ingest_chunks(L), nblocks, ninputs), dim3(256), 0, st,
                     static_cast<const unsigned char *>(raw), raw_stride_bytes, payload, time, time_stride, desc,
                     energy_part, L, nblocks, hist_dst, hist_from);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ================================================================ compute_n0: radio.c:422-465
// Iteration 0 of radio.c:451-462 (sum of |X|^2 over every bin outside the passband) is left by the forward
// transform's epilogue (kh_fft_rows / kh_fft_fwd_lds) as n0_partials(N) partial sums per (input, block).
// kh_n0_pass1 is iteration 1: sum and count of the bins below twice that mean, N0_GX partials per (input, block).
// kh_n0_final adds them up in a fixed order (no atomics: reproducible run to run) and applies the smoothing of
// radio.c:197-202 block by block.
// algorithmic bytes: 8*bincount
__global__ __launch_bounds__(256) void kh_n0_pass1(N0Pass1Args Z) {
  __shared__ float sh[N0_SH];
  n0_pass1_body(Z, blockIdx.x, sh);
}

// (n0_final_body: kh_n0.cuh)
__global__ __launch_bounds__(N0_FT) void kh_n0_final(N0FinalArgs Z) {
  __shared__ float sv[N0_MAXB], se[N0_MAXB];
  n0_final_body(Z, blockIdx.x, sv, se);
}

// per (input, block): first-iteration partials | {sum, count} pairs of the second | direct mode's two words (N0Fuse)
int n0_part_floats() { return N0_SLOTS + 2 * N0_GX + N0_SYNC_STRIDE; } // (bracket mode's candidates: n0_cand_floats(N) more)

N0Plan n0_plan(int N, int samprate, float low, float high) {
  N0Plan p{0, 0, 0};
  int const low_n = (int)((N * low) / samprate); // radio.c:430-431: float arithmetic, truncated
  int const high_n = (int)((N * high) / samprate);
  int const bincount = N - (high_n - low_n);
  p.valid = !(bincount < 0 || bincount > N) && bincount > 0;
  int start = high_n + 1;
  if (start >= N)
    start -= N;
  else if (start < 0)
    start += N;
  p.start = start;
  p.bincount = bincount;
  return p;
}

// second iteration + smoothing + IF power; the forward transform has left the first-iteration partials in `part`.
// n0_pass1_args: what the second iteration needs (valid == 0: compute_n0 off or no bins); the bank hands it to
// launch_chan_wave so that the iteration shares the channel kernel's launch, and then calls launch_n0 with
// pass1_done = 1 for the rest.
int n0_pass1_args(N0Pass1Args *z, const float2 *spec, size_t spec_in_stride, int N, N0Plan plan, float *part, int nblocks,
                  int ninputs, const unsigned *sync, const float *cand) {
  if (spec == nullptr || !plan.valid) return 0;
  // (bracket mode: a workgroup per row tile; they sit at the end of the channel kernel's launch, so no more than needed)
  *z = N0Pass1Args{spec, spec_in_stride, N, plan.start, plan.bincount, part, nblocks, ninputs, n0_partials(N),
                   cand ? n0_partials(N) : N0_GX, sync, cand, n0_cols_tiles(N), n0_partials(N), n0_rows_waves(N)};
  return 1;
}
int launch_n0(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, int samprate, N0Plan plan,
              const float *energy_part, int echunks, int L, float *n0_state, float *part, void *input_status, int nblocks,
              int ninputs, int pass1_done, N0FinalArgs *final_out, const unsigned *sync, const float *cand, int no_reread) {
  if (nblocks <= 0 || ninputs <= 0) return 0;
  if (nblocks > N0_MAXB) {
    set_error("at most %d blocks per launch", N0_MAXB);
    return -1;
  }
  kh_input_status *status = static_cast<kh_input_status *>(input_status);
  int const have_n0 = spec != nullptr;
  N0Pass1Args z;
  int const valid = n0_pass1_args(&z, spec, spec_in_stride, N, plan, part, nblocks, ninputs, sync, cand);
  if (valid) z.no_reread = no_reread;
  if (valid && !pass1_done)
    hipLaunchKernelGGL(kh_n0_pass1, dim3(z.gx * nblocks * ninputs), dim3(256), 0, st, z);
  N0FinalArgs const F{part, energy_part, echunks, L, N, samprate, valid, have_n0, n0_state, status, nblocks, ninputs};
  if (final_out) { // the caller lets it ride in the FM finish launch (launch_demod)
    *final_out = F;
    return 0;
  }
  hipLaunchKernelGGL(kh_n0_final, dim3(ninputs), dim3(N0_FT), 0, st, F);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// (scaleclip / pcm_word, audio.c:22-28,50,101: kh_fm.cuh)

// ================================================================ FM: fm.c:43-139 in two parallel steps
//  (1) kh_filter_out's epilogue (fm_front, kh_fm.cuh), one workgroup per (channel, block), while the block is still
//      in LDS: bb_power, fm_variance and the Rice-corrected SNR (fm.c:53-70); the conjugate-product discriminator
//      and threshold-extension weight of samples 1.. (fm.c:85-93); the de-emphasis integrator run from a ZERO
//      carry-in (fm.c:110-111).  Nothing in it depends on another block.
//  (2) kh_fm_finish, one workgroup per (channel, block): the squelch (fm.c:73-78, which looks back at most one
//      block, so every workgroup replays it from the per-block SNRs); sample 0's discriminator output, which needs
//      the last sample of the block before; the carries of the earlier blocks folded in (s_n = s0_n + carry *
//      decay^n, exact by linearity of fm.c:110-111); scaleclip, output level, frequency offset and the
//      order-dependent peak-deviation update of fm.c:97-102.
constexpr int FMB_NT = 256;

// fm.c:73-78 in closed form.  `squelch_open = 2` spans two blocks and is decremented once per open block, so
// after block j the counter is 1 if snr_j > 4, else max(counter_{j-1} - 1, 0), which is 0 for every j >= 1
// (counter_{j-1} <= 1) and max(so_in - 1, 0) for j = 0.  Block b demodulates iff snr_b > 4 or the counter
// before it is positive.  so_in = counter carried in from the previous launch.
__device__ __forceinline__ int fm_counter_after(const FmBlockStat *__restrict__ fs, int j, int so_in) {
  if (j < 0) return so_in;
  if (fs[j].snr > 4.f) return 1;
  if (j == 0) return so_in - 1 > 0 ? so_in - 1 : 0;
  return 0;
}
__device__ __forceinline__ int fm_is_open(const FmBlockStat *__restrict__ fs, int j, int so_in) {
  return fs[j].snr > 4.f || fm_counter_after(fs, j - 1, so_in) > 0;
}
// sample 0 of block j (fm.c:86-93,110): p = y_j[0] * conj(state), state = last sample of block j-1 if that block was
// demodulated, 0 if its squelch was closed (fm.c:123), the carried state for j == 0.  Returns the integrator
// increment ang*.114*gain*amp; ang0 gets the angle.
__device__ __forceinline__ double fm_sample0(const FmBlockStat *__restrict__ fs, int j, float2 state_in, int so_in,
                                             float gain, float &ang0) {
  float2 prev = state_in;
  if (j > 0) prev = fm_is_open(fs, j - 1, so_in) ? fs[j - 1].last : make_float2(0.f, 0.f);
  float2 const p = cmulc(fs[j].first, prev);
  ang0 = atan2f(p.y, p.x);
  float amp = hypotf(p.x, p.y) / fs[j].variance;
  if (amp > 1) amp = 1;
  return (double)ang0 * .114 * (double)gain * (double)amp;
}

// algorithmic bytes per (channel, block): 8*olen work read + 2*olen PCM written
__global__ __launch_bounds__(FMB_NT) void kh_fm_finish(int olen, int nblocks, const DemodDesc *__restrict__ descs,
                                                       DemodState *__restrict__ states, int par,
                                                       const FmBlockStat *__restrict__ fmstat, int stat_stride,
                                                       const FmWork *__restrict__ work, size_t work_stride,
                                                       const double *__restrict__ carry,
                                                       const double *__restrict__ decay_pow,
                                                       float *__restrict__ audio, int16_t *__restrict__ pcm,
                                                       size_t pcm_stride, int big_endian, kh_status *__restrict__ status,
                                                       int status_stride, const int *__restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) float2 al[]; // (ang, lastaudio seen by the sample), olen entries
  __shared__ double sh_c[2];
  __shared__ float sh_ang0;
  __shared__ float red[FMB_NT / 64];
  int const tid = threadIdx.x;
  int const ch = list[blockIdx.x], b = blockIdx.y;
  const DemodDesc *dd = descs + ch;
  DemodState *st = states + ch;
  const FmBlockStat *fs = fmstat + (size_t)ch * stat_stride;
  int const flat = dd->flat;
  float const dsamprate = dd->dsamprate, gain = dd->fm_gain;
  int const so_in = st->fm_squelch_open[par];
  float2 const state_in = st->fm_state[par];
  int const open = fm_is_open(fs, b, so_in); // uniform: every lane evaluates it
  int const so = fm_counter_after(fs, b, so_in);
  float const lastaudio_in = st->fm_lastaudio[par];
  if (tid < 64) {
    // lastaudio on entry to this block: the carried value and the carry-outs of the earlier open blocks, each
    // decayed over the blocks in between (fm.c:110-111,130): c_b = A^b c_0 + sum_{j<b} A^(b-1-j) E_j with
    // E_j = carry_j + A x0_j for a demodulated block (its own sample 0 enters ahead of the zero-carry run), 0 else
    double acc = 0.0;
    if (!flat) {
      double const Ablk = decay_pow[olen];
      const double *cb = carry + (size_t)ch * stat_stride;
      for (int j = tid - 1; j < b; j += 64) { // j = -1 stands for the carried lastaudio
        double v;
        if (j < 0)
          v = (double)lastaudio_in;
        else if (fm_is_open(fs, j, so_in)) {
          float a0;
          v = cb[j] + Ablk * fm_sample0(fs, j, state_in, so_in, gain, a0);
        } else
          v = 0.0;
        int m = b - 1 - j; // blocks to decay over
        double p = 1.0, q = Ablk;
        while (m) {
          if (m & 1) p *= q;
          q *= q;
          m >>= 1;
        }
        acc += v * p;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    } else {
      acc = (double)lastaudio_in;
    }
    if (tid == 0) {
      float a0 = 0.f;
      double x0 = 0.0;
      if (open) x0 = fm_sample0(fs, b, state_in, so_in, gain, a0);
      sh_c[0] = acc;                     // lastaudio before sample 0
      sh_c[1] = flat ? acc : acc + x0;   // and after its increment
      sh_ang0 = a0;
    }
  }
  __syncthreads();
  double const c_in = sh_c[0], c0 = sh_c[1];
  float const ang0 = sh_ang0;
  const FmWork *wi = work + (size_t)ch * work_stride + (size_t)b * olen;
  size_t const obase = (size_t)b * olen;
  float *ao = audio ? audio + (size_t)ch * pcm_stride + obase : nullptr;
  int16_t *po = pcm + (size_t)ch * pcm_stride + obase;
  float lv = 0.f;
  for (int n = tid; n < olen; n += FMB_NT) {
    float s1, la = 0.f, ang = 0.f;
    if (open) {
      FmWork const w = wi[n];
      ang = n ? w.ang : ang0;
      if (flat) {
        s1 = n ? w.s0 : ang0 * gain; // fm.c:105
        la = (float)c_in;           // lastaudio never changes in flat mode
      } else {
        s1 = (float)((double)w.s0 + c0 * decay_pow[n]);
        if (n > 0) { // lastaudio when sample n is processed = previous sample * decay (fm.c:110-111)
          float const sp = (float)((double)wi[n - 1].s0 + c0 * decay_pow[n - 1]);
          la = (float)((double)sp * FM_DECAY);
        }
      }
    } else {
      s1 = flat ? 0.f : (float)(c_in * decay_pow[n]); // fm.c:124-132: the decaying carry alone
    }
    lv += s1 * s1;
    if (ao) ao[n] = s1;
    po[n] = pcm_word(s1, big_endian);
    al[n] = make_float2(ang, la);
  }
  float const level = block_sum<FMB_NT>(lv, red) / olen; // fm.c:135-138 (its barriers also publish al[])
  if (tid >= 64) return;
  // ---- wave 0: status record; fm.c:81-83,97-102's order-dependent peak-deviation update.
  // pdev_pos is a running maximum (prefix max over lanes); pdev_neg moves to ang_k at the first later sample
  // that is not a new maximum and whose lastaudio is below the current pdev_neg: a find-first over the
  // window, repeated from the hit with the new value.  Same result as the serial loop, sample for sample.
  int const lane = tid;
  float pdev_pos = 0.f, pdev_neg = 0.f;
  if (open) {
    float runmax = -INFINITY;
    for (int w0 = 0; w0 < olen; w0 += 64) {
      int const k = w0 + lane;
      float2 const v = k < olen ? al[k] : make_float2(-INFINITY, INFINITY);
      // exclusive prefix maximum of ang over the block so far
      float inc = v.x;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        float const p = __shfl_up(inc, o, 64);
        if (lane >= o) inc = fmaxf(inc, p);
      }
      float exc = __shfl_up(inc, 1, 64);
      exc = lane == 0 ? runmax : fmaxf(exc, runmax);
      bool const up = v.x > exc;          // fm.c:99: ang > pdev_pos
      float thr = up ? INFINITY : v.y;    // fm.c:101 is only reached when not a new maximum
      if (k == 0) {                       // fm.c:97-98
        thr = INFINITY;
        pdev_neg = v.x;
      }
      if (w0 == 0) pdev_neg = __shfl(pdev_neg, 0, 64);
      unsigned long long live = ~0ull;
      while (true) {
        unsigned long long const hit = __ballot(thr < pdev_neg) & live;
        if (!hit) break;
        int const i = __ffsll((long long)hit) - 1;
        pdev_neg = __shfl(v.x, i, 64);    // fm.c:102
        live = i == 63 ? 0ull : (~0ull << (i + 1));
      }
      runmax = fmaxf(runmax, __shfl(inc, 63, 64));
    }
    pdev_pos = runmax;
  }
  if (lane == 0) {
    kh_status r;
    r.bb_power = fs[b].bb_power;
    r.snr = fs[b].snr;
    r.foffset = NAN; // fm.c:116,121 leave both untouched while the squelch is closed
    r.pdeviation = NAN;
    if (open) {
      float const avg_f = (fs[b].angsum + ang0) / olen; // fm.c:96,114
      r.foffset = (float)((double)(dsamprate * avg_f) * (0.5 * M_1_PI)); // fm.c:116
      pdev_pos -= avg_f;
      pdev_neg -= avg_f;
      r.pdeviation = (float)((double)(dsamprate * fmaxf(pdev_pos, -pdev_neg)) * (0.5 * M_1_PI)); // fm.c:121
    }
    r.cphase = 0.f;
    r.lock_timer = 0.f;
    r.pll_lock = 0;
    r.output_level = level;
    r.agc_gain = NAN;
    r.squelch_open = so;
    r.reserved[0] = r.reserved[1] = 0.f;
    status[(size_t)ch * status_stride + b] = r;
    if (b == nblocks - 1) { // state carried to the next launch (fm.c:26,32,33)
      double c = c_in;
      if (!flat) c = open ? decay_pow[olen] * c0 + carry[(size_t)ch * stat_stride + b] : decay_pow[olen] * c_in;
      // written to the other parity slot: workgroups of this launch may still be reading the current one
      st->fm_lastaudio[par ^ 1] = (float)c;
      st->fm_squelch_open[par ^ 1] = so;
      st->fm_state[par ^ 1] = open ? fs[b].last : make_float2(0.f, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same step with ONE wavefront per (channel, block): lane l owns samples K l .. K l + K - 1 (K =
// ceil(olen / 64)), the earlier blocks' carries are folded by one lane each and summed across the wave, the
// peak-deviation replay uses a wave-wide prefix maximum and a lane-by-lane pass; no LDS.  Four such waves share a workgroup; they never
// wait for each other.  This is the form the bank uses (kh_fm_finish above remains for olen > 1024 or more than 64 blocks per launch).
struct FinishArgs {
  int olen, nblocks;
  const DemodDesc *descs;
  DemodState *states;
  int par;
  const FmBlockStat *fmstat;
  int stat_stride;
  const FmWork *work;
  size_t work_stride;
  const double *carry;
  const double *decay_pow;
  float *audio;
  int16_t *pcm;
  size_t pcm_stride;
  int big_endian;
  kh_status *status;
  int status_stride;
  const int *list;
  int n_fm;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int K>
__device__ __forceinline__ void fm_finish_wave(FinishArgs const &F, int q, int lane) {
  int const olen = F.olen;
  int const ch = F.list[q % F.n_fm], b = q / F.n_fm;
  const DemodDesc *dd = F.descs + ch;
  DemodState *st = F.states + ch;
  const FmBlockStat *fs = F.fmstat + (size_t)ch * F.stat_stride;
  const double *cb = F.carry + (size_t)ch * F.stat_stride;
  int const par = F.par;
  int const flat = dd->flat;
  float const dsamprate = dd->dsamprate, gain = dd->fm_gain;
  int const so_in = st->fm_squelch_open[par];
  float2 const state_in = st->fm_state[par];
  float const lastaudio_in = st->fm_lastaudio[par];
  double const Ablk = F.decay_pow[olen];
  // this lane's samples: issued before anything that depends on the statistics
  const FmWork *wi = F.work + (size_t)ch * F.work_stride + (size_t)b * olen;
  int const n0 = K * lane;
  FmWork w[K];
  double dp[K];
  if ((K % 2) == 0 && (olen % 2) == 0) { // 16-byte loads
#pragma unroll
    for (int j = 0; j < K; j += 2) {
      int const n = n0 + j;
      float4 const v = n < olen ? *reinterpret_cast<const float4 *>(wi + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      double2 const d = n < olen ? *reinterpret_cast<const double2 *>(F.decay_pow + n) : make_double2(0.0, 0.0);
      w[j] = FmWork{v.x, v.y};
      w[j + 1] = FmWork{v.z, v.w};
      dp[j] = d.x;
      dp[j + 1] = d.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; j++) {
      int const n = n0 + j;
      w[j] = n < olen ? wi[n] : FmWork{0.f, 0.f};
      dp[j] = n < olen ? F.decay_pow[n] : 0.0;
    }
  }
  // lane j holds the statistics of block j (nblocks <= 64): everything that looks at another block is a shuffle
  bool const have = lane < F.nblocks;
  FmBlockStat sj = have ? fs[lane] : FmBlockStat{0.f, 1.f, 0.f, 0.f, make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
  double const cj = have ? cb[lane] : 0.0;
  // fm.c:73-78 in closed form (see fm_counter_after): counter after block j
  int cnt_j = sj.snr > 4.f ? 1 : (lane == 0 ? (so_in - 1 > 0 ? so_in - 1 : 0) : 0);
  int cnt_prev = __shfl_up(cnt_j, 1, 64);
  if (lane == 0) cnt_prev = so_in;
  int const open_j = sj.snr > 4.f || cnt_prev > 0;
  int open_prev = __shfl_up(open_j, 1, 64);
  float2 prev = make_float2(__shfl_up(sj.last.x, 1, 64), __shfl_up(sj.last.y, 1, 64));
  if (!open_prev) prev = make_float2(0.f, 0.f); // fm.c:123
  if (lane == 0) prev = state_in;
  // sample 0 of block j (fm.c:86-93,110): p = y_j[0] conj(previous sample)
  float2 const p0 = cmulc(sj.first, prev);
  float const ang0_j = atan2f(p0.y, p0.x);
  float amp0 = hypotf(p0.x, p0.y) / sj.variance;
  if (amp0 > 1) amp0 = 1;
  double const x0_j = (double)ang0_j * .114 * (double)gain * (double)amp0;
  int const open = __shfl(open_j, b, 64), so = __shfl(cnt_j, b, 64);
  // lastaudio on entry to this block (fm.c:110-111,130): c_b = A^b c_0 + sum_{j<b} A^(b-1-j) E_j, E_j = carry_j + A x0_j for
  // a demodulated block (its own sample 0 enters ahead of the zero-carry run), 0 for a closed one
  double acc = (double)lastaudio_in;
  if (!flat) {
    double v = 0.0;
    int m = 0; // blocks to decay over
    if (lane < b) {
      m = b - 1 - lane;
      if (open_j) v = cj + Ablk * x0_j;
    } else if (lane == b) { // the carried lastaudio rides in the first lane that has no block to fold
      m = b;
      v = (double)lastaudio_in;
    }
    double pw = 1.0, qq = Ablk;
    while (m > 0) {
      if (m & 1) pw *= qq;
      qq *= qq;
      m >>= 1;
    }
    acc = wave_sum_f64(v * pw);
  }
  float const ang0 = open ? __shfl(ang0_j, b, 64) : 0.f;
  double const x0 = open ? __shfl(x0_j, b, 64) : 0.0;
  double const c_in = acc;                   // lastaudio before sample 0
  double const c0 = flat ? acc : acc + x0;   // and after its increment
  float const sb_bb = __shfl(sj.bb_power, b, 64), sb_snr = __shfl(sj.snr, b, 64), sb_angsum = __shfl(sj.angsum, b, 64);
  float2 const sb_last = make_float2(__shfl(sj.last.x, b, 64), __shfl(sj.last.y, b, 64));
  double const cb_b = __shfl(cj, b, 64);

  // ---- fm.c:105-111,124-138: audio, scaleclip, output level
  float ang[K], s1[K], la[K];
  float lv = 0.f;
#pragma unroll
  for (int j = 0; j < K; j++) {
    int const n = n0 + j;
    float a = 0.f, s = 0.f;
    if (open) {
      a = n ? w[j].ang : ang0;
      if (flat)
        s = n ? w[j].s0 : ang0 * gain; // fm.c:105
      else
        s = (float)((double)w[j].s0 + c0 * dp[j]);
    } else {
      s = flat ? 0.f : (float)(c_in * dp[j]); // fm.c:124-132: the decaying carry alone
    }
    if (n >= olen) s = 0.f;
    ang[j] = a;
    s1[j] = s;
    lv += s * s;
  }
  { // lastaudio when sample n is processed = previous sample * decay (fm.c:110-111); flat: it never changes
    float const prev_lane = __shfl_up(s1[K - 1], 1, 64);
#pragma unroll
    for (int j = 0; j < K; j++) {
      float const sp = j ? s1[j - 1] : prev_lane;
      la[j] = flat ? (float)c_in : (float)((double)sp * FM_DECAY);
    }
  }
  size_t const obase = (size_t)b * olen + n0;
  int16_t *po = F.pcm + (size_t)ch * F.pcm_stride + obase;
  float *ao = F.audio ? F.audio + (size_t)ch * F.pcm_stride + obase : nullptr;
  if ((K % 8) == 0 && (olen % 8) == 0) { // 16-byte stores
    if (n0 < olen) {
#pragma unroll
      for (int j = 0; j < K; j += 8) {
        unsigned wd[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
          wd[u] = (unsigned)(unsigned short)pcm_word(s1[j + 2 * u], F.big_endian) |
                  ((unsigned)(unsigned short)pcm_word(s1[j + 2 * u + 1], F.big_endian) << 16);
        *reinterpret_cast<uint4 *>(po + j) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; j++)
      if (n0 + j < olen) po[j] = pcm_word(s1[j], F.big_endian);
  }
  if (ao) {
#pragma unroll
    for (int j = 0; j < K; j++)
      if (n0 + j < olen) ao[j] = s1[j];
  }
  float const level = wave_sum(lv) / olen; // fm.c:135-138

  // ---- fm.c:97-102, order dependent: pdev_pos is a running maximum; pdev_neg moves to ang_k at the first later
  // sample that is not a new maximum and whose lastaudio is below the current pdev_neg, then the search goes on
  // from there with the new value.  Same result as the serial loop, sample for sample.
  float pdev_pos = 0.f, pdev_neg = 0.f;
  if (open) {
    float lmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < K; j++)
      if (n0 + j < olen) lmax = fmaxf(lmax, ang[j]);
    float inc = lmax; // inclusive prefix maximum over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float const p = __shfl_up(inc, o, 64);
      if (lane >= o) inc = fmaxf(inc, p);
    }
    float run = __shfl_up(inc, 1, 64);
    if (lane == 0) run = -INFINITY;
    pdev_pos = __shfl(inc, 63, 64);
    float thr[K]; // lastaudio where fm.c:101 is reached, +inf elsewhere
#pragma unroll
    for (int j = 0; j < K; j++) {
      int const n = n0 + j;
      bool const up = ang[j] > run; // fm.c:99: ang > pdev_pos
      thr[j] = (up || n == 0 || n >= olen) ? INFINITY : la[j];
      if (n < olen) run = fmaxf(run, ang[j]);
    }
    float const c_first = __shfl(ang[0], 0, 64); // fm.c:97-98
    // fm.c:101-102 is a chain through the block: pdev_neg = ang_n whenever lastaudio_n < pdev_neg (on a typical signal
    // every other sample moves it).  Lane l maps the value entering its K samples to the value leaving them,
    // g_l(c): for j < K: if (thr[j] < c) c = ang[j]; it hands c on unchanged when c <= min_j thr[j].  The values entering
    // all 64 lanes are found by relaxation: start from "everything passes through", evaluate every g_l on its current
    // guess (in parallel), and give lane l+1 the output of the nearest earlier lane that did not pass its input through
    // (a segmented scan), until nothing changes.  A fixed point satisfies c_{l+1} = g_l(c_l) for every lane, i.e. it IS
    // the serial result, sample for sample; after t rounds lanes 0..t are final, so 64 rounds bound it, and real
    // signals settle in 3 to 14 (a replay lane by lane cost 64 x (2K + 2) instructions).
    float mlane = INFINITY;
#pragma unroll
    for (int j = 0; j < K; j++) mlane = fminf(mlane, thr[j]);
    float cin = c_first, last_val = c_first;
    for (int it = 0; it < 64; it++) {
      float c = cin;
#pragma unroll
      for (int j = 0; j < K; j++)
        if (thr[j] < c) c = ang[j];
      int has = !(cin <= mlane); // this lane's output is its own (not the input handed on)
      int val = __float_as_int(c);
      // inclusive scan over the 64 lanes of "the nearest lane at or before this one that has a value", in DPP moves
      // (no LDS round trips): shifts by 1, 2, 4, 8 inside the rows of 16, then lane 15 of rows 0 and 2 into rows 1 and 3,
      // then lane 31 into the upper half.  Lanes a move does not reach receive has = 0, the identity.
#define KH_PDEV_STEP(CTRL, ROWS)                                                            \
      {                                                                                       \
        int const ph = __builtin_amdgcn_update_dpp(0, has, CTRL, ROWS, 0xf, false);           \
        int const pv = __builtin_amdgcn_update_dpp(0, val, CTRL, ROWS, 0xf, false);           \
        val = has ? val : pv;                                                                 \
        has = has ? has : ph;                                                                 \
      }
      KH_PDEV_STEP(0x111, 0xf) // row_shr:1
      KH_PDEV_STEP(0x112, 0xf) // row_shr:2
      KH_PDEV_STEP(0x114, 0xf) // row_shr:4
      KH_PDEV_STEP(0x118, 0xf) // row_shr:8
      KH_PDEV_STEP(0x142, 0xa) // row_bcast:15 into rows 1, 3
      KH_PDEV_STEP(0x143, 0xc) // row_bcast:31 into rows 2, 3
#undef KH_PDEV_STEP
      int const hp = __builtin_amdgcn_update_dpp(0, has, 0x138, 0xf, 0xf, false); // wave_shr:1 (lane 0 keeps 0)
      int const np_ = __builtin_amdgcn_update_dpp(0, val, 0x138, 0xf, 0xf, false);
      float const ncin = hp ? __int_as_float(np_) : c_first;
      last_val = __builtin_amdgcn_readlane(has, 63) ? __int_as_float(__builtin_amdgcn_readlane(val, 63)) : c_first;
      bool const changed = __float_as_int(ncin) != __float_as_int(cin);
      cin = ncin;
      if (!__builtin_amdgcn_ballot_w64(changed)) break;
    }
    pdev_neg = last_val;
  }
  if (lane == 0) {
    kh_status r;
    r.bb_power = sb_bb;
    r.snr = sb_snr;
    r.foffset = NAN; // fm.c:116,121 leave both untouched while the squelch is closed
    r.pdeviation = NAN;
    if (open) {
      float const avg_f = (sb_angsum + ang0) / olen; // fm.c:96,114
      r.foffset = (float)((double)(dsamprate * avg_f) * (0.5 * M_1_PI)); // fm.c:116
      pdev_pos -= avg_f;
      pdev_neg -= avg_f;
      r.pdeviation = (float)((double)(dsamprate * fmaxf(pdev_pos, -pdev_neg)) * (0.5 * M_1_PI)); // fm.c:121
    }
    r.cphase = 0.f;
    r.lock_timer = 0.f;
    r.pll_lock = 0;
    r.output_level = level;
    r.agc_gain = NAN;
    r.squelch_open = so;
    r.reserved[0] = r.reserved[1] = 0.f;
    F.status[(size_t)ch * F.status_stride + b] = r;
    if (b == F.nblocks - 1) { // state carried to the next launch (fm.c:26,32,33)
      double c = c_in;
      if (!flat) c = open ? Ablk * c0 + cb_b : Ablk * c_in;
      // written to the other parity slot: waves of this launch may still be reading the current one
      st->fm_lastaudio[par ^ 1] = (float)c;
      st->fm_squelch_open[par ^ 1] = so;
      st->fm_state[par ^ 1] = open ? sb_last : make_float2(0.f, 0.f);
    }
  }
}

// workgroups [0, nfin) carry blockDim.x / 64 finish wavefronts each (eight: fewer, larger workgroups dispatch faster); workgroups nfin + i (if any) are kh_n0_final's work for input i
template <int K>
__global__ __launch_bounds__(1024) void kh_fm_finish_w(FinishArgs F, int npairs, int nfin, N0FinalArgs Z) {
  __shared__ float sv[N0_MAXB], se[N0_MAXB];
  if ((int)blockIdx.x >= nfin) {
    n0_final_body(Z, blockIdx.x - nfin, sv, se);
    return;
  }
  int const q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q < npairs) fm_finish_wave<K>(F, q, threadIdx.x & 63);
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

// cispi(x) for |x| <= 1 (osc.c:30 with the PLL's clamp |f| <= 0.49, osc.c:96): quadrant by rounding 2x, Taylor
// series of sin(pi r), cos(pi r) in r^2 on |r| <= 1/4 (nine terms each: 1.1e-16 of the exact values on a 20001-point
// grid).  About 25 fp64 operations; the library's sincospi is several times that, and this sits in the per-sample
// recurrence of every PLL channel.
__device__ __forceinline__ void cispi_small(double x, double *sn, double *cs) {
  double const k = rint(2.0 * x);
  double const r = fma(k, -0.5, x);
  double const t = r * r;
  double ps = 7.952054001475513e-07, pc = 4.303069587032947e-06;
  ps = fma(ps, t, -2.1915353447830217e-05);
  pc = fma(pc, t, -0.0001046381049248457);
  ps = fma(ps, t, 0.00046630280576761255);
  pc = fma(pc, t, 0.0019295743094039231);
  ps = fma(ps, t, -0.0073704309457143504);
  pc = fma(pc, t, -0.02580689139001406);
  ps = fma(ps, t, 0.08214588661112823);
  pc = fma(pc, t, 0.2353306303588932);
  ps = fma(ps, t, -0.5992645293207921);
  pc = fma(pc, t, -1.3352627688545895);
  ps = fma(ps, t, 2.5501640398773455);
  pc = fma(pc, t, 4.0587121264167685);
  ps = fma(ps, t, -5.16771278004997);
  pc = fma(pc, t, -4.934802200544679);
  ps = fma(ps, t, 3.141592653589793);
  pc = fma(pc, t, 1.0);
  ps *= r;
  int const q = (int)k & 3; // x = k/2 + r: rotate (cos, sin) of pi r by q quarter turns
  double const s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q == 2 || q == 3) ? -s0 : s0;
  *cs = (q == 1 || q == 2) ? -c0 : c0;
}

// ================================================================ linear: linear.c:57-156, one lane per channel
// PLL_ONLY: the PLL pass alone (linear.c:64-108: lock counter, the per-sample loop 81-97 with the sample written back as
// linear.c:91 does, the block's cphase / foffset / snr) for channels whose output pass (linear.c:114-148) is left to
// kh_demod_linear_w behind this kernel.  The two passes are separate loops in the reference and the second reads only what
// the first wrote back; here the serial part of a PLL channel -- a recurrence of ~60 dependent operations per sample, most
// of them fp64, on one lane -- sheds the ~100 instructions per sample of the output pass, which a workgroup per channel
// does in parallel (config 5, 256 PLL channels: 18.75 -> 10.77 ms per step).  Writes the PLL's status words and state
// fields only.
template <bool PLL_ONLY>
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
  // shift frequency 0 with the phasor at exactly 1 (the usual case, radio.c:359 with no shift): every product is
  // exact and the phasor never moves, so the multiply and the step are skipped; only the renormalisation count runs on
  bool const shift_active = !(shift.sre == 1.0 && shift.sim == 0.0 && shift.re == 1.0 && shift.im == 0.0);
  int shift_skipped = 0;

  // The PLL pass (linear.c:81-97) and the output pass (linear.c:114-148) of a block are run sample by sample in one
  // loop: the output pass reads only what the PLL pass has just written back (linear.c:91), and nothing the PLL pass
  // uses changes in the output pass, so the order of every operation on every variable is the reference's.  Samples
  // come in groups of G, the next group already in flight while this one is worked on: each lane walks its own row,
  // and a load per sample would put a memory round trip into every step of the recurrence.
  constexpr int G = 8;
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
    }
    float signal = 0, noise = 0, energy = 0, output_level = 0;
    float2 nxt[G];
#pragma unroll
    for (int u = 0; u < G; u++) nxt[u] = u < olen ? yb[u] : make_float2(0.f, 0.f);
    for (int n0 = 0; n0 < olen; n0 += G) {
      float2 cur[G];
#pragma unroll
      for (int u = 0; u < G; u++) cur[u] = nxt[u];
#pragma unroll
      for (int u = 0; u < G; u++) nxt[u] = n0 + G + u < olen ? yb[n0 + G + u] : make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < G; u++) {
        int const n = n0 + u;
        if (n >= olen) break;
        float2 v = cur[u];
        if (d.pll) { // linear.c:81-97
          float2 const ph = make_float2((float)vco.re, (float)vco.im);
          v = cmulc(v, ph); // s *= conjf(pll.vco.phasor)
          float phase;
          if (d.square)
            phase = fast_atan2f(2 * v.x * v.y, v.x * v.x - v.y * v.y); // cargf(s*s)
          else
            phase = fast_atan2f(v.y, v.x); // cargf: kh_fm.cuh, 2.5e-7; the loop filter sees it through gains of ~1e-3
          // run_pll, osc.c:91-107
          float feedback = d.pll_integrator_gain * s.pll_integrator + d.pll_prop_gain * phase;
          s.pll_integrator += phase;
          // osc.c:96: the comparisons are in double against 0.49; the float next above 0.49 is (float)0.49 itself and the
          // one below it is below 0.49, so they are these float comparisons (a NaN passes through, as there)
          feedback = feedback >= 0.49f ? 0.49f : feedback <= -0.49f ? -0.49f : feedback;
          // set_osc(&pll->vco, feedback, 0), osc.c:22-36: is_phasor_init's isnan(re) || isnan(im) || cnrm(phasor) < 0.9 in
          // one comparison (a NaN in either part makes the norm NaN, and NaN >= 0.9 is false)
          if (!(vco.re * vco.re + vco.im * vco.im >= 0.9)) {
            vco.re = 1;
            vco.im = 0;
            vco.steps = 0;
          }
          s.vco_freq = (double)feedback;
          cispi_small(2 * s.vco_freq, &vco.sim, &vco.sre); // cispi(2 f), osc.c:30
          dosc_step(vco);
          signal += v.x * v.x;
          noise += v.y * v.y;
        }
        if constexpr (PLL_ONLY) {
          yb[n] = v; // linear.c:91
          continue;
        }
        // linear.c:114-148
        // complex float * complex double -> double product, rounded to float
        float2 sm = v;
        if (shift_active) {
          sm = make_float2((float)((double)v.x * shift.re - (double)v.y * shift.im),
                           (float)((double)v.x * shift.im + (double)v.y * shift.re));
          dosc_step(shift);
        } else {
          shift_skipped++;
        }
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
          if (d.pll) yb[n] = v; // linear.c:91
        } else if (d.channels == 1) {
          float const smp = sm.x * s.agc_gain;
          output_level += smp * smp;
          if (ao) ao[o] = smp;
          po[o] = pcm_word(smp, big_endian);
          if (d.pll) yb[n] = v; // linear.c:91
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
    }
    if (d.pll) {
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
    if constexpr (PLL_ONLY) { // the PLL's words of the record; kh_demod_linear_w writes the others
      kh_status *const rp = status + (size_t)ch * status_stride + b;
      rp->snr = s.snr;
      rp->foffset = r.foffset;
      rp->cphase = r.cphase;
      rp->lock_timer = r.lock_timer;
      rp->pll_lock = s.pll_lock;
      continue;
    }
    r.output_level = output_level / (olen * d.channels);
    r.bb_power = energy / olen;
    r.snr = s.snr;
    r.pll_lock = s.pll_lock;
    r.agc_gain = s.agc_gain;
    status[(size_t)ch * status_stride + b] = r;
  }
  if constexpr (PLL_ONLY) { // (the PLL's fields; shift oscillator, gain and hang counter belong to kh_demod_linear_w)
    DemodState *const sp = states + ch;
    sp->lock_count = s.lock_count;
    sp->pll_lock = s.pll_lock;
    sp->snr = s.snr;
    sp->pll_integrator = s.pll_integrator;
    sp->vco_freq = s.vco_freq;
    sp->vco_re = vco.re;
    sp->vco_im = vco.im;
    sp->vco_steps = vco.steps;
    return;
  }
  s.shift_re = shift.re;
  s.shift_im = shift.im;
  s.shift_steps = (shift.steps + shift_skipped) % 16384; // Renorm_rate, osc.c:11
  s.vco_re = vco.re;
  s.vco_im = vco.im;
  s.vco_steps = vco.steps;
  states[ch] = s;
}

// ================================================================ linear without PLL: one workgroup per channel
// linear.c:114-148 when opt.pll is off (SSB, IQ, ISB, envelope AM: every modes.txt entry except CAM / DSB-with-PLL).
// Only the AGC (linear.c:127-136) chains from sample to sample, through three numbers: the sample's amplitude, the gain
// and the hang counter.  And between two attacks the gain follows a fixed course: constant while the hang counter runs
// (modes.txt hangs for 1.1 s), then one multiplication by the recovery rate per sample.  So:
//   phase 1  all LW_WAVES wavefronts, a block each at a time: post-detection shift (radio.c:355-361; the oscillator in
//            closed form from the phase the state holds), |s|^2, amplitude; the block's largest amplitude and its energy
//            go to LDS;
//   phase 2  wavefront 0 walks the blocks in order with the state (gain, hang).  A block that the hang counter outlasts
//            and whose largest amplitude does not trip the attack test (fl(a * gain) is monotonic in a for gain >= 0, so
//            then no sample does) leaves the gain alone: note the gain, count the hang down, next block -- a dozen
//            instructions.  Any other block is worked through on the spot, from attack to attack, with the reference's
//            roundings, and its output written;
//   phase 3  all wavefronts: the blocks that phase 2 only noted get their output, sample x (one constant) gain.
// algorithmic bytes per (channel, block): 8 olen read + (2 or 4) olen PCM (+ 8 olen baseband rewritten for stereo,
// linear.c:145); the second read of phase 3 comes from L2.
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// The recovery products G_0 = gain, G_{j+1} = fl(G_j * rate) without walking them.  Within a binade a float's bit
// pattern is linear in its significand m, and the step of the pattern, inc(m) = bits(fl(G rate)) - bits(G), is
// round(m (rate - 1)) with the significand's ulp as unit: monotonic in m (whatever the tie rule picks lies between its
// neighbours' results) and nearly constant -- at 6 dB/s and 39 kHz it changes by one every 190..380 steps.  So the
// products of a block are a handful of RUNS of constant step.  A run from G with step q = inc(G) ends at the first n
// where the REAL product disagrees, bits(fl((G + n q) rate)) - (G + n q) != q, or where G + n q leaves the binade; both
// are monotonic in n, so two ballots find n (64 candidates, then the S between two of them), and by the monotonicity
// every product before it is exactly G + j q: nothing is approximated and nothing is left to check afterwards.  The
// step out of a binade is taken by the real product.  Per run ~30 instructions instead of one dependent multiply per
// sample (a lone wavefront issues one every ~13 cycles).
constexpr int REC_RUNS = 6;
struct RecRuns {
  int n;                     // runs in use
  int start[REC_RUNS];       // first index of the run
  unsigned bits[REC_RUNS];   // G_start as a bit pattern
  unsigned inc[REC_RUNS];    // per-step increment of the pattern (two's complement: rate < 1 steps down)
};
// runs for G_0 .. G_M (M <= 4096); false: more than REC_RUNS runs, or a gain that is not a positive normal number.
// Wave-uniform; every lane of the wavefront must be active.
__device__ __forceinline__ bool rec_build(RecRuns &R, float gain, float rate, int M, int lane) {
  unsigned b = __float_as_uint(gain);
  int j0 = 0;
  bool open = true, good = true;
  R.n = 0;
  static_for<REC_RUNS>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    R.start[r] = 0, R.bits[r] = 0, R.inc[r] = 0;
    if (!open) return; // uniform
    unsigned const ex = b >> 23;
    if (ex < 1 || ex > 253) { // zero, denormal, infinite, NaN, negative
      good = false, open = false;
      return;
    }
    unsigned const q = __float_as_uint(__uint_as_float(b) * rate) - b;
    R.start[r] = j0, R.bits[r] = b, R.inc[r] = q;
    R.n = r + 1;
    int const rem = M - j0;
    if (rem <= 0) {
      open = false;
      return;
    }
    auto pred = [&](int n) { // the pattern has left the binade, or the real product steps differently from here
      unsigned const v = b + (unsigned)n * q;
      return ((v ^ b) >> 23) != 0 || __float_as_uint(__uint_as_float(v) * rate) - v != q;
    };
    int const S = (rem + 63) >> 6;
    int const n1 = (lane + 1) * S;
    unsigned long long const fv = __builtin_amdgcn_ballot_w64(n1 <= rem); // the rem / S multiples of S in reach
    unsigned long long const f1 = __builtin_amdgcn_ballot_w64(n1 <= rem && pred(n1));
    int const l1 = f1 ? __builtin_ctzll(f1) : __builtin_popcountll(fv); // (none: only the tail behind the last multiple is left)
    int const lo = l1 * S;                             // pred(lo) is false, or lo = 0
    int const n2 = lo + 1 + lane;
    unsigned long long const f2 = __builtin_amdgcn_ballot_w64(lane < S && n2 <= rem && pred(n2));
    if (f2 == 0) { // the run reaches the end
      open = false;
      return;
    }
    int const n = lo + 1 + __builtin_ctzll(f2);
    unsigned const v = b + (unsigned)n * q;
    if (((v ^ b) >> 23) != 0) // out of the binade: that one step by the real product
      b = __float_as_uint(__uint_as_float(b + (unsigned)(n - 1) * q) * rate);
    else
      b = v;
    j0 += n;
  });
  return good && !open;
}
__device__ __forceinline__ unsigned rec_at(RecRuns const &R, int j) { // bit pattern of G_j
  unsigned base = R.bits[0], q = R.inc[0];
  int st = 0;
  static_for<REC_RUNS - 1>([&](auto rc) {
    constexpr int r = decltype(rc)::value + 1;
    if (r < R.n) { // uniform
      bool const in = j >= R.start[r];
      base = in ? R.bits[r] : base;
      q = in ? R.inc[r] : q;
      st = in ? R.start[r] : st;
    }
  });
  return base + (unsigned)(j - st) * q;
}
constexpr int LW_WAVES = 8;      // wavefronts per channel (16 are quicker alone, but cost the next launch's transform kernels more registers)
constexpr int LW_MAXB = N0_MAXB; // blocks per launch (LDS notes per block)
template <int K>
__global__ __launch_bounds__(64 * LW_WAVES) void kh_demod_linear_w(float2 *__restrict__ baseband, size_t bb_stride, int olen,
                                                                   int nblocks, const DemodDesc *__restrict__ descs,
                                                                   DemodState *__restrict__ states, float *__restrict__ audio,
                                                                   int16_t *__restrict__ pcm, size_t pcm_stride, int big_endian,
                                                                   kh_status *__restrict__ status, int status_stride,
                                                                   const int *__restrict__ list) {
  __shared__ float sh_max[LW_MAXB], sh_energy[LW_MAXB], sh_gain[LW_MAXB];
  __shared__ int sh_hang[LW_MAXB];
  __shared__ unsigned char sh_done[LW_MAXB]; // 1: phase 2 has written the block itself; 2: a recovery without attack
  __shared__ __attribute__((aligned(16))) float gtab[64 * K + 8];                                   // recovery products of one block (wavefront 0)
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int const ch = list[blockIdx.x];
  DemodDesc const d = descs[ch];
  // (the fields this kernel uses, one by one: a copy of the whole record leaves an alloca behind for the rest, and the
  // compiler parks it in LDS -- 60 KB a workgroup)
  DemodState *const sp = states + ch;
  struct {
    double shift_re, shift_im;
    int shift_steps, hangcount, pll_lock;
    float snr, agc_gain;
  } s = {sp->shift_re, sp->shift_im, sp->shift_steps, sp->hangcount, sp->pll_lock, sp->snr, sp->agc_gain};
  float2 *y = baseband + (size_t)ch * bb_stride;
  float *ao = audio ? audio + (size_t)ch * pcm_stride : nullptr;
  int16_t *po = pcm + (size_t)ch * pcm_stride;
  // the shift oscillator (osc.c:22-50): phasor = cis(2 pi (phase0 + f m)) at sample m of the launch
  bool const shift_active = !(d.shift_freq == 0.0 && s.shift_re == 1.0 && s.shift_im == 0.0);
  double const phase0 = shift_active ? atan2(s.shift_im, s.shift_re) * (0.5 / M_PI) : 0.0;
  int const H = (int)d.hangtime;
  float const rate = d.recovery_rate, headroom = d.headroom;
  bool const runs_apply = rate > 0.f && rate < INFINITY;
  auto rec_prepare = [&](RecRuns &R, float g, int M) -> bool { // the runs of G_0 .. G_M from g
    return runs_apply && M > 0 && rec_build(R, g, rate, M, lane);
  };

  // sample n = lane + 64 j of block b after the shift
  auto load_block = [&](int b, float2 (&sm)[K], float (&amp)[K], float &energy) {
    float2 const *yb = y + (size_t)b * olen;
    energy = 0.f;
#pragma unroll
    for (int j = 0; j < K; j++) {
      int const n = lane + 64 * j;
      float2 v = n < olen ? yb[n] : make_float2(1.f, 0.f);
      if (shift_active) {
        double sn, cs;
        double ph = phase0 + d.shift_freq * (double)((size_t)b * olen + n);
        ph -= rint(ph);
        sincospi(2.0 * ph, &sn, &cs);
        v = make_float2((float)((double)v.x * cs - (double)v.y * sn), (float)((double)v.x * sn + (double)v.y * cs));
      }
      sm[j] = v;
      float const norm = v.x * v.x + v.y * v.y;
      if (n < olen) energy += norm;
      amp[j] = sqrtf(norm);
    }
  };
  // linear.c:137-148 for block b with per-sample gains gn, then the block's status
  auto store_block = [&](int b, float2 const (&sm)[K], float const (&amp)[K], float const (&gn)[K], float energy_sum, float gain_after) {
    float2 *yb = y + (size_t)b * olen;
    float level = 0.f;
#pragma unroll
    for (int j = 0; j < K; j++) {
      int const n = lane + 64 * j;
      if (n >= olen) continue;
      size_t const o = (size_t)b * olen + n;
      if (d.env) {
        float const smp = amp[j] * gn[j];
        level += smp * smp;
        if (ao) ao[o] = smp;
        po[o] = pcm_word(smp, big_endian);
      } else if (d.channels == 1) {
        float const smp = sm[j].x * gn[j];
        level += smp * smp;
        if (ao) ao[o] = smp;
        po[o] = pcm_word(smp, big_endian);
      } else {
        float2 const w = make_float2(sm[j].x * gn[j], sm[j].y * gn[j]);
        yb[n] = w; // linear.c:145
        level += w.x * w.x + w.y * w.y;
        if (ao) reinterpret_cast<float2 *>(ao)[o] = w;
        reinterpret_cast<short2 *>(po)[o] = make_short2(pcm_word(w.x, big_endian), pcm_word(w.y, big_endian));
      }
    }
    level = wave_sum(level);
    if (lane == 0) { // (field by field: a local record here is an alloca that the compiler moves to LDS, 60 KB a workgroup)
      kh_status *const r = status + (size_t)ch * status_stride + b;
      r->bb_power = energy_sum / olen;
      if (!d.pll) { // (a PLL channel's output pass: kh_demod_linear<true> has written these five before this kernel)
        r->snr = s.snr;
        r->foffset = 0.f;
        r->cphase = 0.f;
        r->lock_timer = 0.f;
        r->pll_lock = s.pll_lock;
      }
      r->pdeviation = 0.f;
      r->output_level = level / (olen * d.channels);
      r->agc_gain = gain_after;
      r->squelch_open = 0;
      r->reserved[0] = r->reserved[1] = 0.f;
    }
  };

  // ---- phase 1
  for (int b = wave; b < nblocks; b += LW_WAVES) {
    float2 sm[K];
    float amp[K], energy;
    load_block(b, sm, amp, energy);
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < K; j++)
      if (lane + 64 * j < olen) mx = fmaxf(mx, amp[j]); // (fmaxf drops a NaN amplitude, and a NaN never trips the test either)
    mx = wave_max(mx);
    energy = wave_sum(energy);
    if (lane == 0) {
      sh_max[b] = mx;
      sh_energy[b] = energy;
    }
  }
  __syncthreads();

  // ---- phase 2
  if (wave == 0) {
    float gain = s.agc_gain;
    int hang = s.hangcount;
    for (int b = 0; b < nblocks; b++) {
      bool quiet = !d.agc;
      if (d.agc && hang >= olen && gain >= 0.f) { // (a NaN gain fails the comparison: linear.c:127 attacks at once)
        float const t = sh_max[b] * gain;
        quiet = !(t > headroom);
        if (quiet) hang -= olen;
      }
      if (quiet) {
        if (lane == 0) {
          sh_gain[b] = gain;
          sh_done[b] = 0;
        }
        continue;
      }
      RecRuns R;
      bool closed = false; // R holds the checked runs from (gain, sample `hang`) to the block's end
      if (d.agc && gain >= 0.f && hang < olen) {
        // a block that recovers without an attack: the products are monotonic, so if the block's largest amplitude
        // does not trip the test against the larger of the first and the last, no sample does.  Phase 2 then only owes the last
        // product; phase 3 lays the block's products out again, a wavefront per block.
        int const M0 = olen - hang;
        closed = rec_prepare(R, gain, M0);
        if (closed) {
          float const gend = __uint_as_float(__builtin_amdgcn_readfirstlane(rec_at(R, M0)));
          if (!(sh_max[b] * fmaxf(gain, gend) > headroom)) {
            if (lane == 0) {
              sh_gain[b] = gain;
              sh_hang[b] = hang;
              sh_done[b] = 2;
            }
            gain = gend;
            hang = 0;
            continue;
          }
        }
      }
      // the block in detail: linear.c:127-136 by EVENTS instead of by samples.  From the state (gain, hang) at some sample
      // every later lane can work out, on the assumption that nothing attacks before it, the gain it would be tested
      // against (the iterated fp32 products G_0 = gain, G_{k+1} = fl(G_k * rate), exactly the reference's sequence of
      // roundings) and the gain it would leave.  The first sample whose test fires is a real attack (everything before it
      // held its assumption); it sets gain = headroom / amplitude and hang = hangtime, and the search goes on behind it.
      float2 sm[K];
      float amp[K], gatt[K], gn[K], energy;
      load_block(b, sm, amp, energy);
#pragma unroll
      for (int j = 0; j < K; j++) {
        gatt[j] = headroom / amp[j];
        gn[j] = gain;
      }
      constexpr int AGC_MAX_EVENTS = 6;
      auto agc_row_serial = [&](float &gn_j, float amp_j, float gatt_j, int p, int nvalid) { // the reference's loop
        for (int l = p; l < nvalid; l++) {
          float const a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(amp_j), l));
          float const g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gatt_j), l));
          float const t = isnan(gain) ? INFINITY : a * gain;
          if (t > headroom) {
            gain = g;
            hang = H;
          } else if (hang > 0) {
            hang--;
          } else {
            gain = gain * rate;
          }
          if (lane == l) gn_j = gain;
        }
      };
      int pos = 0, events = 0; // position in the block; passes spent on it
      // While the hang counter outlasts the block (modes.txt: 1.1 s) the gain only moves at attacks, and a sample
      // attacks when it is larger than the one the gain stands for: the attacks are the running maxima, a prefix scan
      // (in noise ln(olen) of them follow the first: event by event they were the block's whole cost).  The scan is
      // only the proposal -- fl(a * fl(headroom / a')) > headroom is not quite a > a' -- so the proposed gains are then
      // held, all samples at once, against linear.c:127-136 itself: the gain before sample n + 1 must be what sample n
      // leaves of the gain before it.  By induction from `gain` that makes every one of them the reference's; if one
      // fails (a near tie) nothing has been touched and the passes below take over.
      float cur_amp = headroom / gain; // the amplitude the gain stands for (exact after an attack in this block)
      bool scan_failed = false;
      auto record_scan = [&]() -> bool {
        float cc = cur_amp, cg = gain; // largest amplitude so far and the gain it set
        float prev_after = gain;
        int last_att = -1;
        bool bad = false;
        static_for<K>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          int const n = lane + 64 * jj;
          bool const active = n >= pos && n < olen;
          int sc = __float_as_int(active ? amp[jj] : -1.f), sg = __float_as_int(gatt[jj]);
          int const ident = __float_as_int(-1.f);
          // inclusive scan over the lanes of (larger amplitude, its gain), the earlier one on a tie
#define KH_REC_STEP(CTRL, ROWS)                                                           \
          {                                                                               \
            int const pc = __builtin_amdgcn_update_dpp(ident, sc, CTRL, ROWS, 0xf, false); \
            int const pg = __builtin_amdgcn_update_dpp(0, sg, CTRL, ROWS, 0xf, false);     \
            bool const mine = __int_as_float(sc) > __int_as_float(pc);                     \
            sc = mine ? sc : pc;                                                           \
            sg = mine ? sg : pg;                                                           \
          }
          // (where a move brings nothing, pc is the identity -1 and only a NaN amplitude "loses": it takes the identity's
          // gain 0 along, and never wins later; the check below sees through it: a NaN never attacks)
          KH_REC_STEP(0x111, 0xf) // row_shr:1
          KH_REC_STEP(0x112, 0xf) // row_shr:2
          KH_REC_STEP(0x114, 0xf) // row_shr:4
          KH_REC_STEP(0x118, 0xf) // row_shr:8
          KH_REC_STEP(0x142, 0xa) // row_bcast:15 into rows 1, 3
          KH_REC_STEP(0x143, 0xc) // row_bcast:31 into rows 2, 3
#undef KH_REC_STEP
          int const ec = __builtin_amdgcn_update_dpp(ident, sc, 0x138, 0xf, 0xf, false); // wave_shr:1: exclusive
          int const eg = __builtin_amdgcn_update_dpp(0, sg, 0x138, 0xf, 0xf, false);
          float const gb = __int_as_float(ec) > cc ? __int_as_float(eg) : cg; // the gain sample n is tested against
          float const tc = __int_as_float(__builtin_amdgcn_readlane(sc, 63)), tg = __int_as_float(__builtin_amdgcn_readlane(sg, 63));
          if (tc > cc) cc = tc, cg = tg; // uniform
          bool const attack = active && amp[jj] * gb > headroom;
          float const gaft = attack ? gatt[jj] : gb;
          int const nb = __builtin_amdgcn_update_dpp(0, __float_as_int(gb), 0x130, 0xf, 0xf, false); // wave_shl:1: lane + 1's
          bad = bad || (lane < 63 && nb != __float_as_int(gaft)) ||
                (lane == 0 && __float_as_int(gb) != __float_as_int(prev_after));
          prev_after = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gaft), 63));
          unsigned long long const f = __builtin_amdgcn_ballot_w64(attack);
          if (f) last_att = 64 * jj + 63 - __builtin_clzll(f);
          if (active) gn[jj] = gaft; // (samples from pos on are rewritten by whatever finishes the block)
        });
        if (__builtin_amdgcn_ballot_w64(bad)) return false;
        gain = prev_after;
        hang = last_att >= 0 ? H - (olen - 1 - last_att) : hang - (olen - pos);
        pos = olen;
        return true;
      };
      while (pos < olen) {     // uniform
        if (H >= olen && hang >= olen - pos && gain >= 0.f && !scan_failed) {
          if (record_scan()) break;
          scan_failed = true;
        }
        if (++events > AGC_MAX_EVENTS) { // attacks come thick: the rest of the block sample by sample
          static_for<K>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            int const lo = 64 * jj, hi = (olen - lo) < 64 ? (olen - lo) : 64;
            if (pos < lo + hi && hi > 0) agc_row_serial(gn[jj], amp[jj], gatt[jj], pos > lo ? pos - lo : 0, hi);
          });
          pos = olen;
          break;
        }
        // one pass over all that is left of the block: sample n >= pos is k = n - pos - hang samples into the recovery
        // (k < 0: still hanging); the recovery products are laid out once, by one lane, one dependent multiply each
        int M = olen - pos - hang;
        M = M < 0 ? 0 : M;
        if (M > 0 && !(pos == 0 && closed)) closed = rec_prepare(R, gain, M); // (pass 1 may inherit the runs)
        if (M > 0 && !closed) {
          if (lane == 0) { // (16-byte LDS writes: a write per product doubled the cost of a step of the chain)
            float g = gain;
            for (int m = 0; m <= M; m += 8) { // the table has room for the overshoot (64 K + 8 entries)
              float4 a, c4;
              a.x = g, g = g * rate;
              a.y = g, g = g * rate;
              a.z = g, g = g * rate;
              a.w = g, g = g * rate;
              c4.x = g, g = g * rate;
              c4.y = g, g = g * rate;
              c4.z = g, g = g * rate;
              c4.w = g, g = g * rate;
              *reinterpret_cast<float4 *>(gtab + m) = a;
              *reinterpret_cast<float4 *>(gtab + m + 4) = c4;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        float const tg = gain;
        float Q[K];
        int first = olen;
        static_for<K>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          int const n = lane + 64 * jj;
          int const k = n - pos - hang;
          bool const rec = M > 0 && k >= 0 && n < olen;
          int const kk = rec ? k : 0;
          float const P = !rec ? tg : closed ? __uint_as_float(rec_at(R, kk)) : gtab[kk];
          Q[jj] = !rec ? tg : closed ? __uint_as_float(rec_at(R, kk + 1)) : gtab[kk + 1];
          float const t = isnan(P) ? INFINITY : amp[jj] * P; // linear.c:127: isnan(gain) || amplitude * gain > headroom
          unsigned long long const f = __builtin_amdgcn_ballot_w64(n >= pos && n < olen && t > headroom);
          if (f != 0 && first == olen) first = 64 * jj + __builtin_ctzll(f);
        });
        float gnew = tg;
        static_for<K>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          int const n = lane + 64 * jj;
          if (n >= pos && n < first && n < olen) gn[jj] = Q[jj];
          if (n == first) gn[jj] = gatt[jj];
          if ((first >> 6) == jj) gnew = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gatt[jj]), first & 63));
        });
        if (first == olen) {
          if (M > 0) gain = __int_as_float(__builtin_amdgcn_readfirstlane(closed ? (int)rec_at(R, M) : __float_as_int(gtab[M])));
          int const left = hang - (olen - pos);
          hang = left > 0 ? left : 0;
          pos = olen;
        } else {
          static_for<K>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            if ((first >> 6) == jj) cur_amp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(amp[jj]), first & 63));
          });
          gain = gnew;
          hang = H;
          pos = first + 1;
        }
        __builtin_amdgcn_wave_barrier(); // (the table is rewritten by the next pass)
      }
      store_block(b, sm, amp, gn, sh_energy[b], gain);
      if (lane == 0) sh_done[b] = 1;
    }
    if (lane == 0) {
      size_t const total = (size_t)nblocks * olen;
      if (shift_active) {
        double sn, cs;
        double ph = phase0 + d.shift_freq * (double)total;
        ph -= rint(ph);
        sincospi(2.0 * ph, &sn, &cs);
        sp->shift_re = cs;
        sp->shift_im = sn;
      }
      sp->shift_steps = (int)((s.shift_steps + total) % 16384); // Renorm_rate, osc.c:11
      sp->agc_gain = gain;
      sp->hangcount = hang;
    }
  }
  __syncthreads();

  // ---- phase 3
  for (int b = wave; b < nblocks; b += LW_WAVES) {
    if (sh_done[b]) continue; // uniform
    float const g = sh_gain[b];
    float2 sm[K];
    float amp[K], gn[K], energy;
    load_block(b, sm, amp, energy);
#pragma unroll
    for (int j = 0; j < K; j++) gn[j] = g;
    store_block(b, sm, amp, gn, sh_energy[b], g);
  }
  for (int b = wave; b < nblocks; b += LW_WAVES) {
    if (sh_done[b] != 2) continue; // uniform
    float const g0 = sh_gain[b];
    int const h0 = sh_hang[b], M = olen - h0; // sample n >= h0 leaves the gain G_{n - h0 + 1}
    RecRuns R;
    rec_build(R, g0, rate, M, lane); // (the runs phase 2 found)
    float2 sm[K];
    float amp[K], gn[K], energy;
    load_block(b, sm, amp, energy);
#pragma unroll
    for (int j = 0; j < K; j++) {
      int const k = lane + 64 * j - h0;
      gn[j] = (k >= 0 && k < M) ? __uint_as_float(rec_at(R, k + 1)) : g0;
    }
    store_block(b, sm, amp, gn, sh_energy[b], __uint_as_float(rec_at(R, M)));
  }
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

// the wave form keeps 8 K registers of samples and one block's statistics per lane
int fm_finish_wave_usable(int olen, int nblocks) {
  static int const wave_env = getenv("KA9Q_HIP_FINISH_WAVE") ? atoi(getenv("KA9Q_HIP_FINISH_WAVE")) : 1; // A/B switch
  return wave_env && (olen + 63) / 64 <= 16 && nblocks <= 64;
}

// ================================================================ PCM packetisation plan: audio.c:32-135
// One workgroup per channel.  Pass 1 (all waves): not_silent of every chunk = OR of its words (audio.c:48-52,99-103).
// Pass 2 (one thread): the serial bookkeeping of audio.c:53-79,104-132 -- timestamp, sequence number, marker.
constexpr int RTP_CHUNK = 480; // PCM_BUFSIZE, audio.c:19
__global__ __launch_bounds__(256) void kh_rtp_plan(const int16_t *__restrict__ pcm, size_t pcm_stride, int olen, int nblocks,
                                                   const DemodDesc *__restrict__ descs, kh_rtp_chunk *__restrict__ plan,
                                                   int plan_stride, kh_rtp_state *__restrict__ states) {
  int const ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DemodDesc const d = descs[ch];
  if (d.demod == KH_DEMOD_NONE) return;
  int const chans = d.demod == KH_DEMOD_LINEAR ? d.channels : 1;
  int const wpb = olen * chans;                         // words per block = one send_*_output call
  int const cpb = (wpb + RTP_CHUNK - 1) / RTP_CHUNK;    // chunks per call
  int const total = nblocks * cpb;
  const int16_t *p = pcm + (size_t)ch * pcm_stride;
  kh_rtp_chunk *out = plan + (size_t)ch * plan_stride;
  for (int c = wave; c < total; c += 4) {
    int const b = c / cpb, j = c - b * cpb;
    int const off = b * wpb + j * RTP_CHUNK;
    int const nw = wpb - j * RTP_CHUNK < RTP_CHUNK ? wpb - j * RTP_CHUNK : RTP_CHUNK;
    int acc = 0;
    for (int k = lane; k < nw; k += 64) acc |= p[off + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o, 64);
    if (lane == 0) {
      kh_rtp_chunk r;
      r.timestamp = 0;
      r.seq = 0;
      r.marker = 0;
      r.send = acc != 0;
      r.offset = (uint32_t)off;
      r.nwords = (uint16_t)nw;
      r.reserved = 0;
      out[c] = r;
    }
  }
  __syncthreads(); // (the records were written by this workgroup: visible to it after the barrier)
  if (tid != 0) return;
  kh_rtp_state s = states[ch];
  for (int c = 0; c < total; c++) {
    kh_rtp_chunk r = out[c];
    r.timestamp = s.timestamp;
    s.timestamp += r.nwords / chans;                 // audio.c:55,106: sample (frame) count
    if (r.send) {
      s.packets++;
      s.bytes += 2u * r.nwords;
      r.marker = s.silent ? 1 : 0;
      s.silent = 0;
      r.seq = s.seq++;
      s.samples += r.nwords / chans;
    } else {
      s.silent = 1;
    }
    out[c] = r;
  }
  states[ch] = s;
}
int launch_rtp_plan(hipStream_t st, const int16_t *pcm, size_t pcm_stride, int olen, int nblocks, int nchan,
                    const DemodDesc *descs, void *plan, int plan_stride, void *states) {
  if (nblocks <= 0 || nchan <= 0) return 0;
  hipLaunchKernelGGL(kh_rtp_plan, dim3(nchan), dim3(256), 0, st, pcm, pcm_stride, olen, nblocks, descs,
                     static_cast<kh_rtp_chunk *>(plan), plan_stride, static_cast<kh_rtp_state *>(states));
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

int launch_demod(hipStream_t st, float2 *baseband, size_t bb_stride, int olen, int nblocks, int nchan,
                 const DemodDesc *desc, DemodState *state, float *audio, int16_t *pcm, size_t pcm_stride,
                 int pcm_big_endian, void *status, int status_stride, const int *fm_list, int n_fm, const int *lin_list,
                 int n_lin, const int *am_list, int n_am, const FmScratch &fm, const N0FinalArgs *n0final, int n_lin_nopll) {
  (void)nchan;
  if (nblocks <= 0) return 0;
  kh_status *stt = static_cast<kh_status *>(status);
  // the first n_lin_nopll entries of lin_list have opt.pll off: one workgroup each (kh_demod_linear_w); the others keep
  // the lane-per-channel loop.  KA9Q_HIP_LINEAR_WAVE=0 sends every linear channel there (A/B switch).
  static int const lin_wave_env = getenv("KA9Q_HIP_LINEAR_WAVE") ? atoi(getenv("KA9Q_HIP_LINEAR_WAVE")) : 1;
  // KA9Q_HIP_PLL_SPLIT=0: PLL channels keep both passes in the lane-per-channel loop.  Default: the PLL pass there
  // (kh_demod_linear<true>), the output pass by kh_demod_linear_w behind it, like every other linear channel's.
  // (read at every launch: tests/test_gpu_bank_linear.py runs one process under both)
  int const pll_split_env = getenv("KA9Q_HIP_PLL_SPLIT") ? atoi(getenv("KA9Q_HIP_PLL_SPLIT")) : 1;
  bool const lw_ok = lin_wave_env && olen <= 1024 && nblocks <= LW_MAXB;
  int const n_nopll = n_lin_nopll < n_lin ? n_lin_nopll : n_lin;
  bool const pll_split = lw_ok && pll_split_env && n_lin > n_nopll;
  int const nw = lw_ok ? (pll_split ? n_lin : n_nopll) : 0;
  if (n_fm > 0) {
    if (olen < 2 || olen > 8192 - 1) {
      set_error("FM demodulator: olen %d outside [2, 8192]", olen);
      return -1;
    }
    FinishArgs const F{olen, nblocks, desc, state, fm.parity, fm.stat, status_stride, fm.work, bb_stride, fm.carry,
                       fm.decay_pow, audio, pcm, pcm_stride, pcm_big_endian, stt, status_stride, fm_list, n_fm};
    int const npairs = n_fm * nblocks;
    // wavefronts per workgroup: fewer, larger workgroups dispatch faster (configs[1], us per step: 1: 98, 2: 77, 4: 68.6,
    // 8: 66.8, 16: 66), but in a bank that also has wave-resident linear channels eight cost the step 15 us (config 3's
    // share 168 -> 182-190: the larger workgroups stand in the way of the next launch's transform kernels for longer)
    static int const wpb_env = getenv("KA9Q_HIP_FINISH_WPB") ? std::min(std::max(atoi(getenv("KA9Q_HIP_FINISH_WPB")), 1), 16) : 0;
    int const wpb = wpb_env ? wpb_env : nw > 0 ? 4 : 8;
    int const nfin = (npairs + wpb - 1) / wpb;
    int const K = (olen + 63) / 64;
    bool const wave = fm_finish_wave_usable(olen, nblocks);
    N0FinalArgs const Z = (wave && n0final) ? *n0final : N0FinalArgs{};
    dim3 const grid(nfin + ((wave && n0final) ? n0final->ninputs : 0));
    if (!wave)
      hipLaunchKernelGGL(kh_fm_finish, dim3(n_fm, nblocks), dim3(FMB_NT), olen * sizeof(float2), st, olen, nblocks, desc,
                         state, fm.parity, fm.stat, status_stride, fm.work, bb_stride, fm.carry, fm.decay_pow, audio, pcm,
                         pcm_stride, pcm_big_endian, stt, status_stride, fm_list);
    else if (K <= 4)
      hipLaunchKernelGGL(kh_fm_finish_w<4>, grid, dim3(64 * wpb), 0, st, F, npairs, nfin, Z);
    else if (K <= 8)
      hipLaunchKernelGGL(kh_fm_finish_w<8>, grid, dim3(64 * wpb), 0, st, F, npairs, nfin, Z);
    else
      hipLaunchKernelGGL(kh_fm_finish_w<16>, grid, dim3(64 * wpb), 0, st, F, npairs, nfin, Z);
  }
  if (pll_split) // the PLL pass of the channels that have one, ahead of their output pass
    hipLaunchKernelGGL(kh_demod_linear<true>, dim3((n_lin - n_nopll + 63) / 64), dim3(64), 0, st, baseband, bb_stride, olen, nblocks,
                       desc, state, audio, pcm, pcm_stride, pcm_big_endian, stt, status_stride, lin_list + n_nopll, n_lin - n_nopll);
  if (nw > 0) {
    auto const kern = olen <= 256 ? kh_demod_linear_w<4> : olen <= 512 ? kh_demod_linear_w<8> : kh_demod_linear_w<16>;
    hipLaunchKernelGGL(kern, dim3(nw), dim3(64 * LW_WAVES), 0, st, baseband, bb_stride, olen, nblocks, desc, state, audio, pcm, pcm_stride,
                       pcm_big_endian, stt, status_stride, lin_list);
  }
  if (n_lin - nw > 0)
    hipLaunchKernelGGL(kh_demod_linear<false>, dim3((n_lin - nw + 63) / 64), dim3(64), 0, st, baseband, bb_stride, olen, nblocks,
                       desc, state, audio, pcm, pcm_stride, pcm_big_endian, stt, status_stride, lin_list + nw, n_lin - nw);
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

// ---- the whole cascade in one launch (hackrf.c:495-509: log2(D) half-band stages per rail, in place) ----
// A workgroup owns DEC tile: TO final outputs of one rail.  It loads the raw samples they depend on into LDS (TO << ld
// of them plus the 13 (2^ld - 1) the 15-tap stages reach back), runs every stage from LDS to LDS with the formulas of
// kh_hb15 / kh_hb3 above (same expressions, so the same floats), and stores TO floats.  Samples before the start of the
// call come from each stage's carried state exactly where the staged kernels take them (hb15_hist).  The workgroup of
// the last tile leaves every stage's new state (decimate.c:145-147, :162) in a second state buffer: the first tiles may
// still be reading the old one.
// algorithmic bytes per rail and call: 4 (chunk << ld) read + 4 chunk written; the halo re-reads hit L2.
constexpr int DEC_LDS_FLOATS = 15872; // 62 KB: raw tile + halo (A), and the first stage's output (B)

__device__ __forceinline__ float dec_hist(const Hb15Dev *s15, const float *s3, int slot, bool is15, int idx) {
  if (idx < -14) return 0.f; // never read: the lowest tap of an output k >= 0 is x[-13]
  if (is15) return hb15_hist(s15[slot], idx);
  return idx == -1 ? s3[slot] : 0.f;
}

// one stage, LDS to LDS: dst[k - dlo] for k in [dlo, dhi] from src[idx - slo].  EDGE: the tile reaches back before the
// start of the call (only the first tile of a rail does), so taps and outputs at negative indices come from the states.
template <bool EDGE>
__device__ __forceinline__ void dec_stage(const float *src, float *dst, int slo, int dlo, int dhi, bool is15, int slot, bool n15,
                                          const Hb15Dev *__restrict__ s15_in, const float *__restrict__ s3_in) {
  int const tid = threadIdx.x;
  auto X = [&](int idx) -> float {
    if (EDGE && idx < 0) return dec_hist(s15_in, s3_in, slot, is15, idx);
    return src[idx - slo];
  };
  if (is15) {
    float const c0 = s15_in[slot].coeffs[0], c1 = s15_in[slot].coeffs[1], c2 = s15_in[slot].coeffs[2], c3 = s15_in[slot].coeffs[3];
    for (int t = tid; t <= dhi - dlo; t += 256) {
      int const k = dlo + t;
      float result;
      if (!EDGE || k >= 0) {
        result = X(2 * k - 6);
        result += (X(2 * k + 1) + X(2 * k - 13)) * c0;
        result += (X(2 * k - 1) + X(2 * k - 11)) * c1;
        result += (X(2 * k - 3) + X(2 * k - 9)) * c2;
        result += (X(2 * k - 5) + X(2 * k - 7)) * c3;
      } else {
        result = dec_hist(s15_in, s3_in, slot - 1, n15, k); // what the next stage's state holds for this sample
      }
      dst[t] = result;
    }
  } else {
    for (int t = tid; t <= dhi - dlo; t += 256) {
      int const k = dlo + t;
      dst[t] = (!EDGE || k >= 0) ? 2 * X(2 * k) + X(2 * k + 1) + X(2 * k - 1) : dec_hist(s15_in, s3_in, slot - 1, n15, k);
    }
  }
}

__global__ __launch_bounds__(256) void kh_decimate_fused(const float *__restrict__ in, size_t in_stride, float *__restrict__ out,
                                                         size_t out_stride, int chunk, int ld, int thr, int TO, int a_cap,
                                                         const Hb15Dev *__restrict__ s15_in, Hb15Dev *__restrict__ s15_out,
                                                         const float *__restrict__ s3_in, float *__restrict__ s3_out) {
  extern __shared__ float dec_lds[];
  float *const A = dec_lds, *const B = dec_lds + a_cap;
  int const r = blockIdx.y, tid = threadIdx.x;
  int const k0 = blockIdx.x * TO, k1 = min(chunk, k0 + TO) - 1;
  bool const last = blockIdx.x == gridDim.x - 1;
  // [lo, hi] of stage j's output (j = ld: the raw samples), in that stream's own index space
  auto range = [&](int j, int &lo, int &hi) {
    lo = k0, hi = k1;
    for (int q = 0; q < j; q++) {
      lo = 2 * lo - (q < thr ? 13 : 1);
      hi = 2 * hi + 1;
    }
  };
  int slo, shi;
  range(ld, slo, shi);
  bool const edge = slo < 0; // then some stage reads before the call's first sample; otherwise none does
  slo = max(slo, -14);
  {
    const float *x = in + (size_t)r * in_stride;
    int const slot = r * ld + (ld - 1);
    bool const is15 = (ld - 1) < thr;
    for (int t = tid; t <= shi - slo; t += 256) {
      int const idx = slo + t;
      A[t] = idx >= 0 ? x[idx] : dec_hist(s15_in, s3_in, slot, is15, idx);
    }
  }
  __syncthreads();
  float *src = A, *dst = B;
  for (int j = ld - 1; j >= 0; j--) {
    int const slot = r * ld + j;
    bool const is15 = j < thr;
    if (last && tid == 0) { // the carried state of stage j after cnt outputs, from the stage's input stream
      int const cnt = chunk << j;
      auto X = [&](int idx) -> float { return idx >= 0 ? src[idx - slo] : dec_hist(s15_in, s3_in, slot, is15, idx); };
      Hb15Dev n = s15_in[slot];
      float n3 = s3_in[slot];
      if (is15) {
        n.even_samples[0] = X(2 * (cnt - 1));
        n.even_samples[1] = X(2 * (cnt - 1));
        n.even_samples[2] = X(2 * (cnt - 2));
        n.even_samples[3] = X(2 * (cnt - 3));
        n.odd_samples[0] = X(2 * (cnt - 1) + 1);
        n.odd_samples[1] = X(2 * (cnt - 1) + 1);
        n.odd_samples[2] = X(2 * (cnt - 2) + 1);
        n.odd_samples[3] = X(2 * (cnt - 3) + 1);
        n.old_odd_samples[3] = X(2 * (cnt - 4) + 1);
        n.old_odd_samples[2] = X(2 * (cnt - 5) + 1);
        n.old_odd_samples[1] = X(2 * (cnt - 6) + 1);
        n.old_odd_samples[0] = X(2 * (cnt - 7) + 1);
      } else {
        n3 = X(2 * cnt - 1);
      }
      s15_out[slot] = n;
      s3_out[slot] = n3;
    }
    int dlo, dhi;
    range(j, dlo, dhi);
    dlo = max(dlo, -14);
    bool const n15 = (j - 1) < thr; // the type of the stage that consumes this one's output (j >= 1)
    if (edge)
      dec_stage<true>(src, dst, slo, dlo, dhi, is15, slot, n15, s15_in, s3_in);
    else
      dec_stage<false>(src, dst, slo, dlo, dhi, is15, slot, n15, s15_in, s3_in);
    __syncthreads();
    float *const tmp = src;
    src = dst;
    dst = tmp;
    slo = dlo, shi = dhi;
  }
  for (int t = tid; t <= k1 - k0; t += 256) out[(size_t)r * out_stride + k0 + t] = src[t];
}

// The tile: TO final outputs from raw = TO << ld samples plus the halo.  The largest raw tile that fits the LDS budget
// wastes the fewest re-reads, a smaller one gives short calls enough workgroups to fill the chip and more workgroups per
// CU; raw tiles of 1024-2048 measured best for long rails (1.3-1.4 TB/s of algorithmic bytes against 0.74 at 4096-8192) (tools/decimate_bench.py).  Returns 0 when the cascade is too deep.
static int decimate_tile(int ld, int thr, int nrails, int chunk, int *a_cap, int *b_cap) {
  long halo = 0;
  for (int q = 0; q < ld; q++) halo = 2 * halo + (q < thr ? 13 : 1);
  static int const pref = getenv("KA9Q_HIP_DECIMATE_TILE") ? atoi(getenv("KA9Q_HIP_DECIMATE_TILE")) : 2048;
  int best = 0;
  for (int raw = 8192; raw >= (1 << ld) && raw >= 256; raw >>= 1) {
    long const a = raw + halo + 16, b = raw / 2 + halo / 2 + 16;
    if (a + b > DEC_LDS_FLOATS) continue;
    if (best == 0) best = raw; // the largest that fits
    // go smaller while the launch would leave CUs idle (fewer than two workgroups each) or down to the preferred size
    long const wgs = (long)nrails * (((long)chunk << ld) + raw - 1) / raw;
    if (raw > pref || wgs < 512) best = raw;
    else break;
  }
  if (best == 0) return 0;
  *a_cap = (int)(best + halo + 16);
  *b_cap = (int)(best / 2 + halo / 2 + 16);
  return best >> ld;
}
bool decimate_can_fuse(int log_decimate, int stage_threshold) {
  int a, b;
  return log_decimate >= 1 && log_decimate <= 12 && decimate_tile(log_decimate, stage_threshold, 1, 1 << 20, &a, &b) >= 1;
}
// in: nrails rails of chunk << ld floats (device); out: nrails rails of chunk floats (device, not overlapping `in`);
// states_*: nrails*ld entries each, in != out
int launch_decimate_fused(hipStream_t st, const float *in, size_t in_stride, float *out, size_t out_stride, int nrails, int chunk,
                          int log_decimate, int stage_threshold, const void *hb15_in, void *hb15_out, const float *hb3_in,
                          float *hb3_out) {
  if (nrails <= 0 || chunk <= 0 || log_decimate <= 0) return 0;
  int a_cap = 0, b_cap = 0;
  int const TO = decimate_tile(log_decimate, stage_threshold, nrails, chunk, &a_cap, &b_cap);
  if (TO < 1) {
    set_error("launch_decimate_fused: 2^%d:1 does not fit one launch", log_decimate);
    return -1;
  }
  hipLaunchKernelGGL(kh_decimate_fused, dim3((chunk + TO - 1) / TO, nrails), dim3(256), (a_cap + b_cap) * sizeof(float), st, in,
                     in_stride, out, out_stride, chunk, log_decimate, stage_threshold, TO, a_cap,
                     static_cast<const Hb15Dev *>(hb15_in), static_cast<Hb15Dev *>(hb15_out), hb3_in, hb3_out);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
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
