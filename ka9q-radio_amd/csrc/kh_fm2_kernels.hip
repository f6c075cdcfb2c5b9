// kh_fm2_kernels.hip -- demod_fm (fm.c:43-139) behind kh_chan2: the Rice-corrected SNR one lane per (channel, block),
// then one workgroup per channel that only demodulates the blocks whose squelch is open.
//
// Reference behaviour restated here (citations into the reference tree):
//   fm_snr                 fm.c:190-205 (the fixed-point iteration; xi from kh_fm.cuh's polynomial pieces)
//   squelch                fm.c:73-78 (open for two blocks after the SNR last exceeded 4), 123 (state zeroed when closed)
//   discriminator          fm.c:85-93 (conjugate product, cargf, amplitude weight cabsf(p) / fm_variance clipped to 1)
//   peak deviation         fm.c:97-102, 114-121 (order dependent), de-emphasis fm.c:105-111, closed squelch fm.c:124-132
//   output level           fm.c:135-138, scaleclip audio.c:22-28
//
// Why the split (profiles/r03_base_*_sq.txt): the channel wavefronts of round 2 ran discriminator, integrator scan and
// fm_snr's iteration for every (channel, block) -- ~1000 vector instructions a pair -- although a bank is mostly idle
// channels whose squelch is closed, and fm_snr ran wave-uniform (one value per 64 lanes' issue slots) exactly where it
// is slowest (amplitude of pure noise: avg^2/var sits at fm_snr's threshold pi/(4-pi); 10 rounds on average, 65 at most).
//   kh_fm_snr_lanes   a lane per (channel, block): 64 iterations share a wavefront's issue slots
//   kh_fm_demod2      a workgroup per channel, G = olen/8 lanes per block.  Closed blocks cost the squelch logic and
//                     the decaying carry (fm.c:124-132); open blocks read their baseband, run fm.c:85-111 with the
//                     integrator started from zero, hand the block's carry-out through LDS, and add lastaudio's
//                     contribution c_in decay^n (fm.c:110-111 is linear in lastaudio) after one barrier.
// Roofline: HBM; algorithmic bytes per open (channel, block): 8*olen baseband read + 2*olen PCM written.
#include "kh_internal.h"
#include "kh_fft.cuh"
#include "kh_fm.cuh"
#include "kh_n0.cuh"
#include "../../include/ka9q_hip.h"
#include <algorithm>
#include <cstdlib>

namespace kh {

// fm_snr (fm.c:190-205) with a different argument in every lane.  Same polynomial pieces as kh_fm.cuh's fm_snr(), same
// iteration, same stopping rule, hence the same value for the same r; the pieces' coefficients come from LDS (`coef`:
// KH_XI_SEGS x 12 doubles, copied there by the caller), and the loop is cut in two so that the round itself -- 20 fp64
// operations on one lane's chain, ~50 rounds for the slowest lane of a wavefront of noise blocks -- carries one exit test:
// a lone wavefront pays ~10 cycles per instruction, scalar bookkeeping included (profiles/r03_s1_c4all_sq.txt: 1250
// vector + 1570 scalar instructions per wavefront in 12.5 us before this).
static __device__ double fm_snr_lane(double r, const double *coef) {
  double thetasq = r;
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  double const r1 = 1 + r;
  int guard = 0; // the reference loop has no bound; NaN input falls through both tests above and exits here
  bool done = false;
  while (!done) {
    if (!(thetasq >= 0.0 && thetasq <= 100.0)) { // (not reached from fm_snr's own range checks: the series itself)
      double const o = thetasq;
      thetasq = fm_xi(thetasq) * r1 - 2;
      done = !(fabs(thetasq - o) > .001) || ++guard >= 10000;
      continue;
    }
    constexpr double E[KH_XI_SEGS + 1] = KH_XI_EDGES;
    int seg = 0;
#pragma unroll
    for (int k = 1; k < KH_XI_SEGS; k++) seg += thetasq >= E[k] ? 1 : 0;
    double lo = E[0], e1 = E[1];
#pragma unroll
    for (int k = 1; k < KH_XI_SEGS; k++)
      if (seg == k) lo = E[k], e1 = E[k + 1];
    double const hi = seg == KH_XI_SEGS - 1 ? 101.0 : e1; // thetasq == 100 belongs to the last segment
    double const mid = 0.5 * (lo + e1), scale = 2.0 / (e1 - lo);
    double c[KH_XI_DEG + 1];
    const double2 *cs = reinterpret_cast<const double2 *>(coef + seg * (KH_XI_DEG + 1));
#pragma unroll
    for (int k = 0; k <= KH_XI_DEG; k += 2) {
      double2 const t = cs[k / 2];
      c[k] = t.x;
      c[k + 1] = t.y;
    }
    bool inside;
    do {
      double const o = thetasq;
      double const u = (thetasq - mid) * scale;
      double const u2 = u * u, u4 = u2 * u2, u8 = u4 * u4;
      double const p01 = fma(c[1], u, c[0]), p23 = fma(c[3], u, c[2]), p45 = fma(c[5], u, c[4]);
      double const p67 = fma(c[7], u, c[6]), p89 = fma(c[9], u, c[8]), pab = fma(c[11], u, c[10]);
      double const q0 = fma(p23, u2, p01), q1 = fma(p67, u2, p45), q2 = fma(pab, u2, p89);
      double const xi = fma(q2, u8, fma(q1, u4, q0));
      thetasq = xi * r1 - 2;
      done = !(fabs(thetasq - o) > .001) || ++guard >= 10000;
      inside = thetasq >= lo && thetasq < hi;
    } while (!done && inside);
  }
  return thetasq;
}

struct Fm2Args {
  int nblocks;
  const DemodDesc *descs;
  DemodState *states;
  int par;
  const FmBlockStat *stat;
  int stat_stride;
  const float2 *bb;
  size_t bb_stride;
  const double *decay_pow;
  float *audio;
  int16_t *pcm;
  size_t pcm_stride;
  int big_endian;
  kh_status *status;
  int status_stride;
  const int *list;
  // splits > 1 (one-channel workgroups only): a channel's blocks are dealt in contiguous runs over `splits` neighbouring
  // workgroups; the carry-outs of the runs before a workgroup's own reach it through xch ([channel][stat_stride]),
  // marked with this launch's epoch
  int splits;
  unsigned epoch;
  Fm2Carry *xch;
};

template <int G>
__device__ __forceinline__ float fm2_group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}

#ifdef KH_FM2_TIMES
__device__ long long kh_fm2_times[8192 * 8];
#define KH_T(k) if (threadIdx.x == 0 && blockIdx.x < 8192) kh_fm2_times[blockIdx.x * 8 + (k)] = wall_clock64();
#else
#define KH_T(k)
#endif
constexpr int FM2_MAXT = 64;  // (channels per workgroup) x (blocks per launch) <= 64: a workgroup's items are one wavefront's lanes
constexpr int FM2_MAXCPW = 8; // channels per workgroup
struct Fm2Chan { // what a block's group needs of its channel, in LDS
  int ch, flat, so_in;
  float dsamprate, gain, lastaudio_in;
  float2 state_in;
};
// fm.c:73-78 asks one thing of the SNR: is it above 4.  fm_snr (fm.c:190-205) of r = avg^2/variance passes 4 at
// r = 6.174478 (scanned over [pi/(4-pi), 100] in steps of 2.4e-5, and in steps of 5e-8 around the crossing; the stopping
// rule makes it a sawtooth with teeth of at most 0.001): 3.971 at 6.15, 4.029 at 6.20, monotone outside up to those
// teeth.  So the squelch of every block outside [6.15, 6.20] is known from r alone, without the iteration.
constexpr float FM2_R_LO = 6.15f, FM2_R_HI = 6.20f;
// all of a status record but .snr (the iteration's result: written by the lane that ran it)
__device__ __forceinline__ void fm2_store_status(kh_status *p, kh_status const &r) {
  float *w = reinterpret_cast<float *>(p);
  static_assert(sizeof(kh_status) == 48 && offsetof(kh_status, snr) == 4, "word 1 is left to the SNR lane");
  w[0] = r.bb_power;
  *reinterpret_cast<float2 *>(w + 2) = make_float2(r.foffset, r.pdeviation);
  *reinterpret_cast<float4 *>(w + 4) = make_float4(r.cphase, r.lock_timer, __int_as_float(r.pll_lock), r.output_level);
  *reinterpret_cast<float4 *>(w + 8) = make_float4(r.agc_gain, __int_as_float(r.squelch_open), r.reserved[0], r.reserved[1]);
}
// A workgroup serves `cpw` FM channels (1, or up to 8 in a large bank); its items are their (channel, block) pairs, at most
// 64.  After one barrier (the items' statistics are in LDS) its wavefronts split into two roles that never wait for each
// other:
//   SNR wavefronts (the last four of a one-channel workgroup, one per SIMD; the last one otherwise): fm_snr's iteration, a
//      lane per item, and the status word it yields.  0 .. 65 rounds of ~26 dependent fp64 instructions at a block's
//      threshold, where idle channels sit: 7 to 15 us for the slowest lane, and a wavefront waits for its slowest lane.
//   work wavefronts: the first of them, a lane per item: the squelch from r (above; the iteration only for an r inside
//      [6.15, 6.20]), the list of OPEN items by ballot, and all there is to an idle item (closed, integrator at zero:
//      nearly every item of a large bank): its status.  Then a group of G = olen/8 lanes per open item, taken from the list
//      wherever they are -- one busy channel among idle ones gets every group: discriminator and de-emphasis from a zero
//      carry, the block's carry-out to LDS with a "ready" bit; the carry-in's contribution (linear, fm.c:110-111) once the
//      channel's earlier blocks are ready (they stand earlier in the list: whoever has them publishes before it waits, so
//      the wait ends; it is bounded all the same).  Then the closed items that carry a decaying lastaudio (fm.c:124-132),
//      then 16-byte zero stores for the idle ones.  An SNR wavefront that is done takes items as well.
// A workgroup of one wavefront does the work first and iterates after it.
//
// Small banks (F.splits > 1): a workgroup per channel keeps only as many CUs busy as the bank has FM channels, and a busy
// channel's 32 blocks are ~1200 vector instructions each on ONE CU's four SIMDs (64 channels x 32 open blocks: 18 us of
// issue time there while three quarters of the chip idle).  The channel's blocks then go to `splits` workgroups with
// neighbouring numbers, a contiguous run of blocks each.  Everything that depends on the whole launch is cheap and every
// workgroup of the channel repeats it (statistics, squelch, which items are idle); what a block needs of the blocks before
// it is their carry-outs for a zero carry-in (fm.c:110-111 is linear in lastaudio), which do NOT depend on each other:
// a group publishes its block's in global memory (value, then this launch's epoch as the flag: agent-scope atomics, no
// cache line that anybody reads with plain loads) and fetches those of the runs before its own.  A workgroup only ever
// waits for workgroups with LOWER numbers, which the dispatcher has started before it; the wait is bounded all the same.
// Same arithmetic in the same order whatever the split: tests/test_gpu_bank_fm.py holds the arrangements to bit identity.
template <int G>
__global__ __launch_bounds__(1024) void kh_fm_demod2(Fm2Args F, int nfm_wg, N0FinalArgs Z, int n_fm, int cpw, float r_lo, float r_hi) {
  constexpr int K = 8, OLEN = K * G;
  static_assert(G == 32 || G == 64, "olen = 256 or 512");
  __shared__ double shB[FM2_MAXT]; // a block's integrator carry-out for a zero carry-in (0: closed)
  __shared__ __attribute__((aligned(16))) double sh_coef[KH_XI_SEGS * (KH_XI_DEG + 1)];
  __shared__ float sh_r[FM2_MAXT], sh_bbp[FM2_MAXT], sh_var[FM2_MAXT];
  __shared__ unsigned char sh_list[FM2_MAXT]; // the open items in ascending order, then the closed ones
  __shared__ int sh_count[3];
  __shared__ unsigned sh_gt4[2], sh_ready[2], sh_listed; // per item: SNR above 4; carry-out published; (flag) the list stands
  __shared__ int sh_next[2];                              // tickets: the next entry of the list (open + decaying; idle)
  __shared__ unsigned sh_open[2];                         // per item: squelch open (every item of the channels, not only this workgroup's)
  __shared__ Fm2Chan sh_ch[FM2_MAXCPW];
  int const S = F.splits; // (1 whenever cpw > 1)
  if ((int)blockIdx.x >= nfm_wg * S) { // kh_n0_final's work (one workgroup per input) rides in this launch
    n0_final_body(Z, blockIdx.x - nfm_wg * S, sh_r, sh_bbp);
    return;
  }
  KH_T(0)
  int const lane64 = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  int const par = F.par, nblocks = F.nblocks, items = cpw * nblocks; // items <= 64
  // this workgroup's place among the channels' workgroups, and its run of blocks [b_lo, b_hi)
  int const wgc = S > 1 ? (int)blockIdx.x / S : (int)blockIdx.x, part = (int)blockIdx.x - wgc * S;
  int const b_lo = S > 1 ? part * nblocks / S : 0, b_hi = S > 1 ? (part + 1) * nblocks / S : nblocks;
  double const Ablk = F.decay_pow[OLEN];
  // roles
  int const nws = (cpw > 1 || nwv < 4) ? 1 : (nwv < 8 ? 2 : 4); // wavefronts that iterate: no more than half, one per SIMD
  bool const split = nwv > nws;
  int const nwork = split ? nwv - nws : nwv, ws = split ? wave - nwork : wave; // ws in [0, nws): iterates items ws, ws + nws, ..
  bool const works = wave < nwork;
  int const l = lane64 % G;
  double const dpl = F.decay_pow[K * l]; // decay^(K l)
  // ---- fm.c:53-70: the statistics kh_chan2 left, every item's by the lane that will iterate it
  int const per = (ws >= 0 && ws < nws) ? (items + nws - 1) / nws : 0;
  int const es = lane64 < per ? lane64 * nws + ws : items;
  float r_it = 0.f; // fm_snr's argument avg^2 / variance
  bool it_valid = false;
  {
    if ((int)threadIdx.x < cpw) {
      int const chi = threadIdx.x * nfm_wg + wgc; // neighbouring channels (busy together, if any are) go to different workgroups
      Fm2Chan c{};
      c.ch = -1;
      if (chi < n_fm) {
        c.ch = F.list ? F.list[chi] : chi; // (null: every channel of the bank is an FM channel, the list is the identity)
        const DemodDesc *dd = F.descs + c.ch;
        const DemodState *st = F.states + c.ch;
        c.flat = dd->flat, c.dsamprate = dd->dsamprate, c.gain = dd->fm_gain;
        c.so_in = st->fm_squelch_open[par], c.state_in = st->fm_state[par], c.lastaudio_in = st->fm_lastaudio[par];
      }
      sh_ch[threadIdx.x] = c;
    }
    for (int k = threadIdx.x; k < KH_XI_SEGS * (KH_XI_DEG + 1); k += blockDim.x) sh_coef[k] = (&kh_xi_coeffs[0][0])[k];
    if (threadIdx.x == 0) sh_listed = 0u, sh_next[0] = 0, sh_next[1] = 0;
    if (es < items) {
      int const ce = es / nblocks, be = es - ce * nblocks, che = ce * nfm_wg + wgc;
      if (che < n_fm) {
        const FmBlockStat *fs = F.stat + (size_t)(F.list ? F.list[che] : che) * F.stat_stride + be;
        float const bbp = fs->bb_power, var = fs->variance, r = fs->snr;
        sh_bbp[es] = bbp, sh_var[es] = var, sh_r[es] = r;
        it_valid = be >= b_lo && be < b_hi; // (the iteration and its status word: this workgroup's blocks only)
        r_it = it_valid ? r : 0.f;
      }
    }
    __syncthreads(); // the only barrier
    KH_T(1)
  }
  // the iteration, and the status word it yields
  auto iterate = [&]() {
    if (lane64 < per) { // (spare lanes carry r = 0 and leave at once)
      float const snr = fmaxf(0.0f, (float)fm_snr_lane((double)r_it, sh_coef));
      if (it_valid) {
        int const ce = es / nblocks, be = es - ce * nblocks;
        F.status[(size_t)sh_ch[ce].ch * F.status_stride + be].snr = snr;
      }
    }
  };
  if (!works) iterate(); // (uniform per wavefront; afterwards these wavefronts take items like the others)
  // fm.c:73-78 in closed form (fm_counter_after, kh_demod_kernels.hip): the counter after block j is 1 if snr_j > 4, else
  // max(so_in - 1, 0) for j = 0 and 0 for j >= 1; a block demodulates iff snr > 4 or the counter before it is positive
  auto squelch = [&](unsigned long long g4, int e, int b, int so_in, bool &open, bool &open_p, int &so) {
    bool const gt_b = g4 >> e & 1, gt_p = b >= 1 && (g4 >> (e - 1) & 1), gt_pp = b >= 2 && (g4 >> (e - 2) & 1);
    int const dec_in = so_in - 1 > 0 ? so_in - 1 : 0;
    int const cnt_pp = b >= 2 ? (gt_pp ? 1 : (b == 2 ? dec_in : 0)) : so_in; // counter before block b - 1
    int const cnt_p = b >= 1 ? (gt_p ? 1 : (b == 1 ? dec_in : 0)) : so_in;  // counter before block b
    open = gt_b || cnt_p > 0;
    open_p = b >= 1 && (gt_p || cnt_pp > 0);
    so = gt_b ? 1 : (b == 0 ? dec_in : 0);
  };
  if (wave == 0) { // (a work wavefront in either arrangement)
    int const e = lane64, ce = e / nblocks, b = e - ce * nblocks;
    bool const valid = e < items && ce * nfm_wg + wgc < n_fm;
    bool const mine = valid && b >= b_lo && b < b_hi;
    float const r = valid ? sh_r[e] : 0.f;
    bool gt = r > r_hi; // (also r > 100, where fm_snr returns r)
    if (valid && !gt && r >= r_lo) gt = fmaxf(0.0f, (float)fm_snr_lane((double)r, sh_coef)) > 4.f; // (~8 rounds there)
    unsigned long long const g4 = __builtin_amdgcn_ballot_w64(valid && gt);
    bool open = false, open_p;
    int so = 0;
    Fm2Chan const C = sh_ch[ce < cpw ? ce : 0];
    if (valid) squelch(g4, e, b, C.so_in, open, open_p, so);
    unsigned long long const mo = __builtin_amdgcn_ballot_w64(valid && open);
    // idle: closed, and the integrator holds 0 on entry (no open block of the channel before this one, lastaudio 0 from
    // the launch before; fma(A, 0, 0) = 0) or is not in use (flat).  All there is to such a block is its status here and
    // olen zeros of PCM below; in a large bank that is nearly every item.
    bool const idle = valid && !open &&
                      (C.flat || (C.lastaudio_in == 0.f && !(mo & ((1ull << e) - (1ull << (e - b))))));
    // the lists hold this workgroup's items only (all of them unless the channel is split)
    unsigned long long const mo_m = __builtin_amdgcn_ballot_w64(mine && open);
    unsigned long long const mg = __builtin_amdgcn_ballot_w64(mine && !open && !idle), mi = __builtin_amdgcn_ballot_w64(mine && idle);
    unsigned long long const below = (1ull << e) - 1;
    int const no = __popcll(mo_m), ng = __popcll(mg);
    if (mine)
      sh_list[open ? __popcll(mo_m & below) : !idle ? no + __popcll(mg & below) : no + ng + __popcll(mi & below)] = (unsigned char)e;
    shB[e] = 0.0;
    if (e == 0) {
      sh_count[0] = no, sh_count[1] = no + ng, sh_count[2] = no + ng + __popcll(mi);
      sh_gt4[0] = (unsigned)g4, sh_gt4[1] = (unsigned)(g4 >> 32);
      sh_open[0] = (unsigned)mo, sh_open[1] = (unsigned)(mo >> 32);
      // a closed block's carry-out is 0: ready; so is, as far as the wait inside the workgroup goes, a block of another
      // run (its carry-out is fetched from global memory before that wait)
      sh_ready[0] = ~(unsigned)mo_m, sh_ready[1] = ~(unsigned)(mo_m >> 32);
    }
    // (a wavefront's LDS operations complete in order: the flag follows everything above)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (e == 0) __hip_atomic_store(&sh_listed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (mine && idle) {
      kh_status st_{};
      st_.bb_power = sh_bbp[e];
      st_.foffset = NAN; // fm.c:116,121 leave both untouched while the squelch is closed
      st_.pdeviation = NAN;
      st_.agc_gain = NAN;
      st_.squelch_open = so;
      fm2_store_status(F.status + (size_t)C.ch * F.status_stride + b, st_);
      if (b == nblocks - 1) { // state carried to the next launch (fm.c:26,32,33), into the other parity slot
        DemodState *st = F.states + C.ch;
        st->fm_lastaudio[par ^ 1] = C.flat ? C.lastaudio_in : 0.f;
        st->fm_squelch_open[par ^ 1] = so;
        st->fm_state[par ^ 1] = make_float2(0.f, 0.f);
      }
    }
  }
  // the other work wavefronts wait for the list (bounded: the first one is resident with them and waits for nothing)
  for (int spin = 0; spin < (1 << 22) && !__hip_atomic_load(&sh_listed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); spin++)
    __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  KH_T(2)
  int const n_gen = sh_count[1], n_all = sh_count[2]; // (sh_count[0]: the open items, the first of the n_gen)
  unsigned long long const g4 = (unsigned long long)sh_gt4[0] | ((unsigned long long)sh_gt4[1] << 32);
  bool fetched = false; // (split channels) this group has the carry-outs of the runs before b_lo in shB
  // the open items, then the closed ones that carry a decaying lastaudio: a group takes the next one of the list (in list
  // order: what an item waits for was taken before it, by a group that publishes before it waits)
  for (;;) {
    int k = 0;
    if (l == 0) k = __hip_atomic_fetch_add(&sh_next[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    k = __shfl(k, 0, G);
    if (k >= n_gen) break;
    int const e = sh_list[k];
    int const ci = e / nblocks, b = e - ci * nblocks;
    Fm2Chan const C = sh_ch[ci];
    int const ch = C.ch, flat = C.flat;
    bool open, open_p;
    int so;
    squelch(g4, e, b, C.so_in, open, open_p, so);
    const float2 *bbc = F.bb + (size_t)ch * F.bb_stride;

    float2 y[K];
    float ang[K], wgt[K];
    double x[K];
    double cin0 = 0.0, Bsum = 0.0;
    float angsum = 0.f;
    float2 ylast = make_float2(0.f, 0.f);
    if (open) {
      const float4 *yi = reinterpret_cast<const float4 *>(bbc + (size_t)b * OLEN + K * l);
#pragma unroll
      for (int j = 0; j < K; j += 2) {
        float4 const t = yi[j / 2];
        y[j] = make_float2(t.x, t.y);
        y[j + 1] = make_float2(t.z, t.w);
      }
      // fm.c:86-87: the sample before; block 0: the carried state; a closed block before leaves 0 (fm.c:123)
      float2 prev = make_float2(__shfl_up(y[K - 1].x, 1, G), __shfl_up(y[K - 1].y, 1, G));
      if (l == 0) {
        prev = make_float2(0.f, 0.f);
        if (b == 0)
          prev = C.state_in;
        else if (open_p)
          prev = bbc[(size_t)b * OLEN - 1];
      }
      ylast = make_float2(__shfl(y[K - 1].x, G - 1, G), __shfl(y[K - 1].y, G - 1, G));
      float const inv_var = 1.f / sh_var[e];
#pragma unroll
      for (int j = 0; j < K; j++) {
        float2 const p = cmulc(y[j], j ? y[j - 1] : prev);
        float const a = fast_atan2f(p.y, p.x);                                 // cargf, fm.c:89
        float w = __builtin_amdgcn_sqrtf(p.x * p.x + p.y * p.y) * inv_var;     // cabsf(p) / fm_variance, fm.c:91
        if (w > 1) w = 1;
        ang[j] = a;
        wgt[j] = w;
        angsum += a;
      }
      angsum = fm2_group_sum<G>(angsum);
      if (!flat) {
        double B = 0.0; // this lane's chunk as the map carry_out = decay^K carry_in + B
#pragma unroll
        for (int j = 0; j < K; j++) {
          x[j] = (double)ang[j] * .114 * (double)C.gain * (double)wgt[j]; // fm.c:110
          B = (B + x[j]) * FM_DECAY;
        }
        // inclusive scan over the group's lanes; every lane's multiplier is decay^K, so at distance o it is decay^(K o)
        double Bi = B;
        double Apow = FM_DECAY_K8;
        static_assert(K == 8, "FM_DECAY_K8 is decay^8");
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
          double const Bp = __shfl_up(Bi, o, G);
          if (l >= o) Bi = fma(Apow, Bp, Bi);
          Apow *= Apow;
        }
        cin0 = __shfl_up(Bi, 1, G); // lastaudio entering this lane's chunk for a zero carry into the block
        if (l == 0) cin0 = 0.0;
        Bsum = __shfl(Bi, G - 1, G);
        if (l == 0) shB[e] = Bsum;
      }
    }
    if (open && !flat) { // publish the carry-out
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (l == 0) {
        __hip_atomic_fetch_or(&sh_ready[e >> 5], 1u << (e & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (part < S - 1) { // the runs after this one need it: the value, and once it has landed the flag
          Fm2Carry *xc = F.xch + (size_t)ch * F.stat_stride + b;
          __hip_atomic_store(&xc->B, (unsigned long long)__double_as_longlong(Bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(&xc->epoch, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // lastaudio on entry to this block: the channel's blocks before it, each c <- decay^olen c + B (fm.c:110-111,130);
    // the open ones among them stand earlier in the list
    double c_in = (double)C.lastaudio_in;
    if (!flat) {
      if (b_lo > 0 && !fetched) { // (S > 1, hence one channel: item e is block e) the open blocks of the runs before
        fetched = true;
        unsigned long long const mo_all = (unsigned long long)sh_open[0] | ((unsigned long long)sh_open[1] << 32);
        for (int j = l; j < b_lo; j += G)
          if (mo_all >> j & 1) {
            Fm2Carry *xc = F.xch + (size_t)ch * F.stat_stride + j;
            for (int spin = 0; spin < (1 << 22); spin++) {
              if (__hip_atomic_load(&xc->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == F.epoch) break;
              __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            shB[j] = __longlong_as_double((long long)__hip_atomic_load(&xc->B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          }
        // (every group of the workgroup writes the same values; a wavefront's LDS operations complete in order, so the
        // group's own lanes read below what its lanes wrote here)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      }
      unsigned long long const need = (1ull << e) - (1ull << (e - b));
      for (int spin = 0; spin < (1 << 22); spin++) {
        unsigned long long const rd = (unsigned long long)__hip_atomic_load(&sh_ready[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) |
                                      ((unsigned long long)__hip_atomic_load(&sh_ready[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << 32);
        if ((rd & need) == need) break;
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double *Bc = shB + ci * nblocks;
      for (int j = 0; j < b; j++) c_in = fma(Ablk, c_in, Bc[j]);
    }

    // ---- fm.c:105-111 (open) / 124-132 (closed): audio, scaleclip, output level
    float s1[K], la[K];
    size_t const obase = (size_t)b * OLEN + K * l;
    float level = 0.f;
    if (!open && (flat || c_in == 0.0)) {
      // a closed block of a channel whose integrator has run down to zero (an idle channel, i.e. most of a large bank):
      // every sample is 0 (fm.c:124-132 with lastaudio == 0), so is its PCM word in either byte order
      *reinterpret_cast<uint4 *>(F.pcm + (size_t)ch * F.pcm_stride + obase) = make_uint4(0u, 0u, 0u, 0u);
      if (F.audio) {
        float4 *ao = reinterpret_cast<float4 *>(F.audio + (size_t)ch * F.pcm_stride + obase);
        ao[0] = ao[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      float lv = 0.f;
      if (open) {
        if (flat) {
#pragma unroll
          for (int j = 0; j < K; j++) s1[j] = ang[j] * C.gain; // fm.c:105
        } else {
          double cin = fma(c_in, dpl, cin0);
#pragma unroll
          for (int j = 0; j < K; j++) {
            cin += x[j];     // samples[n] = lastaudio += ...
            s1[j] = (float)cin;
            cin *= FM_DECAY; // lastaudio *= 0.99376949
          }
        }
      } else {
        double v = c_in * dpl;
#pragma unroll
        for (int j = 0; j < K; j++) {
          s1[j] = (float)v; // fm.c:124-132: the decaying carry alone
          v *= FM_DECAY;
        }
      }
#pragma unroll
      for (int j = 0; j < K; j++) lv += s1[j] * s1[j];
      {
        int16_t *po = F.pcm + (size_t)ch * F.pcm_stride + obase;
        unsigned wd[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
          wd[u] = (unsigned)(unsigned short)pcm_word(s1[2 * u], F.big_endian) |
                  ((unsigned)(unsigned short)pcm_word(s1[2 * u + 1], F.big_endian) << 16);
        *reinterpret_cast<uint4 *>(po) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
      if (F.audio) {
        float4 *ao = reinterpret_cast<float4 *>(F.audio + (size_t)ch * F.pcm_stride + obase);
        ao[0] = make_float4(s1[0], s1[1], s1[2], s1[3]);
        ao[1] = make_float4(s1[4], s1[5], s1[6], s1[7]);
      }
      level = fm2_group_sum<G>(lv) / OLEN; // fm.c:135-138
    }

    // ---- fm.c:97-102, order dependent: pdev_pos is a running maximum; pdev_neg moves to ang_k at the first later
    // sample that is not a new maximum and whose lastaudio is below the current pdev_neg (see fm_finish_wave in
    // kh_demod_kernels.hip for the relaxation and why its fixed point is the serial result)
    float pdev_pos = 0.f, pdev_neg = 0.f;
    if (open) {
      { // lastaudio when sample n is processed = previous sample * decay (fm.c:110-111); flat: it never changes
        float const prev_lane = __shfl_up(s1[K - 1], 1, G);
#pragma unroll
        for (int j = 0; j < K; j++) {
          float const sp = j ? s1[j - 1] : prev_lane;
          la[j] = flat ? (float)c_in : (float)((double)sp * FM_DECAY);
        }
      }
      float lmax = ang[0];
#pragma unroll
      for (int j = 1; j < K; j++) lmax = fmaxf(lmax, ang[j]);
      float inc = lmax; // inclusive prefix maximum over the group's lanes
#pragma unroll
      for (int o = 1; o < G; o <<= 1) {
        float const p = __shfl_up(inc, o, G);
        if (l >= o) inc = fmaxf(inc, p);
      }
      float run = __shfl_up(inc, 1, G);
      if (l == 0) run = -INFINITY;
      pdev_pos = __shfl(inc, G - 1, G);
      float thr[K]; // lastaudio where fm.c:101 is reached, +inf elsewhere
#pragma unroll
      for (int j = 0; j < K; j++) {
        bool const up = ang[j] > run; // fm.c:99: ang > pdev_pos
        thr[j] = (up || (l == 0 && j == 0)) ? INFINITY : la[j];
        run = fmaxf(run, ang[j]);
      }
      float const c_first = __shfl(ang[0], 0, G); // fm.c:97-98
      float mlane = INFINITY;
#pragma unroll
      for (int j = 0; j < K; j++) mlane = fminf(mlane, thr[j]);
      float cin = c_first;
      int has_end = 0, val_end = 0;
      for (int it = 0; it < G; it++) {
        float c = cin;
#pragma unroll
        for (int j = 0; j < K; j++)
          if (thr[j] < c) c = ang[j];
        int has = !(cin <= mlane); // this lane's output is its own (not the input handed on)
        int val = __float_as_int(c);
        // inclusive scan over the group's lanes of "the nearest lane at or before this one that has a value" in DPP
        // moves: shifts by 1, 2, 4, 8 inside the rows of 16, lane 15 of rows 0 and 2 into rows 1 and 3, and (G = 64)
        // lane 31 into the upper half.  Lanes a move does not reach receive has = 0, the identity.
#define KH_PDEV_STEP(CTRL, ROWS)                                                            \
        {                                                                                     \
          int const ph = __builtin_amdgcn_update_dpp(0, has, CTRL, ROWS, 0xf, false);         \
          int const pv = __builtin_amdgcn_update_dpp(0, val, CTRL, ROWS, 0xf, false);         \
          val = has ? val : pv;                                                               \
          has = has ? has : ph;                                                               \
        }
        KH_PDEV_STEP(0x111, 0xf) // row_shr:1
        KH_PDEV_STEP(0x112, 0xf) // row_shr:2
        KH_PDEV_STEP(0x114, 0xf) // row_shr:4
        KH_PDEV_STEP(0x118, 0xf) // row_shr:8
        KH_PDEV_STEP(0x142, 0xa) // row_bcast:15 into rows 1, 3
        if constexpr (G == 64) KH_PDEV_STEP(0x143, 0xc) // row_bcast:31 into rows 2, 3
#undef KH_PDEV_STEP
        int const hp = __builtin_amdgcn_update_dpp(0, has, 0x138, 0xf, 0xf, false); // wave_shr:1
        int const np_ = __builtin_amdgcn_update_dpp(0, val, 0x138, 0xf, 0xf, false);
        float const ncin = (hp && l != 0) ? __int_as_float(np_) : c_first;
        has_end = has, val_end = val; // (of the last round run: read once, after the loop)
        bool const changed = __float_as_int(ncin) != __float_as_int(cin);
        cin = ncin;
        if (!__builtin_amdgcn_ballot_w64(changed)) break;
      }
      pdev_neg = __shfl(has_end, G - 1, G) ? __int_as_float(__shfl(val_end, G - 1, G)) : c_first;
    }
    if (l == 0) {
      kh_status r{};
      r.bb_power = sh_bbp[e];
      r.foffset = NAN; // fm.c:116,121 leave both untouched while the squelch is closed
      r.pdeviation = NAN;
      if (open) {
        float const avg_f = angsum / OLEN; // fm.c:96,114
        r.foffset = (float)((double)(C.dsamprate * avg_f) * (0.5 * M_1_PI)); // fm.c:116
        pdev_pos -= avg_f;
        pdev_neg -= avg_f;
        r.pdeviation = (float)((double)(C.dsamprate * fmaxf(pdev_pos, -pdev_neg)) * (0.5 * M_1_PI)); // fm.c:121
      }
      r.output_level = level;
      r.agc_gain = NAN;
      r.squelch_open = so;
      fm2_store_status(F.status + (size_t)ch * F.status_stride + b, r);
      if (b == nblocks - 1) { // state carried to the next launch (fm.c:26,32,33), into the other parity slot
        DemodState *st = F.states + ch;
        double c = c_in;
        if (!flat) c = fma(Ablk, c_in, Bsum);
        st->fm_lastaudio[par ^ 1] = (float)c;
        st->fm_squelch_open[par ^ 1] = so;
        st->fm_state[par ^ 1] = open ? ylast : make_float2(0.f, 0.f);
      }
    }
  }
  for (;;) { // the idle items: fm.c:124-132 with lastaudio == 0: zeros, in either byte order
    int k = 0;
    if (l == 0) k = __hip_atomic_fetch_add(&sh_next[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    k = n_gen + __shfl(k, 0, G);
    if (k >= n_all) break;
    int const e = sh_list[k], ci = e / nblocks, b = e - ci * nblocks;
    size_t const o = (size_t)sh_ch[ci].ch * F.pcm_stride + (size_t)b * OLEN + K * l;
    *reinterpret_cast<uint4 *>(F.pcm + o) = make_uint4(0u, 0u, 0u, 0u);
    if (F.audio) {
      float4 *ao = reinterpret_cast<float4 *>(F.audio + o);
      ao[0] = ao[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  KH_T(3)
  if (!split) iterate(); // (too few wavefronts for two roles)
}

// KA9Q_HIP_BACK2=0: round 2's kernels (kh_back_a + kh_fm_finish_w), the A/B reference.  One path for every bank size on
// purpose: a channel's PCM then does not depend on how many channels share its bank, i.e. on how a band is sharded over
// GPUs (tests/test_gpu_configs.py::test_config4_full_size_properties).  What that costs: a workgroup per channel keeps
// only as many CUs busy as there are FM channels, so at configs[1] (64 channels x 32 open blocks) the demodulator stage
// takes ~20 us against 13 us for round 2's wavefront per (channel, block); the step there is bound by the front half.
int fm2_usable(int ND, int olen, int nblocks, int n_fm) {
  (void)n_fm;
  static int const env = getenv("KA9Q_HIP_BACK2") ? atoi(getenv("KA9Q_HIP_BACK2")) : 1;
  return env != 0 && chan_wave_supported(ND, olen) && nblocks >= 1 && nblocks <= FM2_MAXT;
}

// n0final (may be null): kh_n0_final's work rides in the launch
int launch_fm2(hipStream_t st, const float2 *baseband, size_t bb_stride, int olen, int nblocks, const DemodDesc *desc,
               DemodState *state, float *audio, int16_t *pcm, size_t pcm_stride, int pcm_big_endian, void *status,
               int status_stride, const int *fm_list, int n_fm, FmBlockStat *stat, const double *decay_pow, int parity,
               const N0FinalArgs *n0final, Fm2Carry *xch, unsigned epoch) {
  if (nblocks <= 0 || n_fm <= 0) return 0;
  if ((olen != 256 && olen != 512) || nblocks > FM2_MAXT) {
    set_error("FM demodulator (workgroup per channel): olen %d / %d blocks unsupported", olen, nblocks);
    return -1;
  }
  N0FinalArgs const Z = n0final ? *n0final : N0FinalArgs{};
  Fm2Args F{nblocks, desc, state, parity, stat, status_stride, baseband, bb_stride, decay_pow, audio, pcm, pcm_stride,
            pcm_big_endian, static_cast<kh_status *>(status), status_stride, fm_list, 1, 0u, nullptr};
  int const G = olen / 8, ppw = 64 / G;
  // Channels per workgroup: one, or in a large bank (>= 1024 FM channels) as many as fill a wavefront's lanes with items.
  // Wavefronts: no more than give every item a group, and about as many as keep every workgroup resident at once (110
  // VGPRs: 4 wavefronts per SIMD, 4096 on the chip).  Measured over the bench shapes (tools/micro/fm2_waves_sweep.sh): a quarter more
  // than fit is still better -- idle channels' workgroups are gone after 6 us, and a busy channel with 32 open blocks
  // needs groups more than it needs an early start (640 channels x 32 blocks: 4 wavefronts 45 us, 8: 31, 16: 42).
  // (KA9Q_HIP_FM2_WAVES, KA9Q_HIP_FM2_CPW: read at every launch -- the tests run one bank under several arrangements)
  const char *const nw_s = getenv("KA9Q_HIP_FM2_WAVES"), *const cpw_s = getenv("KA9Q_HIP_FM2_CPW");
  int const nw_env = nw_s ? atoi(nw_s) : 0, cpw_env = cpw_s ? atoi(cpw_s) : 0;
  int cpw = cpw_env > 0 ? cpw_env : (n_fm >= 1024 ? FM2_MAXCPW : 1);
  if (cpw > FM2_MAXCPW) cpw = FM2_MAXCPW;
  while (cpw > 1 && cpw * nblocks > FM2_MAXT) cpw >>= 1;
  int const nwg = (n_fm + cpw - 1) / cpw;
  // Wavefronts per workgroup: a power of two (they then spread evenly over a CU's four SIMDs)
  auto waves_for = [&](int s) {
    int const wg_items = (cpw * nblocks + s - 1) / s;
    int nw = 1;
    while (nw < 16 && 2 * nw * nwg * s <= 5120 && nw * ppw < wg_items) nw *= 2;
    return nw;
  };
  // A small bank's channels are split over neighbouring workgroups (see the kernel): as many as give every CU one
  // workgroup at most, runs of at least four blocks, and no more wavefronts in all than half of what the chip holds --
  // the other half belongs to the front half's kernels and, in a mixed bank, to the linear demodulators beside this one
  // (config 3's share, 128 FM channels: two workgroups of 16 wavefronts per channel took the whole chip and the step
  // went from 138 to 182 us).  xch == nullptr (a launch captured into a hipGraph would bake the epoch in): never.
  // (KA9Q_HIP_FM2_SPLIT: read at every launch; 1: never, n: n workgroups per channel wherever there are that many blocks)
  const char *const sp_s = getenv("KA9Q_HIP_FM2_SPLIT");
  int const sp_env = sp_s ? atoi(sp_s) : 0;
  int S = 1;
  if (cpw == 1 && xch != nullptr) {
    if (sp_env > 0)
      S = std::min(std::min(sp_env, nblocks), 16);
    else
      for (int s = 2; s <= 8; s *= 2)
        if (s * nwg <= 256 && nblocks / s >= 4 && waves_for(s) * nwg * s <= 2048) S = s;
  }
  F.splits = S, F.epoch = epoch, F.xch = xch;
  int nw = waves_for(S);
  if (nw_env > 0) nw = nw_env > 16 ? 16 : nw_env;
  // KA9Q_HIP_FM2_ITERATE_ALL=1 (test switch, read at every launch): the squelch of every block from the iteration, none from r alone
  const char *const it_env = getenv("KA9Q_HIP_FM2_ITERATE_ALL");
  int const it_all = it_env ? atoi(it_env) : 0;
  float const r_lo = it_all ? 0.f : FM2_R_LO, r_hi = it_all ? INFINITY : FM2_R_HI;
  dim3 const grid(nwg * S + (n0final ? n0final->ninputs : 0));
  if (G == 32)
    hipLaunchKernelGGL(kh_fm_demod2<32>, grid, dim3(64 * nw), 0, st, F, nwg, Z, n_fm, cpw, r_lo, r_hi);
  else
    hipLaunchKernelGGL(kh_fm_demod2<64>, grid, dim3(64 * nw), 0, st, F, nwg, Z, n_fm, cpw, r_lo, r_hi);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
#ifdef KH_FM2_TIMES
extern "C" int kh_debug_fm2_times(long long *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kh::kh_fm2_times), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#endif
