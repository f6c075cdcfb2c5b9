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
// A workgroup serves `cpw` FM channels (1, or up to 8 in a large bank); its items are their (channel, block) pairs, at most
// 64.  It works in three steps:
//   1. a lane per item: the block's statistics, fm_snr's iteration, the squelch; the open items go to a list in LDS.
//      (The iteration costs a wavefront its slowest lane whatever the number of lanes in use, hence cpw > 1 in a large
//      bank; with cpw == 1 a channel's blocks are dealt over all wavefronts instead, which shortens that wait.)
//   2. the OPEN items, a group of G = olen/8 lanes each, as many per pass as the workgroup has groups: discriminator and
//      de-emphasis from a zero carry, the block's carry-out to LDS, one barrier, then the carry-in's contribution (linear).
//      A bank is mostly idle channels, so the groups of a workgroup go to the blocks that need them: one busy channel
//      among idle ones gets every group (measured with time stamps per workgroup, 4096 channels x 16 blocks, one
//      wavefront per channel: idle channels done after 15 us, the busy one after 34 us -- eight passes of its own).
//   3. the closed items, a group each, no barrier: zeros (or the decaying carry, fm.c:124-132) and the status.
template <int G>
__global__ __launch_bounds__(1024) void kh_fm_demod2(Fm2Args F, int nfm_wg, N0FinalArgs Z, int n_fm, int cpw) {
  constexpr int K = 8, OLEN = K * G, PPW = 64 / G;
  static_assert(G == 32 || G == 64, "olen = 256 or 512");
  __shared__ double shB[FM2_MAXT]; // a block's integrator carry-out for a zero carry-in (0: closed)
  __shared__ __attribute__((aligned(16))) double sh_coef[KH_XI_SEGS * (KH_XI_DEG + 1)];
  __shared__ float sh_snr[FM2_MAXT], sh_bbp[FM2_MAXT], sh_var[FM2_MAXT];
  __shared__ unsigned char sh_list[FM2_MAXT]; // the open items in ascending order, then the closed ones
  __shared__ int sh_count[3];
  __shared__ Fm2Chan sh_ch[FM2_MAXCPW];
  if ((int)blockIdx.x >= nfm_wg) { // kh_n0_final's work (one workgroup per input) rides in this launch
    n0_final_body(Z, blockIdx.x - nfm_wg, sh_snr, sh_bbp);
    return;
  }
  KH_T(0)
  int const lane64 = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  int const l = lane64 % G, gid = wave * PPW + lane64 / G, NG = nwv * PPW;
  int const par = F.par, nblocks = F.nblocks, items = cpw * nblocks; // items <= 64
  double const Ablk = F.decay_pow[OLEN];
  double const dpl = F.decay_pow[K * l]; // decay^(K l)
  // ---- step 1.  fm.c:53-70: statistics and the Rice-corrected SNR of every item
  {
    if ((int)threadIdx.x < cpw) {
      int const chi = threadIdx.x * nfm_wg + blockIdx.x; // neighbouring channels (busy together, if any are) go to different workgroups
      Fm2Chan c{};
      c.ch = -1;
      if (chi < n_fm) {
        c.ch = F.list[chi];
        const DemodDesc *dd = F.descs + c.ch;
        const DemodState *st = F.states + c.ch;
        c.flat = dd->flat, c.dsamprate = dd->dsamprate, c.gain = dd->fm_gain;
        c.so_in = st->fm_squelch_open[par], c.state_in = st->fm_state[par], c.lastaudio_in = st->fm_lastaudio[par];
      }
      sh_ch[threadIdx.x] = c;
    }
    for (int k = threadIdx.x; k < KH_XI_SEGS * (KH_XI_DEG + 1); k += blockDim.x) sh_coef[k] = (&kh_xi_coeffs[0][0])[k];
    // Which wavefronts iterate.  A round is ~26 dependent vector instructions whatever the number of lanes in use, and the
    // wavefronts that share a SIMD take turns (time stamps, 64 channels x 32 blocks dealt over 16 wavefronts, four to a SIMD:
    // 270 ns a round; one wavefront per SIMD: ~60 ns).  So: cpw == 1: the first four wavefronts, one per SIMD, the items
    // dealt round-robin (the blocks of idle channels sit at fm_snr's threshold, where the round count varies most, 0 .. 65:
    // fewer lanes per wavefront shorten the wait for the slowest); cpw > 1: one wavefront, a different one in the
    // workgroups that are likely to share a CU (they are dealt over 8 XCDs x 32 CUs).
    int const nws = cpw > 1 ? 1 : (nwv < 4 ? nwv : 4), w0 = cpw > 1 ? (int)(blockIdx.x >> 8) % nwv : 0;
    int const ws = wave - w0; // 0 .. nws-1: iterates
    int const per = (ws >= 0 && ws < nws) ? (items + nws - 1) / nws : 0;
    int const e = lane64 < per ? lane64 * nws + ws : items;
    float r = 0.f; // kh_chan2 left avg^2 / variance in .snr
    if (e < items) {
      int const ce = e / nblocks, be = e - ce * nblocks, che = ce * nfm_wg + blockIdx.x;
      if (che < n_fm) {
        const FmBlockStat *fs = F.stat + (size_t)F.list[che] * F.stat_stride + be;
        float const bbp = fs->bb_power, var = fs->variance;
        r = fs->snr;
        sh_bbp[e] = bbp, sh_var[e] = var;
      }
    }
    __syncthreads();
    KH_T(1)
    if (lane64 < per) { // (spare lanes carry r = 0 and leave at once)
      float const snr = fmaxf(0.0f, (float)fm_snr_lane((double)r, sh_coef));
      if (e < items) sh_snr[e] = snr;
    }
  }
  __syncthreads();
  // fm.c:73-78 in closed form (fm_counter_after, kh_demod_kernels.hip): the counter after block j is 1 if snr_j > 4, else
  // max(so_in - 1, 0) for j = 0 and 0 for j >= 1; a block demodulates iff snr > 4 or the counter before it is positive
  auto squelch = [&](int e, int b, int so_in, bool &open, bool &open_p, int &so) {
    float const snr_b = sh_snr[e];
    float const snr_p = b >= 1 ? sh_snr[e - 1] : 0.f;
    float const snr_pp = b >= 2 ? sh_snr[e - 2] : 0.f;
    int const dec_in = so_in - 1 > 0 ? so_in - 1 : 0;
    int const cnt_pp = b >= 2 ? (snr_pp > 4.f ? 1 : (b == 2 ? dec_in : 0)) : so_in; // counter before block b - 1
    int const cnt_p = b >= 1 ? (snr_p > 4.f ? 1 : (b == 1 ? dec_in : 0)) : so_in;  // counter before block b
    open = snr_b > 4.f || cnt_p > 0;
    open_p = b >= 1 && (snr_p > 4.f || cnt_pp > 0);
    so = snr_b > 4.f ? 1 : (b == 0 ? dec_in : 0);
  };
  // the status of a block whose squelch is closed and whose audio is all zeros
  auto idle_status = [&](int ch, int b, int e, int so) {
    kh_status r;
    r.bb_power = sh_bbp[e];
    r.snr = sh_snr[e];
    r.foffset = NAN; // fm.c:116,121 leave both untouched while the squelch is closed
    r.pdeviation = NAN;
    r.cphase = 0.f;
    r.lock_timer = 0.f;
    r.pll_lock = 0;
    r.output_level = 0.f;
    r.agc_gain = NAN;
    r.squelch_open = so;
    r.reserved[0] = r.reserved[1] = 0.f;
    F.status[(size_t)ch * F.status_stride + b] = r;
  };
  if (wave == 0) {
    int const e = lane64, ce = e / nblocks, b = e - ce * nblocks;
    bool const valid = e < items && ce * nfm_wg + (int)blockIdx.x < n_fm;
    bool open = false, open_p;
    int so = 0;
    Fm2Chan const C = sh_ch[ce < cpw ? ce : 0];
    if (valid) squelch(e, b, C.so_in, open, open_p, so);
    unsigned long long const mo = __builtin_amdgcn_ballot_w64(valid && open);
    // idle: closed, and the integrator holds 0 on entry (no open block of the channel before this one, lastaudio 0 from
    // the launch before; fma(A, 0, 0) = 0) or is not in use (flat).  All there is to such a block is its status here and
    // olen zeros of PCM below; in a large bank that is nearly every item.
    bool const idle = valid && !open &&
                      (C.flat || (C.lastaudio_in == 0.f && !(mo & ((1ull << e) - (1ull << (e - b))))));
    unsigned long long const mg = __builtin_amdgcn_ballot_w64(valid && !open && !idle), mi = __builtin_amdgcn_ballot_w64(idle);
    unsigned long long const below = (1ull << e) - 1;
    int const no = __popcll(mo), ng = __popcll(mg);
    if (valid)
      sh_list[open ? __popcll(mo & below) : !idle ? no + __popcll(mg & below) : no + ng + __popcll(mi & below)] = (unsigned char)e;
    if (e == 0) sh_count[0] = no, sh_count[1] = no + ng, sh_count[2] = no + ng + __popcll(mi);
    shB[e] = 0.0;
    if (idle) {
      idle_status(C.ch, b, e, so);
      if (b == nblocks - 1) { // state carried to the next launch (fm.c:26,32,33), into the other parity slot
        DemodState *st = F.states + C.ch;
        st->fm_lastaudio[par ^ 1] = C.flat ? C.lastaudio_in : 0.f;
        st->fm_squelch_open[par ^ 1] = so;
        st->fm_state[par ^ 1] = make_float2(0.f, 0.f);
      }
    }
  }
  __syncthreads();
  KH_T(2)
  int const n_open = sh_count[0], n_gen = sh_count[1], n_all = sh_count[2];
  for (int k = n_gen + gid; k < n_all; k += NG) { // fm.c:124-132 with lastaudio == 0: zeros, in either byte order
    int const e = sh_list[k], ci = e / nblocks, b = e - ci * nblocks;
    size_t const o = (size_t)sh_ch[ci].ch * F.pcm_stride + (size_t)b * OLEN + K * l;
    *reinterpret_cast<uint4 *>(F.pcm + o) = make_uint4(0u, 0u, 0u, 0u);
    if (F.audio) {
      float4 *ao = reinterpret_cast<float4 *>(F.audio + o);
      ao[0] = ao[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // passes over the open items (uniform bounds: every thread meets every barrier), then the closed ones that carry a
  // decaying lastaudio (a pass is all open items or all closed ones: the closed ones start at a pass of their own)
  for (int p0 = 0; p0 < n_gen; p0 = (p0 < n_open && p0 + NG > n_open) ? n_open : p0 + NG) {
    bool const open_pass = p0 < n_open;
    int const k = p0 + gid;
    bool const have = k < (open_pass ? n_open : n_gen);
    int const e = have ? sh_list[k] : 0;
    int const ci = e / nblocks, b = e - ci * nblocks;
    Fm2Chan const C = sh_ch[ci];
    int const ch = C.ch < 0 ? 0 : C.ch, flat = C.flat;
    bool open = false, open_p = false;
    int so = 0;
    if (have) squelch(e, b, C.so_in, open, open_p, so);
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
    if (open_pass) __syncthreads(); // (uniform: p0 and n_open are)
    if (open_pass && p0 == 0) { KH_T(4) }
    if (!have) continue;
    // lastaudio on entry to this block: the channel's blocks before it, each c <- decay^olen c + B (fm.c:110-111,130);
    // the open ones among them stand earlier in the list, so their B is in LDS by now
    double c_in = (double)C.lastaudio_in;
    if (!flat) {
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

    if (open_pass && p0 == 0) { KH_T(5) }
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
      float cin = c_first, last_val = c_first;
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
        last_val = __shfl(has, G - 1, G) ? __int_as_float(__shfl(val, G - 1, G)) : c_first;
        bool const changed = __float_as_int(ncin) != __float_as_int(cin);
        cin = ncin;
        if (!__builtin_amdgcn_ballot_w64(changed)) break;
#ifdef KH_FM2_TIMES
        if (threadIdx.x == 0 && blockIdx.x < 8192) kh_fm2_times[blockIdx.x * 8 + 7] = it;
#endif
      }
      pdev_neg = last_val;
    }
    if (open_pass && p0 == 0) { KH_T(6) }
    if (l == 0) {
      kh_status r;
      r.bb_power = sh_bbp[e];
      r.snr = sh_snr[e];
      r.foffset = NAN; // fm.c:116,121 leave both untouched while the squelch is closed
      r.pdeviation = NAN;
      if (open) {
        float const avg_f = angsum / OLEN; // fm.c:96,114
        r.foffset = (float)((double)(C.dsamprate * avg_f) * (0.5 * M_1_PI)); // fm.c:116
        pdev_pos -= avg_f;
        pdev_neg -= avg_f;
        r.pdeviation = (float)((double)(C.dsamprate * fmaxf(pdev_pos, -pdev_neg)) * (0.5 * M_1_PI)); // fm.c:121
      }
      r.cphase = 0.f;
      r.lock_timer = 0.f;
      r.pll_lock = 0;
      r.output_level = level;
      r.agc_gain = NAN;
      r.squelch_open = so;
      r.reserved[0] = r.reserved[1] = 0.f;
      F.status[(size_t)ch * F.status_stride + b] = r;
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
  KH_T(3)
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
               const N0FinalArgs *n0final) {
  if (nblocks <= 0 || n_fm <= 0) return 0;
  if ((olen != 256 && olen != 512) || nblocks > FM2_MAXT) {
    set_error("FM demodulator (workgroup per channel): olen %d / %d blocks unsupported", olen, nblocks);
    return -1;
  }
  N0FinalArgs const Z = n0final ? *n0final : N0FinalArgs{};
  Fm2Args const F{nblocks, desc, state, parity, stat, status_stride, baseband, bb_stride, decay_pow, audio, pcm, pcm_stride,
                  pcm_big_endian, static_cast<kh_status *>(status), status_stride, fm_list};
  int const G = olen / 8, ppw = 64 / G;
  // Channels per workgroup: one, or in a large bank (>= 1024 FM channels) as many as fill a wavefront's lanes with items.
  // Wavefronts: as many as keep every workgroup resident at once (5 wavefronts per SIMD, 5120 on the chip) -- a workgroup
  // is a chain of dependent round trips, so a second round of workgroups doubles the stage -- and no more than give every
  // item a group.
  static int const nw_env = getenv("KA9Q_HIP_FM2_WAVES") ? atoi(getenv("KA9Q_HIP_FM2_WAVES")) : 0;
  static int const cpw_env = getenv("KA9Q_HIP_FM2_CPW") ? atoi(getenv("KA9Q_HIP_FM2_CPW")) : 0;
  int cpw = cpw_env > 0 ? cpw_env : (n_fm >= 1024 ? FM2_MAXCPW : 1);
  if (cpw > FM2_MAXCPW) cpw = FM2_MAXCPW;
  while (cpw > 1 && cpw * nblocks > FM2_MAXT) cpw >>= 1;
  int const nwg = (n_fm + cpw - 1) / cpw;
  int nw = 1; // a power of two: a workgroup's wavefronts then spread evenly over a CU's four SIMDs, and the last
               // workgroups do not wait for a slot (1024 workgroups x 5 wavefronts: one in ten started 15 us late)
  while (nw < 16 && 2 * nw * nwg <= 5120 && nw * ppw < cpw * nblocks) nw *= 2;
  if (nw_env > 0) nw = nw_env > 16 ? 16 : nw_env;
  dim3 const grid(nwg + (n0final ? n0final->ninputs : 0));
  if (G == 32)
    hipLaunchKernelGGL(kh_fm_demod2<32>, grid, dim3(64 * nw), 0, st, F, nwg, Z, n_fm, cpw);
  else
    hipLaunchKernelGGL(kh_fm_demod2<64>, grid, dim3(64 * nw), 0, st, F, nwg, Z, n_fm, cpw);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
#ifdef KH_FM2_TIMES
extern "C" int kh_debug_fm2_times(long long *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kh::kh_fm2_times), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#endif
