// kh_chan2_kernels.hip -- filter output for complex channels with N_dec = 1024 or 512 and olen = N_dec/2: 32 points per lane.
//
// Reference behaviour restated here (citations into the reference tree):
//   execute_filter_output  filter.c:240-289 (complex -> complex bin select/rotate x response), 371-382 (CROSS_CONJ
//                          fold, backward DFT of N/D points), filter.c:197,206 (the last olen points are the output)
//   demod_fm               fm.c:53-68 (block statistics: baseband power, amplitude mean and variance)
//
// Why a second form of kh_chan_kernels.hip's kernel: that one (16 points per lane, the 32/64-point second stage split
// two/four ways with every lane re-reading its whole row) issues ~1030 (N_dec = 512, two pairs) to ~1240 (1024, one
// pair) vector instructions per wavefront before any demodulator work, and the back half of a large bank is bound by
// exactly that (profiles/r03_base_*_sq.txt: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = 0.8-0.95 of the kernel's duration).
// Here RB = N_dec/32 lanes own one (channel, block), 64/RB pairs share a wavefront, and both stages are whole
// register transforms with nothing computed twice:
//   stage A  lane l < RB: 32-point transform over a of f[RB a + l], twiddle W_ND^{l c} from a table laid out [c][l]
//   exchange in two halves of 16 rows through LDS (half the footprint, so four waves per SIMD fit)
//   stage B  RB = 32: lane l transforms row c = l (32 points); RB = 16: rows l and l + 16 (16 points each)
// and the kept half of the outputs ends up as y[m], m < 16: sample l + RB m of the block.  The FM demodulator proper
// no longer rides here (kh_fm2_kernels.hip): an idle channel's block then costs the filter and three sums, not a
// discriminator, an integrator scan and fm_snr's iteration.
//
// Roofline: HBM.  algorithmic bytes per (channel, block): 8*ND spectrum bins + 8*ND response + 8*olen baseband.
#include "kh_internal.h"
#include "kh_fft.cuh"
#include "kh_regfft.cuh"
#include "kh_n0.cuh"
#include <cstdlib>

namespace kh {

enum { C2_COMPLEX = 1, C2_CROSS_CONJ = 2 }; // filter.h:18-23

struct Chan2Args {
  const float2 *spec;
  size_t spec_in_stride;
  int N;
  const ChanDesc *chans;
  const float2 *response;
  float2 *out;
  size_t out_stride;
  const float2 *tw;   // chan2_twiddle(ND): [32][RB], exp(+2 pi i l c / ND)
  FmBlockStat *stat;  // FM channels: bb_power, variance and (in .snr) the ratio fm_snr is fed, avg^2 / variance; may be null
  int stat_stride;
  int nchan, nblocks, npairs;
  int store_all;      // 0: FM channels' blocks whose squelch is certainly closed are not stored
  unsigned magic;     // ceil(2^32 / nblocks): pair -> channel without a division
};

__device__ __forceinline__ void c2_wave_sync() { // see wave_sync() in kh_chan_kernels.hip
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int R>
__device__ __forceinline__ float c2_group_sum(float v) { // over the R lanes of a group; every lane gets the total
#pragma unroll
  for (int o = R / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, R);
  return v;
}

constexpr int C2_RA = 32;
template <int RB>
struct C2Shape {
  static constexpr int ND = C2_RA * RB, OLEN = ND / 2, PAIRS = 64 / RB, PITCH = RB + 2, H = ND / 2 + 1;
  static constexpr int PAIR_LDS = 16 * PITCH;       // float2 per pair: half the rows at a time
  static constexpr int WAVE_LDS = PAIRS * PAIR_LDS; // 1088 (RB = 32) or 1152 (RB = 16) float2
};
constexpr int C2_WAVE_LDS = 1152;

// ---- filter.c:274-289: f[p] = response[p] * fdomain[m], p = (ND/2+1+i) mod ND, m = (N-ND/2+1+rotate+i) mod N.
// Lane l holds p = RB a + l, a < 32.  i = p - H mod ND is linear in l except where it wraps: i = RB a + l + ND - H
// for a < 16, RB a + l - H for a > 16, (l - 1) mod ND for a = 16; and m wraps at N only for channels at the band's
// edge.  When no pair of the wavefront wraps (FAST), the 31 regular loads are one base address and immediate offsets.
// rp: response + l, in global memory or in LDS (the two calls are separate so that each keeps its address space).
template <int RB, bool FAST>
__device__ __forceinline__ void c2_products(v2f (&v)[C2_RA], const float2 *__restrict__ rp, const float2 *__restrict__ fd, int m0,
                                            int l, int N) {
  constexpr int ND = C2_RA * RB, H = ND / 2 + 1;
  if constexpr (FAST) {
    const float2 *fl = fd + m0 + l;
#pragma unroll
    for (int a = 0; a < C2_RA; a++) {
      if (a == 16) {
        v[a] = pk::cmul(pk::load(rp[RB * a]), pk::load(fd[m0 + ((l - 1) & (ND - 1))]));
      } else {
        int const off = RB * a + (a < 16 ? ND - H : -H); // >= 0
        v[a] = pk::cmul(pk::load(rp[RB * a]), pk::load(fl[off]));
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < C2_RA; a++) {
      int const i = (RB * a + l - H) & (ND - 1);
      int m = m0 + i;
      if (m >= N) m -= N;
      v[a] = pk::cmul(pk::load(rp[RB * a]), pk::load(fd[m]));
    }
  }
}

// The same in two steps, for the wavefronts whose response comes through LDS: the bins are REQUESTED before the response
// and the twiddles are parked in LDS (a wavefront otherwise goes through four memory round trips one after the other --
// twiddles, channel descriptor, response, bins -- which was half of its 8 us at 4096 channels), and multiplied afterwards.
// Same operands in the same order as c2_products: bit-identical.
template <int RB, bool FAST>
__device__ __forceinline__ void c2_load_bins(float2 (&bins)[C2_RA], const float2 *__restrict__ fd, int m0, int l, int N) {
  constexpr int ND = C2_RA * RB, H = ND / 2 + 1;
  if constexpr (FAST) {
    const float2 *fl = fd + m0 + l;
#pragma unroll
    for (int a = 0; a < C2_RA; a++) {
      if (a == 16) {
        bins[a] = fd[m0 + ((l - 1) & (ND - 1))];
      } else {
        int const off = RB * a + (a < 16 ? ND - H : -H); // >= 0
        bins[a] = fl[off];
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < C2_RA; a++) {
      int const i = (RB * a + l - H) & (ND - 1);
      int m = m0 + i;
      if (m >= N) m -= N;
      bins[a] = fd[m];
    }
  }
}
template <int RB>
__device__ __forceinline__ void c2_mul(v2f (&v)[C2_RA], const float2 *rp /*LDS*/, float2 const (&bins)[C2_RA]) {
#pragma unroll
  for (int a = 0; a < C2_RA; a++) v[a] = pk::cmul(pk::load(rp[RB * a]), pk::load(bins[a]));
}

// pairs are numbered block-fastest: pair q = ch * nblocks + b.  A wavefront's pairs are then blocks b .. b+3 of one
// channel (one response fetch serves them) when a launch has several blocks, and neighbouring channels when it has one.
// Called by EVERY wavefront of a channel workgroup (there is a workgroup barrier inside): `active` false = a wavefront beyond
// the launch's last, which computes on the last pair and stores nothing.  twr: this thread's share of the stage-A twiddle
// table, requested by the caller before anything else; it is parked in twl (LDS) here, behind the requests for the bins.
template <int RB, int TWK>
__device__ __forceinline__ void chan2_wave(Chan2Args const &A, float2 *exw, float2 *twl, float2 const (&twr)[TWK], int tw_stride,
                                           int w, int lane64, bool active) {
  using S = C2Shape<RB>;
  constexpr int RA = C2_RA, ND = S::ND, OLEN = S::OLEN, PITCH = S::PITCH, H = S::H;
  static_assert(RB == 32 || RB == 16, "N_dec = 1024 or 512");
  int const g = lane64 / RB, l = lane64 % RB;
  int q = w * S::PAIRS + g;
  bool const valid = active && q < A.npairs;
  if (!valid) q = A.npairs - 1; // padding pair: computes on the last pair, stores nothing
  int const ch = A.magic ? (int)__umulhi((unsigned)q, A.magic) : q, b = q - ch * A.nblocks; // (magic == 0: one block per launch)
  float2 *ex = exw + g * S::PAIR_LDS;
  ChanDesc const cd = A.chans[ch];
  int const N = A.N;
  const float2 *fd = A.spec + (size_t)cd.input * A.spec_in_stride + (size_t)b * N;
  // The response: each lane needs response[RB a + l], a < 32 -- the same 32 values in every group of the wavefront when its
  // pairs are blocks of one channel (block-fastest numbering).  Fetching them per lane costs the texture path a 128-byte
  // access per group and a; instead the wavefront loads the N_dec values once, 512 contiguous bytes an instruction,
  // parks them in its exchange region and every lane picks its 32 from LDS (profiles/r03_s2_c4all_mem.txt: 1680 L1
  // accesses per wavefront before this, more cycles on the CU's one texture unit than the kernel's VALU work on its four SIMDs).
  bool const one_channel = __builtin_amdgcn_ballot_w64(ch != __builtin_amdgcn_readfirstlane(ch)) == 0;
  int m0 = N - ND / 2 + 1 + cd.rotate;
  if (m0 >= N) m0 -= N;
  if (m0 < 0) m0 += N;
  bool const fast = __builtin_amdgcn_ballot_w64(m0 + ND > N) == 0;
  v2f v[RA]; // (packed pairs: kh_regfft.cuh)
  float2 rr[ND / 64], bins[RA];
  if (one_channel) { // requests only: the response (contiguous), then the bins
    const float2 *rc = A.response + (size_t)ch * ND + lane64;
#pragma unroll
    for (int k = 0; k < ND / 64; k++) rr[k] = rc[64 * k];
    if (fast)
      c2_load_bins<RB, true>(bins, fd, m0, l, N);
    else
      c2_load_bins<RB, false>(bins, fd, m0, l, N);
  }
  // the twiddles (every wavefront of the workgroup its share) and this wavefront's response go to LDS
#pragma unroll
  for (int k = 0; k < TWK; k++) twl[threadIdx.x + tw_stride * k] = twr[k];
  if (one_channel) {
#pragma unroll
    for (int k = 0; k < ND / 64; k++) exw[lane64 + 64 * k] = rr[k];
  }
  __syncthreads(); // (the workgroup's only barrier: twl; it orders this wavefront's own region as well)
  if (one_channel) {
    c2_mul<RB>(v, exw + l, bins);
    c2_wave_sync(); // (the exchange below reuses the region)
  } else {
    const float2 *rp = A.response + (size_t)ch * ND + l;
    if (fast)
      c2_products<RB, true>(v, rp, fd, m0, l, N);
    else
      c2_products<RB, false>(v, rp, fd, m0, l, N);
  }
  if (__builtin_amdgcn_ballot_w64(cd.out_type == C2_CROSS_CONJ) != 0) { // filter.c:371-381, on the fly: the mirrored bin as well
    if (cd.out_type == C2_CROSS_CONJ) {
#pragma unroll
      for (int a = 0; a < RA; a++) {
        int const p = RB * a + l;
        if (p != 0 && p != ND / 2) {
          int const dn = ND - p;
          int const i = (dn - H) & (ND - 1);
          int m = m0 + i;
          if (m >= N) m -= N;
          float2 const o = cmul(A.response[(size_t)ch * ND + dn], fd[m]);
          float2 const va = pk::store(v[a]);
          v[a] = pk::load(p < ND / 2 ? cadd(va, cconj(o)) : csub(va, cconj(o)));
        }
      }
    }
  }

  // ---- stage A
  pk::reg_fft<RA, +1>(v);
  {
    const float2 *tw = twl + l;
#pragma unroll
    for (int c = 1; c < RA; c++) v[c] = pk::cmul(v[c], pk::load(tw[c * RB]));
  }

  // ---- exchange + stage B
  float2 y[16]; // sample l + RB m of the block (filter.c:197,206: the last OLEN points)
  if constexpr (RB == 32) {
    v2f u[32];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int c = 0; c < 16; c++) ex[c * PITCH + l] = pk::store(v[16 * h + c]);
      c2_wave_sync();
      if ((l >> 4) == h) {
        const float4 *row = reinterpret_cast<const float4 *>(ex + (l & 15) * PITCH);
#pragma unroll
        for (int j = 0; j < 16; j++) {
          float4 const t = row[j];
          u[2 * j] = (v2f){t.x, t.y};
          u[2 * j + 1] = (v2f){t.z, t.w};
        }
      }
      c2_wave_sync();
    }
    pk::reg_fft<32, +1>(u); // X[l + 32 d]; d >= 16 is the block's output (the unused half is dead code to the compiler)
#pragma unroll
    for (int m = 0; m < 16; m++) y[m] = pk::store(u[16 + m]);
  } else {
    v2f u0[16], u1[16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int c = 0; c < 16; c++) ex[c * PITCH + l] = pk::store(v[16 * h + c]);
      c2_wave_sync();
      const float4 *row = reinterpret_cast<const float4 *>(ex + l * PITCH);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float4 const t = row[j];
        if (h == 0) {
          u0[2 * j] = (v2f){t.x, t.y};
          u0[2 * j + 1] = (v2f){t.z, t.w};
        } else {
          u1[2 * j] = (v2f){t.x, t.y};
          u1[2 * j + 1] = (v2f){t.z, t.w};
        }
      }
      c2_wave_sync();
    }
    pk::reg_fft<16, +1>(u0); // X[l + 32 d]
    pk::reg_fft<16, +1>(u1); // X[l + 16 + 32 d]
#pragma unroll
    for (int e = 0; e < 8; e++) {
      y[2 * e] = pk::store(u0[8 + e]);
      y[2 * e + 1] = pk::store(u1[8 + e]);
    }
  }

  bool const fm_wave = A.stat != nullptr && __builtin_amdgcn_ballot_w64(cd.fm_stats != 0) != 0;
  bool store = valid;
  if (fm_wave) {
    // ---- fm.c:53-68: baseband power, amplitude mean and variance; fm_snr's argument (kh_fm_demod2 takes it from there)
    float amp[16];
    float bb = 0.f, sa = 0.f;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      float const t = y[m].x * y[m].x + y[m].y * y[m].y;
      bb += t;
      amp[m] = __builtin_amdgcn_sqrtf(t);
      sa += amp[m];
    }
    bb = c2_group_sum<RB>(bb);
    sa = c2_group_sum<RB>(sa);
    float const avg = sa / OLEN;
    float var = 0.f;
#pragma unroll
    for (int m = 0; m < 16; m++) var += (amp[m] - avg) * (amp[m] - avg);
    var = c2_group_sum<RB>(var) / (OLEN - 1);
    float const r = avg * avg / var;
    if (valid && cd.fm_stats && l == 0)
      A.stat[(size_t)ch * A.stat_stride + b] = FmBlockStat{bb / OLEN, var, r, 0.f, make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    // A block whose squelch is closed is never read (kh_fm_demod2), and a block is open only if its SNR or the block
    // before's exceeds 4 (fm.c:73-78), i.e. fm_snr's argument exceeds ~6.1 (fm_snr is increasing; fm_snr(5) = 2.5).  When
    // the caller does not want the FM channels' baseband, blocks that are certainly closed skip the store; the block
    // before is known here when it sits in the neighbouring group of this wavefront.
    float const r_prev = __shfl(r, (lane64 & ~(RB - 1)) - RB, 64);
    bool const prev_known = g > 0 && b > 0;
    if (cd.fm_stats && !A.store_all && prev_known && r <= 5.f && r_prev <= 5.f) store = false;
  }
  if (store) {
    float2 *o = A.out + (size_t)ch * A.out_stride + (size_t)b * OLEN + l;
#pragma unroll
    for (int m = 0; m < 16; m++) o[RB * m] = y[m];
  }
}

template <int RB>
__host__ __device__ constexpr int chan2_waves(int npairs) { return (npairs + C2Shape<RB>::PAIRS - 1) / C2Shape<RB>::PAIRS; }

// Workgroups [0, nchan_wg) carry C2_WAVES channel wavefronts each; the rest are compute_n0's second iteration
// (radio.c:451-462, kh_n0.cuh) over N0_GX slices per (input, block), as in kh_back_a.
// Four wavefronts, one per SIMD: beside the transform's passes (four wavefronts of 105-109 VGPRs fill a SIMD's register
// file; one of this kernel's 152-VGPR wavefronts displaces a whole workgroup of a pass from its CU) a resident workgroup
// then costs the pass exactly one of its own.  With two: channel stage 14.4 instead of 13.1 us alone at configs[1].
constexpr int C2_WAVES = 4;
template <int RB, int OCC>
__global__ __launch_bounds__(64 * C2_WAVES, OCC) void kh_chan2(Chan2Args A, N0Pass1Args Z, int nchan_wg, int nwaves) {
  __shared__ __attribute__((aligned(16))) float2 ex[C2_WAVES * C2_WAVE_LDS];
  __shared__ float2 twl[C2_RA * RB]; // stage A's twiddles, [c][l]: one coalesced copy per workgroup instead of 31 loads per lane
  if ((int)blockIdx.x < nchan_wg) {
    constexpr int TWK = C2_RA * RB / (64 * C2_WAVES);
    static_assert(TWK * 64 * C2_WAVES == C2_RA * RB, "the twiddle table divides evenly over the workgroup");
    float2 twr[TWK]; // requested first, parked in LDS inside chan2_wave
#pragma unroll
    for (int k = 0; k < TWK; k++) twr[k] = A.tw[threadIdx.x + 64 * C2_WAVES * k];
    // workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), each with an L2 of its own: XCD x takes a
    // contiguous eighth of the pairs, so that the bins neighbouring channels share (a 12.5 kHz raster reads every bin
    // 2.5 times) are fetched into ONE L2 once instead of into several from HBM (speed only; any placement is correct)
    int const per = nchan_wg >> 3; // nchan_wg is a multiple of 8
    int const wg = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    int const wave = threadIdx.x >> 6;
    int const w = wg * C2_WAVES + wave;
    bool const active = w < nwaves; // (uniform per wavefront; the others go through the motions for the barrier's sake)
    chan2_wave<RB, TWK>(A, ex + wave * C2_WAVE_LDS, twl, twr, 64 * C2_WAVES, active ? w : nwaves - 1, threadIdx.x & 63, active);
  } else {
    // (bracket mode's tasks were also tried inside the channel workgroups -- requested before the channel work, finished
    // after it: the 40 registers they hold across it cost the 4096-channel bank 12 us and gained 2 us here)
    n0_pass1_body(Z, blockIdx.x - nchan_wg, reinterpret_cast<float *>(ex));
  }
}

template <int RB>
static void launch_chan2_rb(hipStream_t st, Chan2Args const &A, const N0Pass1Args *n0) {
  int const nwaves = chan2_waves<RB>(A.npairs);
  int const nchan_wg = (((nwaves + C2_WAVES - 1) / C2_WAVES) + 7) & ~7;
  N0Pass1Args const Z = n0 ? *n0 : N0Pass1Args{};
  int const extra = n0 ? n0->gx * n0->nblocks * n0->ninputs : 0;
  static int const occ = getenv("KA9Q_HIP_C2_OCC") ? atoi(getenv("KA9Q_HIP_C2_OCC")) : 1; // experiment: 3 forces three wavefronts per SIMD (4 spilled and was slower: removed)
  if (occ == 3)
    hipLaunchKernelGGL((kh_chan2<RB, 3>), dim3(nchan_wg + extra), dim3(64 * C2_WAVES), 0, st, A, Z, nchan_wg, nwaves);
  else
    hipLaunchKernelGGL((kh_chan2<RB, 1>), dim3(nchan_wg + extra), dim3(64 * C2_WAVES), 0, st, A, Z, nchan_wg, nwaves);
}

int launch_chan2(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, const ChanDesc *chans,
                 const float2 *response, int ND, int olen, float2 *out, size_t out_stride, int nblocks, int nchan,
                 FmBlockStat *stat, int stat_stride, int store_all, const N0Pass1Args *n0) {
  if (!chan_wave_supported(ND, olen)) {
    set_error("filter output (32 points per lane): N_dec %d / olen %d unsupported", ND, olen);
    return -1;
  }
  if (nblocks <= 0 || nchan <= 0) return 0;
  if ((long long)nchan * nblocks >= (1ll << 26)) {
    set_error("filter output: %d channels x %d blocks exceed one launch", nchan, nblocks);
    return -1;
  }
  const float2 *tw = chan2_twiddle(ND);
  if (!tw) return -1;
  unsigned const magic = (unsigned)(((1ull << 32) + (unsigned)nblocks - 1) / (unsigned)nblocks);
  Chan2Args const A{spec, spec_in_stride, N, chans, response, out, out_stride, tw, stat, stat_stride, nchan, nblocks,
                    nchan * nblocks, store_all, nblocks == 1 ? 0u : magic};
  if (ND == 1024)
    launch_chan2_rb<32>(st, A, n0);
  else
    launch_chan2_rb<16>(st, A, n0);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
