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

// pairs are numbered block-fastest: pair q = ch * nblocks + b.  A wavefront's pairs are then blocks b .. b+3 of one
// channel (one response fetch serves them) when a launch has several blocks, and neighbouring channels when it has one.
template <int RB>
__device__ __forceinline__ void chan2_wave(Chan2Args const &A, float2 *exw, int w, int lane64) {
  using S = C2Shape<RB>;
  constexpr int RA = C2_RA, ND = S::ND, OLEN = S::OLEN, PITCH = S::PITCH, H = S::H;
  static_assert(RB == 32 || RB == 16, "N_dec = 1024 or 512");
  int const g = lane64 / RB, l = lane64 % RB;
  int q = w * S::PAIRS + g;
  bool const valid = q < A.npairs;
  if (!valid) q = A.npairs - 1; // padding pair: computes on the last pair, stores nothing
  int const ch = A.magic ? (int)__umulhi((unsigned)q, A.magic) : q, b = q - ch * A.nblocks; // (magic == 0: one block per launch)
  float2 *ex = exw + g * S::PAIR_LDS;
  ChanDesc const cd = A.chans[ch];
  int const N = A.N;
  const float2 *fd = A.spec + (size_t)cd.input * A.spec_in_stride + (size_t)b * N;
  const float2 *rp = A.response + (size_t)ch * ND + l;

  // ---- filter.c:274-289: f[p] = response[p] * fdomain[m], p = (ND/2+1+i) mod ND, m = (N-ND/2+1+rotate+i) mod N.
  // Lane l holds p = RB a + l, a < 32.  i = p - H mod ND is linear in l except where it wraps: i = RB a + l + ND - H
  // for a < 16, RB a + l - H for a > 16, (l - 1) mod ND for a = 16; and m wraps at N only for channels at the band's
  // edge.  When no pair of the wavefront wraps, the 31 regular loads are one base address and immediate offsets.
  int m0 = N - ND / 2 + 1 + cd.rotate;
  if (m0 >= N) m0 -= N;
  if (m0 < 0) m0 += N;
  float2 v[RA];
  if (__builtin_amdgcn_ballot_w64(m0 + ND > N) == 0) {
    const float2 *fl = fd + m0 + l;
#pragma unroll
    for (int a = 0; a < RA; a++) {
      if (a == 16) {
        v[a] = cmul(rp[RB * a], fd[m0 + ((l - 1) & (ND - 1))]);
      } else {
        int const off = RB * a + (a < 16 ? ND - H : -H); // >= 0
        v[a] = cmul(rp[RB * a], fl[off]);
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < RA; a++) {
      int const i = (RB * a + l - H) & (ND - 1);
      int m = m0 + i;
      if (m >= N) m -= N;
      v[a] = cmul(rp[RB * a], fd[m]);
    }
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
          v[a] = p < ND / 2 ? cadd(v[a], cconj(o)) : csub(v[a], cconj(o));
        }
      }
    }
  }

  // ---- stage A
  reg_fft<RA, +1>(v);
  {
    const float2 *tw = A.tw + l;
#pragma unroll
    for (int c = 1; c < RA; c++) v[c] = cmul(v[c], tw[c * RB]);
  }

  // ---- exchange + stage B
  float2 y[16]; // sample l + RB m of the block (filter.c:197,206: the last OLEN points)
  if constexpr (RB == 32) {
    float2 u[32];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int c = 0; c < 16; c++) ex[c * PITCH + l] = v[16 * h + c];
      c2_wave_sync();
      if ((l >> 4) == h) {
        const float4 *row = reinterpret_cast<const float4 *>(ex + (l & 15) * PITCH);
#pragma unroll
        for (int j = 0; j < 16; j++) {
          float4 const t = row[j];
          u[2 * j] = make_float2(t.x, t.y);
          u[2 * j + 1] = make_float2(t.z, t.w);
        }
      }
      c2_wave_sync();
    }
    reg_fft<32, +1>(u); // X[l + 32 d]; d >= 16 is the block's output (the unused half is dead code to the compiler)
#pragma unroll
    for (int m = 0; m < 16; m++) y[m] = u[16 + m];
  } else {
    float2 u0[16], u1[16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int c = 0; c < 16; c++) ex[c * PITCH + l] = v[16 * h + c];
      c2_wave_sync();
      const float4 *row = reinterpret_cast<const float4 *>(ex + l * PITCH);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float4 const t = row[j];
        if (h == 0) {
          u0[2 * j] = make_float2(t.x, t.y);
          u0[2 * j + 1] = make_float2(t.z, t.w);
        } else {
          u1[2 * j] = make_float2(t.x, t.y);
          u1[2 * j + 1] = make_float2(t.z, t.w);
        }
      }
      c2_wave_sync();
    }
    reg_fft<16, +1>(u0); // X[l + 32 d]
    reg_fft<16, +1>(u1); // X[l + 16 + 32 d]
#pragma unroll
    for (int e = 0; e < 8; e++) {
      y[2 * e] = u0[8 + e];
      y[2 * e + 1] = u1[8 + e];
    }
  }

  if (valid) {
    float2 *o = A.out + (size_t)ch * A.out_stride + (size_t)b * OLEN + l;
#pragma unroll
    for (int m = 0; m < 16; m++) o[RB * m] = y[m];
  }
  if (A.stat == nullptr || __builtin_amdgcn_ballot_w64(cd.fm_stats != 0) == 0) return;

  // ---- fm.c:53-68: baseband power, amplitude mean and variance; fm_snr's argument (kh_fm_snr_lanes takes it from there)
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
  if (valid && cd.fm_stats && l == 0)
    A.stat[(size_t)ch * A.stat_stride + b] =
        FmBlockStat{bb / OLEN, var, avg * avg / var, 0.f, make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
}

template <int RB>
__host__ __device__ constexpr int chan2_waves(int npairs) { return (npairs + C2Shape<RB>::PAIRS - 1) / C2Shape<RB>::PAIRS; }

// Workgroups [0, nchan_wg) carry C2_WAVES channel wavefronts each; the rest are compute_n0's second iteration
// (radio.c:451-462, kh_n0.cuh) over N0_GX slices per (input, block), as in kh_back_a.
constexpr int C2_WAVES = 2;
template <int RB>
__global__ __launch_bounds__(64 * C2_WAVES) void kh_chan2(Chan2Args A, N0Pass1Args Z, int nchan_wg, int nwaves) {
  __shared__ __attribute__((aligned(16))) float2 ex[C2_WAVES * C2_WAVE_LDS];
  if ((int)blockIdx.x < nchan_wg) {
    int const wave = threadIdx.x >> 6;
    int const w = blockIdx.x * C2_WAVES + wave;
    if (w < nwaves) chan2_wave<RB>(A, ex + wave * C2_WAVE_LDS, w, threadIdx.x & 63);
  } else {
    n0_pass1_body(Z, blockIdx.x - nchan_wg, reinterpret_cast<float *>(ex));
  }
}

template <int RB>
static void launch_chan2_rb(hipStream_t st, Chan2Args const &A, const N0Pass1Args *n0) {
  int const nwaves = chan2_waves<RB>(A.npairs);
  int const nchan_wg = (nwaves + C2_WAVES - 1) / C2_WAVES;
  N0Pass1Args const Z = n0 ? *n0 : N0Pass1Args{};
  int const extra = n0 ? n0->gx * n0->nblocks * n0->ninputs : 0;
  hipLaunchKernelGGL(kh_chan2<RB>, dim3(nchan_wg + extra), dim3(64 * C2_WAVES), 0, st, A, Z, nchan_wg, nwaves);
}

int launch_chan2(hipStream_t st, const float2 *spec, size_t spec_in_stride, int N, const ChanDesc *chans,
                 const float2 *response, int ND, int olen, float2 *out, size_t out_stride, int nblocks, int nchan,
                 FmBlockStat *stat, int stat_stride, const N0Pass1Args *n0) {
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
                    nchan * nblocks, nblocks == 1 ? 0u : magic};
  if (ND == 1024)
    launch_chan2_rb<32>(st, A, n0);
  else
    launch_chan2_rb<16>(st, A, n0);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

} // namespace kh
