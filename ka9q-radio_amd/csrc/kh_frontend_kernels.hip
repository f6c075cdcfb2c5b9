// kh_frontend_kernels.hip -- the per-sample corrections the SDR front ends apply before decimate.c:
//   hackrf.c:822-919 rx_callback   int8 I/Q -> clip at -1, DC removal, I/Q gain balance, phase correction, Fs/4 rotation
//   funcube.c:364-428              int16 I/Q -> the same corrections -> network-order int16
//   hackrf.c:511-573               decimated rails x Filter_atten -> IQ_PT / IQ_PT12 / IQ_PT8 payload + output energy
// The reference updates its estimates (DC, imbalance, sin phi) once per block from sums over that block's *corrected*
// samples, and corrects the next block with them: a chain through the blocks.  The sums are polynomials in the raw
// integer moments (sum i, sum q, sum i^2, sum q^2, sum iq) and the state, so here the moments of ALL blocks are taken
// first, in parallel and exactly (integers), one thread then walks the chain block by block (a few dozen flops each),
// and a third kernel applies every block's parameters to its samples in parallel.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kh_internal.h"
#include "../../include/ka9q_hip.h"

namespace kh {

constexpr int FE_NT = 256, FE_SPT = 16, FE_CHUNK = FE_NT * FE_SPT; // samples per workgroup
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// WIDE = 0: int8 pairs, -128 clipped to -127 (hackrf.c:841-848; SCALE8 * -128 < -1); WIDE = 1: int16 pairs.
// part: [nblocks][nchunks][6] = sum i, sum q, sum i^2, sum q^2, sum iq, clips.  algorithmic bytes: the input once.
template <int WIDE>
__global__ __launch_bounds__(FE_NT) void kh_fe_moments(const void *__restrict__ raw, int samples, int nchunks,
                                                       long long *__restrict__ part) {
  __shared__ long long red[FE_NT / 64][6];
  int const c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  long long m[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < FE_SPT; u++) {
    int const n = c * FE_CHUNK + u * FE_NT + tid;
    if (n >= samples) continue;
    size_t const idx = (size_t)b * samples + n;
    int i, q;
    if (WIDE) {
      short2 const v = static_cast<const short2 *>(raw)[idx];
      i = v.x;
      q = v.y;
    } else {
      char2 const v = static_cast<const char2 *>(raw)[idx];
      i = v.x;
      q = v.y;
      if (q < -127) {
        q = -127;
        m[5]++;
      }
      if (i < -127) {
        i = -127;
        m[5]++;
      }
    }
    m[0] += i;
    m[1] += q;
    m[2] += (long long)i * i;
    m[3] += (long long)q * q;
    m[4] += (long long)i * q;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) m[k] = wave_sum_i64(m[k]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < 6; k++) red[tid >> 6][k] = m[k];
  __syncthreads();
  if (tid < 6) {
    long long t = 0;
    for (int w = 0; w < FE_NT / 64; w++) t += red[w][tid];
    part[((size_t)b * nchunks + c) * 6 + tid] = t;
  }
}

// The chain of hackrf.c:906-917 / funcube.c:417-428 over the launch's blocks.  One workgroup: all threads add up the
// chunk partials of a tile of blocks into LDS, then thread 0 walks that tile's blocks in order.  params[b] = what block
// b is corrected with; *st is left as the reference leaves it after the last block.
constexpr int FE_TILE = 512;
__global__ __launch_bounds__(FE_NT) void kh_fe_chain(const long long *__restrict__ part, int nchunks, int nblocks,
                                                     int samples, int funcube, int offset, float rate_factor,
                                                     kh_fe_state *__restrict__ st, FeParams *__restrict__ params) {
  __shared__ long long tot[FE_TILE][6];
  __shared__ kh_fe_state s;
  if (threadIdx.x == 0) s = *st;
  double const S = funcube ? (double)(float)(1. / 32767) : (double)(float)(1. / 127); // SCALE16 / SCALE8 as floats
  float const DC_alpha = funcube ? 1.0e-6f : .001f;                                    // funcube.c:65, hackrf.c:66
  for (int b0 = 0; b0 < nblocks; b0 += FE_TILE) {
    int const nb = nblocks - b0 < FE_TILE ? nblocks - b0 : FE_TILE;
    __syncthreads();
    for (int e = threadIdx.x; e < nb * 6; e += FE_NT) {
      int const b = e / 6, k = e - 6 * b;
      long long t = 0;
      for (int c = 0; c < nchunks; c++) t += part[((size_t)(b0 + b) * nchunks + c) * 6 + k];
      tot[b][k] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int b = 0; b < nb; b++) {
        FeParams p;
        p.dc_re = s.dc_re;
        p.dc_im = s.dc_im;
        p.gain_i = s.gain_i;
        p.gain_q = s.gain_q;
        p.secphi = s.secphi;
        p.tanphi = s.tanphi;
        p.rotate_phase = s.rotate_phase;
        p.pad = 0;
        params[b0 + b] = p;
        const long long *m = tot[b];
        double const n = (double)samples, dr = s.dc_re, di = s.dc_im;
        double const si = S * (double)m[0], sq = S * (double)m[1];
        float const i_energy = (float)(S * S * (double)m[2] - 2.0 * dr * si + n * dr * dr);
        float const q_energy = (float)(S * S * (double)m[3] - 2.0 * di * sq + n * di * di);
        float const dotprod =
            (float)((double)s.gain_i * (double)s.gain_q * (S * S * (double)m[4] - di * si - dr * sq + n * dr * di));
        s.clips += (int)m[5];
        if (funcube) { // funcube.c:417
          s.dc_re += DC_alpha * ((float)si - samples * s.dc_re);
          s.dc_im += DC_alpha * ((float)sq - samples * s.dc_im);
        } else {       // hackrf.c:906
          s.dc_re += DC_alpha * ((float)si / samples - s.dc_re);
          s.dc_im += DC_alpha * ((float)sq / samples - s.dc_im);
        }
        float const block_energy = i_energy + q_energy;
        if (block_energy > 0) {
          float const rf = funcube ? rate_factor : rate_factor * samples; // funcube.c:421 vs hackrf.c:910
          s.in_power = block_energy / samples;
          s.imbalance += rf * ((i_energy / q_energy) - s.imbalance);
          float const dpn = 2 * dotprod / block_energy;
          s.sinphi += rf * (dpn - s.sinphi);
          s.gain_q = sqrtf((float)(0.5 * (double)(1 + s.imbalance)));
          s.gain_i = sqrtf((float)(0.5 * (1 + 1. / (double)s.imbalance)));
          s.secphi = 1 / sqrtf(1 - s.sinphi * s.sinphi);
          s.tanphi = s.sinphi * s.secphi;
        }
        s.rotate_phase = (s.rotate_phase + (int)(((long long)offset * samples) & 3)) & 3; // hackrf.c:897-898, per sample
      }
    }
  }
  if (threadIdx.x == 0) *st = s;
}

// algorithmic bytes: hackrf 2 in + 8 out per sample; funcube 4 in + 4 out
template <int WIDE>
__global__ __launch_bounds__(FE_NT) void kh_fe_apply(const void *__restrict__ raw, int samples, int offset,
                                                     const FeParams *__restrict__ params, float *__restrict__ out_i,
                                                     float *__restrict__ out_q, short2 *__restrict__ out_be) {
  int const b = blockIdx.y;
  int const n = blockIdx.x * FE_NT + threadIdx.x;
  if (n >= samples) return;
  FeParams const p = params[b];
  size_t const idx = (size_t)b * samples + n;
  float re, im;
  if (WIDE) {
    short2 const v = static_cast<const short2 *>(raw)[idx];
    re = (float)v.x * (float)(1. / 32767); // CMPLXF(i, q) * SCALE16, funcube.c:369
    im = (float)v.y * (float)(1. / 32767);
  } else {
    char2 const v = static_cast<const char2 *>(raw)[idx];
    re = (float)(1. / 127) * (float)v.x; // hackrf.c:838-848
    im = (float)(1. / 127) * (float)v.y;
    if (im < -1) im = -1;
    if (re < -1) re = -1;
  }
  re -= p.dc_re; // hackrf.c:852-867 / funcube.c:372-387
  im -= p.dc_im;
  re *= p.gain_i;
  im *= p.gain_q;
  im = p.secphi * im - p.tanphi * re;
  if (WIDE) { // funcube.c:390-391: (signed short)round(x * SHRT_MAX), then htons
    short const a = (short)round((double)(re * 32767)), c = (short)round((double)(im * 32767));
    unsigned short const ua = (unsigned short)a, uc = (unsigned short)c;
    out_be[idx] = make_short2((short)(unsigned short)((ua << 8) | (ua >> 8)), (short)(unsigned short)((uc << 8) | (uc >> 8)));
  } else {
    int const ph = (p.rotate_phase + (int)(((long long)offset * n) & 3)) & 3; // hackrf.c:878-898
    float oi, oq;
    switch (ph) {
    default:
    case 0: oi = re; oq = im; break;
    case 1: oi = -im; oq = re; break;
    case 2: oi = -re; oq = -im; break;
    case 3: oi = im; oq = -re; break;
    }
    out_i[idx] = oi;
    out_q[idx] = oq;
  }
}

// hackrf.c:511-560.  energy_part: one float per workgroup (added in order on the host side of the ABI).
__device__ __forceinline__ int fe_clip(float s, int lo, int hi) { return s >= hi ? hi : s <= lo ? lo : (int)s; }
__global__ __launch_bounds__(FE_NT) void kh_fe_pack(int pt, const float *__restrict__ ip, const float *__restrict__ qp,
                                                    int chunk, float atten, unsigned char *__restrict__ out,
                                                    float *__restrict__ energy_part) {
  __shared__ float red[FE_NT / 64];
  int const n = blockIdx.x * FE_NT + threadIdx.x;
  float e = 0.f;
  if (n < chunk) {
    float const s = ip[n] * atten, t = qp[n] * atten;
    e = s * s;
    e += t * t;
    if (pt == KH_PT_IQ16LE) {
      short const si = (short)fe_clip(s, -32768, 32767), sq = (short)fe_clip(t, -32768, 32767);
      reinterpret_cast<short2 *>(out)[n] = make_short2(si, sq);
    } else if (pt == KH_PT_IQ12) {
      short const si = (short)fe_clip(s, -32768, 32767), sq = (short)fe_clip(t, -32768, 32767);
      unsigned char *dp = out + 3 * (size_t)n;
      dp[0] = (unsigned char)(si >> 8);
      dp[1] = (unsigned char)((si & 0xf0) | ((sq >> 12) & 0xf));
      dp[2] = (unsigned char)(sq >> 4);
    } else {
      reinterpret_cast<char2 *>(out)[n] = make_char2((signed char)fe_clip(s, -128, 127), (signed char)fe_clip(t, -128, 127));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < FE_NT / 64; w++) t += red[w];
    energy_part[blockIdx.x] = t;
  }
}

int fe_chunks(int samples) { return (samples + FE_CHUNK - 1) / FE_CHUNK; }
int fe_pack_groups(int chunk) { return (chunk + FE_NT - 1) / FE_NT; }

int launch_frontend(hipStream_t st, int funcube, const void *raw, int samples, int nblocks, int offset, float rate_factor,
                    kh_fe_state *state, long long *part, FeParams *params, float *out_i, float *out_q, void *out_be) {
  if (samples <= 0 || nblocks <= 0) return 0;
  int const nchunks = fe_chunks(samples);
  dim3 const gm(nchunks, nblocks), ga((samples + FE_NT - 1) / FE_NT, nblocks);
  if (funcube)
    hipLaunchKernelGGL(kh_fe_moments<1>, gm, dim3(FE_NT), 0, st, raw, samples, nchunks, part);
  else
    hipLaunchKernelGGL(kh_fe_moments<0>, gm, dim3(FE_NT), 0, st, raw, samples, nchunks, part);
  hipLaunchKernelGGL(kh_fe_chain, dim3(1), dim3(FE_NT), 0, st, part, nchunks, nblocks, samples, funcube, offset, rate_factor,
                     state, params);
  if (funcube)
    hipLaunchKernelGGL(kh_fe_apply<1>, ga, dim3(FE_NT), 0, st, raw, samples, offset, params, out_i, out_q,
                       static_cast<short2 *>(out_be));
  else
    hipLaunchKernelGGL(kh_fe_apply<0>, ga, dim3(FE_NT), 0, st, raw, samples, offset, params, out_i, out_q,
                       static_cast<short2 *>(out_be));
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}
int launch_fe_pack(hipStream_t st, int pt, const float *ip, const float *qp, int chunk, float atten, void *out,
                   float *energy_part) {
  if (chunk <= 0) return 0;
  hipLaunchKernelGGL(kh_fe_pack, dim3(fe_pack_groups(chunk)), dim3(FE_NT), 0, st, pt, ip, qp, chunk, atten,
                     static_cast<unsigned char *>(out), energy_part);
  return KH_HIP(hipGetLastError()) ? 0 : -1;
}

// ---------------------------------------------------------------- kh_hackrf_pack_noalloc
// kh_hackrf_pack for callers that cannot allocate (HIP graph capture, tight serving loops): the same pack kernel, its
// per-workgroup energy partial sums in a static device array instead of the library's growing scratch allocation, and
// the final sum (kh_hackrf_pack adds the partials on the host, in order) by one more kernel that adds them in the same
// order.  Every GPU holds its own copy of the array at its own address: the host only ever uses the address
// hipGetSymbolAddress returned for the device (g_pack_scratch[device]), never the array's name.
constexpr int PACK_NOALLOC_GROUPS = KH_PACK_NOALLOC_MAX_CHUNK / FE_NT; // 4096 floats = 16 KiB per GPU
static_assert(PACK_NOALLOC_GROUPS * FE_NT == KH_PACK_NOALLOC_MAX_CHUNK, "the maximum is a whole number of workgroups");
static __device__ __align__(256) float scratch[PACK_NOALLOC_GROUPS];
constexpr int PACK_MAX_DEVICES = 64;
static float *g_pack_scratch[PACK_MAX_DEVICES] = {};

// once per device, with that device current (called from the library's init)
int pack_noalloc_setup(int device) {
  if (device < 0 || device >= PACK_MAX_DEVICES) return 0; // such a device simply has no no-alloc entry point
  float *ptr = nullptr;
  if (!KH_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)))) return -1;
  g_pack_scratch[device] = ptr;
  return 0;
}

// the partials of kh_fe_pack, added in ascending order by one lane as the host loop of kh_hackrf_pack adds them, so
// that the two entry points agree to the last bit.  Cost: a dependent chain of `groups` additions on one lane, ~5 ns each:
// 20 us at the maximum chunk (4096 partials), 0.6 us at a USB transfer's 131072 samples (128 partials).
__global__ __launch_bounds__(FE_NT) void kh_fe_energy_sum(const float *__restrict__ part, int groups, float *__restrict__ out) {
  __shared__ float sh[PACK_NOALLOC_GROUPS];
  for (int k = threadIdx.x; k < groups; k += FE_NT) sh[k] = part[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < groups; k++) t += sh[k];
    *out = t;
  }
}

// What a kernel on `device` can dereference: device memory OF THAT DEVICE, managed memory, and pinned host memory that
// is mapped (through its device-side address).  Returns that address, or nullptr: pageable host memory and another
// GPU's allocation would be a memory fault in the kernel, not an error code.
static void *kernel_address(const void *p, int device) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  switch (a.type) {
  case hipMemoryTypeDevice: return a.device == device ? a.devicePointer : nullptr;
  case hipMemoryTypeManaged: return a.devicePointer ? a.devicePointer : const_cast<void *>(p);
  case hipMemoryTypeHost: return a.devicePointer; // null unless the allocation is mapped
  default: return nullptr;
  }
}

} // namespace kh

extern "C" int kh_hackrf_pack_noalloc(int payload, const float *ip, const float *qp, unsigned int chunk, float filter_atten,
                                      void *out, float *output_energy, void *stream) {
  using namespace kh;
  if (!ip || !qp || !out) {
    set_error("kh_hackrf_pack_noalloc: NULL argument");
    return -1;
  }
  if (payload != KH_PT_IQ16LE && payload != KH_PT_IQ12 && payload != KH_PT_IQ8) {
    set_error("kh_hackrf_pack_noalloc: payload type %d is not one of IQ_PT, IQ_PT12, IQ_PT8", payload);
    return -1;
  }
  if (chunk > KH_PACK_NOALLOC_MAX_CHUNK) {
    set_error("kh_hackrf_pack_noalloc: chunk %u above the maximum of %d samples (use kh_hackrf_pack)", chunk,
              KH_PACK_NOALLOC_MAX_CHUNK);
    return -1;
  }
  int const device = initialised_device();
  if (device < 0) { // initialising creates a stream: not something this entry point may do
    set_error("kh_hackrf_pack_noalloc: call ka9q_hip_init first (this entry point allocates nothing, so it cannot do it)");
    return -1;
  }
  float *const part = device < PACK_MAX_DEVICES ? g_pack_scratch[device] : nullptr;
  if (!part) {
    set_error("kh_hackrf_pack_noalloc: no scratch address for device %d", device);
    return -1;
  }
  if (!KH_HIP(hipSetDevice(device))) return -1;
  // a host pointer handed to a kernel is a GPU memory fault, not an error code: refuse it here
  ip = static_cast<const float *>(kernel_address(ip, device));
  qp = static_cast<const float *>(kernel_address(qp, device));
  out = kernel_address(out, device);
  float *const energy_in = output_energy;
  if (output_energy) output_energy = static_cast<float *>(kernel_address(output_energy, device));
  if (!ip || !qp || !out || (energy_in && !output_energy)) {
    set_error("kh_hackrf_pack_noalloc: ip, qp, out and output_energy must be memory a kernel on device %d can address "
              "(that device's memory, managed memory, or mapped pinned host memory)", device);
    return -1;
  }
  hipStream_t const st = stream ? static_cast<hipStream_t>(stream) : default_stream();
  if (chunk == 0) {
    if (output_energy && !KH_HIP(hipMemsetAsync(output_energy, 0, sizeof(float), st))) return -1;
    return 0;
  }
  int const groups = fe_pack_groups((int)chunk);
  if (launch_fe_pack(st, payload, ip, qp, (int)chunk, filter_atten, out, part)) return -1;
  if (output_energy) {
    hipLaunchKernelGGL(kh_fe_energy_sum, dim3(1), dim3(FE_NT), 0, st, part, groups, output_energy);
    if (!KH_HIP(hipGetLastError())) return -1;
  }
  return 0;
}
