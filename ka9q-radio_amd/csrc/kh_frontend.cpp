// kh_frontend.cpp -- host side of the front-end corrections and payload packing (hackrf.c, funcube.c).
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/ka9q_hip.h"
#include "kh_internal.h"

using namespace kh;

namespace {
bool on_device(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError(); // plain host memory: not an error
    return false;
  }
  return a.type == hipMemoryTypeDevice;
}
struct DevBuf { // a device view of caller memory: the pointer itself, or a staged copy
  void *d = nullptr;
  bool owned = false;
  bool ok = true;
  DevBuf(const void *p, size_t bytes, bool upload, hipStream_t st) {
    if (on_device(p)) {
      d = const_cast<void *>(p);
      return;
    }
    owned = true;
    ok = KH_HIP(hipMalloc(&d, bytes ? bytes : 16));
    if (ok && upload) ok = KH_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, st));
  }
  bool download(void *p, size_t bytes, hipStream_t st) {
    return !owned || KH_HIP(hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, st));
  }
  ~DevBuf() {
    if (owned && d) hipFree(d);
  }
};

// Scratch of the calls below (block moments, per-block parameters, the state, energy partials): one device buffer
// per process that only grows, so that a steady stream of calls allocates nothing.  Calls are serialised on it.
std::mutex g_scratch_mutex;
char *g_scratch = nullptr;
size_t g_scratch_bytes = 0;
char *scratch(size_t bytes) {
  if (bytes <= g_scratch_bytes) return g_scratch;
  if (g_scratch) hipFree(g_scratch);
  g_scratch = nullptr;
  g_scratch_bytes = 0;
  size_t const want = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
  if (!KH_HIP(hipMalloc((void **)&g_scratch, want))) return nullptr;
  g_scratch_bytes = want;
  return g_scratch;
}
size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

int correct(int funcube, struct kh_fe_state *state, void *raw, size_t in_bytes, unsigned samples, unsigned nblocks,
            int offset, float rate_factor, float *out_i, float *out_q) {
  if (!state || !raw || (!funcube && (!out_i || !out_q))) {
    set_error("front-end correction: NULL argument");
    return -1;
  }
  if (ensure_init()) return -1;
  if (samples == 0 || nblocks == 0) return 0;
  if ((unsigned long long)samples * nblocks > (1ull << 31) - 1) {
    set_error("front-end correction: more than 2^31 samples in one call");
    return -1;
  }
  hipStream_t st = default_stream();
  size_t const total = (size_t)samples * nblocks;
  DevBuf in(raw, in_bytes, true, st);
  // hackrf: the two float rails; funcube corrects in place (these two then just alias the input view)
  DevBuf oi(funcube ? in.d : (void *)out_i, funcube ? 0 : total * sizeof(float), false, st);
  DevBuf oq(funcube ? in.d : (void *)out_q, funcube ? 0 : total * sizeof(float), false, st);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  size_t const part_bytes = up256((size_t)nblocks * fe_chunks((int)samples) * 6 * sizeof(long long));
  size_t const par_bytes = up256(nblocks * sizeof(FeParams));
  char *sc = scratch(part_bytes + par_bytes + 256);
  long long *part = reinterpret_cast<long long *>(sc);
  FeParams *params = reinterpret_cast<FeParams *>(sc + part_bytes);
  kh_fe_state *dstate = reinterpret_cast<kh_fe_state *>(sc + part_bytes + par_bytes);
  bool ok = in.ok && oi.ok && oq.ok && sc != nullptr;
  ok = ok && KH_HIP(hipMemcpyAsync(dstate, state, sizeof *state, hipMemcpyHostToDevice, st));
  // funcube corrects in place: every sample is read once by the apply kernel's own thread before it is overwritten
  ok = ok && launch_frontend(st, funcube, in.d, (int)samples, (int)nblocks, offset, rate_factor, dstate, part, params,
                             funcube ? nullptr : (float *)oi.d, funcube ? nullptr : (float *)oq.d, funcube ? in.d : nullptr) == 0;
  ok = ok && KH_HIP(hipMemcpyAsync(state, dstate, sizeof *state, hipMemcpyDeviceToHost, st));
  if (funcube)
    ok = ok && in.download(raw, in_bytes, st);
  else
    ok = ok && oi.download(out_i, total * sizeof(float), st) && oq.download(out_q, total * sizeof(float), st);
  ok = ok && KH_HIP(hipStreamSynchronize(st));
  return ok ? 0 : -1;
}
} // namespace

extern "C" {

int kh_hackrf_correct(struct kh_fe_state *state, const int8_t *iq, unsigned int samples, unsigned int nblocks, int offset,
                      float rate_factor, float *out_i, float *out_q) {
  return correct(0, state, const_cast<int8_t *>(iq), (size_t)samples * nblocks * 2, samples, nblocks, offset, rate_factor,
                 out_i, out_q);
}

int kh_funcube_correct(struct kh_fe_state *state, int16_t *sampbuf, unsigned int blocksize, unsigned int nblocks,
                       float rate_factor) {
  return correct(1, state, sampbuf, (size_t)blocksize * nblocks * 4, blocksize, nblocks, 0, rate_factor, nullptr, nullptr);
}

int kh_hackrf_pack(int payload, const float *ip, const float *qp, unsigned int chunk, float filter_atten, void *out,
                   float *output_energy) {
  if (!ip || !qp || !out) {
    set_error("kh_hackrf_pack: NULL argument");
    return -1;
  }
  size_t bps;
  switch (payload) {
  case KH_PT_IQ16LE: bps = 4; break;
  case KH_PT_IQ12: bps = 3; break;
  case KH_PT_IQ8: bps = 2; break;
  default: set_error("kh_hackrf_pack: payload type %d is not one of IQ_PT, IQ_PT12, IQ_PT8", payload); return -1;
  }
  if (ensure_init()) return -1;
  if (output_energy) *output_energy = 0.f;
  if (chunk == 0) return 0;
  hipStream_t st = default_stream();
  DevBuf di(ip, chunk * sizeof(float), true, st), dq(qp, chunk * sizeof(float), true, st), dout(out, chunk * bps, false, st);
  int const groups = fe_pack_groups((int)chunk);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  float *epart = reinterpret_cast<float *>(scratch(groups * sizeof(float)));
  bool ok = di.ok && dq.ok && dout.ok && epart != nullptr;
  ok = ok && launch_fe_pack(st, payload, (const float *)di.d, (const float *)dq.d, (int)chunk, filter_atten, dout.d, epart) == 0;
  std::vector<float> e(groups);
  ok = ok && KH_HIP(hipMemcpyAsync(e.data(), epart, groups * sizeof(float), hipMemcpyDeviceToHost, st));
  ok = ok && dout.download(out, chunk * bps, st);
  ok = ok && KH_HIP(hipStreamSynchronize(st));
  if (!ok) return -1;
  float total = 0.f;
  for (float v : e) total += v;
  if (output_energy) *output_energy = total;
  return 0;
}

} // extern "C"
