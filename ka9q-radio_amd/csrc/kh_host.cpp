// kh_host.cpp -- library state, error reporting, twiddle tables, and the filter.h / decimate.h drop-in ABI.
//
// Mirrors the reference's host interface (filter.c:57-702, decimate.h:10-11) with every transform and
// multiply on the GPU.  There is no CPU fallback: without a usable HIP device every call fails.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/ka9q_hip.h"
#include "kh_internal.h"

namespace kh {

// ---------------------------------------------------------------- errors / init
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  if (getenv("KA9Q_HIP_VERBOSE")) fprintf(stderr, "libka9q_hip: %s\n", g_err);
}
bool check_hip(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  set_error("%s failed: %s", what, hipGetErrorString(e));
  return false;
}

static std::mutex g_mutex;
static int g_device = -1;
static hipStream_t g_stream = nullptr;

static int init_locked(int device) {
  if (g_device >= 0) return 0;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device available: libka9q_hip has no CPU fallback");
    return -1;
  }
  if (device < 0 || device >= count) {
    set_error("device %d out of range (%d devices)", device, count);
    return -1;
  }
  if (!KH_HIP(hipSetDevice(device))) return -1;
  hipDeviceProp_t prop;
  if (!KH_HIP(hipGetDeviceProperties(&prop, device))) return -1;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; libka9q_hip is built for gfx950 only", device, prop.gcnArchName);
    return -1;
  }
  if (!KH_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking))) return -1;
  if (pack_noalloc_setup(device)) return -1; // device address of kh_hackrf_pack_noalloc's static scratch
  g_device = device;
  return 0;
}
int initialised_device() {
  std::lock_guard<std::mutex> lk(g_mutex);
  return g_device;
}
int ensure_init() {
  std::lock_guard<std::mutex> lk(g_mutex);
  if (g_device >= 0) {
    hipSetDevice(g_device); // per-thread current device
    return 0;
  }
  const char *env = getenv("KA9Q_HIP_DEVICE");
  return init_locked(env ? atoi(env) : 0);
}
hipStream_t default_stream() { return g_stream; }

// ---------------------------------------------------------------- twiddles
static std::map<int, float2 *> g_tw;
static std::map<int, BigTwiddle> g_bigtw;

static float2 *upload_table(const std::vector<float2> &h) {
  float2 *d = nullptr;
  if (!KH_HIP(hipMalloc(&d, h.size() * sizeof(float2)))) return nullptr;
  if (!KH_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice))) {
    hipFree(d);
    return nullptr;
  }
  return d;
}
const float2 *twiddle_table(int n) {
  std::lock_guard<std::mutex> lk(g_mutex);
  auto it = g_tw.find(n);
  if (it != g_tw.end()) return it->second;
  std::vector<float2> h(n);
  for (int k = 0; k < n; k++) {
    double const a = -2.0 * M_PI * (double)k / (double)n;
    h[k] = make_float2((float)cos(a), (float)sin(a));
  }
  float2 *d = upload_table(h);
  if (d) g_tw[n] = d;
  return d;
}
BigTwiddle big_twiddle(int n) {
  std::lock_guard<std::mutex> lk(g_mutex);
  auto it = g_bigtw.find(n);
  if (it != g_bigtw.end()) return it->second;
  int const nc = n >> 10 > 0 ? n >> 10 : 1;
  std::vector<float2> hc(nc), hf(1024);
  for (int j = 0; j < nc; j++) {
    double const a = -2.0 * M_PI * (double)((long)j << 10) / (double)n;
    hc[j] = make_float2((float)cos(a), (float)sin(a));
  }
  for (int k = 0; k < 1024; k++) {
    double const a = -2.0 * M_PI * (double)k / (double)n;
    hf[k] = make_float2((float)cos(a), (float)sin(a));
  }
  BigTwiddle bt;
  bt.coarse = upload_table(hc);
  bt.fine = upload_table(hf);
  if (bt.coarse && bt.fine) g_bigtw[n] = bt;
  return bt;
}

static std::map<std::array<int, 3>, ColsTwiddle> g_colstw;
ColsTwiddle cols_twiddle(int N2, int RA, int RB) {
  std::lock_guard<std::mutex> lk(g_mutex);
  std::array<int, 3> const key{N2, RA, RB};
  auto it = g_colstw.find(key);
  if (it != g_colstw.end()) return it->second;
  double const n = (double)N2 * RA * RB;
  int const nq = RB / 8;
  auto w = [n](long idx) {
    double const a = -2.0 * M_PI * (double)idx / n;
    return make_float2((float)cos(a), (float)sin(a));
  };
  std::vector<float2> hu((size_t)N2 * 8), ht((size_t)N2 * RA * nq);
  for (long n2 = 0; n2 < N2; n2++) {
    for (int j = 0; j < 8; j++) hu[n2 * 8 + j] = w(n2 * RA * j);
    for (int c = 0; c < RA; c++)
      for (int q = 0; q < nq; q++) ht[(n2 * RA + c) * nq + q] = w(n2 * (c + RA * 8 * q));
  }
  ColsTwiddle ct{upload_table(hu), upload_table(ht)};
  if (ct.uj && ct.tq) g_colstw[key] = ct;
  return ct;
}

static std::map<int, float2 *> g_c2tw;
const float2 *chan2_twiddle(int ND) {
  std::lock_guard<std::mutex> lk(g_mutex);
  auto it = g_c2tw.find(ND);
  if (it != g_c2tw.end()) return it->second;
  int const RB = ND / 32;
  std::vector<float2> h((size_t)32 * RB);
  for (int c = 0; c < 32; c++)
    for (int l = 0; l < RB; l++) {
      double const a = 2.0 * M_PI * (double)(l * c) / (double)ND;
      h[(size_t)c * RB + l] = make_float2((float)cos(a), (float)sin(a));
    }
  float2 *d = upload_table(h);
  if (d) g_c2tw[ND] = d;
  return d;
}

static bool is_pow2(unsigned v) { return v && !(v & (v - 1)); }

AnyPlan make_any_plan(int n) {
  AnyPlan p{};
  if (n < 16 || n > KH_MAX_SMALL_N || (n & 1)) return p; // (even: the REAL layouts and the bin split use n/2)
  int m = n, k = 0;
  for (int r : {4, 2, 3, 5, 7})
    while (m % r == 0 && k < 14) {
      p.radix[k++] = r;
      m /= r;
    }
  if (m != 1) return p;
  p.n = n;
  p.npass = k;
  return p;
}

// ---------------------------------------------------------------- filter.h private state
struct FilterInImpl {
  int N, L, M;
  bool real;
  size_t esize;       // bytes per time sample
  char *d_time[2];    // ping-pong overlap-save buffers, N samples each
  int cur;
  float2 *d_spec[2];  // spectrum of block k lives in d_spec[k & 1]
  float2 *d_tmp;
  hipStream_t stream;
  size_t mapped_bytes; // create_filter_input_file: fdomain is a mapping of this size, not pinned memory
};
struct FilterOutImpl {
  int N, ND, olen;
  float2 *d_resp[2];  // active = d_resp[active]
  int active;
  kh_cf32 *resp_seen; // host response pointer whose contents d_resp[active] holds
  void *d_out;        // olen outputs
  ChanDesc *chan;     // pinned, device visible
  float *d_scalar;
  hipStream_t stream;
};

static int upload_response(struct filter_out *slave, FilterOutImpl *o) {
  // caller holds response_mutex
  bool const rr = slave->master->in_type == REAL && slave->out_type == REAL;
  size_t const count = rr ? o->ND / 2 + 1 : o->ND;
  if (!KH_HIP(hipMemcpyAsync(o->d_resp[o->active], slave->response, count * sizeof(float2), hipMemcpyHostToDevice,
                             o->stream)))
    return -1;
  o->resp_seen = slave->response;
  return 0;
}

static float noise_gain_locked(struct filter_out const *f, FilterOutImpl *o) { // filter.c:631-656
  struct filter_in *master = f->master;
  int const N = master->ilen + master->impulse_length - 1;
  int const ND = N / f->decimate;
  bool const rr = master->in_type == REAL && f->out_type == REAL;
  int const count = rr ? ND / 2 + 1 : ND;
  float const scale = (f->out_type == REAL || f->out_type == CROSS_CONJ) ? 2.f * N : (float)N;
  if (launch_noise_gain(o->stream, o->d_resp[o->active], count, scale, o->d_scalar)) return NAN;
  float v = NAN;
  if (!KH_HIP(hipMemcpyAsync(&v, o->d_scalar, sizeof(float), hipMemcpyDeviceToHost, o->stream))) return NAN;
  if (!KH_HIP(hipStreamSynchronize(o->stream))) return NAN;
  return v;
}

} // namespace kh

using namespace kh;

extern "C" {

int ka9q_hip_init(int device) {
  std::lock_guard<std::mutex> lk(g_mutex);
  if (g_device >= 0) {
    if (device != g_device) {
      set_error("already initialised on device %d", g_device);
      return -1;
    }
    return 0;
  }
  return init_locked(device);
}
const char *ka9q_hip_last_error(void) { return g_err; }

void *ka9q_hip_alloc(size_t bytes) {
  if (ensure_init()) return nullptr;
  void *p = nullptr;
  if (!KH_HIP(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault))) return nullptr;
  return p;
}
void ka9q_hip_free(void *p) {
  if (p) hipHostFree(p);
}

// ------------------------------------------------------------------ filter.c:57
static struct filter_in *create_input_common(unsigned int L, unsigned int M, enum filtertype in_type, const char *file);
struct filter_in *create_filter_input(unsigned int L, unsigned int M, enum filtertype in_type) {
  return create_input_common(L, M, in_type, nullptr);
}
// ------------------------------------------------------------------ filter.c:96
struct filter_in *create_filter_input_file(unsigned int L, unsigned int M, enum filtertype in_type, char *file) {
  if (file == nullptr) {
    set_error("create_filter_input_file: NULL file name");
    return nullptr;
  }
  return create_input_common(L, M, in_type, file);
}
static struct filter_in *create_input_common(unsigned int L, unsigned int M, enum filtertype in_type, const char *file) {
  if (ensure_init()) return nullptr;
  unsigned const N = L + M - 1;
  if (L == 0 || M == 0 || !((is_pow2(N) && N >= 256 && N <= (1u << 20)) || make_any_plan((int)N).npass)) {
    set_error("create_filter_input: N = L+M-1 = %u must be a power of two in [256, 2^20] or have only the factors 2, 3, 5, 7 "
              "and be at most %d", N, KH_MAX_SMALL_N);
    return nullptr;
  }
  if (in_type == REAL && N > KH_MAX_SMALL_N) {
    set_error("create_filter_input: REAL input supports N <= %d", KH_MAX_SMALL_N);
    return nullptr;
  }
  struct filter_in *master = (struct filter_in *)calloc(1, sizeof(*master));
  FilterInImpl *im = new FilterInImpl();
  if (!master) return nullptr;
  pthread_mutex_init(&master->filter_mutex, nullptr);
  pthread_cond_init(&master->filter_cond, nullptr);
  master->blocknum = 0;
  if (in_type != REAL && in_type != COMPLEX) {
    fprintf(stderr, "Filter input type %d, assuming complex\n", in_type); // filter.c:73
    in_type = COMPLEX;
  }
  master->in_type = in_type;
  master->ilen = L;
  master->impulse_length = M;
  im->N = N;
  im->L = L;
  im->M = M;
  im->real = in_type == REAL;
  im->esize = im->real ? sizeof(float) : sizeof(float2);
  im->cur = 0;
  bool ok = KH_HIP(hipStreamCreateWithFlags(&im->stream, hipStreamNonBlocking));
  size_t const nbins = im->real ? N / 2 + 1 : N;
  master->fd = -1;
  im->mapped_bytes = 0;
  if (file == nullptr) {
    master->fdomain = (kh_cf32 *)ka9q_hip_alloc(nbins * sizeof(kh_cf32));
  } else { // filter.c:111-124,133-146: the spectrum lives in a shared mapping of the file
    master->fd = open(file, O_CREAT | O_TRUNC | O_RDWR, 0644);
    void *map = MAP_FAILED;
    if (master->fd >= 0 && ftruncate(master->fd, (off_t)(nbins * sizeof(kh_cf32))) == 0)
      map = mmap(nullptr, nbins * sizeof(kh_cf32), PROT_READ | PROT_WRITE, MAP_SHARED, master->fd, 0);
    if (map == MAP_FAILED) {
      set_error("create_filter_input_file: cannot create and map %s", file);
      ok = false;
    } else {
      master->fdomain = (kh_cf32 *)map;
      im->mapped_bytes = nbins * sizeof(kh_cf32);
    }
  }
  void *ib = ka9q_hip_alloc(N * im->esize);
  ok = ok && master->fdomain && ib;
  if (ib) memset(ib, 0, N * im->esize); // "Clear earlier state", filter.c:79
  if (im->real) {
    master->input_buffer.r = (float *)ib;
    master->input.r = master->input_buffer.r + M - 1;
  } else {
    master->input_buffer.c = (kh_cf32 *)ib;
    master->input.c = master->input_buffer.c + M - 1;
  }
  for (int k = 0; k < 2 && ok; k++) {
    ok = ok && KH_HIP(hipMalloc(&im->d_time[k], N * im->esize));
    ok = ok && KH_HIP(hipMemset(im->d_time[k], 0, N * im->esize));
    ok = ok && KH_HIP(hipMalloc(&im->d_spec[k], N * sizeof(float2)));
  }
  if (ok && N > KH_MAX_SMALL_N) ok = KH_HIP(hipMalloc(&im->d_tmp, N * sizeof(float2)));
  master->fwd_plan = im;
  if (!ok) {
    delete_filter_input(master);
    return nullptr;
  }
  return master;
}

// ------------------------------------------------------------------ filter.c:163
struct filter_out *create_filter_output(struct filter_in *master, kh_cf32 *response, unsigned int decimate,
                                        enum filtertype out_type) {
  if (master == nullptr) return nullptr;
  if (ensure_init()) return nullptr;
  unsigned const N = master->ilen + master->impulse_length - 1;
  if (decimate == 0) {
    set_error("create_filter_output: decimate == 0");
    return nullptr;
  }
  unsigned const ND = N / decimate;
  if ((N % decimate) != 0)
    fprintf(stderr, "Warning: FFT size %'u is not divisible by decimation ratio %d\n", N, decimate); // filter.c:172
  if ((N % decimate) != 0 || !((is_pow2(ND) && ND >= 256 && ND <= 8192) || make_any_plan((int)ND).npass)) {
    set_error("create_filter_output: N/decimate = %u must be a power of two in [256, 8192] or have only the factors 2, 3, 5, 7", ND);
    return nullptr;
  }
  struct filter_out *slave = (struct filter_out *)calloc(1, sizeof(*slave));
  if (!slave) return nullptr;
  FilterOutImpl *o = new FilterOutImpl();
  pthread_mutex_init(&slave->response_mutex, nullptr);
  slave->master = master;
  slave->out_type = out_type;
  slave->decimate = decimate;
  slave->olen = master->ilen / decimate;
  slave->response = response;
  slave->rev_plan = o;
  o->N = N;
  o->ND = ND;
  o->olen = slave->olen;
  o->active = 0;
  o->resp_seen = nullptr;
  bool ok = KH_HIP(hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
  for (int k = 0; k < 2 && ok; k++) {
    ok = ok && KH_HIP(hipMalloc(&o->d_resp[k], ND * sizeof(float2)));
    ok = ok && KH_HIP(hipMemset(o->d_resp[k], 0, ND * sizeof(float2)));
  }
  ok = ok && KH_HIP(hipMalloc(&o->d_out, slave->olen * sizeof(float2) + 16));
  ok = ok && KH_HIP(hipMalloc(&o->d_scalar, sizeof(float)));
  o->chan = (ChanDesc *)ka9q_hip_alloc(sizeof(ChanDesc));
  ok = ok && o->chan;
  // host mirrors, same shapes as filter.c:193-207
  size_t const fbins = out_type == REAL ? ND / 2 + 1 : ND;
  slave->f_fdomain = (kh_cf32 *)ka9q_hip_alloc(fbins * sizeof(kh_cf32)); // kept for layout; not filled
  if (out_type == REAL) {
    slave->output_buffer.r = (float *)ka9q_hip_alloc(ND * sizeof(float));
    if (slave->output_buffer.r) memset(slave->output_buffer.r, 0, ND * sizeof(float));
    slave->output.r = slave->output_buffer.r + ND - slave->olen;
  } else {
    slave->output_buffer.c = (kh_cf32 *)ka9q_hip_alloc(ND * sizeof(kh_cf32));
    if (slave->output_buffer.c) memset(slave->output_buffer.c, 0, ND * sizeof(kh_cf32));
    slave->output.c = slave->output_buffer.c + ND - slave->olen;
  }
  ok = ok && slave->f_fdomain && slave->output_buffer.c;
  if (!ok) {
    slave->response = nullptr; // ownership stays with the caller on failure
    delete_filter_output(slave);
    return nullptr;
  }
  if (response != nullptr) {
    pthread_mutex_lock(&slave->response_mutex);
    upload_response(slave, o);
    slave->noise_gain = noise_gain_locked(slave, o);
    pthread_mutex_unlock(&slave->response_mutex);
  } else
    slave->noise_gain = NAN;
  return slave;
}

// ------------------------------------------------------------------ filter.c:212
int execute_filter_input(struct filter_in *master) {
  if (master == nullptr) return -1;
  FilterInImpl *im = (FilterInImpl *)master->fwd_plan;
  if (im == nullptr || ensure_init()) return -1;
  int const N = im->N, L = im->L, M = im->M;
  char *dt = im->d_time[im->cur];
  const void *hnew = im->real ? (const void *)master->input.r : (const void *)master->input.c;
  if (!KH_HIP(hipMemcpyAsync(dt + (size_t)(M - 1) * im->esize, hnew, (size_t)L * im->esize, hipMemcpyHostToDevice,
                             im->stream)))
    return -1;
  unsigned const next = master->blocknum + 1;
  float2 *spec = im->d_spec[next & 1];
  if (launch_forward_fft(im->stream, dt, 0, im->real, spec, im->d_tmp, 0, N, L, 1, 1, nullptr, N0Fuse{nullptr, 0, 0})) return -1;
  // overlap-save history for the next block (filter.c:230), on the device copy ...
  if (launch_carry(im->stream, dt + (size_t)L * im->esize, im->d_time[im->cur ^ 1], 0, (size_t)(M - 1) * im->esize, 1))
    return -1;
  im->cur ^= 1;
  size_t const nbins = im->real ? N / 2 + 1 : N;
  if (!KH_HIP(hipMemcpyAsync(master->fdomain, spec, nbins * sizeof(float2), hipMemcpyDeviceToHost, im->stream)))
    return -1;
  if (!KH_HIP(hipStreamSynchronize(im->stream))) return -1;
  // ... and on the caller-visible buffer
  if (im->real)
    memmove(master->input_buffer.r, master->input_buffer.r + L, (M - 1) * sizeof(float));
  else
    memmove(master->input_buffer.c, master->input_buffer.c + L, (M - 1) * sizeof(kh_cf32));
  pthread_mutex_lock(&master->filter_mutex); // filter.c:220-223
  master->blocknum++;
  pthread_cond_broadcast(&master->filter_cond);
  pthread_mutex_unlock(&master->filter_mutex);
  return 0;
}

// ------------------------------------------------------------------ filter.c:240
int execute_filter_output(struct filter_out *slave, int rotate) {
  if (slave == nullptr) return -1;
  struct filter_in *master = slave->master;
  FilterOutImpl *o = (FilterOutImpl *)slave->rev_plan;
  if (master == nullptr || o == nullptr || ensure_init()) return -1;
  FilterInImpl *im = (FilterInImpl *)master->fwd_plan;

  pthread_mutex_lock(&master->filter_mutex); // filter.c:260-264
  while (slave->blocknum == master->blocknum) pthread_cond_wait(&master->filter_cond, &master->filter_mutex);
  slave->blocknum = master->blocknum;
  unsigned const blk = slave->blocknum;
  pthread_mutex_unlock(&master->filter_mutex);

  if (rotate <= -(int)o->N || rotate >= (int)o->N) {
    set_error("execute_filter_output: |rotate| must be < N");
    return -1;
  }
  int rc = 0;
  pthread_mutex_lock(&slave->response_mutex); // filter.c:266
  if (slave->response == nullptr) {
    pthread_mutex_unlock(&slave->response_mutex);
    set_error("execute_filter_output: response not set");
    return -1;
  }
  if (slave->response != o->resp_seen) rc = upload_response(slave, o);
  int const out_type = slave->out_type == NONE ? COMPLEX : slave->out_type;
  o->chan->input = 0;
  o->chan->rotate = rotate;
  o->chan->out_type = out_type;
  o->chan->fm_stats = 0;
  if (!rc)
    rc = launch_filter_output(o->stream, im->d_spec[blk & 1], 0, o->N, master->in_type == REAL, o->chan,
                              o->d_resp[o->active], o->ND, o->olen, o->d_out, 0, 1, 1, FmFront{}, out_type != REAL);
  if (!rc) {
    if (out_type == REAL)
      rc = KH_HIP(hipMemcpyAsync(slave->output.r, o->d_out, o->olen * sizeof(float), hipMemcpyDeviceToHost, o->stream))
               ? 0
               : -1;
    else
      rc = KH_HIP(hipMemcpyAsync(slave->output.c, o->d_out, o->olen * sizeof(float2), hipMemcpyDeviceToHost,
                                 o->stream))
               ? 0
               : -1;
  }
  if (!rc) rc = KH_HIP(hipStreamSynchronize(o->stream)) ? 0 : -1;
  pthread_mutex_unlock(&slave->response_mutex); // filter.c:369
  return rc;
}

int delete_filter_input(struct filter_in *master) { // filter.c:386
  if (master == nullptr) return 0;
  FilterInImpl *im = (FilterInImpl *)master->fwd_plan;
  if (im && im->mapped_bytes) { // create_filter_input_file
    munmap(master->fdomain, im->mapped_bytes);
    master->fdomain = nullptr;
  }
  if (master->fd >= 0) close(master->fd);
  if (im) {
    if (im->stream) {
      hipStreamSynchronize(im->stream);
      hipStreamDestroy(im->stream);
    }
    for (int k = 0; k < 2; k++) {
      if (im->d_time[k]) hipFree(im->d_time[k]);
      if (im->d_spec[k]) hipFree(im->d_spec[k]);
    }
    if (im->d_tmp) hipFree(im->d_tmp);
    delete im;
  }
  ka9q_hip_free(master->input_buffer.c);
  ka9q_hip_free(master->fdomain);
  free(master);
  return 0;
}
int delete_filter_output(struct filter_out *slave) { // filter.c:396
  if (slave == nullptr) return 0;
  FilterOutImpl *o = (FilterOutImpl *)slave->rev_plan;
  if (o) {
    if (o->stream) {
      hipStreamSynchronize(o->stream);
      hipStreamDestroy(o->stream);
    }
    for (int k = 0; k < 2; k++)
      if (o->d_resp[k]) hipFree(o->d_resp[k]);
    if (o->d_out) hipFree(o->d_out);
    if (o->d_scalar) hipFree(o->d_scalar);
    ka9q_hip_free(o->chan);
    delete o;
  }
  pthread_mutex_destroy(&slave->response_mutex);
  ka9q_hip_free(slave->output_buffer.c);
  ka9q_hip_free(slave->response);
  ka9q_hip_free(slave->f_fdomain);
  free(slave);
  return 0;
}

// ------------------------------------------------------------------ filter.c:481
int make_kaiser(float *window, unsigned int M, float beta) {
  if (window == nullptr) return -1;
  if (ensure_init()) return -1;
  if (M == 0) return 0;
  float *d = nullptr;
  if (!KH_HIP(hipMalloc(&d, M * sizeof(float)))) return -1;
  int rc = launch_make_kaiser(default_stream(), d, (int)M, beta);
  if (!rc) rc = KH_HIP(hipMemcpyAsync(window, d, M * sizeof(float), hipMemcpyDeviceToHost, default_stream())) ? 0 : -1;
  if (!rc) rc = KH_HIP(hipStreamSynchronize(default_stream())) ? 0 : -1;
  hipFree(d);
  return rc;
}

static int window_common(int L, int M, kh_cf32 *response, float beta, int real_filter) {
  if (response == nullptr) return -1;
  if (ensure_init()) return -1;
  int const N = L + M - 1;
  if (!((N >= 256 && N <= 8192 && is_pow2((unsigned)N)) || make_any_plan(N).npass)) {
    set_error("window_filter: N = L+M-1 = %d must be a power of two in [256, 8192] or have only the factors 2, 3, 5, 7", N);
    return -1;
  }
  size_t const count = real_filter ? N / 2 + 1 : N;
  float2 *d = nullptr;
  if (!KH_HIP(hipMalloc(&d, N * sizeof(float2)))) return -1;
  hipStream_t st = default_stream();
  int rc = KH_HIP(hipMemcpyAsync(d, response, count * sizeof(float2), hipMemcpyHostToDevice, st)) ? 0 : -1;
  if (!rc) rc = launch_window_filter(st, d, L, M, beta, real_filter);
  if (!rc) rc = KH_HIP(hipMemcpyAsync(response, d, count * sizeof(float2), hipMemcpyDeviceToHost, st)) ? 0 : -1;
  if (!rc) rc = KH_HIP(hipStreamSynchronize(st)) ? 0 : -1;
  hipFree(d);
  return rc;
}
int window_filter(int L, int M, kh_cf32 *response, float beta) { return window_common(L, M, response, beta, 0); }   // filter.c:509
int window_rfilter(int L, int M, kh_cf32 *response, float beta) { return window_common(L, M, response, beta, 1); }  // filter.c:573

float noise_gain(struct filter_out const *filter) { // filter.c:631
  if (filter == nullptr) return NAN;
  FilterOutImpl *o = (FilterOutImpl *)filter->rev_plan;
  if (o == nullptr || filter->response == nullptr || ensure_init()) return NAN;
  struct filter_out *f = const_cast<struct filter_out *>(filter);
  pthread_mutex_lock(&f->response_mutex);
  float v = NAN;
  if (filter->response == o->resp_seen || upload_response(f, o) == 0) v = noise_gain_locked(filter, o);
  pthread_mutex_unlock(&f->response_mutex);
  return v;
}

// ------------------------------------------------------------------ filter.c:660
int set_filter(struct filter_out *slave, float low, float high, float kaiser_beta) {
  if (slave == nullptr || std::isnan(low) || std::isnan(high) || std::isnan(kaiser_beta)) return -1;
  FilterOutImpl *o = (FilterOutImpl *)slave->rev_plan;
  if (o == nullptr || ensure_init()) return -1;
  if (fabsf(low) > 0.5f || fabsf(high) > 0.5f) { // the reference asserts (filter.c:664-665)
    set_error("set_filter: |low|, |high| must be <= 0.5");
    return -1;
  }
  struct filter_in *master = slave->master;
  int const L_dec = slave->olen;
  int const M_dec = (master->impulse_length - 1) / slave->decimate + 1;
  int const N_dec = L_dec + M_dec - 1;
  int const N = master->ilen + master->impulse_length - 1;
  if (N_dec != o->ND) {
    set_error("set_filter: L/D + (M-1)/D = %d differs from N/D = %d", N_dec, o->ND);
    return -1;
  }
  float gain = 1.f / ((float)N);
  if (slave->out_type == REAL || slave->out_type == CROSS_CONJ) gain *= (float)M_SQRT1_2;
  kh_cf32 *response = (kh_cf32 *)ka9q_hip_alloc(N_dec * sizeof(kh_cf32));
  if (!response) return -1;
  // design into the inactive device buffer on the control stream, then swap under the mutex (filter.c:694-698)
  hipStream_t st = default_stream();
  float2 *d = o->d_resp[o->active ^ 1];
  int rc = launch_brickwall(st, d, N_dec, low, high, gain);
  if (!rc) rc = launch_window_filter(st, d, L_dec, M_dec, kaiser_beta, 0);
  if (!rc) rc = KH_HIP(hipMemcpyAsync(response, d, N_dec * sizeof(float2), hipMemcpyDeviceToHost, st)) ? 0 : -1;
  if (!rc) rc = KH_HIP(hipStreamSynchronize(st)) ? 0 : -1;
  if (rc) {
    ka9q_hip_free(response);
    return -1;
  }
  pthread_mutex_lock(&slave->response_mutex);
  kh_cf32 *tmp = slave->response;
  slave->response = response;
  o->active ^= 1;
  o->resp_seen = response;
  slave->noise_gain = noise_gain_locked(slave, o);
  pthread_mutex_unlock(&slave->response_mutex);
  ka9q_hip_free(tmp);
  return 0;
}

// ------------------------------------------------------------------ filter.c:707-727
struct notchfilter *notch_create(double f, float bw) {
  struct notchfilter *nf = (struct notchfilter *)calloc(1, sizeof(struct notchfilter));
  if (nf == nullptr) return nullptr;
  nf->osc_phase.re = 1; // filter.c:712
  nf->osc_phase.im = 0;
  double sn, cs;
  sincos(2 * M_PI * f, &sn, &cs); // cispi(2 f), dsp.h:37-50: a constant of the control path
  nf->osc_step.re = cs;
  nf->osc_step.im = sn;
  nf->bw = bw;
  return nf;
}
int kh_notch_block(struct notchfilter *nf, kh_cf32 *buf, int n) {
  if (nf == nullptr || (buf == nullptr && n > 0) || n < 0) {
    set_error("kh_notch_block: bad argument");
    return -1;
  }
  if (ensure_init()) return -1;
  if (n == 0) return 0;
  hipStream_t st = default_stream();
  hipPointerAttribute_t attr;
  bool on_dev = hipPointerGetAttributes(&attr, buf) == hipSuccess && attr.type == hipMemoryTypeDevice;
  (void)hipGetLastError();
  char *scratch = nullptr; // state, then the samples when they come from the host
  size_t const bytes = (size_t)n * sizeof(float2);
  if (!KH_HIP(hipMalloc(&scratch, 256 + (on_dev ? 0 : bytes)))) return -1;
  NotchState hs{nf->osc_phase.re, nf->osc_phase.im, nf->osc_step.re, nf->osc_step.im, nf->dcstate.re, nf->dcstate.im, nf->bw, 0};
  float2 *d = on_dev ? reinterpret_cast<float2 *>(buf) : reinterpret_cast<float2 *>(scratch + 256);
  bool ok = KH_HIP(hipMemcpyAsync(scratch, &hs, sizeof hs, hipMemcpyHostToDevice, st));
  if (ok && !on_dev) ok = KH_HIP(hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, st));
  ok = ok && launch_notch(st, reinterpret_cast<NotchState *>(scratch), d, n) == 0;
  ok = ok && KH_HIP(hipMemcpyAsync(&hs, scratch, sizeof hs, hipMemcpyDeviceToHost, st));
  if (ok && !on_dev) ok = KH_HIP(hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, st));
  ok = ok && KH_HIP(hipStreamSynchronize(st));
  hipFree(scratch);
  if (!ok) return -1;
  nf->osc_phase.re = hs.pr;
  nf->osc_phase.im = hs.pi;
  nf->dcstate.re = hs.dr;
  nf->dcstate.im = hs.di;
  return 0;
}
kh_cf32 notch(struct notchfilter *nf, kh_cf32 s) {
  if (nf == nullptr || kh_notch_block(nf, &s, 1)) s.re = s.im = NAN; // filter.c:720-721
  return s;
}

// ------------------------------------------------------------------ decimate.h
// buf/outbuf/state pointers may be host or device memory.  The device-side working memory is kept between calls and
// only grows (hackrf.c calls this once per USB transfer: no allocation on that path after the first transfer); calls
// are serialised on it.
struct DecScratch {
  float *in = nullptr, *out = nullptr; // rails x len staged input (host callers, deep cascades), rails x chunk output
  size_t in_cap = 0, out_cap = 0;      // floats
  char *s15[2] = {nullptr, nullptr};   // hb15 states before / after the call
  float *s3[2] = {nullptr, nullptr};
  int nstate_cap = 0;
};
static DecScratch g_dec;
static std::mutex g_dec_mutex;

static bool dec_reserve(float **p, size_t *cap, size_t need) {
  if (need <= *cap) return true;
  if (*p) (void)hipFree(*p);
  *p = nullptr, *cap = 0;
  if (!KH_HIP(hipMalloc((void **)p, need * sizeof(float)))) return false;
  *cap = need;
  return true;
}
static bool is_device_pointer(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError(); // plain host memory the runtime has never seen
    return false;
  }
  return a.type == hipMemoryTypeDevice;
}

static int decimate_host(struct hb15_state *hb15, float *hb3, int log_decimate, int stage_threshold, float *buf,
                         size_t rail_stride, int rails, int chunk, float *outbuf) {
  if (ensure_init()) return -1;
  if (rails <= 0 || chunk <= 0 || log_decimate <= 0) return 0;
  std::lock_guard<std::mutex> lk(g_dec_mutex);
  hipStream_t st = default_stream();
  size_t const len = (size_t)chunk << log_decimate;
  int const nstate = rails * log_decimate;
  DecScratch &D = g_dec;
  if (nstate > D.nstate_cap) {
    for (int k = 0; k < 2; k++) {
      if (D.s15[k]) (void)hipFree(D.s15[k]);
      if (D.s3[k]) (void)hipFree(D.s3[k]);
      D.s15[k] = nullptr, D.s3[k] = nullptr;
    }
    D.nstate_cap = 0;
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++)
      ok = KH_HIP(hipMalloc((void **)&D.s15[k], nstate * sizeof(struct hb15_state))) && KH_HIP(hipMalloc((void **)&D.s3[k], nstate * sizeof(float)));
    if (!ok) return -1;
    D.nstate_cap = nstate;
  }
  const char *const e = getenv("KA9Q_HIP_DECIMATE_FUSED"); // 0: one launch pair per stage (the round-1 route; tests compare the two)
  bool const fuse = !(e && atoi(e) == 0) && decimate_can_fuse(log_decimate, stage_threshold);
  bool const direct = fuse && is_device_pointer(buf); // the fused kernel reads the caller's rails where they lie
  // staged cascade: input and ping-pong half in one buffer; fused from host memory: the input only
  if (!direct && !dec_reserve(&D.in, &D.in_cap, (fuse ? 1 : 2) * (size_t)rails * len)) return -1;
  if (fuse && !dec_reserve(&D.out, &D.out_cap, (size_t)rails * chunk)) return -1;
  bool ok = true;
  for (int r = 0; r < rails && ok && !direct; r++)
    ok = KH_HIP(hipMemcpyAsync(D.in + r * len, buf + r * rail_stride, len * sizeof(float), hipMemcpyDefault, st));
  // a state the caller does not pass still has to be defined where a stage would read it: zero
  if (ok) ok = hb15 ? KH_HIP(hipMemcpyAsync(D.s15[0], hb15, nstate * sizeof(struct hb15_state), hipMemcpyDefault, st))
                    : KH_HIP(hipMemsetAsync(D.s15[0], 0, nstate * sizeof(struct hb15_state), st));
  if (ok) ok = hb3 ? KH_HIP(hipMemcpyAsync(D.s3[0], hb3, nstate * sizeof(float), hipMemcpyDefault, st))
                   : KH_HIP(hipMemsetAsync(D.s3[0], 0, nstate * sizeof(float), st));
  int rc = ok ? 0 : -1;
  const float *result = nullptr;
  size_t result_stride = 0;
  int after = 0; // which state buffer holds the states after the call
  if (!rc && fuse) {
    rc = launch_decimate_fused(st, direct ? buf : D.in, direct ? rail_stride : len, D.out, chunk, rails, chunk, log_decimate,
                               stage_threshold, D.s15[0], D.s15[1], D.s3[0], D.s3[1]);
    result = D.out, result_stride = chunk, after = 1;
  } else if (!rc) {
    rc = launch_decimate(st, D.in, len, rails, chunk, log_decimate, stage_threshold, D.s15[0], D.s3[0]);
    result = D.in, result_stride = len;
  }
  for (int r = 0; r < rails && !rc; r++)
    rc = KH_HIP(hipMemcpyAsync((outbuf ? outbuf : buf) + r * rail_stride, result + r * result_stride, chunk * sizeof(float),
                               hipMemcpyDefault, st))
             ? 0
             : -1;
  if (!rc && hb15) rc = KH_HIP(hipMemcpyAsync(hb15, D.s15[after], nstate * sizeof(struct hb15_state), hipMemcpyDefault, st)) ? 0 : -1;
  if (!rc && hb3) rc = KH_HIP(hipMemcpyAsync(hb3, D.s3[after], nstate * sizeof(float), hipMemcpyDefault, st)) ? 0 : -1;
  if (!rc) rc = KH_HIP(hipStreamSynchronize(st)) ? 0 : -1;
  return rc;
}
int kh_decimate_rails(struct hb15_state *hb15, float *hb3, int log_decimate, int stage_threshold, float *buf,
                      size_t rail_stride, int rails, int chunk) {
  return decimate_host(hb15, hb3, log_decimate, stage_threshold, buf, rail_stride, rails, chunk, nullptr);
}
void hb15_block(struct hb15_state *state, float *output, float *input, int cnt) { // decimate.c:113
  if (state == nullptr || cnt <= 0) return;
  // one hb15 stage: log_decimate 1, threshold above it
  if (decimate_host(state, nullptr, 1, 1, input, 0, 1, cnt, output))
    fprintf(stderr, "hb15_block: %s\n", ka9q_hip_last_error());
}
void hb3_block(float *state, float *output, float *input, int cnt) { // decimate.c:150
  if (state == nullptr || cnt <= 0) return;
  if (decimate_host(nullptr, state, 1, 0, input, 0, 1, cnt, output))
    fprintf(stderr, "hb3_block: %s\n", ka9q_hip_last_error());
}

} // extern "C"
