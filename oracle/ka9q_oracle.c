/* ka9q_oracle.c -- CPU restatement of ka9q-radio's receive DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ka9q_oracle.h).  Plain C11, single threaded, no GPU.
 * Every function cites the reference file:line (under /root/reference) whose behaviour it restates.
 * Float arithmetic is written in the reference's evaluation order; the reference itself is built with
 * -funsafe-math-optimizations (Makefile:2) so bitwise float equality with it is not defined.
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <complex.h>
#include "ka9q_oracle.h"

#ifndef M_SQRT1_2
#define M_SQRT1_2 0.70710678118654752440
#endif
#define ORC_M_1_2PI (0.5 * M_1_PI) /* dsp.h:19 */

static inline float cnrmf_(float complex x) { /* dsp.h:65 */
  return crealf(x) * crealf(x) + cimagf(x) * cimagf(x);
}
static inline double cnrm_(double complex x) { /* dsp.h:68 */
  return creal(x) * creal(x) + cimag(x) * cimag(x);
}
static inline double complex cispi_(double x) { /* dsp.h:27,57-61: sincos(x*M_PI) */
  double s, c;
  sincos(x * M_PI, &s, &c);
  return CMPLX(c, s);
}

/* =========================================================== FFT */
/* Unnormalised DFT, sign -1 forward / +1 backward, as FFTW3's fftwf_plan_dft_1d computes
 * (call sites filter.c:81,198,517-518).  Stockham autosort, mixed radix 4/2/3/5/7/generic. */
struct orc_fft_plan {
  int n;
  int nfac;
  int fac[40];
  float complex *tw; /* tw[k] = exp(-2 pi i k / n), computed in double */
  float complex *work;
};

orc_fft_plan *orc_fft_plan_create(int n) {
  if (n < 1) return NULL;
  orc_fft_plan *p = calloc(1, sizeof(*p));
  p->n = n;
  int m = n;
  while (m % 4 == 0) { p->fac[p->nfac++] = 4; m /= 4; }
  while (m % 2 == 0) { p->fac[p->nfac++] = 2; m /= 2; }
  for (int f = 3; m > 1;) {
    if (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
    else f += 2;
  }
  p->tw = malloc(sizeof(float complex) * (size_t)n);
  p->work = malloc(sizeof(float complex) * (size_t)n);
  for (int k = 0; k < n; k++) {
    double s, c;
    sincos(-2.0 * M_PI * (double)k / (double)n, &s, &c);
    p->tw[k] = CMPLXF((float)c, (float)s);
  }
  return p;
}
void orc_fft_plan_destroy(orc_fft_plan *p) {
  if (!p) return;
  free(p->tw);
  free(p->work);
  free(p);
}

static inline float complex twf(const orc_fft_plan *p, long idx, int sign) {
  float complex w = p->tw[idx % p->n];
  return sign < 0 ? w : conjf(w);
}

/* one Stockham pass: length-n sub-transforms with stride s, radix r; x -> y */
static void pass(const orc_fft_plan *pl, int n, int s, int r, const float complex *x, float complex *y, int sign) {
  int const m = n / r;
  long const N = pl->n;
  long const tstep = N / n; /* W_n^k = tw[k*tstep] */
  if (r == 2) {
    for (int p = 0; p < m; p++) {
      float complex const w1 = twf(pl, (long)p * tstep, sign);
      for (int q = 0; q < s; q++) {
        float complex const a = x[q + s * (p + 0)];
        float complex const b = x[q + s * (p + m)];
        y[q + s * (2 * p + 0)] = a + b;
        y[q + s * (2 * p + 1)] = (a - b) * w1;
      }
    }
  } else if (r == 4) {
    for (int p = 0; p < m; p++) {
      float complex const w1 = twf(pl, (long)p * tstep, sign);
      float complex const w2 = twf(pl, 2L * p * tstep, sign);
      float complex const w3 = twf(pl, 3L * p * tstep, sign);
      for (int q = 0; q < s; q++) {
        float complex const a = x[q + s * (p + 0)];
        float complex const b = x[q + s * (p + m)];
        float complex const c = x[q + s * (p + 2 * m)];
        float complex const d = x[q + s * (p + 3 * m)];
        float complex const apc = a + c, amc = a - c, bpd = b + d;
        float complex const bmd = b - d;
        /* multiply by -i (forward) or +i (backward) */
        float complex const jbmd = sign < 0 ? CMPLXF(cimagf(bmd), -crealf(bmd)) : CMPLXF(-cimagf(bmd), crealf(bmd));
        y[q + s * (4 * p + 0)] = apc + bpd;
        y[q + s * (4 * p + 1)] = (amc + jbmd) * w1;
        y[q + s * (4 * p + 2)] = (apc - bpd) * w2;
        y[q + s * (4 * p + 3)] = (amc - jbmd) * w3;
      }
    }
  } else {
    float complex a[r];
    long const rstep = N / r; /* W_r^k = tw[k*rstep] */
    for (int p = 0; p < m; p++) {
      for (int q = 0; q < s; q++) {
        for (int k = 0; k < r; k++) a[k] = x[q + s * (p + k * m)];
        for (int j = 0; j < r; j++) {
          float complex acc = a[0];
          for (int k = 1; k < r; k++) acc += a[k] * twf(pl, (long)((j * k) % r) * rstep, sign);
          y[q + s * (r * p + j)] = acc * twf(pl, (long)p * j * tstep, sign);
        }
      }
    }
  }
}

void orc_fft_exec(orc_fft_plan *p, const float complex *in, float complex *out, int sign) {
  int const n = p->n;
  if (n == 1) { out[0] = in[0]; return; }
  /* ping-pong between out and work so that the final pass lands in out */
  float complex *bufs[2] = {out, p->work};
  int cur; /* index of buffer receiving the next pass */
  const float complex *src = in;
  /* if in == out the first pass must not write to out */
  cur = (p->nfac % 2 == 1) ? 0 : 1;
  float complex *tmp_in = NULL;
  if (in == out && cur == 0) {
    tmp_in = malloc(sizeof(float complex) * (size_t)n);
    memcpy(tmp_in, in, sizeof(float complex) * (size_t)n);
    src = tmp_in;
  }
  int len = n, s = 1;
  for (int i = 0; i < p->nfac; i++) {
    int const r = p->fac[i];
    pass(p, len, s, r, src, bufs[cur], sign);
    src = bufs[cur];
    cur ^= 1;
    len /= r;
    s *= r;
  }
  free(tmp_in);
}

void orc_fft(int n, const float complex *in, float complex *out, int sign) {
  orc_fft_plan *p = orc_fft_plan_create(n);
  orc_fft_exec(p, in, out, sign);
  orc_fft_plan_destroy(p);
}

/* fftwf_plan_dft_r2c_1d semantics (filter.c:90): n reals -> n/2+1 bins of the forward DFT */
void orc_fft_r2c(int n, const float *in, float complex *out) {
  float complex *t = malloc(sizeof(float complex) * (size_t)n);
  float complex *u = malloc(sizeof(float complex) * (size_t)n);
  for (int i = 0; i < n; i++) t[i] = in[i];
  orc_fft(n, t, u, -1);
  memcpy(out, u, sizeof(float complex) * (size_t)(n / 2 + 1));
  free(t);
  free(u);
}
/* fftwf_plan_dft_c2r_1d semantics (filter.c:207): n/2+1 bins -> n reals, imaginary parts of DC/Nyquist ignored */
void orc_fft_c2r(int n, const float complex *in, float *out) {
  float complex *t = malloc(sizeof(float complex) * (size_t)n);
  float complex *u = malloc(sizeof(float complex) * (size_t)n);
  t[0] = crealf(in[0]);
  for (int k = 1; k <= n / 2; k++) {
    if (2 * k == n) {
      t[k] = crealf(in[k]);
    } else {
      t[k] = in[k];
      t[n - k] = conjf(in[k]);
    }
  }
  orc_fft(n, t, u, +1);
  for (int i = 0; i < n; i++) out[i] = crealf(u[i]);
  free(t);
  free(u);
}

/* =========================================================== filter.c */
struct orc_filter_in *orc_create_filter_input(unsigned L, unsigned M, int in_type) { /* filter.c:57-94 */
  int const N = L + M - 1;
  struct orc_filter_in *m = calloc(1, sizeof(*m));
  m->in_type = in_type == ORC_REAL ? ORC_REAL : ORC_COMPLEX; /* default: "assuming complex", filter.c:72-74 */
  m->ilen = L;
  m->impulse_length = M;
  m->plan = orc_fft_plan_create(N);
  if (m->in_type == ORC_REAL) {
    m->fdomain = calloc(N / 2 + 1, sizeof(float complex));
    m->input_buffer_r = calloc(N, sizeof(float));
    m->input_r = m->input_buffer_r + M - 1;
  } else {
    m->fdomain = calloc(N, sizeof(float complex));
    m->input_buffer_c = calloc(N, sizeof(float complex));
    m->input_c = m->input_buffer_c + M - 1;
  }
  return m;
}

struct orc_filter_out *orc_create_filter_output(struct orc_filter_in *master, float complex *response,
                                                unsigned decimate, int out_type) { /* filter.c:163-211 */
  if (master == NULL) return NULL;
  int const N = master->ilen + master->impulse_length - 1;
  int const N_dec = N / decimate;
  struct orc_filter_out *s = calloc(1, sizeof(*s));
  s->master = master;
  s->out_type = out_type;
  s->decimate = decimate;
  s->olen = master->ilen / decimate;
  s->response = response;
  s->noise_gain = response ? orc_noise_gain(s) : NAN;
  s->plan = orc_fft_plan_create(N_dec);
  if (out_type == ORC_REAL) {
    s->f_fdomain = calloc(N_dec / 2 + 1, sizeof(float complex));
    s->output_buffer_r = calloc(N_dec, sizeof(float));
    s->output_r = s->output_buffer_r + N_dec - s->olen;
  } else {
    s->f_fdomain = calloc(N_dec, sizeof(float complex));
    s->output_buffer_c = calloc(N_dec, sizeof(float complex));
    s->output_c = s->output_buffer_c + N_dec - s->olen;
  }
  return s;
}

int orc_execute_filter_input(struct orc_filter_in *m) { /* filter.c:212-238 */
  if (m == NULL) return -1;
  int const N = m->ilen + m->impulse_length - 1;
  if (m->in_type == ORC_REAL) {
    orc_fft_r2c(N, m->input_buffer_r, m->fdomain);
    memmove(m->input_buffer_r, m->input_buffer_r + m->ilen, (m->impulse_length - 1) * sizeof(float));
  } else {
    orc_fft_exec(m->plan, m->input_buffer_c, m->fdomain, -1);
    memmove(m->input_buffer_c, m->input_buffer_c + m->ilen, (m->impulse_length - 1) * sizeof(float complex));
  }
  m->blocknum++;
  return 0;
}

int orc_execute_filter_output(struct orc_filter_out *slave, int rotate) { /* filter.c:240-384 */
  if (slave == NULL) return -1;
  struct orc_filter_in *master = slave->master;
  int const N = master->ilen + master->impulse_length - 1;
  int const N_dec = N / slave->decimate;
  float complex *f = slave->f_fdomain;
  float complex const *resp = slave->response;
  float complex const *fd = master->fdomain;

  if (master->in_type != ORC_REAL && slave->out_type != ORC_REAL) { /* 274-289 */
    int p = N_dec / 2 + 1;
    int m = N - N_dec / 2 + 1 + rotate;
    if (m >= N) m -= N;
    if (m < 0) m += N;
    for (int i = 0; i < N_dec; i++) {
      f[p] = resp[p] * fd[m];
      if (++p == N_dec) p = 0;
      if (++m == N) m = 0;
    }
  } else if (master->in_type == ORC_REAL && slave->out_type == ORC_REAL) { /* 290-299 */
    int m = rotate;
    if (m < 0) m += N;
    for (int p = 0; p <= N_dec / 2; p++) {
      f[p] = resp[p] * fd[m];
      if (++m == N) m = 0;
    }
  } else if (master->in_type != ORC_REAL && slave->out_type == ORC_REAL) { /* 300-317 */
    int m = rotate;
    if (m < 0) m += N;
    f[0] = resp[0] * fd[m];
    if (++m == N) m = 0;
    for (int p = 1; p < N_dec / 2; p++) {
      f[p] = resp[p] * fd[m] + conjf(resp[N_dec - p] * fd[m]);
      if (++m == N) m = 0;
    }
    f[N_dec / 2] = resp[N_dec / 2] * fd[m];
  } else { /* real -> complex, 318-335 */
    int m = rotate;
    if (m < 0) m += N;
    f[0] = resp[0] * fd[m];
    if (++m == N) m = 0;
    for (int p = 1;; p++) {
      f[p] = resp[p] * fd[m];
      if (p == N_dec / 2) break;
      f[N_dec - p] = resp[N_dec - p] * conjf(fd[m]);
      if (++m == N) m = 0;
    }
  }
  if (slave->out_type == ORC_CROSS_CONJ) { /* 371-381 */
    for (int p = 1, dn = N_dec - 1; p < N_dec / 2; p++, dn--) {
      float complex const pos = f[p];
      float complex const neg = f[dn];
      f[p] = pos + conjf(neg);
      f[dn] = neg - conjf(pos);
    }
  }
  if (slave->out_type == ORC_REAL)
    orc_fft_c2r(N_dec, f, slave->output_buffer_r);
  else
    orc_fft_exec(slave->plan, f, slave->output_buffer_c, +1);
  return 0;
}

int orc_delete_filter_input(struct orc_filter_in *m) { /* filter.c:386-395 */
  if (m == NULL) return 0;
  orc_fft_plan_destroy(m->plan);
  free(m->input_buffer_c);
  free(m->input_buffer_r);
  free(m->fdomain);
  free(m);
  return 0;
}
int orc_delete_filter_output(struct orc_filter_out *s) { /* filter.c:396-407 */
  if (s == NULL) return 0;
  orc_fft_plan_destroy(s->plan);
  free(s->output_buffer_c);
  free(s->output_buffer_r);
  free(s->response);
  free(s->f_fdomain);
  free(s);
  return 0;
}

float orc_i0f(float z) { /* filter.c:410-421 */
  float const t = (z * z) / 4;
  float sum = 1 + t;
  float term = t;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * k);
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return sum;
}
float orc_i1f(float z) { /* filter.c:424-436 */
  float const t = (z * z) / 4;
  float term = 1;
  float sum = term;
  for (int k = 1; k < 40; k++) {
    term *= t / (k * (k + 1));
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return 0.5 * z * sum;
}

int orc_make_kaiser(float *window, unsigned M, float beta) { /* filter.c:481-501 */
  if (window == NULL) return -1;
  float const numc = M_PI * beta;
  float const inv_denom = 1. / orc_i0f(numc);
  float const pc = 2.0 / (M - 1);
  for (int n = 0; n < (int)(M / 2); n++) {
    float const p = pc * n - 1;
    window[M - 1 - n] = window[n] = orc_i0f(numc * sqrtf(1 - p * p)) * inv_denom;
  }
  if (M & 1) window[(M - 1) / 2] = 1;
  return 0;
}

int orc_window_filter(int L, int M, float complex *response, float beta) { /* filter.c:509-568 */
  if (response == NULL) return -1;
  int const N = L + M - 1;
  float complex *buffer = malloc(sizeof(float complex) * (size_t)N);
  float complex *tmp = malloc(sizeof(float complex) * (size_t)N);
  orc_fft(N, response, buffer, +1); /* to time domain, 521-522 */
  float *kaiser_window = malloc(sizeof(float) * (size_t)M);
  orc_make_kaiser(kaiser_window, M, beta);
  float const gain = 1. / N;
  for (int n = M - 1; n >= 0; n--) /* 543-544: in place, n descending */
    buffer[n] = buffer[(n - M / 2 + N) % N] * kaiser_window[n] * gain;
  memset(buffer + M, 0, (N - M) * sizeof(*buffer)); /* 546 */
  orc_fft(N, buffer, tmp, -1);                      /* 555 */
  memcpy(response, tmp, N * sizeof(*response));
  free(kaiser_window);
  free(buffer);
  free(tmp);
  return 0;
}

int orc_window_rfilter(int L, int M, float complex *response, float beta) { /* filter.c:573-628 */
  if (response == NULL) return -1;
  int const N = L + M - 1;
  float *timebuf = malloc(sizeof(float) * (size_t)N);
  orc_fft_c2r(N, response, timebuf); /* 589-590 */
  float *kaiser_window = malloc(sizeof(float) * (size_t)M);
  orc_make_kaiser(kaiser_window, M, beta);
  float const gain = 1. / N;
  for (int n = M - 1; n >= 0; n--) /* 604-605 */
    timebuf[n] = timebuf[(n - M / 2 + N) % N] * kaiser_window[n] * gain;
  memset(timebuf + M, 0, (N - M) * sizeof(*timebuf));
  orc_fft_r2c(N, timebuf, response); /* 616, 625 */
  free(kaiser_window);
  free(timebuf);
  return 0;
}

float orc_noise_gain(struct orc_filter_out const *filter) { /* filter.c:631-656 */
  if (filter == NULL) return NAN;
  struct orc_filter_in *master = filter->master;
  int const N = master->ilen + master->impulse_length - 1;
  int const N_dec = N / filter->decimate;
  float sum = 0;
  if (master->in_type == ORC_REAL && filter->out_type == ORC_REAL) {
    for (int i = 0; i < N_dec / 2 + 1; i++) sum += cnrmf_(filter->response[i]);
  } else {
    for (int i = 0; i < N_dec; i++) sum += cnrmf_(filter->response[i]);
  }
  if (filter->out_type == ORC_REAL || filter->out_type == ORC_CROSS_CONJ)
    return 2 * N * sum;
  else
    return N * sum;
}

int orc_set_filter(struct orc_filter_out *slave, float low, float high, float kaiser_beta) { /* filter.c:660-702 */
  if (slave == NULL || isnan(low) || isnan(high) || isnan(kaiser_beta)) return -1;
  struct orc_filter_in *master = slave->master;
  int const L_dec = slave->olen;
  int const M_dec = (master->impulse_length - 1) / slave->decimate + 1;
  int const N_dec = L_dec + M_dec - 1;
  int const N = master->ilen + master->impulse_length - 1;
  float gain = 1. / ((float)N);
  if (slave->out_type == ORC_REAL || slave->out_type == ORC_CROSS_CONJ) gain *= M_SQRT1_2;
  float complex *response = malloc(sizeof(float complex) * (size_t)N_dec);
  for (int n = 0; n < N_dec; n++) {
    float f;
    if (n <= N_dec / 2)
      f = (float)n / N_dec;
    else
      f = (float)(n - N_dec) / N_dec;
    if (f >= low && f <= high)
      response[n] = gain;
    else
      response[n] = 0;
  }
  orc_window_filter(L_dec, M_dec, response, kaiser_beta);
  float complex *tmp = slave->response;
  slave->response = response;
  slave->noise_gain = orc_noise_gain(slave);
  free(tmp);
  return 0;
}

float complex *orc_filter_in_input(struct orc_filter_in *f) { return f->input_c; }
float *orc_filter_in_input_r(struct orc_filter_in *f) { return f->input_r; }
float complex *orc_filter_in_fdomain(struct orc_filter_in *f) { return f->fdomain; }
float complex *orc_filter_out_output(struct orc_filter_out *f) { return f->output_c; }
float *orc_filter_out_output_r(struct orc_filter_out *f) { return f->output_r; }
float complex *orc_filter_out_response(struct orc_filter_out *f) { return f->response; }
float orc_filter_out_noise_gain(struct orc_filter_out *f) { return f->noise_gain; }
void orc_filter_out_set_type(struct orc_filter_out *f, int t) { f->out_type = t; }

void orc_notch_init(struct orc_notchfilter *nf, double f, float bw) { /* filter.c:707-717 */
  nf->osc_phase = 1;
  nf->osc_step = cispi_(2 * f);
  nf->dcstate = 0;
  nf->bw = bw;
}
void orc_notch_block(struct orc_notchfilter *nf, float complex *buf, int n) { /* filter.c:719-727 per sample */
  for (int i = 0; i < n; i++) {
    float complex s = buf[i];
    s = s * conj(nf->osc_phase) - nf->dcstate;
    nf->dcstate += nf->bw * s;
    s *= nf->osc_phase;
    nf->osc_phase *= nf->osc_step;
    buf[i] = s;
  }
}

/* =========================================================== osc.c */
#define ORC_RENORM_RATE 16384 /* osc.c:11 */

int orc_is_phasor_init(double complex x) { /* osc.c:14-18 */
  if (isnan(creal(x)) || isnan(cimag(x)) || cnrm_(x) < 0.9) return 0;
  return 1;
}
void orc_set_osc(struct orc_osc *osc, double f, double r) { /* osc.c:22-36 */
  if (!orc_is_phasor_init(osc->phasor)) {
    osc->phasor = 1;
    osc->steps = 0;
  }
  osc->freq = f;
  osc->rate = r;
  osc->phasor_step = cispi_(2 * osc->freq);
  if (osc->rate != 0)
    osc->phasor_step_step = cispi_(2 * osc->rate);
  else
    osc->phasor_step_step = 1;
}
void orc_renorm_osc(struct orc_osc *osc) { /* osc.c:52-58 */
  osc->steps = 0;
  osc->phasor /= cabs(osc->phasor);
  if (osc->rate != 0) osc->phasor_step /= cabs(osc->phasor_step);
}
double complex orc_step_osc(struct orc_osc *osc) { /* osc.c:39-50 */
  double complex r = osc->phasor;
  if (osc->rate != 0) osc->phasor_step *= osc->phasor_step_step;
  osc->phasor *= osc->phasor_step;
  if (++osc->steps == ORC_RENORM_RATE) orc_renorm_osc(osc);
  return r;
}
void orc_step_osc_block(struct orc_osc *osc, double complex *out, int n) {
  for (int i = 0; i < n; i++) out[i] = orc_step_osc(osc);
}
void orc_init_pll(struct orc_pll *pll, float nf, float damping, double freq, float samprate) { /* osc.c:64-85 */
  pll->samptime = 1. / samprate;
  freq *= pll->samptime;
  nf *= pll->samptime;
  float const vcogain = 2 * M_PI;
  float const pdgain = 1;
  float const natfreq = nf * 2 * M_PI;
  float const tau1 = vcogain * pdgain / (natfreq * natfreq);
  float const tau2 = 2 * damping / natfreq;
  pll->prop_gain = tau2 / tau1;
  pll->integrator_gain = 1 / tau1;
  pll->integrator = freq * pll->samptime / pll->integrator_gain;
}
float orc_run_pll(struct orc_pll *pll, float phase) { /* osc.c:91-107 */
  float feedback = pll->integrator_gain * pll->integrator + pll->prop_gain * phase;
  pll->integrator += phase;
  feedback = feedback > 0.49 ? 0.49 : feedback < -0.49 ? -0.49 : feedback;
  orc_set_osc(&pll->vco, feedback, 0);
  orc_step_osc(&pll->vco);
  return feedback;
}

/* =========================================================== decimate.c (portable version, 113-163) */
void orc_hb15_block(struct orc_hb15_state *state, float *output, const float *input, int cnt) { /* decimate.c:113-148 */
  float even_samples[4], odd_samples[4], old_odd_samples[4], coeffs[4];
  memcpy(coeffs, state->coeffs, sizeof(coeffs));
  memcpy(even_samples, state->even_samples, sizeof(even_samples));
  memcpy(odd_samples, state->odd_samples, sizeof(odd_samples));
  memcpy(old_odd_samples, state->old_odd_samples, sizeof(old_odd_samples));
  while (cnt--) {
    even_samples[0] = *input++;
    odd_samples[0] = *input++;
    float result = even_samples[3];
    for (int i = 2; i >= 0; i--) even_samples[i + 1] = even_samples[i];
    for (int i = 0; i < 4; i++) result += (odd_samples[i] + old_odd_samples[i]) * coeffs[i];
    *output++ = result;
    for (int i = 0; i < 3; i++) old_odd_samples[i] = old_odd_samples[i + 1];
    old_odd_samples[3] = odd_samples[3];
    for (int i = 2; i >= 0; i--) odd_samples[i + 1] = odd_samples[i];
  }
  memcpy(state->even_samples, even_samples, sizeof(even_samples));
  memcpy(state->odd_samples, odd_samples, sizeof(odd_samples));
  memcpy(state->old_odd_samples, old_odd_samples, sizeof(old_odd_samples));
}
void orc_hb3_block(float *state, float *output, const float *input, int cnt) { /* decimate.c:150-163 */
  float oldstate = *state;
  while (cnt--) {
    float in0 = *input++;
    float in1 = *input++;
    *output++ = 2 * in0 + in1 + oldstate;
    oldstate = in1;
  }
  *state = oldstate;
}
void orc_hb15_init_f8(struct orc_hb15_state *s) { /* hackrf.c:442-451 */
  memset(s, 0, sizeof(*s));
  s->coeffs[3] = 490. / 802;
  s->coeffs[2] = -116. / 802;
  s->coeffs[1] = 33. / 802;
  s->coeffs[0] = -6. / 802;
}
void orc_decimate_rail(struct orc_hb15_state *hb15, float *hb3, int log_decimate, int stage_threshold,
                       float *buf, int chunk) { /* hackrf.c:495-501 */
  int j;
  for (j = log_decimate - 1; j >= stage_threshold; j--) orc_hb3_block(&hb3[j], buf, buf, chunk << j);
  for (; j >= 0; j--) orc_hb15_block(&hb15[j], buf, buf, chunk << j);
}

/* =========================================================== radio.c */
static float const SCALE16 = 1. / SHRT_MAX; /* radio.c:34 */
static float const SCALE8 = 1. / INT8_MAX;  /* radio.c:35 */

int orc_sampcount(int pt, int size) { /* radio.c:80-99 (size after the IQ_PT 24-byte header is removed) */
  switch (pt) {
  case ORC_IQ_PT: return size / 4;
  case ORC_IQ_PT8: return size / 2;
  case ORC_PCM_STEREO_PT: return size / 4;
  case ORC_IQ_PT12: return size / 3;
  default: return -1;
  }
}

int orc_convert_samples(int pt, const unsigned char *dp, int size, float gain_factor, float complex *out) { /* radio.c:136-178 */
  int const sampcount = orc_sampcount(pt, size);
  if (sampcount < 0) return -1;
  switch (pt) {
  case ORC_IQ_PT12: {
    float gain = SCALE16 * gain_factor;
    for (int i = 0; i < sampcount; i++) {
      float re = gain * (short)(((dp[0] << 8) | dp[1]) & 0xfff0);
      float im = gain * (short)(((dp[1] << 8) | dp[2]) << 4);
      out[i] = CMPLXF(re, im);
      dp += 3;
    }
  } break;
  case ORC_PCM_STEREO_PT: {
    float gain = SCALE16 * gain_factor;
    for (int i = 0; i < sampcount; i++) {
      float re = gain * (signed short)((dp[0] << 8) | dp[1]);
      float im = gain * (signed short)((dp[2] << 8) | dp[3]);
      out[i] = CMPLXF(re, im);
      dp += 4;
    }
  } break;
  case ORC_IQ_PT: {
    float gain = SCALE16 * gain_factor;
    for (int i = 0; i < sampcount; i++) {
      float re = gain * (signed short)(dp[0] | (dp[1] << 8));
      float im = gain * (signed short)(dp[2] | (dp[3] << 8));
      out[i] = CMPLXF(re, im);
      dp += 4;
    }
  } break;
  case ORC_IQ_PT8: {
    float gain = SCALE8 * gain_factor;
    for (int i = 0; i < sampcount; i++) {
      float re = gain * (signed char)dp[0];
      float im = gain * (signed char)dp[1];
      out[i] = CMPLXF(re, im);
      dp += 2;
    }
  } break;
  }
  return sampcount;
}

float orc_mix_samples(struct orc_osc *doppler, struct orc_osc *lo2, const float complex *in, float complex *out,
                      int n) { /* radio.c:179-191 */
  float block_energy = 0;
  for (int i = 0; i < n; i++) {
    float complex x = in[i];
    if (doppler != NULL && doppler->freq != 0) x *= orc_step_osc(doppler); /* 180-184; complex float *= complex double */
    float complex samp = x * orc_step_osc(lo2);                            /* 187 */
    block_energy += cnrmf_(samp);                                          /* 188 */
    out[i] = samp;
  }
  return block_energy;
}
void orc_skip_samples(struct orc_osc *doppler, struct orc_osc *lo2, int n) { /* radio.c:118-122 */
  for (int i = 0; i < n; i++) {
    (void)orc_step_osc(lo2);
    (void)orc_step_osc(doppler);
  }
}

float orc_compute_n0(const float complex *fdomain, int N, float low, float high, int samprate) { /* radio.c:422-465 */
  int low_n = (N * low) / samprate;
  int high_n = (N * high) / samprate;
  int bincount = N - (high_n - low_n);
  if (bincount < 0 || bincount > N) return 0;
  float *power_spectrum = malloc(sizeof(float) * (size_t)(bincount > 0 ? bincount : 1));
  int in = high_n + 1;
  for (int n = 0; n < bincount; n++) {
    if (in >= N)
      in -= N;
    else if (in < 0)
      in += N;
    power_spectrum[n] = cnrmf_(fdomain[in++]);
  }
  float avg_n = INFINITY;
  for (int iter = 0; iter < 2; iter++) {
    int noisebins = 0;
    float new_avg_n = 0;
    for (int n = 0; n < bincount; n++) {
      float const s = power_spectrum[n];
      if (s < avg_n * 2) {
        new_avg_n += s;
        noisebins++;
      }
    }
    avg_n = new_avg_n / noisebins;
  }
  free(power_spectrum);
  return avg_n / ((float)N * samprate);
}

/* =========================================================== fm.c */
static double fm_i0(double z) { /* fm.c:152-163 */
  const double t = 0.25 * (z * z);
  double sum = 1 + t;
  double term = t;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * k);
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return sum;
}
static double fm_i1(double z) { /* fm.c:166-179 */
  const double t = 0.25 * (z * z);
  double term = 0.5 * t;
  double sum = 1 + term;
  for (int k = 2; k < 40; k++) {
    term *= t / (k * (k + 1));
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return 0.5 * z * sum;
}
static double fm_xi(double thetasq) { /* fm.c:180-186 */
  double t = (2 + thetasq) * fm_i0(0.25 * thetasq) + thetasq * fm_i1(0.25 * thetasq);
  t *= t;
  return 2 + thetasq - (0.125 * M_PI) * exp(-0.5 * thetasq) * t;
}
double orc_fm_snr(double r) { /* fm.c:190-205 */
  double thetasq = r;
  double othetasq = r + 10;
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  while (fabs(thetasq - othetasq) > .001) {
    othetasq = thetasq;
    thetasq = fm_xi(thetasq) * (1 + r) - 2;
  }
  return thetasq;
}

void orc_demod_fm_block(const struct orc_fm_params *prm, struct orc_fm_state *st, const float complex *y, int olen,
                        float *samples, struct orc_sig *sig) { /* fm.c:43-139 */
  float const dsamprate = (float)prm->input_samprate / prm->decimate;                       /* fm.c:27 */
  float const gain = (prm->headroom * M_1_PI * dsamprate) / fabsf(prm->low - prm->high);    /* fm.c:47 */
  float *amplitudes = malloc(sizeof(float) * (size_t)olen);
  sig->bb_power = 0;
  float avg_amp = 0;
  for (int n = 0; n < olen; n++) { /* 55-59 */
    float t = cnrmf_(y[n]);
    sig->bb_power += t;
    avg_amp += amplitudes[n] = sqrtf(t);
  }
  sig->bb_power /= olen;
  avg_amp /= olen;
  float fm_variance = 0;
  for (int n = 0; n < olen; n++) fm_variance += (amplitudes[n] - avg_amp) * (amplitudes[n] - avg_amp);
  fm_variance /= (olen - 1);
  sig->snr = orc_fm_snr(avg_amp * avg_amp / fm_variance);
  sig->snr = fmaxf(0.0f, sig->snr);
  float const thresh = 4;
  if (sig->snr > thresh) st->squelch_open = 2;
  if (st->squelch_open > 0) {
    st->squelch_open--;
    float pdev_pos = 0, pdev_neg = 0, avg_f = 0;
    for (int n = 0; n < olen; n++) { /* 85-113 */
      float complex p = y[n] * conjf(st->state);
      st->state = y[n];
      float ang = cargf(p);
      float amp = cabsf(p) / fm_variance;
      if (amp > 1) amp = 1;
      avg_f += ang;
      if (n == 0)
        pdev_pos = pdev_neg = ang;
      else if (ang > pdev_pos)
        pdev_pos = ang;
      else if (st->lastaudio < pdev_neg)
        pdev_neg = ang;
      if (prm->flat) {
        samples[n] = ang * gain;
      } else {
        samples[n] = st->lastaudio += ang * .114 * gain * amp;
        st->lastaudio *= 0.99376949;
      }
    }
    avg_f /= olen;
    sig->foffset = dsamprate * avg_f * ORC_M_1_2PI;
    pdev_pos -= avg_f;
    pdev_neg -= avg_f;
    sig->pdeviation = dsamprate * fmaxf(pdev_pos, -pdev_neg) * ORC_M_1_2PI;
  } else {
    st->state = 0;
    for (int n = 0; n < olen; n++) { /* 124-132 */
      if (prm->flat)
        samples[n] = 0;
      else {
        samples[n] = st->lastaudio;
        st->lastaudio *= 0.99376949;
      }
    }
  }
  float output_level = 0;
  for (int n = 0; n < olen; n++) output_level += samples[n] * samples[n];
  sig->output_level = output_level / olen;
  free(amplitudes);
}

/* =========================================================== linear.c */
void orc_linear_init(const struct orc_linear_params *p, struct orc_linear_state *st, float agc_gain0,
                     double shift_cycles_per_sample) { /* linear.c:26-44 */
  memset(st, 0, sizeof(*st));
  orc_init_pll(&st->pll, p->loop_bw, (float)M_SQRT1_2, 0.0, (float)p->output_samprate); /* linear.c:33,41 */
  st->pll.vco.phasor = 1; /* quirk: reference leaves it uninitialised (linear.c:40); first set_osc makes it 1 */
  st->agc_gain = agc_gain0;
  st->snr = 0; /* linear.c:44 */
  orc_set_osc(&st->shift, shift_cycles_per_sample, 0.0); /* radio.c:359 */
}

void orc_demod_linear_block(const struct orc_linear_params *prm, struct orc_linear_state *st, float complex *y,
                            int olen, float *samples, struct orc_sig *sig) { /* linear.c:57-156 */
  float const snrthresh = 2;                                   /* 32 */
  int const lock_limit = 1 /*lock_time*/ * prm->output_samprate; /* 34-36 */
  if (prm->pll) {
    if (st->snr < snrthresh) /* 64-68; NaN compares false -> += */
      st->lock_count -= olen;
    else
      st->lock_count += olen;
    if (st->lock_count >= lock_limit) {
      st->lock_count = lock_limit;
      st->pll_lock = 1;
    }
    if (st->lock_count <= -lock_limit) {
      st->lock_count = -lock_limit;
      st->pll_lock = 0;
    }
    sig->lock_timer = st->lock_count;
    float signal = 0, noise = 0;
    for (int n = 0; n < olen; n++) { /* 81-97 */
      float complex s = y[n];
      s *= conjf(st->pll.vco.phasor); /* conjf of a complex double: converts to complex float first */
      float phase;
      if (prm->square)
        phase = cargf(s * s);
      else
        phase = cargf(s);
      orc_run_pll(&st->pll, phase);
      y[n] = s;
      float rp = crealf(s) * crealf(s);
      float ip = cimagf(s) * cimagf(s);
      signal += rp;
      noise += ip;
    }
    sig->cphase = carg(st->pll.vco.phasor);
    if (prm->square) sig->cphase /= 2;
    sig->foffset = st->pll.vco.freq * prm->output_samprate;
    if (noise != 0) {
      st->snr = (signal / noise) - 1;
      if (st->snr < 0) st->snr = 0;
    } else
      st->snr = NAN;
  }
  sig->pll_lock = st->pll_lock;
  sig->snr = st->snr;
  float energy = 0, output_level = 0;
  for (int n = 0; n < olen; n++) { /* 114-148 */
    float complex s = y[n] * orc_step_osc(&st->shift);
    float norm = cnrmf_(s);
    energy += norm;
    float amplitude = sqrtf(norm);
    if (prm->agc) {
      if (isnan(st->agc_gain) || amplitude * st->agc_gain > prm->headroom) {
        st->agc_gain = prm->headroom / amplitude;
        st->hangcount = prm->hangtime;
      } else if (st->hangcount > 0) {
        st->hangcount--;
      } else {
        st->agc_gain *= prm->recovery_rate;
      }
    }
    if (prm->env) {
      samples[n] = amplitude * st->agc_gain;
      output_level += samples[n] * samples[n];
    } else if (prm->channels == 1) {
      samples[n] = crealf(s) * st->agc_gain;
      output_level += samples[n] * samples[n];
    } else {
      y[n] = s * st->agc_gain;
      output_level += cnrmf_(y[n]);
    }
  }
  sig->output_level = output_level / (olen * prm->channels);
  sig->bb_power = energy / olen;
  sig->agc_gain = st->agc_gain;
}

/* =========================================================== am.c (stale in the reference; see header) */
void orc_am_init(const struct orc_am_params *p, struct orc_am_state *st, float agc_gain0) { /* am.c:21-35 */
  float const samptime = p->decimate / (float)p->input_samprate;
  st->hangcount = 0;
  st->recovery_rate_per_sample = powf(10., (p->recovery_rate * samptime) / 20.); /* dB2voltage, dsp.h:75 */
  st->prev_recovery_rate = p->recovery_rate;
  st->DC_filter = 0;
  st->agc_gain = agc_gain0;
}
void orc_demod_am_block(const struct orc_am_params *p, struct orc_am_state *st, const float complex *y, int olen,
                        float *samples, struct orc_sig *sig) { /* am.c:42-91 */
  float const samptime = p->decimate / (float)p->input_samprate;
  float const DC_filter_coeff = .0001;
  if (p->recovery_rate != st->prev_recovery_rate) { /* 56-59 */
    st->recovery_rate_per_sample = powf(10., (p->recovery_rate * samptime) / 20.);
    st->prev_recovery_rate = p->recovery_rate;
  }
  float signal = 0;
  for (int n = 0; n < olen; n++) {
    float const sampsq = cnrmf_(y[n]);
    signal += sampsq;
    float samp = sqrtf(sampsq);
    st->DC_filter += DC_filter_coeff * (samp - st->DC_filter);
    if (isnan(st->agc_gain)) {
      st->agc_gain = p->headroom / st->DC_filter;
    } else if (st->agc_gain * st->DC_filter > p->headroom) {
      st->agc_gain = p->headroom / st->DC_filter;
      st->hangcount = p->hangtime * p->output_samprate; /* am.c:80 (quirk 15) */
    } else if (st->hangcount != 0) {
      st->hangcount--;
    } else {
      st->agc_gain *= st->recovery_rate_per_sample;
    }
    samples[n] = (samp - st->DC_filter) * st->agc_gain;
  }
  sig->bb_power = signal / olen;
  sig->agc_gain = st->agc_gain;
}

/* =========================================================== audio.c */
short orc_scaleclip(float x) { /* audio.c:22-28 */
  if (x >= 1.0)
    return SHRT_MAX;
  else if (x <= -1.0)
    return SHRT_MIN;
  return (short)(SHRT_MAX * x);
}
void orc_scaleclip_block(const float *in, int16_t *out, int n) {
  for (int i = 0; i < n; i++) out[i] = orc_scaleclip(in[i]);
}
#define ORC_PCM_BUFSIZE 480 /* audio.c:19 */
static inline uint16_t swap16(uint16_t v) { return (uint16_t)((v << 8) | (v >> 8)); }

int orc_packetise(struct orc_pcm_state *st, const float *buffer, int size, int stereo, int16_t *pcm_be, int *sent,
                  int *marker, uint32_t *ts, int maxchunks) { /* audio.c:32-82 (stereo), 85-135 (mono) */
  int nchunks = 0;
  while (size > 0 && nchunks < maxchunks) {
    int not_silent = 0;
    int chunk = stereo ? (ORC_PCM_BUFSIZE < 2 * size ? ORC_PCM_BUFSIZE : 2 * size)
                       : (ORC_PCM_BUFSIZE < size ? ORC_PCM_BUFSIZE : size);
    for (int i = 0; i < chunk; i++) {
      int16_t v = (int16_t)swap16((uint16_t)orc_scaleclip(*buffer++)); /* htons on a little-endian host */
      *pcm_be++ = v;
      not_silent |= v;
    }
    int const frames = stereo ? chunk / 2 : chunk;
    ts[nchunks] = st->timestamp;
    st->timestamp += frames;
    marker[nchunks] = 0;
    if (not_silent) {
      st->packets++;
      st->bytes += 2 * chunk;
      if (st->silent) {
        st->silent = 0;
        marker[nchunks] = 1;
      }
      st->seq++;
      st->samples += frames;
      sent[nchunks] = 1;
    } else {
      st->silent = 1;
      sent[nchunks] = 0;
    }
    nchunks++;
    size -= frames;
  }
  return nchunks;
}

/* =========================================================== packet.c (AFSK-1200 demodulator, HDLC), ax25.c (CRC) */
void orc_pcm_to_real(const int16_t *pcm_be, float *out, int n) { /* packet.c:340-343 */
  float const SCALE = 1. / 32768; /* packet.c:43 */
  for (int i = 0; i < n; i++) {
    /* ntohs() yields an UNSIGNED short, so a negative sample s arrives as s + 65536: the real signal is
     * s/32768 + 2 [s < 0].  The 100 Hz .. 4 kHz analytic filter removes the mean; what is left acts like a limiter. */
    uint16_t const host = swap16((uint16_t)pcm_be[i]);
    out[i] = host * SCALE;
  }
}
int orc_crc_good(const unsigned char *frame, int length) { /* ax25.c:140-156 */
  unsigned int const crc_poly = 0x8408;
  unsigned short crc = 0xffff;
  while (length-- > 0) {
    unsigned char byte = *frame++;
    for (int i = 0; i < 8; i++) {
      unsigned short feedback = 0;
      if ((crc ^ byte) & 1) feedback = crc_poly;
      crc = (crc >> 1) ^ feedback;
      byte >>= 1;
    }
  }
  return crc == 0xf0b8;
}
void orc_afsk_init(struct orc_afsk_state *st) { /* packet.c:410-435 */
  float const Samprate = 48000; /* packet.c:49 */
  memset(st, 0, sizeof *st);
  orc_set_osc(&st->mark, -1200. / Samprate, 0.0);  /* packet.c:414 */
  orc_set_osc(&st->space, -2200. / Samprate, 0.0); /* packet.c:419 */
}
/* ms (may be NULL; tests only): the tone products of every sample, {Re, Im} of y * mark and of y * space, taken as given
 * instead of formed here.  With a device's own products the loop below is that device's arithmetic operation for
 * operation (sequential float additions, cnrmf_ without contraction): the sweeps' referee for a bit clock that parts
 * from the oracle's at a knife edge of packet.c:469 (tools/fuzz_afsk.py). */
int orc_afsk_block_ms(struct orc_afsk_state *st, const float complex *y, const float *ms, int n_in, int sample_base,
                      struct orc_ax25_frame *out, int maxframes, struct orc_afsk_bit *bits, int maxbits, int *nbits) {
  int const Samppbit = ORC_AFSK_SAMPPBIT;
  int nframes = 0, nb = 0;
  for (int n = 0; n < n_in; n++) { /* packet.c:440-544 */
    float complex s;
    if (ms)
      s = ms[4 * n] + I * ms[4 * n + 1];
    else
      s = y[n] * orc_step_osc(&st->mark); /* complex float x complex double, rounded to float on assignment */
    st->mark_accum += s;
    st->mark_offset_accum += s;
    if (ms)
      s = ms[4 * n + 2] + I * ms[4 * n + 3];
    else
      s = y[n] * orc_step_osc(&st->space);
    st->space_accum += s;
    st->space_offset_accum += s;
    if (++st->symphase == Samppbit / 2) {
      st->mid_val = cnrmf_(st->mark_offset_accum) - cnrmf_(st->space_offset_accum);
      st->mark_offset_accum = st->space_offset_accum = 0;
    }
    if (st->symphase < Samppbit) continue;
    st->symphase = 0;
    float cur_val = cnrmf_(st->mark_accum) - cnrmf_(st->space_accum);
    st->mark_accum = st->space_accum = 0;
    if (bits && nb < maxbits) {
      bits[nb].sample = sample_base + n;
      bits[nb].cur_val = cur_val;
    }
    nb++;
    if (cur_val * st->last_val < 0) {
      st->symphase += ((cur_val - st->last_val) * st->mid_val) > 0 ? +1 : -1; /* packet.c:469 */
      if (st->ones == 6) { /* flag */
        if (st->flagsync) {
          st->frame_bit -= 7;
          int bytes = st->frame_bit / 8;
          /* the reference indexes hdlc_frame[1024] without a bound; frames that would not fit are dropped here */
          if (bytes > 0 && bytes <= ORC_AFSK_FRAME_MAX && orc_crc_good(st->hdlc_frame, bytes)) {
            if (nframes < maxframes) {
              struct orc_ax25_frame *f = &out[nframes];
              f->sample = sample_base + n;
              f->bytes = bytes;
              f->seq = st->out_seq;
              f->timestamp = st->out_timestamp;
              memset(f->data, 0, sizeof f->data);
              memcpy(f->data, st->hdlc_frame, bytes);
              nframes++;
            }
            st->out_seq++;                 /* packet.c:498 */
            st->out_timestamp += bytes;    /* packet.c:501 */
            st->out_packets++;             /* packet.c:510-511 */
            st->out_bytes += bytes;
          }
        }
        memset(st->hdlc_frame, 0, sizeof st->hdlc_frame); /* packet.c:514-517 */
        st->frame_bit = 0;
        st->flagsync = 1;
      } else if (st->ones == 5) {
        /* stuffed zero */
      } else if (st->ones < 5) {
        if (st->flagsync) st->frame_bit++;
      }
      st->ones = 0;
    } else {
      if (++st->ones == 7) { /* abort */
        memset(st->hdlc_frame, 0, sizeof st->hdlc_frame);
        st->frame_bit = 0;
        st->flagsync = 0;
      } else if (st->flagsync) {
        if (st->frame_bit >= 0 && st->frame_bit / 8 < ORC_AFSK_FRAME_MAX)
          st->hdlc_frame[st->frame_bit / 8] |= 1 << (st->frame_bit % 8);
        st->frame_bit++;
      }
    }
    st->last_val = cur_val;
  }
  if (nbits) *nbits = nb;
  return nframes;
}
int orc_afsk_block(struct orc_afsk_state *st, const float complex *y, int n_in, int sample_base,
                   struct orc_ax25_frame *out, int maxframes, struct orc_afsk_bit *bits, int maxbits, int *nbits) {
  return orc_afsk_block_ms(st, y, NULL, n_in, sample_base, out, maxframes, bits, maxbits, nbits);
}

/* =========================================================== pl.c */
const float orc_pl_tones[ORC_PL_NTONES] = { /* pl.c:36-44 */
    67.0,  69.3,  71.9,  74.4,  77.0,  79.7,  82.5,  85.4,  88.5,  91.5,  94.8,  97.4,
    100.0, 103.5, 107.2, 110.9, 114.8, 118.8, 123.0, 127.3, 131.8, 136.5, 141.3, 146.2,
    150.0, 151.4, 156.7, 162.2, 167.9, 173.8, 179.9, 186.2, 192.8, 199.5, 206.5, 213.8,
    221.3, 229.1, 237.1, 245.5, 254.1, 159.8, 165.5, 171.3, 177.3, 183.5, 189.9, 196.6,
    203.5, 210.7, 218.1, 225.7, 233.6, 241.8, 250.3,
};
static const float dtmf_low_tones[4] = {697, 770, 852, 941};      /* pl.c:48 */
static const float dtmf_high_tones[4] = {1209, 1336, 1477, 1633}; /* pl.c:49 */
static const char dtmf_matrix[4][4] = {{'1', '2', '3', 'A'}, {'4', '5', '6', 'B'}, {'7', '8', '9', 'C'}, {'*', '0', '#', 'D'}};
#define PL_SAMPRATE 48000 /* pl.c:25 */

void orc_pl_init(struct orc_pl_session *sp) { /* make_session, pl.c:305,317-329 */
  memset(sp, 0, sizeof *sp);
  for (int n = 0; n < ORC_PL_NTONES; n++)
    orc_set_osc(&sp->pl_osc[n], orc_pl_tones[n] / PL_SAMPRATE, 9); /* rate 9 cycles/sample^2: cispi(18) is 1 - 2.2e-15 i */
  for (int n = 0; n < 4; n++) {
    orc_set_osc(&sp->dtmf_low_osc[n], dtmf_low_tones[n] / PL_SAMPRATE, 0);
    orc_set_osc(&sp->dtmf_high_osc[n], dtmf_high_tones[n] / PL_SAMPRATE, 0);
  }
}
static void process_pl(struct orc_pl_session *sp, struct orc_pl_result *r) { /* pl.c:360-387 */
  sp->pl_audio_count = 0;
  /* qsort in descending energy (pl.c:369): only the two largest are used */
  float e0 = -1, e1 = -1;
  int i0 = 0;
  for (int n = 0; n < ORC_PL_NTONES; n++) {
    float const e = cnrmf_(sp->pl_integrators[n]);
    sp->pl_integrators[n] = 0;
    if (e > e0) {
      e1 = e0;
      e0 = e;
      i0 = n;
    } else if (e > e1)
      e1 = e;
  }
  float const pl_snr = 10 * log10f(e0) - 10 * log10f(e1); /* power2dB, dsp.h:74 */
  r->tone = pl_snr >= 6 ? orc_pl_tones[i0] : 0;
  r->snr = pl_snr;
  r->e0 = e0;
  r->e1 = e1;
  r->index = i0;
}
static int dtmf_group(float complex *integ, float *snr_out) { /* pl.c:395-410 */
  int index = -1;
  float max_energy = 0, total_energy = 0;
  for (int n = 0; n < 4; n++) {
    float energy = cnrmf_(integ[n]);
    integ[n] = 0;
    total_energy += energy;
    if (energy >= max_energy) {
      max_energy = energy;
      index = n;
    }
  }
  float const snr = max_energy / (total_energy - max_energy);
  if (snr < 10) index = -1; /* (a NaN from digital silence does not take this branch) */
  *snr_out = snr;
  return index;
}
static void process_dtmf(struct orc_pl_session *sp, struct orc_dtmf_result *r) { /* pl.c:390-439 */
  sp->dtmf_audio_count = 0;
  r->low_index = dtmf_group(sp->dtmf_low_integrators, &r->low_snr);
  r->high_index = dtmf_group(sp->dtmf_high_integrators, &r->high_snr);
  r->digit = (r->low_index != -1 && r->high_index != -1) ? dtmf_matrix[r->low_index][r->high_index] : 0;
}
void orc_pl_samples(struct orc_pl_session *sp, const int16_t *pcm_be, int count, struct orc_pl_result *pl, int maxpl,
                    int *npl, struct orc_dtmf_result *dt, int maxdt, int *ndt) {
  float const SCALE16 = 1. / SHRT_MAX; /* pl.c:29 */
  int np = 0, nd = 0;
  for (int k = 0; k < count; k++) { /* pl.c:249-276 */
    float samp = SCALE16 * (short)swap16((uint16_t)pcm_be[k]);
    for (int n = 0; n < ORC_PL_NTONES; n++) sp->pl_integrators[n] += samp * orc_step_osc(&sp->pl_osc[n]);
    if (++sp->pl_audio_count >= ORC_PL_BLOCKSIZE) {
      struct orc_pl_result r;
      process_pl(sp, &r);
      if (np < maxpl) pl[np] = r;
      np++;
    }
    for (int n = 0; n < 4; n++) {
      sp->dtmf_low_integrators[n] += samp * orc_step_osc(&sp->dtmf_low_osc[n]);
      sp->dtmf_high_integrators[n] += samp * orc_step_osc(&sp->dtmf_high_osc[n]);
    }
    if (++sp->dtmf_audio_count >= ORC_DTMF_BLOCKSIZE) {
      struct orc_dtmf_result r;
      process_dtmf(sp, &r);
      if (nd < maxdt) dt[nd] = r;
      nd++;
    }
  }
  if (npl) *npl = np;
  if (ndt) *ndt = nd;
}

/* =========================================================== front-end corrections: hackrf.c, funcube.c */
void orc_fe_init(struct orc_fe_state *st, float imbalance0) {
  memset(st, 0, sizeof *st);
  st->imbalance = imbalance0;
  st->gain_i = st->gain_q = st->secphi = 1; /* hackrf.c:136-139, funcube.c:331-334 */
  st->tanphi = 0;
}
static void fe_update_gains(struct orc_fe_state *st) { /* hackrf.c:913-916, funcube.c:424-427 */
  st->gain_q = sqrtf(0.5 * (1 + st->imbalance));
  st->gain_i = sqrtf(0.5 * (1 + 1. / st->imbalance));
  st->secphi = 1 / sqrtf(1 - st->sinphi * st->sinphi);
  st->tanphi = st->sinphi * st->secphi;
}
void orc_hackrf_rx_block(struct orc_fe_state *st, const int8_t *dp, int samples, int offset, float Rate_factor,
                         float *out_i, float *out_q) { /* hackrf.c:822-919 */
  float const SCALE8 = 1. / INT8_MAX; /* hackrf.c:64 */
  float const DC_alpha = .001;        /* hackrf.c:66 */
  float complex samp_sum = 0;
  float i_energy = 0, q_energy = 0, dotprod = 0;
  for (int n = 0; n < samples; n++) {
    float re = SCALE8 * (float)*dp++;
    float im = SCALE8 * (float)*dp++;
    if (im < -1) {
      st->clips++;
      im = -1;
    }
    if (re < -1) {
      st->clips++;
      re = -1;
    }
    float complex samp = CMPLXF(re, im);
    samp_sum += samp;
    samp -= st->DC;
    re = crealf(samp);
    im = cimagf(samp);
    i_energy += re * re;
    q_energy += im * im;
    re *= st->gain_i;
    im *= st->gain_q;
    dotprod += re * im;
    im = st->secphi * im - st->tanphi * re;
    switch (st->rotate_phase) { /* hackrf.c:878-896 */
    default:
    case 0: out_i[n] = re; out_q[n] = im; break;
    case 1: out_i[n] = -im; out_q[n] = re; break;
    case 2: out_i[n] = -re; out_q[n] = -im; break;
    case 3: out_i[n] = im; out_q[n] = -re; break;
    }
    st->rotate_phase += offset;
    st->rotate_phase &= 3;
  }
  st->DC += DC_alpha * (samp_sum / samples - st->DC); /* hackrf.c:906 */
  float block_energy = i_energy + q_energy;
  if (block_energy > 0) {
    st->in_power = block_energy / samples;
    st->imbalance += Rate_factor * samples * ((i_energy / q_energy) - st->imbalance);
    float dpn = 2 * dotprod / block_energy;
    st->sinphi += Rate_factor * samples * (dpn - st->sinphi);
    fe_update_gains(st);
  }
}
void orc_funcube_block(struct orc_fe_state *st, int16_t *sampbuf, int Blocksize, float rate_factor) { /* funcube.c:364-428 */
  float const SCALE16 = 1. / SHRT_MAX; /* funcube.c:64 */
  float const DC_alpha = 1.0e-6;       /* funcube.c:65 */
  float i_energy = 0, q_energy = 0, dotprod = 0;
  float complex samp_sum = 0;
  for (int i = 0; i < 2 * Blocksize; i += 2) {
    float complex samp = CMPLXF(sampbuf[i], sampbuf[i + 1]) * SCALE16;
    samp_sum += samp;
    samp -= st->DC;
    float re = crealf(samp), im = cimagf(samp);
    i_energy += re * re;
    q_energy += im * im;
    re *= st->gain_i;
    im *= st->gain_q;
    dotprod += re * im;
    im = st->secphi * im - st->tanphi * re;
    sampbuf[i] = (int16_t)swap16((uint16_t)(signed short)round(re * SHRT_MAX));     /* htons, funcube.c:390-391 */
    sampbuf[i + 1] = (int16_t)swap16((uint16_t)(signed short)round(im * SHRT_MAX));
  }
  st->DC += DC_alpha * (samp_sum - Blocksize * st->DC); /* funcube.c:417 */
  float block_energy = i_energy + q_energy;
  if (block_energy > 0) {
    st->in_power = block_energy / Blocksize;
    st->imbalance += rate_factor * ((i_energy / q_energy) - st->imbalance);
    float dpn = 2 * dotprod / block_energy;
    st->sinphi += rate_factor * (dpn - st->sinphi);
    fe_update_gains(st);
  }
}
static inline int fe_clip(float s, int lo, int hi) { /* "Clip?" (hackrf.c:518): an out-of-range float -> short conversion is
  undefined in C; the restatement saturates, and the tests stay inside the range */
  return s >= hi ? hi : s <= lo ? lo : (int)s;
}
float orc_hackrf_pack(int pt, const float *ip, const float *qp, int chunk, float Filter_atten, unsigned char *dp) {
  float output_energy = 0;
  for (int i = 0; i < chunk; i++) { /* hackrf.c:511-560 */
    float s = *ip++ * Filter_atten;
    output_energy += s * s;
    float t = *qp++ * Filter_atten;
    output_energy += t * t;
    switch (pt) {
    case 97: { /* IQ_PT: 16-bit little endian */
      short si = (short)fe_clip(s, -32768, 32767), sq = (short)fe_clip(t, -32768, 32767);
      dp[0] = si & 0xff; dp[1] = (si >> 8) & 0xff; dp[2] = sq & 0xff; dp[3] = (sq >> 8) & 0xff;
      dp += 4;
    } break;
    case 95: { /* IQ_PT12: packed big endian */
      short si = (short)fe_clip(s, -32768, 32767), sq = (short)fe_clip(t, -32768, 32767);
      dp[0] = si >> 8;
      dp[1] = (si & 0xf0) | ((sq >> 12) & 0xf);
      dp[2] = sq >> 4;
      dp += 3;
    } break;
    default: { /* IQ_PT8 */
      dp[0] = (unsigned char)(signed char)fe_clip(s, -128, 127);
      dp[1] = (unsigned char)(signed char)fe_clip(t, -128, 127);
      dp += 2;
    } break;
    }
  }
  return output_energy;
}

/* =========================================================== misc.c */
uint32_t orc_nextfastfft(uint32_t n) { /* misc.c:129-148 */
  uint64_t result = 4288306050ULL;
  if (n >= result) return 0;
  for (uint64_t f7 = 1; f7 < result; f7 *= 7)
    for (uint64_t f5 = f7; f5 < result; f5 *= 5)
      for (uint64_t f3 = f5; f3 < result; f3 *= 3)
        for (uint64_t f2 = f3; f2 < result; f2 *= 2)
          if (f2 > n) {
            result = f2;
            break;
          }
  return (uint32_t)result;
}
