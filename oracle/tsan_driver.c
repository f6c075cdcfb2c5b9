/* tsan_driver.c -- TEST INFRASTRUCTURE ONLY (make -C oracle tsan).
 * The oracle is single threaded, but oracle/pipeline.py's `threads=` mode (bench.py's cpu_baseline experiments, the
 * threaded parity runs) calls it from a thread pool: one shared orc_filter_in whose forward transform the main thread
 * runs per block, then the channels' orc_execute_filter_output + orc_demod_fm_block in parallel, each on a filter_out
 * and demodulator state of its own, all reading the master's frequency domain.  This driver does exactly that under
 * -fsanitize=thread: any hidden shared mutable state in the restatement (a cached plan, a static scratch buffer) would be
 * reported as a data race.  Exit status 0 and no report = clean. */
#include "ka9q_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { NTHREADS = 4, CH_PER_THREAD = 3, BLOCKS = 6 };
static struct orc_filter_in *master;
struct worker {
  struct orc_filter_out *out[CH_PER_THREAD];
  struct orc_fm_params p;
  struct orc_fm_state st[CH_PER_THREAD];
  int rotate[CH_PER_THREAD];
  double checksum;
};
static void *run_block(void *arg) {
  struct worker *w = arg;
  float samples[4096];
  struct orc_sig sig;
  for (int c = 0; c < CH_PER_THREAD; c++) {
    orc_execute_filter_output(w->out[c], w->rotate[c]);
    int const olen = (int)w->out[c]->olen;
    orc_demod_fm_block(&w->p, &w->st[c], orc_filter_out_output(w->out[c]), olen, samples, &sig);
    for (int n = 0; n < olen; n++) w->checksum += samples[n];
    w->checksum += sig.snr;
  }
  return NULL;
}
int main(void) {
  unsigned const L = 3840, M = 4353, D = 4; /* the reference's default sizes (main.c:326-330) */
  master = orc_create_filter_input(L, M, ORC_COMPLEX);
  if (!master) return 2;
  struct worker w[NTHREADS];
  memset(w, 0, sizeof w);
  for (int t = 0; t < NTHREADS; t++) {
    w[t].p.input_samprate = 192000;
    w[t].p.decimate = (int)D;
    w[t].p.headroom = 0.178f;
    w[t].p.low = -5500.f;
    w[t].p.high = 5500.f;
    for (int c = 0; c < CH_PER_THREAD; c++) {
      w[t].out[c] = orc_create_filter_output(master, NULL, D, ORC_COMPLEX);
      if (!w[t].out[c]) return 2;
      orc_set_filter(w[t].out[c], -5500.f / 48000.f, 5500.f / 48000.f, 3.0f);
      w[t].rotate[c] = 40 * (t * CH_PER_THREAD + c) - 200;
    }
  }
  unsigned seed = 1;
  for (int b = 0; b < BLOCKS; b++) {
    float complex *in = orc_filter_in_input(master);
    for (unsigned n = 0; n < L; n++) {
      seed = seed * 1664525u + 1013904223u;
      float const nz = ((seed >> 8) & 0xffff) / 65536.f - .5f;
      in[n] = 0.3f * cexpf(I * (float)(2 * M_PI * 0.01 * (b * L + n))) + 0.01f * nz;
    }
    orc_execute_filter_input(master);
    pthread_t th[NTHREADS];
    for (int t = 0; t < NTHREADS; t++) pthread_create(&th[t], NULL, run_block, &w[t]);
    for (int t = 0; t < NTHREADS; t++) pthread_join(th[t], NULL);
  }
  double sum = 0;
  for (int t = 0; t < NTHREADS; t++) sum += w[t].checksum;
  printf("tsan_driver: %d threads x %d channels x %d blocks, checksum %.6g\n", NTHREADS, CH_PER_THREAD, BLOCKS, sum);
  for (int t = 0; t < NTHREADS; t++)
    for (int c = 0; c < CH_PER_THREAD; c++) orc_delete_filter_output(w[t].out[c]);
  orc_delete_filter_input(master);
  return isfinite(sum) ? 0 : 3;
}
