/* abi_threads.c -- a C11 caller of libka9q_hip that knows nothing but include/ka9q_hip.h, driving the filter.h API the
 * way `radio` does: a producer thread fills input.c[] and calls execute_filter_input (radio.c:186-195), a consumer
 * thread blocks in execute_filter_output (fm.c:43) and reads output.c[].  The numbers checked are the known answers
 * SURVEY.md 8(c) recorded from the unmodified reference (filter.c + osc.c): fs = 192 kS/s, L = 3840, M = 4353,
 * decimate 4, set_filter(-8000/48000, +8000/48000, 3.0), input = unit tone at +1 kHz with phase 0 at n = 0.
 *
 *   gcc -std=gnu11 -O2 -pthread -Iinclude tests/c/abi_threads.c -Lka9q-radio_amd -lka9q_hip -lm -o tests/c/abi_threads
 */
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdio.h>
#include <stdlib.h>

#include <ka9q_hip.h>

enum { L = 3840, M = 4353, D = 4, NBLOCKS = 4 };

static struct filter_in *Fin;
static struct filter_out *Fout;
static sem_t Consumed;
static float Power[NBLOCKS];
static complex float First[NBLOCKS];
static int Consumer_rc;

static void *consumer(void *arg) { /* demod_fm's loop, fm.c:35-56, without the demodulator */
  (void)arg;
  for (int b = 0; b < NBLOCKS; b++) {
    if (execute_filter_output(Fout, 0) != 0) { /* blocks until the producer has run execute_filter_input */
      Consumer_rc = -1;
      sem_post(&Consumed);
      return NULL;
    }
    float p = 0;
    for (unsigned n = 0; n < Fout->olen; n++) {
      complex float const y = Fout->output.c[n];
      p += crealf(y) * crealf(y) + cimagf(y) * cimagf(y);
    }
    Power[b] = p / Fout->olen;
    First[b] = Fout->output.c[0];
    sem_post(&Consumed); /* the reference skips blocks a slow consumer misses (filter.c:261-263): hand back in step */
  }
  return NULL;
}

static int near(double got, double want, double tol, const char *what) {
  if (fabs(got - want) <= tol) return 0;
  fprintf(stderr, "FAIL %s: got %.7g, want %.7g (+-%g)\n", what, got, want, tol);
  return 1;
}

int main(void) {
  if (ka9q_hip_init(0) != 0) {
    fprintf(stderr, "ka9q_hip_init: %s\n", ka9q_hip_last_error());
    return 2;
  }
  Fin = create_filter_input(L, M, COMPLEX);                 /* main.c:332 */
  Fout = create_filter_output(Fin, NULL, D, COMPLEX);       /* main.c:335 */
  if (Fin == NULL || Fout == NULL || set_filter(Fout, -8000.f / 48000, 8000.f / 48000, 3.0f) != 0) { /* main.c:336-339 */
    fprintf(stderr, "set-up: %s\n", ka9q_hip_last_error());
    return 2;
  }
  int bad = 0;
  bad += near(Fout->noise_gain, 0.0831292, 2e-6, "noise_gain");
  bad += near(noise_gain(Fout), 0.0831292, 2e-6, "noise_gain()");
  bad += Fout->olen != L / D || Fin->ilen != L || Fin->impulse_length != M;
  sem_init(&Consumed, 0, 0);
  pthread_t th;
  pthread_create(&th, NULL, consumer, NULL);
  double const f = 1000.0 / 192000.0; /* cycles per sample */
  for (int b = 0; b < NBLOCKS; b++) { /* proc_samples, radio.c:186-195 */
    for (int n = 0; n < L; n++) {
      double const ph = 2 * M_PI * fmod(f * ((double)b * L + n), 1.0);
      Fin->input.c[n] = (float)cos(ph) + I * (float)sin(ph);
    }
    Fout->out_type = COMPLEX; /* radio.c:194 writes it every block */
    if (execute_filter_input(Fin) != 0) {
      fprintf(stderr, "execute_filter_input: %s\n", ka9q_hip_last_error());
      return 2;
    }
    sem_wait(&Consumed);
    if (Consumer_rc) {
      fprintf(stderr, "execute_filter_output: %s\n", ka9q_hip_last_error());
      return 2;
    }
  }
  pthread_join(th, NULL);
  /* block 0 carries the start-up transient, from block 1 on unity passband gain and a group delay of
   * floor(M_dec/2) = 544 output samples */
  bad += near(Power[0], 0.432622, 2e-5, "block 0 mean power");
  bad += near(cabsf(First[0]), 0.0, 1e-6, "block 0 y[0]");
  bad += near(Power[1], 1.0, 2e-5, "block 1 mean power");
  bad += near(crealf(First[1]), -0.499974, 2e-5, "block 1 Re y[0]");
  bad += near(cimagf(First[1]), -0.866029, 2e-5, "block 1 Im y[0]");
  bad += near(crealf(First[2]), -0.5, 2e-5, "block 2 Re y[0]");
  bad += near(cimagf(First[2]), -0.866025, 2e-5, "block 2 Im y[0]");
  bad += near(Power[3], 1.0, 2e-5, "block 3 mean power");
  bad += Fin->blocknum != NBLOCKS || Fout->blocknum != NBLOCKS;
  bad += delete_filter_output(Fout) != 0; /* outputs before their master, filter.c:161-162 */
  bad += delete_filter_input(Fin) != 0;
  bad += delete_filter_output(NULL) != 0 || delete_filter_input(NULL) != 0 || execute_filter_input(NULL) != -1;
  printf(bad ? "abi_threads: %d check(s) failed\n" : "abi_threads: ok\n", bad);
  return bad ? 1 : 0;
}
