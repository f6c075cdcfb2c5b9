/* bank_san.c -- a C caller of the channel bank (include/ka9q_hip.h only) that walks the HOST side of kh_bank_* through its
 * less travelled paths, meant to run against a build of libka9q_hip whose host code is instrumented
 * (-Xarch_host -fsanitize=address,undefined or =thread: `make -C tests/c san`); device code is never instrumented.
 *   - the raw ring's read offset: a full ring consumed block by block, top-ups that force the move of what is left
 *     (by fewer and by more samples than are left), a refused overflow, a refused launch on an empty ring
 *   - launches queued without a sync, a channel retuned and the keep_* switches flipped in between
 *   - the streaming ring (kh_bank_stream_*) with several submits in flight, read-backs in between, close and reopen
 *   - KA9Q_HIP_ENQ_THREAD=1 (set by the caller's environment): all of the above with the back halves queued by the bank's
 *     own thread, which every other entry point has to wait for
 * Prints a checksum of all PCM read; the test compares it between runs (thread on / off, sanitizer or not).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <ka9q_hip.h>

enum { L = 8192, M = 8193, D = 32, C = 8, T = 6, OLEN = L / D };
static uint64_t Sum;
static int Fail;

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    if ((call) != 0) {                                                                               \
      fprintf(stderr, "FAIL %s:%d %s: %s\n", __FILE__, __LINE__, #call, ka9q_hip_last_error());      \
      Fail = 1;                                                                                      \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)

static void fold(const int16_t *p, size_t n) {
  for (size_t i = 0; i < n; i++) Sum = Sum * 1099511628211ull + (uint16_t)p[i];
}

static void make_signal(int16_t *raw, size_t nsamples, size_t n0) { /* s16le I/Q: three FM-ish carriers and an AM one */
  for (size_t n = 0; n < nsamples; n++) {
    double const t = (double)(n0 + n);
    double re = 0, im = 0;
    for (int k = 0; k < 4; k++) {
      double const f = (k - 1.5) * 12500.0 / 1e6, dev = 0.3 * sin(2 * M_PI * t * (300.0 + 100 * k) / 1e6) / ((300.0 + 100 * k) / 2500.0);
      double const a = (k == 3) ? 0.05 * (1 + 0.5 * sin(2 * M_PI * t * 400.0 / 1e6)) : 0.05;
      re += a * cos(2 * M_PI * f * t + (k < 3 ? dev : 0));
      im += a * sin(2 * M_PI * f * t + (k < 3 ? dev : 0));
    }
    raw[2 * n] = (int16_t)lrint(re * 20000);
    raw[2 * n + 1] = (int16_t)lrint(im * 20000);
  }
}

static int set_channels(kh_bank *b, int shift) {
  for (unsigned c = 0; c < C; c++) {
    struct kh_channel_config cc;
    memset(&cc, 0, sizeof cc);
    int const k = (int)c - 4 + shift;
    cc.rotate = (int)lrint((k + 0.5) * 12500.0 * (L + M - 1) / 1e6);
    cc.out_type = COMPLEX;
    cc.low = -5500, cc.high = 5500, cc.kaiser_beta = 3;
    cc.headroom = 0.178f;
    cc.agc_gain = 1000;
    cc.recovery_rate = 1.00002f;
    cc.hangtime = 30000;
    cc.loop_bw = 5;
    cc.channels = 1;
    if (c < C - 2)
      cc.demod = KH_DEMOD_FM;
    else
      cc.demod = KH_DEMOD_LINEAR, cc.agc = 1, cc.env = (c == C - 1);
    CHECK(kh_bank_set_channel(b, c, &cc));
  }
  return 0;
}

int main(void) {
  if (ka9q_hip_init(0) != 0) {
    fprintf(stderr, "ka9q_hip_init: %s\n", ka9q_hip_last_error());
    return 2;
  }
  struct kh_bank_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.L = L, cfg.M = M, cfg.decimate = D, cfg.n_inputs = 1, cfg.n_channels = C, cfg.max_blocks = T;
  cfg.samprate = 1000000, cfg.payload = KH_PT_IQ16LE, cfg.compute_n0 = 1, cfg.n0_low = -5500, cfg.n0_high = 5500;
  kh_bank *b = kh_bank_create(&cfg);
  if (!b) {
    fprintf(stderr, "kh_bank_create: %s\n", ka9q_hip_last_error());
    return 2;
  }
  if (set_channels(b, 0)) return 1;
  size_t const total = 40;
  int16_t *raw = malloc(sizeof(int16_t) * 2 * L * total);
  int16_t *pcm = malloc(sizeof(int16_t) * C * T * OLEN);
  struct kh_status st[T];
  make_signal(raw, L * total, 0);
  size_t w = 0; /* blocks written so far */

  /* 1. a full ring, consumed block by block; then top-ups that force the unread samples to the start of the ring */
  CHECK(kh_bank_write_input(b, 0, raw + 2 * L * w, (size_t)T * L));
  w += T;
  for (int k = 0; k < 1; k++) CHECK(kh_bank_execute(b, 1));                 /* offset 1, 5 left */
  CHECK(kh_bank_write_input(b, 0, raw + 2 * L * w, L));                     /* moved by 1 block in 5 pieces */
  w += 1;
  CHECK(kh_bank_execute(b, 1));
  CHECK(kh_bank_execute(b, 2));                                             /* offset 3, 3 left */
  CHECK(kh_bank_read_pcm_all(b, pcm, 2 * OLEN));
  fold(pcm, C * 2 * OLEN);
  CHECK(kh_bank_write_zeros(b, 0, 2 * L));                                  /* lost samples: offset 3 + 3 + 2 > 6: moved */
  if (kh_bank_write_input(b, 0, raw, 2 * L) == 0) {                         /* 5 + 2 > 6 */
    fprintf(stderr, "FAIL: a write beyond the ring was accepted\n");
    return 1;
  }
  CHECK(kh_bank_execute(b, 3));
  CHECK(kh_bank_execute(b, 2));                                             /* empty: offset back to 0 */
  CHECK(kh_bank_read_status(b, 1, st, 2));
  if (kh_bank_execute(b, 1) == 0) {
    fprintf(stderr, "FAIL: a launch on an empty ring was accepted\n");
    return 1;
  }

  /* 2. launches without a sync; a retune and the switches in between */
  for (int it = 0; it < 6; it++) {
    unsigned const nb = 1 + (unsigned)(it * 5) % T;
    CHECK(kh_bank_write_input(b, 0, raw + 2 * L * (w % (total - T)), (size_t)nb * L));
    w += nb;
    CHECK(kh_bank_execute(b, nb));
    if (it == 1 && set_channels(b, 1)) return 1;
    if (it == 2) CHECK(kh_bank_keep_baseband(b, 0));
    if (it == 3) CHECK(kh_bank_keep_spectrum(b, 0));
    if (it == 4) CHECK(kh_bank_keep_audio(b, 0));
    if (it % 2) {
      CHECK(kh_bank_read_pcm(b, (unsigned)it % C, pcm, nb * OLEN));
      fold(pcm, nb * OLEN);
    }
  }
  CHECK(kh_bank_sync(b));

  /* 3. the streaming ring */
  for (int round = 0; round < 2; round++) {
    CHECK(kh_bank_stream_open(b, round ? 2 : 3));
    for (int j = 0; j < 7; j++) {
      unsigned const nb = 1 + (unsigned)(j * 3) % T;
      size_t cap = 0;
      void *slot = kh_bank_stream_slot(b, 0, &cap);
      if (!slot || cap < (size_t)nb * L) {
        fprintf(stderr, "FAIL kh_bank_stream_slot: %s\n", ka9q_hip_last_error());
        return 1;
      }
      memcpy(slot, raw + 2 * L * ((w + j) % (total - T)), sizeof(int16_t) * 2 * L * nb);
      CHECK(kh_bank_stream_submit(b, nb));
      if (j % 3 == 2) {
        CHECK(kh_bank_read_pcm_all(b, pcm, nb * OLEN));
        fold(pcm, C * nb * OLEN);
      }
    }
    CHECK(kh_bank_stream_close(b));
  }
  /* and plain launches again on the same bank */
  CHECK(kh_bank_write_input(b, 0, raw, (size_t)T * L));
  CHECK(kh_bank_execute(b, T));
  CHECK(kh_bank_read_pcm_all(b, pcm, T * OLEN));
  fold(pcm, C * T * OLEN);
  CHECK(kh_bank_destroy(b));
  free(raw);
  free(pcm);
  printf("bank_san: ok, checksum %016llx\n", (unsigned long long)Sum);
  return Fail;
}
