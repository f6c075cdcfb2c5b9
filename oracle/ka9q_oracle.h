/* ka9q_oracle.h -- CPU restatement of ka9q-radio's receive DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product library (ka9q-radio_amd/,
 * include/) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   osc.c, decimate.c   : PINNED -- the restatement is compared with the real reference
 *                         objects compiled from /root/reference into oracle/_ref/ (tests/test_oracle_ref.py)
 *                         and with golden vectors those objects produced (tests/golden/ref_*.json).
 *   filter.c, fm.c, linear.c, am.c, audio.c, radio.c : PARITY UNPINNED -- these sources include
 *                         <fftw3.h>, which this image lacks, and the reference holds no tests or
 *                         golden vectors.  They are anchored only on (a) the known answers recorded
 *                         in SURVEY.md 8(c) and (b) float64 numpy cross-checks of the DFT /
 *                         linear-convolution definition.
 *
 * All file:line citations are into /root/reference.
 */
#ifndef KA9Q_ORACLE_H
#define KA9Q_ORACLE_H 1

#include <complex.h>
#include <stdint.h>

#ifdef __cplusplus
#error "C only"
#endif

/* ---------------------------------------------------------------- FFT (replaces FFTW3 single precision) */
/* FFTW 3.3.x (INSTALLING.md:13 names libfftw3-dev; not vendored in /root/reference) computes the
 * unnormalised DFT  X[k] = sum_n x[n] exp(sign*2*pi*i*n*k/N), sign=-1 forward, +1 backward.
 * orc_fft is a float32 mixed-radix (2,3,4,5,7,generic) Stockham transform with double-computed twiddles. */
typedef struct orc_fft_plan orc_fft_plan;
orc_fft_plan *orc_fft_plan_create(int n);
void orc_fft_plan_destroy(orc_fft_plan *p);
/* out-of-place or in-place (in==out allowed) */
void orc_fft_exec(orc_fft_plan *p, const float complex *in, float complex *out, int sign);
/* convenience one-shot */
void orc_fft(int n, const float complex *in, float complex *out, int sign);
/* r2c: n real -> n/2+1 complex ; c2r: n/2+1 complex -> n real (unnormalised, FFTW conventions) */
void orc_fft_r2c(int n, const float *in, float complex *out);
void orc_fft_c2r(int n, const float complex *in, float *out);

/* ---------------------------------------------------------------- filter.c / filter.h */
enum orc_filtertype { ORC_NONE = 0, ORC_COMPLEX = 1, ORC_CROSS_CONJ = 2, ORC_REAL = 3 }; /* filter.h:18-23 */

struct orc_filter_in {          /* filter.h:55-68 */
  int in_type;
  unsigned ilen;                /* L */
  unsigned impulse_length;      /* M */
  float complex *fdomain;       /* N (COMPLEX) or N/2+1 (REAL) bins */
  float complex *input_buffer_c;/* N points (COMPLEX) */
  float *input_buffer_r;        /* N points (REAL) */
  float complex *input_c;       /* input_buffer_c + M-1 */
  float *input_r;               /* input_buffer_r + M-1 */
  orc_fft_plan *plan;
  unsigned blocknum;
};
struct orc_filter_out {         /* filter.h:69-82 */
  struct orc_filter_in *master;
  int out_type;
  float complex *response;      /* owned */
  float complex *f_fdomain;
  float noise_gain;
  float complex *output_buffer_c;
  float *output_buffer_r;
  float complex *output_c;      /* output_buffer_c + N_dec - olen */
  float *output_r;
  orc_fft_plan *plan;
  unsigned decimate;
  unsigned olen;
};

struct orc_filter_in *orc_create_filter_input(unsigned L, unsigned M, int in_type);          /* filter.c:57 */
struct orc_filter_out *orc_create_filter_output(struct orc_filter_in *master, float complex *response,
                                                unsigned decimate, int out_type);             /* filter.c:163 */
int orc_execute_filter_input(struct orc_filter_in *);                                         /* filter.c:212 */
int orc_execute_filter_output(struct orc_filter_out *, int rotate);                           /* filter.c:240 */
int orc_delete_filter_input(struct orc_filter_in *);                                          /* filter.c:386 */
int orc_delete_filter_output(struct orc_filter_out *);                                        /* filter.c:396 */
float orc_i0f(float z);                                                                       /* filter.c:410 */
float orc_i1f(float z);                                                                       /* filter.c:424 */
int orc_make_kaiser(float *window, unsigned M, float beta);                                   /* filter.c:481 */
int orc_window_filter(int L, int M, float complex *response, float beta);                     /* filter.c:509 */
int orc_window_rfilter(int L, int M, float complex *response, float beta);                    /* filter.c:573 */
float orc_noise_gain(struct orc_filter_out const *);                                          /* filter.c:631 */
int orc_set_filter(struct orc_filter_out *, float low, float high, float beta);               /* filter.c:660 */
/* accessors for ctypes users */
float complex *orc_filter_in_input(struct orc_filter_in *f);
float *orc_filter_in_input_r(struct orc_filter_in *f);
float complex *orc_filter_in_fdomain(struct orc_filter_in *f);
float complex *orc_filter_out_output(struct orc_filter_out *f);
float *orc_filter_out_output_r(struct orc_filter_out *f);
float complex *orc_filter_out_response(struct orc_filter_out *f);
float orc_filter_out_noise_gain(struct orc_filter_out *f);
void orc_filter_out_set_type(struct orc_filter_out *f, int t);   /* radio.c:194 writes out_type every block */

struct orc_notchfilter {        /* filter.h:99-104 */
  double complex osc_phase;
  double complex osc_step;
  float complex dcstate;
  float bw;
};
void orc_notch_init(struct orc_notchfilter *nf, double f, float bw);                          /* filter.c:707 */
void orc_notch_block(struct orc_notchfilter *nf, float complex *s, int n);                    /* filter.c:719 */

/* ---------------------------------------------------------------- osc.c / osc.h */
struct orc_osc {                /* osc.h:9-17 (mutex dropped: the oracle is single threaded) */
  double freq;
  double rate;
  double complex phasor;
  double complex phasor_step;
  double complex phasor_step_step;
  int steps;
};
struct orc_pll {                /* osc.h:19-25 */
  float samptime;
  struct orc_osc vco;
  float integrator_gain;
  float prop_gain;
  float integrator;
};
int orc_is_phasor_init(double complex x);                                                     /* osc.c:14 */
void orc_set_osc(struct orc_osc *osc, double f, double r);                                    /* osc.c:22 */
double complex orc_step_osc(struct orc_osc *osc);                                             /* osc.c:39 */
void orc_renorm_osc(struct orc_osc *osc);                                                     /* osc.c:52 */
void orc_step_osc_block(struct orc_osc *osc, double complex *out, int n);
void orc_init_pll(struct orc_pll *pll, float bw, float damping, double freq, float samprate); /* osc.c:64 */
float orc_run_pll(struct orc_pll *pll, float phase);                                          /* osc.c:91 */

/* ---------------------------------------------------------------- decimate.c / decimate.h */
struct orc_hb15_state {         /* decimate.h:4-9 */
  float coeffs[4];
  float even_samples[4];
  float odd_samples[4];
  float old_odd_samples[4];
};
void orc_hb15_block(struct orc_hb15_state *state, float *output, const float *input, int cnt); /* decimate.c:113 */
void orc_hb3_block(float *state, float *output, const float *input, int cnt);                  /* decimate.c:150 */
/* the caller pattern of hackrf.c:497-509: in-place cascade of log2(D) stages on one rail */
void orc_decimate_rail(struct orc_hb15_state *hb15, float *hb3, int log_decimate, int stage_threshold,
                       float *buf, int chunk);
void orc_hb15_init_f8(struct orc_hb15_state *s);                                               /* hackrf.c:442-451 */

/* ---------------------------------------------------------------- radio.c proc_samples / compute_n0 */
enum { ORC_IQ_PT12 = 95, ORC_IQ_PT = 97, ORC_IQ_PT8 = 98, ORC_PCM_STEREO_PT = 10 };           /* multicast.h:19-24 */
/* radio.c:136-178: wire payload (after the 24-byte header of IQ_PT has been skipped) -> complex float * gain.
 * returns sample count. */
int orc_sampcount(int pt, int size);                                                          /* radio.c:80-99 */
int orc_convert_samples(int pt, const unsigned char *dp, int size, float gain_factor, float complex *out);
/* radio.c:179-191: optional Doppler multiply, LO2 multiply, energy accumulation.  Returns block energy sum. */
float orc_mix_samples(struct orc_osc *doppler, struct orc_osc *lo2, const float complex *in,
                      float complex *out, int n);
/* radio.c:112-123 zero fill: advance both oscillators n steps */
void orc_skip_samples(struct orc_osc *doppler, struct orc_osc *lo2, int n);
/* radio.c:422-465 */
float orc_compute_n0(const float complex *fdomain, int N, float low, float high, int samprate);

/* ---------------------------------------------------------------- fm.c */
struct orc_fm_params {
  int input_samprate;           /* demod->input.samprate */
  int decimate;                 /* demod->filter.decimate */
  float headroom;               /* demod->agc.headroom (amplitude ratio) */
  float low, high;              /* demod->filter.low/high, Hz */
  int flat;                     /* demod->opt.flat */
};
struct orc_fm_state {           /* fm.c:26,32,33 */
  float complex state;
  float lastaudio;
  int squelch_open;
};
struct orc_sig {                /* radio.h:129-140 subset written by the demods */
  float bb_power, snr, foffset, pdeviation, cphase, lock_timer;
  int pll_lock;
  float output_level;           /* demod->output.level */
  float agc_gain;               /* demod->agc.gain after the block */
  float if_power, n0;
};
double orc_fm_snr(double r);                                                                   /* fm.c:190 */
/* fm.c:43-139, one block: y[olen] -> samples[olen] */
void orc_demod_fm_block(const struct orc_fm_params *p, struct orc_fm_state *st, const float complex *y,
                        int olen, float *samples, struct orc_sig *sig);

/* ---------------------------------------------------------------- linear.c */
struct orc_linear_params {
  int output_samprate;          /* demod->output.samprate */
  int pll, square, agc, env;    /* demod->opt.* */
  int channels;                 /* demod->output.channels */
  float headroom, hangtime, recovery_rate; /* demod->agc.* (hangtime in samples, rate per sample) */
  float loop_bw;                /* forced to 5 at thread start, linear.c:26 */
};
struct orc_linear_state {
  int hangcount;                /* linear.c:29 */
  int lock_count;               /* linear.c:38 */
  struct orc_pll pll;           /* linear.c:40-41 (vco.phasor defined = 1, quirk 9) */
  struct orc_osc shift;         /* demod->shift */
  float agc_gain;               /* demod->agc.gain */
  float snr;                    /* demod->sig.snr carried to the next block's lock detector */
  int pll_lock;
};
void orc_linear_init(const struct orc_linear_params *p, struct orc_linear_state *st, float agc_gain0,
                     double shift_cycles_per_sample);
/* linear.c:57-156, one block.  y is modified in place exactly as filter->output.c is (linear.c:91,145).
 * samples[] receives olen mono samples, or (stereo) is untouched and y holds the stereo frames. */
void orc_demod_linear_block(const struct orc_linear_params *p, struct orc_linear_state *st, float complex *y,
                            int olen, float *samples, struct orc_sig *sig);

/* ---------------------------------------------------------------- am.c (stale in the reference, named by north_star) */
struct orc_am_params {
  int input_samprate, decimate, output_samprate;
  float headroom, hangtime, recovery_rate;
};
struct orc_am_state {
  int hangcount;
  float recovery_rate_per_sample, prev_recovery_rate; /* am.c:31-32 */
  float DC_filter;              /* am.c:35 */
  float agc_gain;
};
void orc_am_init(const struct orc_am_params *p, struct orc_am_state *st, float agc_gain0);
void orc_demod_am_block(const struct orc_am_params *p, struct orc_am_state *st, const float complex *y,
                        int olen, float *samples, struct orc_sig *sig);                         /* am.c:42-91 */

/* ---------------------------------------------------------------- audio.c */
short orc_scaleclip(float x);                                                                  /* audio.c:22 */
void orc_scaleclip_block(const float *in, int16_t *out, int n);
struct orc_pcm_state { uint32_t timestamp; uint16_t seq; int silent; long long packets, bytes, samples; };
/* audio.c:85-135 / 32-82 with send() replaced by capture: pcm_be receives every chunk's big-endian int16
 * words (sent or not), sent[] one flag per chunk, marker[] the RTP marker bit, ts[] the RTP timestamp.
 * Returns the number of chunks. */
int orc_packetise(struct orc_pcm_state *st, const float *buffer, int size, int stereo, int16_t *pcm_be,
                  int *sent, int *marker, uint32_t *ts, int maxchunks);

/* ---------------------------------------------------------------- packet.c (AFSK-1200 + HDLC) and ax25.c (CRC) */
#define ORC_AFSK_AL 1000        /* packet.c:46 */
#define ORC_AFSK_AM 1049        /* packet.c:48 */
#define ORC_AFSK_SAMPPBIT 40    /* packet.c:52 */
#define ORC_AFSK_FRAME_MAX 1024 /* packet.c:431 */
struct orc_afsk_state {         /* the locals of decode_task, packet.c:410-435, and sp->rtp_state_out */
  struct orc_osc mark, space;
  int symphase;
  float complex mark_accum, space_accum, mark_offset_accum, space_offset_accum;
  float last_val, mid_val;
  unsigned char hdlc_frame[ORC_AFSK_FRAME_MAX];
  int frame_bit, flagsync, ones;
  uint32_t out_timestamp;
  uint16_t out_seq;
  long long out_packets, out_bytes;
};
struct orc_ax25_frame {         /* one frame that passed crc_good (packet.c:477), as handed to send() (506-509) */
  int sample;                   /* index, within the call, of the sample whose bit closed the frame */
  int bytes;                    /* frame length including the two CRC bytes */
  uint32_t timestamp;           /* RTP header fields, packet.c:498-502 */
  uint16_t seq;
  unsigned char data[ORC_AFSK_FRAME_MAX];
};
struct orc_afsk_bit {           /* trace of every finished bit, packet.c:462-463 (tests only) */
  int sample;
  float cur_val;
};
void orc_pcm_to_real(const int16_t *pcm_be, float *out, int n);                                /* packet.c:340-343 */
int orc_crc_good(const unsigned char *frame, int length);                                      /* ax25.c:140-156 */
void orc_afsk_init(struct orc_afsk_state *st);                                                 /* packet.c:410-435 */
/* packet.c:440-544 over y[0..n): returns the number of frames written to out (at most maxframes; further good frames
 * are counted in the RTP state but not stored).  bits/maxbits/nbits: optional trace. */
int orc_afsk_block(struct orc_afsk_state *st, const float complex *y, int n, int sample_base,
                   struct orc_ax25_frame *out, int maxframes, struct orc_afsk_bit *bits, int maxbits, int *nbits);
/* the same loop over given tone products (4 floats per sample) instead of y * oscillator: the sweeps' referee */
int orc_afsk_block_ms(struct orc_afsk_state *st, const float complex *y, const float *ms, int n, int sample_base,
                      struct orc_ax25_frame *out, int maxframes, struct orc_afsk_bit *bits, int maxbits, int *nbits);

/* ---------------------------------------------------------------- pl.c (CTCSS "PL" tone and DTMF correlator banks) */
#define ORC_PL_NTONES 55          /* pl.c:36-44 */
#define ORC_PL_BLOCKSIZE 12000    /* pl.c:26: 250 ms at 48 kS/s */
#define ORC_DTMF_BLOCKSIZE 2400   /* pl.c:27: 50 ms */
extern const float orc_pl_tones[ORC_PL_NTONES];
struct orc_pl_session {           /* the DSP members of pl.c's struct session, 62-83 */
  float complex pl_integrators[ORC_PL_NTONES];
  struct orc_osc pl_osc[ORC_PL_NTONES];
  float complex dtmf_low_integrators[4], dtmf_high_integrators[4];
  struct orc_osc dtmf_low_osc[4], dtmf_high_osc[4];
  int pl_audio_count, dtmf_audio_count;
};
struct orc_pl_result {            /* process_pl, pl.c:360-387 */
  float tone;                     /* Hz, 0 = none */
  float snr;                      /* dB between the two strongest correlators */
  float e0, e1;                   /* their energies */
  int index;                      /* of the strongest */
};
struct orc_dtmf_result {          /* process_dtmf, pl.c:390-439 */
  int digit;                      /* character, 0 = none */
  int low_index, high_index;
  float low_snr, high_snr;
};
void orc_pl_init(struct orc_pl_session *sp);                                                   /* pl.c:317-329 */
/* pl.c:248-276 over n network-order samples; results of every integration interval that ends inside */
void orc_pl_samples(struct orc_pl_session *sp, const int16_t *pcm_be, int n, struct orc_pl_result *pl, int maxpl,
                    int *npl, struct orc_dtmf_result *dt, int maxdt, int *ndt);

/* ---------------------------------------------------------------- front-end corrections (hackrf.c, funcube.c) */
struct orc_fe_state {             /* sdr->DC/sinphi/imbalance/clips/in_power and the file-scope gains of hackrf.c:136-139,815 */
  float complex DC;
  float sinphi, imbalance;
  float gain_i, gain_q, secphi, tanphi;
  int rotate_phase;
  int clips;
  float in_power;
};
void orc_fe_init(struct orc_fe_state *st, float imbalance0); /* gains 1,1,1,0; hackrf.c leaves imbalance at 0 */
/* rx_callback, hackrf.c:822-919: `samples` int8 I/Q pairs -> the two float rails; Offset = rotate_phase increment */
void orc_hackrf_rx_block(struct orc_fe_state *st, const int8_t *iq, int samples, int offset, float rate_factor,
                         float *out_i, float *out_q);
/* funcube.c:364-428: Blocksize int16 pairs in host order, corrected in place into network order */
void orc_funcube_block(struct orc_fe_state *st, int16_t *sampbuf, int blocksize, float rate_factor);
/* hackrf.c:511-560: decimated rails x Filter_atten -> IQ_PT (97, s16le) / IQ_PT12 (95) / IQ_PT8 (98); returns output_energy */
float orc_hackrf_pack(int pt, const float *ip, const float *qp, int chunk, float filter_atten, unsigned char *out);

/* ---------------------------------------------------------------- misc.c */
uint32_t orc_nextfastfft(uint32_t n);                                                          /* misc.c:129 */

#endif
