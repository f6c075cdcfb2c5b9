/* ka9q_hip.h -- C ABI of libka9q_hip: ka9q-radio's receive DSP hot path on one MI355X (gfx950).
 *
 * Part 1 is the drop-in boundary: the same names, argument meaning, struct field order and error
 * behaviour as the reference's filter.h (filter.h:18-108), so that `radio`, `packet` and `modulate`
 * link against libka9q_hip.so instead of filter.o.  All arithmetic runs in hand-written HIP kernels;
 * there is no CPU fallback: every entry point fails (NULL / -1 and a message in ka9q_hip_last_error())
 * when no gfx950 device is usable.
 *
 * Part 2 is the channel-bank extension the reference lacks (its `rotate` argument is "for eventual
 * use in a demodulator channel bank", filter.c:272): one shared forward FFT per block, C channels,
 * fused demodulators with device-resident state, many blocks per launch.
 *
 * All file:line citations are into the reference tree.
 */
#ifndef KA9Q_HIP_H
#define KA9Q_HIP_H 1

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>

#ifdef __cplusplus
typedef struct kh_cf32 { float re, im; } kh_cf32;   /* layout of C's `complex float` */
typedef struct kh_cf64 { double re, im; } kh_cf64;  /* layout of C's `complex double` */
extern "C" {
#else
#include <complex.h>
typedef float _Complex kh_cf32;
typedef double _Complex kh_cf64;
#endif

/* ------------------------------------------------------------------ library */
/* replaces the global FFTW hooks of main.c:124-127,310 (fftwf_init_threads, ...).  Optional: every
 * create_* call initialises lazily on device 0.  Returns 0, or -1 if there is no gfx950 device. */
int ka9q_hip_init(int device);
const char *ka9q_hip_last_error(void);
/* replaces fftwf_alloc_complex / fftwf_free for arrays whose ownership passes to a filter
 * (response of create_filter_output, filter.c:42,183,403; modulate.c:121).  Pinned host memory. */
void *ka9q_hip_alloc(size_t bytes);
void ka9q_hip_free(void *p);

/* ------------------------------------------------------------------ Part 1: filter.h */
enum filtertype { NONE, COMPLEX, CROSS_CONJ, REAL }; /* filter.h:18-23 */

union rc { /* filter.h:26-29 */
  float *r;
  kh_cf32 *c;
};

/* filter.h:55-68.  Field order and types are ABI: callers write input.c[] sample by sample
 * (radio.c:119,191; packet.c:343), read ilen / impulse_length / fdomain[] (radio.c:124,427-446).
 * fwd_plan (an fftwf_plan in the reference) holds this library's private handle.
 * Host arrays are pinned; fdomain[] is coherent when execute_filter_input returns. */
struct filter_in {
  enum filtertype in_type;
  unsigned int ilen;           /* L */
  unsigned int impulse_length; /* M */
  kh_cf32 *fdomain;            /* N bins (COMPLEX) or N/2+1 (REAL) */
  union rc input_buffer;       /* N points */
  union rc input;              /* input_buffer + M-1, L points */
  void *fwd_plan;
  unsigned int blocknum;
  pthread_mutex_t filter_mutex;
  pthread_cond_t filter_cond;
  int fd;
};

/* filter.h:69-82.  out_type is written by the producer every block (radio.c:126,194); output.c[] is
 * read (fm.c:56,86) and written (linear.c:91,145) by the demodulators; noise_gain is read by
 * radio_status.c:173.  rev_plan holds this library's private handle. */
struct filter_out {
  struct filter_in *master;
  enum filtertype out_type;
  kh_cf32 *response;
  pthread_mutex_t response_mutex;
  kh_cf32 *f_fdomain;
  float noise_gain;
  union rc output_buffer;      /* N/decimate points */
  union rc output;             /* output_buffer + N_dec - olen */
  void *rev_plan;
  unsigned int decimate;
  unsigned int olen;
  unsigned int blocknum;
};

/* filter.c:57.  NULL on failure (the reference only asserts).  N = L+M-1 must be a power of two in
 * [256, 2^20], or an even number up to 8192 whose only prime factors are 2, 3, 5, 7 -- the sizes the
 * reference's nextfastfft (misc.c:129-148) hands to main.c:326-330, e.g. 7680 at 192 kS/s (the
 * reference itself accepts any FFTW size). */
struct filter_in *create_filter_input(unsigned int L, unsigned int M, enum filtertype in_type);
/* filter.c:163.  response may be NULL (set later with set_filter); if not NULL it must come from
 * ka9q_hip_alloc and becomes filter-owned.  NULL if master is NULL.  N/decimate must be a power of
 * two in [256, 8192] or an even 2,3,5,7-smooth number up to 8192 (1920 for the default above). */
struct filter_out *create_filter_output(struct filter_in *master, kh_cf32 *response, unsigned int decimate,
                                        enum filtertype out_type);
/* filter.c:212: forward transform of the N-point buffer, blocknum++, wake the outputs, move the
 * last M-1 samples to the head.  0, or -1 on NULL. */
int execute_filter_input(struct filter_in *);
/* filter.c:240: block until a new spectrum exists, select/rotate bins, multiply by the response,
 * CROSS_CONJ fold, decimating inverse transform.  0, or -1 on NULL. */
int execute_filter_output(struct filter_out *, int rotate);
int delete_filter_input(struct filter_in *);   /* filter.c:386; NULL -> 0 */
int delete_filter_output(struct filter_out *); /* filter.c:396; NULL -> 0 */
int make_kaiser(float *window, unsigned int M, float beta);           /* filter.c:481 */
int window_filter(int L, int M, kh_cf32 *response, float beta);       /* filter.c:509 */
int window_rfilter(int L, int M, kh_cf32 *response, float beta);      /* filter.c:573 */
int set_filter(struct filter_out *, float low, float high, float kaiser_beta); /* filter.c:660 */
float noise_gain(struct filter_out const *);                          /* filter.c:631 */

/* filter.c:96-160 (experimental, no live caller: main.c:334 is commented out): as create_filter_input, but
 * fdomain[] is a shared mapping of `file` (created/truncated, N or N/2+1 bins), so that other processes can
 * read every block's spectrum; master->fd holds the descriptor.  delete_filter_input unmaps and closes it
 * (the reference leaks both). */
struct filter_in *create_filter_input_file(unsigned int L, unsigned int M, enum filtertype in_type, char *file);

/* filter.h:97-108, filter.c:707-727: experimental one-pole complex notch (no caller anywhere in the reference).
 * The state is in the caller-visible struct, as there. */
struct notchfilter {
  kh_cf64 osc_phase; /* phase of the local complex mixer */
  kh_cf64 osc_step;  /* mixer phase increment (frequency) */
  kh_cf32 dcstate;   /* average signal at the mixer frequency */
  float bw;          /* relative bandwidth of the notch */
};
struct notchfilter *notch_create(double f, float bw); /* filter.c:707; f in cycles/sample */
#define notch_delete(x) free(x)                       /* filter.h:107 */
/* filter.c:719, one sample per call as the reference declares it (NaN for a NULL filter).  Each call is a
 * host-to-device round trip: use kh_notch_block for a stream. */
kh_cf32 notch(struct notchfilter *, kh_cf32 s);
/* the same recurrence over n samples in place (buf: host or device memory); 0 or -1.  Not in the reference. */
int kh_notch_block(struct notchfilter *, kh_cf32 *buf, int n);

/* osc.h:32-39 (set_osc, step_osc, renorm_osc, is_phasor_init, init_pll, run_pll) is NOT exported: those are
 * per-sample host calls on caller-owned structs, and a receiver built on this library keeps the reference's
 * osc.o for them.  Inside the library the same oscillators run in closed form on the
 * device (kh_bank_set_lo / kh_bank_set_doppler, the PLL of kh_bank's linear demodulator). */

/* ------------------------------------------------------------------ decimate.h */
struct hb15_state { /* decimate.h:4-9 */
  float coeffs[4];
  float even_samples[4];
  float odd_samples[4];
  float old_odd_samples[4];
};
/* decimate.h:10-11 with host pointers (staged through the device); in-place (output == input) allowed. */
void hb15_block(struct hb15_state *state, float *output, float *input, int cnt);
void hb3_block(float *state, float *output, float *input, int cnt);
/* The call pattern of hackrf.c:495-509 / airspy.c:508-522 as one launch: `rails` planar float rails
 * of chunk<<log_decimate samples each (host pointers, rail r at buf + r*rail_stride), decimated in
 * place by 2^log_decimate; stages j >= stage_threshold use hb3, the rest hb15; hb15[r*log_decimate+j],
 * hb3[r*log_decimate+j] carry stage j of rail r.  0 or -1. */
int kh_decimate_rails(struct hb15_state *hb15, float *hb3, int log_decimate, int stage_threshold, float *buf,
                      size_t rail_stride, int rails, int chunk);

/* ------------------------------------------------------------------ Part 2: channel bank */
typedef struct kh_bank kh_bank;

enum kh_demod { KH_DEMOD_NONE = 0, KH_DEMOD_FM = 1, KH_DEMOD_LINEAR = 2, KH_DEMOD_AM = 3 };
/* wire formats of radio.c:80-99,136-178 plus the internal cf32 */
enum kh_payload { KH_PT_CF32 = 0, KH_PT_IQ12 = 95, KH_PT_IQ16LE = 97, KH_PT_IQ8 = 98, KH_PT_PCM16BE = 10 };

struct kh_bank_config {
  unsigned int L, M;        /* block and impulse length at the input rate (main.c:326-330) */
  unsigned int decimate;    /* bank-wide D; N_dec = (L+M-1)/D */
  unsigned int n_inputs;    /* independent input streams (1 for a wideband bank; P for P pipelines) */
  unsigned int n_channels;  /* filter_out + demod instances */
  unsigned int max_blocks;  /* blocks per execute call (T) */
  int samprate;             /* demod->input.samprate */
  int payload;              /* enum kh_payload of the raw input ring */
  int pcm_big_endian;       /* 1: PCM words in network order as audio.c:50,101 sends them */
  int compute_n0;           /* 1: run compute_n0 (radio.c:422) per input and block */
  float n0_low, n0_high;    /* demod->filter.low/high (Hz) used by compute_n0 */
};

struct kh_channel_config {
  unsigned int input;       /* which input stream */
  int rotate;               /* execute_filter_output's rotate (filter.c:240) */
  int out_type;             /* COMPLEX or CROSS_CONJ (demod->filter.isb, radio.c:194) */
  int demod;                /* enum kh_demod */
  float low, high;          /* demod->filter.low/high in Hz (set_filter gets low/out_rate, radio.c:408-411) */
  float kaiser_beta;
  /* FM (fm.c) */
  int flat;                 /* opt.flat */
  float headroom;           /* agc.headroom, amplitude ratio (all demods) */
  /* linear (linear.c) / AM (am.c) */
  int pll, square, agc, env;/* opt.* */
  int channels;             /* output.channels: 1 mono, 2 stereo (linear only) */
  float hangtime;           /* agc.hangtime, samples */
  float recovery_rate;      /* agc.recovery_rate, amplitude ratio per sample (am.c: see quirk in DESIGN.md) */
  float agc_gain;           /* initial agc.gain (main.c:139: +80 dB) */
  float loop_bw;            /* forced to 5 Hz by linear.c:26 */
  double shift;             /* post-detection shift in cycles per output sample (radio.c:355-361) */
};

/* one record per channel per block: the demod->sig / output / agc fields the demod loops write
 * (fm.c:53-70,116-121,138; linear.c:71-77,98-108,149,156; am.c:90) */
struct kh_status {
  float bb_power, snr, foffset, pdeviation, cphase, lock_timer;
  int pll_lock;
  float output_level, agc_gain;
  int squelch_open;
  float reserved[2];
};
/* one record per input per block (radio.c:197-202) */
struct kh_input_status {
  float if_power, n0;
};

kh_bank *kh_bank_create(const struct kh_bank_config *cfg);
int kh_bank_destroy(kh_bank *);
int kh_bank_set_channel(kh_bank *, unsigned int ch, const struct kh_channel_config *);
/* set_second_LO / set_doppler semantics (radio.c:233-240,342-352): f in cycles/sample, rate in
 * cycles/sample^2, phase kept (osc.c:24-27).  lo2_freq is exactly what radio.c:350 hands to set_osc
 * (second_LO/samprate, NOT negated: tune.second_LO already carries set_freq's sign, radio.c:267,289);
 * doppler_freq / doppler_rate are what radio.c:238 hands to set_osc, i.e. the caller negates both. */
int kh_bank_set_lo(kh_bank *, unsigned int input, double lo2_freq, double doppler_freq, double doppler_rate);
/* the two halves of kh_bank_set_lo on their own: set_second_LO touches only the LO2 oscillator (radio.c:350),
 * set_doppler only the Doppler one and restarts its sweep from doppler_freq (radio.c:238, osc.c:28-35) */
int kh_bank_set_second_lo(kh_bank *, unsigned int input, double lo2_freq);
int kh_bank_set_doppler(kh_bank *, unsigned int input, double doppler_freq, double doppler_rate);
/* proc_samples' per-packet work (radio.c:132-178) as a bulk upload: append `count` samples of wire
 * format cfg.payload to input's raw ring.  `payload` may be a host or a device pointer. */
int kh_bank_write_input(kh_bank *, unsigned int input, const void *payload, size_t count);
/* lost samples (radio.c:112-131): append `count` zero samples; the NCOs keep stepping */
int kh_bank_write_zeros(kh_bank *, unsigned int input, size_t count);
int kh_bank_set_gain(kh_bank *, unsigned int input, float gain_factor);
/* device address and capacity (in samples) of input's raw ring, for producers that are already on
 * the GPU.  After filling nblocks*L samples call kh_bank_commit_input. */
int kh_bank_raw_device_ptr(kh_bank *, unsigned int input, void **dptr, size_t *capacity_samples);
int kh_bank_commit_input(kh_bank *, unsigned int input, size_t count);
/* Streaming ingest: proc_samples' hand-over (radio.c:132-191) when the samples arrive in host memory.  The library
 * owns a pinned host ring of `depth` slots (each max_blocks*L samples of the wire format, per input).  The producer
 * asks for the current slot of each input (kh_bank_stream_slot: blocks only if the transfer that last left that slot,
 * depth submits ago, is still running), writes nblocks*L samples there -- e.g. recvfrom() straight into it -- and calls
 * kh_bank_stream_submit: a hipMemcpyAsync on a copy stream of its own into one of two device raw rings, and
 * kh_bank_execute(nblocks) ordered behind it.  Submit returns at once; the transfer of launch k+1 overlaps launch k.
 * Do not mix with kh_bank_write_input / kh_bank_commit_input on the same bank while the stream is open. */
int kh_bank_stream_open(kh_bank *, unsigned int depth);   /* 2..64 */
void *kh_bank_stream_slot(kh_bank *, unsigned int input, size_t *capacity_samples);
int kh_bank_stream_submit(kh_bank *, unsigned int nblocks);
int kh_bank_stream_close(kh_bank *);                      /* waits for everything in flight; NULL -> 0 */
/* one pass of the hot path over nblocks (<= max_blocks) blocks of every input: convert + mix (a7),
 * forward FFT (a2), compute_n0 (a8), per channel bin-select x response x IFFT (a4), demod (a9-a12),
 * scaleclip (a14).  Asynchronous on the bank's streams; kh_bank_sync waits.  A caller may run ahead of the GPU by three
 * launches: the call blocks until the launch three before this one has finished (its spectrum slot is the one this
 * launch overwrites).  0 or -1. */
int kh_bank_execute(kh_bank *, unsigned int nblocks);
int kh_bank_sync(kh_bank *);
/* results of the last execute (host pointers; these calls synchronise) */
int kh_bank_read_baseband(kh_bank *, unsigned int ch, kh_cf32 *out, size_t nsamples);       /* nblocks*olen */
int kh_bank_read_pcm(kh_bank *, unsigned int ch, int16_t *out, size_t nwords);               /* nblocks*olen*channels */
int kh_bank_read_audio(kh_bank *, unsigned int ch, float *out, size_t nwords);               /* float before scaleclip */
/* the PCM of ALL channels in one transfer: channel ch at out + ch*nwords */
int kh_bank_read_pcm_all(kh_bank *, int16_t *out, size_t nwords);
int kh_bank_read_status(kh_bank *, unsigned int ch, struct kh_status *out, size_t nblocks);
int kh_bank_read_input_status(kh_bank *, unsigned int input, struct kh_input_status *out, size_t nblocks);
int kh_bank_read_spectrum(kh_bank *, unsigned int input, unsigned int block, kh_cf32 *out);  /* N bins */
int kh_bank_read_response(kh_bank *, unsigned int ch, kh_cf32 *out, float *noise_gain);      /* N_dec bins */
/* device-resident views for consumers on the GPU */
int kh_bank_pcm_device_ptr(kh_bank *, void **dptr, size_t *words_per_channel);
/* 1 (default): also keep the float audio that scaleclip consumed (kh_bank_read_audio); 0: PCM only */
int kh_bank_keep_audio(kh_bank *, int enable);
/* 1 (default): FM channels also store their complex baseband (kh_bank_read_baseband); 0: they go from the filter
 * output straight to the discriminator inside one kernel and only PCM + status leave it.  Channels with another
 * demodulator (or none) always store it. */
int kh_bank_keep_baseband(kh_bank *, int enable);
/* 1 (default): every launch leaves the whole N-bin spectrum of every block (kh_bank_read_spectrum; filter.c:217's fdomain).
 * 0: four-step transforms (N > 8192) store only the rows of N1 bins that the channels' filter outputs read
 * (filter.c:274-335) -- at BASELINE configs[1] a tenth of the spectrum -- and compute_n0 (radio.c:422-465) is finished from
 * the transform's registers and scratch, so PCM, status and input status are what they are with 1; kh_bank_read_spectrum
 * then fails.  No counterpart in the reference, whose single filter_out reads one master spectrum. */
int kh_bank_keep_spectrum(kh_bank *, int enable);
/* 1: the two halves of kh_bank_execute's kernel sequence are captured into hipGraphs the second time a launch shape
 * (nblocks, buffer parity) is seen and replayed from then on.  0 (default): plain launches -- with four to six
 * kernels per launch the replay costs the host more than the launches do (measured, tools/hostbound.py). */
int kh_bank_use_graph(kh_bank *, int enable);
/* PCM packetisation plan (audio.c:32-135): send_mono_output / send_stereo_output cut every block's PCM into chunks
 * of at most 480 16-bit words, skip all-zero chunks (the RTP timestamp still advances), set the marker bit on the
 * first packet after silence and number the packets that are sent.  With the plan enabled the bank leaves one record
 * per chunk, so the host only prepends the RTP header and calls send(). */
struct kh_rtp_chunk {
  uint32_t timestamp;  /* rtp.timestamp of the chunk (audio.c:54,105) */
  uint16_t seq;        /* rtp.seq, valid when send != 0 (audio.c:64,116) */
  uint8_t marker;      /* 1: first packet after silence (audio.c:59-63,111-115) */
  uint8_t send;        /* 0: all-zero chunk, nothing is sent (audio.c:53,77-78) */
  uint32_t offset;     /* first word of the chunk in kh_bank_read_pcm's array */
  uint16_t nwords;     /* 16-bit words in the chunk (<= 480) */
  uint16_t reserved;
};
struct kh_rtp_state {   /* demod->output.rtp.{timestamp,seq,packets,bytes}, output.silent, output.samples */
  uint32_t timestamp;
  uint16_t seq;
  uint16_t silent;
  uint64_t packets, bytes, samples;
};
/* The demodulator state of one channel -- what the reference keeps in its demod thread's locals and in struct demod
 * between blocks (fm.c:26-33 state/lastaudio/squelch; linear.c:29-44 AGC gain, hang count, PLL integrator, VCO phasor
 * and frequency, lock count, shift oscillator; am.c:26-35) -- as an opaque record of kh_bank_channel_state_size() bytes,
 * valid between launches.  A channel moves to another bank (another GPU's process fed by the same stream) by
 * kh_bank_set_channel there with the same parameters, then kh_bank_set_channel_state: its next block continues where the
 * last one here ended.  Both synchronise.  The RTP counters travel separately (kh_bank_rtp_set_state). */
size_t kh_bank_channel_state_size(void);
int kh_bank_get_channel_state(kh_bank *, unsigned int ch, void *out, size_t bytes);
int kh_bank_set_channel_state(kh_bank *, unsigned int ch, const void *in, size_t bytes);
int kh_bank_rtp_enable(kh_bank *, int enable);                                   /* default 0 */
int kh_bank_rtp_set_state(kh_bank *, unsigned int ch, const struct kh_rtp_state *);
/* chunks of the last execute for channel ch, in order; *nchunks = nblocks * ceil(olen*channels/480) (0 for a channel
 * without demodulator); *after = state after the last chunk.  Synchronises. */
int kh_bank_read_rtp(kh_bank *, unsigned int ch, struct kh_rtp_chunk *out, size_t max_chunks, size_t *nchunks,
                     struct kh_rtp_state *after);
/* ---- packet.c's receive side: AFSK-1200 demodulator + HDLC + CRC for many PCM sessions at once.
 * The reference gives every RTP session (ssrc) a REAL filter_in (L = 1000, M = 1049 -> N = 2048), a COMPLEX
 * filter_out (decimate 1, 100 Hz .. 4 kHz, beta 3) and a decode thread (packet.c:315-326, 402-408); this object is
 * that, `sessions` times over, with all sessions advanced together by kh_afsk_execute.  Input is what `packet`
 * takes from the network: 48 kS/s mono PCM in network byte order (packet.c:312,340-343), e.g. straight from
 * kh_bank_pcm_device_ptr of a bank whose FM channels run at 48 kS/s. */
#define KH_AFSK_L 1000   /* packet.c:46 */
#define KH_AFSK_M 1049   /* packet.c:48 */
#define KH_AX25_MAX 1024 /* packet.c:431: hdlc_frame[1024]; the reference has no bound check, longer frames are dropped here */
typedef struct kh_afsk kh_afsk;
struct kh_ax25_frame {   /* a frame that passed crc_good (packet.c:477), as packet.c:494-509 hands it to send() */
  uint64_t sample;       /* samples consumed by the session up to and including the one that closed the frame */
  uint32_t timestamp;    /* RTP header fields of the AX25_PT packet, packet.c:498-502 */
  uint16_t seq;
  uint16_t bytes;        /* length including the two CRC bytes */
  uint8_t data[KH_AX25_MAX];
};
struct kh_afsk_bit {     /* trace of every finished bit (packet.c:462-463); for tests */
  uint64_t sample;
  float cur_val;
  uint32_t reserved;
};
struct kh_afsk_status {  /* sp->rtp_state_out and the HDLC state after the last launch */
  uint32_t timestamp;
  uint16_t seq;
  uint16_t flagsync;
  uint64_t packets, bytes;
  uint32_t frames_last_launch; /* good frames of the last launch (may exceed what max_frames kept) */
  uint32_t bits_last_launch;
};
kh_afsk *kh_afsk_create(unsigned int sessions, unsigned int max_blocks, unsigned int max_frames, unsigned int max_bits);
int kh_afsk_destroy(kh_afsk *); /* NULL -> 0 */
/* nwords network-order int16 words appended to the session's pending input (host memory) */
int kh_afsk_write_pcm(kh_afsk *, unsigned int session, const int16_t *pcm_be, size_t nwords);
/* device-resident producers: row `session` of the pending-input array starts at dptr + session*stride_words */
int kh_afsk_pcm_device_ptr(kh_afsk *, void **dptr, size_t *stride_words);
int kh_afsk_commit_pcm(kh_afsk *, unsigned int session, size_t nwords);
/* consumes nblocks*KH_AFSK_L words of every session (all must have them); asynchronous like kh_bank_execute */
int kh_afsk_execute(kh_afsk *, unsigned int nblocks);
int kh_afsk_sync(kh_afsk *);
int kh_afsk_read_frames(kh_afsk *, unsigned int session, struct kh_ax25_frame *out, size_t max, size_t *n);
int kh_afsk_read_bits(kh_afsk *, unsigned int session, struct kh_afsk_bit *out, size_t max, size_t *n);
int kh_afsk_read_status(kh_afsk *, unsigned int session, struct kh_afsk_status *out);
int kh_afsk_read_baseband(kh_afsk *, unsigned int session, kh_cf32 *out, size_t nsamples);
/* diagnostic (no counterpart in packet.c): the tone products of the last launch, 4 floats per sample {Re, Im of y x mark,
 * Re, Im of y x space} -- what the decoder's integrate-and-dump accumulators (packet.c:442-451) add up */
int kh_afsk_read_products(kh_afsk *, unsigned int session, float *out, size_t nsamples); /* analytic signal, last launch */

/* ---- pl.c: CTCSS ("PL") tone and DTMF decoders for many 48 kS/s mono PCM sessions.
 * The reference steps 55 + 8 tone oscillators per audio sample per session and looks at the integrators every 250 ms
 * (PL, pl.c:26,255,360-387) and 50 ms (DTMF, pl.c:27,270,390-439).  A launch takes a whole number of 50 ms intervals
 * (KH_DTMF_BLOCK samples) from every session; PL intervals may straddle launches. */
#define KH_PL_NTONES 55    /* pl.c:36-46 */
#define KH_PL_BLOCK 12000  /* pl.c:26 */
#define KH_DTMF_BLOCK 2400 /* pl.c:27 */
typedef struct kh_tones kh_tones;
struct kh_pl_result {   /* process_pl, pl.c:360-387 */
  float tone;           /* Hz; 0: the strongest correlator is less than 6 dB above the next */
  float snr;            /* dB between the two */
  float e0, e1;         /* their energies */
  int32_t index;        /* of the strongest, into PL_tones (pl.c:36-44) */
};
struct kh_dtmf_result { /* process_dtmf, pl.c:390-439 */
  int32_t digit;        /* character; 0: none */
  int32_t low_index, high_index; /* -1: below 10 (pl.c:408,426) */
  float low_snr, high_snr;
};
kh_tones *kh_tones_create(unsigned int sessions, unsigned int max_windows /* 50 ms intervals per launch */);
int kh_tones_destroy(kh_tones *); /* NULL -> 0 */
int kh_tones_write_pcm(kh_tones *, unsigned int session, const int16_t *pcm_be, size_t nwords);
int kh_tones_pcm_device_ptr(kh_tones *, void **dptr, size_t *stride_words);
int kh_tones_commit_pcm(kh_tones *, unsigned int session, size_t nwords);
int kh_tones_execute(kh_tones *, unsigned int nwindows); /* takes nwindows*KH_DTMF_BLOCK words of every session */
int kh_tones_sync(kh_tones *);
int kh_tones_read_pl(kh_tones *, unsigned int session, struct kh_pl_result *out, size_t max, size_t *n);
int kh_tones_read_dtmf(kh_tones *, unsigned int session, struct kh_dtmf_result *out, size_t max, size_t *n);

/* ---- SDR front-end corrections in front of decimate.c.  Pointers may be host or device memory
 * (device pointers skip the copies, so correct -> kh_decimate_rails -> pack can stay on the GPU). */
struct kh_fe_state {   /* sdr->DC, sinphi, imbalance, clips, in_power and the gains of hackrf.c:136-139,815 / funcube.c:331-334 */
  float dc_re, dc_im;
  float sinphi, imbalance;  /* hackrf.c leaves imbalance at 0 on start (zeroed global); funcube.c likewise */
  float gain_i, gain_q, secphi, tanphi; /* start at 1, 1, 1, 0 */
  int32_t rotate_phase;
  int32_t clips;
  float in_power;
};
/* rx_callback, hackrf.c:822-919, for nblocks consecutive transfers of `samples` int8 I/Q pairs: clip at -1, DC removal,
 * gain balance, phase correction, Fs/4 rotation (offset = Offset, hackrf.c:77) into the two float rails
 * (Sampbuffer_i/q); the estimates are updated after every transfer as the reference does (906-917).
 * rate_factor = 1/(ADC_samprate*Power_alpha), hackrf.c:277. */
int kh_hackrf_correct(struct kh_fe_state *state, const int8_t *iq, unsigned int samples, unsigned int nblocks, int offset,
                      float rate_factor, float *out_i, float *out_q);
/* funcube.c:364-428 for nblocks blocks of `blocksize` int16 pairs (host order in, network order out, in place).
 * rate_factor = Blocksize/(ADC_samprate*Power_alpha), funcube.c:340. */
int kh_funcube_correct(struct kh_fe_state *state, int16_t *sampbuf, unsigned int blocksize, unsigned int nblocks,
                       float rate_factor);
/* hackrf.c:511-573: chunk decimated samples x filter_atten -> payload bytes (KH_PT_IQ16LE: 4/sample, KH_PT_IQ12: 3,
 * KH_PT_IQ8: 2); *output_energy receives the sum the reference divides into sdr->out_power (566-573).
 * Out-of-range values saturate (the reference's float -> short conversion is undefined there: "Clip?"). */
int kh_hackrf_pack(int payload, const float *ip, const float *qp, unsigned int chunk, float filter_atten, void *out,
                   float *output_energy);

/* NO COUNTERPART IN THE REFERENCE.  kh_hackrf_pack for callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and same results as
 * kh_hackrf_pack, but: ip, qp, out and output_energy (may be NULL) must be memory a kernel on the library's device can
 * address -- that device's own memory, managed memory, or mapped pinned host memory (another GPU's allocation and
 * pageable host memory are refused with -1, they would be a memory fault); the work is queued on `stream`
 * (a hipStream_t; NULL: the library's own stream) and NOT synchronised; the per-workgroup energy partial sums live in
 * a static 16 KiB device array that every GPU the library runs on holds until the process ends.  No hipMalloc /
 * hipFree / allocating routine is called, so ka9q_hip_init must have run before (this call cannot initialise lazily).
 * chunk > KH_PACK_NOALLOC_MAX_CHUNK is refused (-1 and a message).  All calls on one GPU share the array: calls that
 * overlap in time on one GPU (two streams, two threads) race on it -- serialise them.  output_energy is the sum of the
 * partials in ascending order on one lane (bit-equal with kh_hackrf_pack): 0.6 us at 131072 samples, ~20 us at the maximum. */
#define KH_PACK_NOALLOC_MAX_CHUNK 1048576
int kh_hackrf_pack_noalloc(int payload, const float *ip, const float *qp, unsigned int chunk, float filter_atten, void *out,
                           float *output_energy, void *stream);

/* diagnostic: one character per stream / event of the bank ('.' idle or complete, 'P' pending, 'E' error) */
int kh_bank_debug_pending(kh_bank *, char *buf, size_t n);
void *kh_bank_stream(kh_bank *); /* hipStream_t */
/* per-kernel timing with HIP events.  kh_bank_profile(bank, 1): every stage alone on the chip (both halves on the
 * bank's one stream, an event pair around every stage); 2: the normal two-stream pipeline with the same event pairs, each
 * on the stream its stage runs on (what a stage takes beside the other half's kernels; the events themselves cost the
 * step a few microseconds); 0: off.  Stages: */
enum kh_stage { KH_STAGE_INGEST = 0, KH_STAGE_FFT1, KH_STAGE_FFT2, KH_STAGE_N0, KH_STAGE_CHAN, KH_STAGE_DEMOD,
                KH_STAGE_CARRY, KH_STAGE_COUNT };
int kh_bank_profile(kh_bank *, int enable);
/* sum of elapsed ms and number of launches per stage since profiling was enabled (synchronises) */
int kh_bank_stage_times(kh_bank *, float ms[KH_STAGE_COUNT], unsigned int launches[KH_STAGE_COUNT]);
const char *kh_stage_kernel_name(int stage);

#ifdef __cplusplus
}
#endif
#endif
