/*
 * fxrx.h -- C ABI of libfxrx.so, the MI355X-native flexframe receive path.
 *
 * Two layers:
 *
 *  (1) DROP-IN NAMES.  The exact liquid-dsp entry points that gr::liquiddsp's blocks call, with
 *      ABI-compatible signatures, so that gnuradio-liquiddsp can be linked against libfxrx.so for
 *      this path.  Each declaration cites the reference line it serves.
 *
 *  (2) BATCHED API (fxrx_*).  Many independent IQ streams per call, device or host pointers,
 *      results returned as plain records.  This is the throughput path (BASELINE configs 2-5);
 *      the reference has no counterpart (one flexframesync handle per block instance,
 *      /root/reference/lib/flex_rx_impl.cc:49).
 *
 * No torch / C++ types cross this boundary.  Nothing here throws; failures are status codes
 * (or NULL handles) and fxrx_last_error().  The library needs a HIP device: every constructor
 * fails loudly (NULL + error text) when none is usable -- there is no CPU fallback.
 */
#ifndef FXRX_H
#define FXRX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* liquid's `float complex` is two packed floats; gr_complex (std::complex<float>) likewise.  A caller that wants its own
 * spelling of that type in these prototypes defines FXRX_COMPLEX_TYPE before including this header (include/liquid/liquid.h
 * does: std::complex<float> in C++, float _Complex in C); layout and calling convention are the same. */
#ifdef FXRX_COMPLEX_TYPE
typedef FXRX_COMPLEX_TYPE fx_complex;
#else
typedef struct { float re, im; } fx_complex;
#endif

/* ------------------------------------------------------------------------------------------
 * (1) drop-in names
 * ------------------------------------------------------------------------------------------ */

/* framesyncstats_s -- read at /root/reference/lib/flex_rx_impl.cc:195,198,218,232-234 */
typedef struct {
    float        evm;            /* error vector magnitude [dB] */
    float        rssi;           /* received signal strength [dB] */
    float        cfo;            /* carrier offset estimate [rad/sample] */
    fx_complex  *framesyms;      /* payload symbols after carrier recovery */
    unsigned int num_framesyms;
    unsigned int mod_scheme;     /* LIQUID_MODEM_* value */
    unsigned int mod_bps;
    unsigned int check;          /* LIQUID_CRC_*   */
    unsigned int fec0;           /* LIQUID_FEC_* (first / "inner" in the reference's naming) */
    unsigned int fec1;           /* LIQUID_FEC_* (second / "outer") */
} framesyncstats_s;

/* callback type -- /root/reference/lib/flex_rx_impl.h:48-55 */
typedef int (*framesync_callback)(unsigned char *header, int header_valid, unsigned char *payload,
                                  unsigned int payload_len, int payload_valid, framesyncstats_s stats,
                                  void *userdata);

typedef struct fxrx_sync_s *flexframesync;

/* /root/reference/lib/flex_rx_impl.cc:49 */
flexframesync flexframesync_create(framesync_callback callback, void *userdata);
/* /root/reference/lib/flex_rx_impl.cc:71 */
void flexframesync_destroy(flexframesync q);
/* /root/reference/lib/flex_rx_impl.cc:213.  Samples are queued and run through the GPU in blocks
 * (fxrx_sync_set_block); at most ONE completed frame is handed to the callback per call, which is
 * what the reference's work() loop can consume (it tests a single flag after each call, :216).
 * Buffers passed to the callback stay valid until the next call on the same handle. */
void flexframesync_execute(flexframesync q, fx_complex *x, unsigned int n);
void flexframesync_reset(flexframesync q);
/* extensions (additive).  Samples are collected in page-locked buffers of `block` samples (default 2^20; FXRX_SYNC_BLOCK) and
 * run through the GPU as consecutive blocks of one continuing stream, up to FXRX_SYNC_DEPTH (default 3) of them in flight while
 * the next buffer fills: flexframesync_execute never waits for the GPU unless all of those are still busy.  Frames therefore
 * reach the callback in order, one per call, between one and `depth` + 1 blocks after their last sample was handed in.
 * fxrx_sync_flush: run whatever is queued now and wait for everything in flight (frames are then pending: n = 0 calls
 * deliver them).  fxrx_sync_set_block: flushes, then changes the block length. */
void fxrx_sync_flush(flexframesync q);
void fxrx_sync_set_block(flexframesync q, unsigned int samples);
void fxrx_sync_set_threshold(flexframesync q, float threshold);
void fxrx_sync_set_equalizer(flexframesync q, int on);   /* re-creates the context like fxrx_sync_set_threshold */
void fxrx_sync_set_soft(flexframesync q, int on);        /* likewise: soft-decision payload decoding */
unsigned int fxrx_sync_pending(flexframesync q);     /* completed frames not yet delivered */
/* the liquid signatures return void: when a block fails on the GPU its samples (and those of the blocks in flight with it) are
 * dropped -- never fed twice --, the synchroniser restarts freshly reset behind the gap, the text stays in fxrx_last_error(),
 * one line goes to stderr and this counter goes up */
unsigned int fxrx_sync_errors(flexframesync q);
struct fxrx_ctx_s;
struct fxrx_ctx_s *fxrx_sync_context(flexframesync q);   /* the batched context underneath (tests, statistics) */

/* m-sequence -- /root/reference/lib/frame_detector_cc_impl.cc:47,49,50,52 */
typedef struct fxrx_mseq_s *msequence;
msequence    msequence_create(unsigned int m, unsigned int g, unsigned int a);
unsigned int msequence_advance(msequence ms);
void         msequence_destroy(msequence ms);

/* detector -- /root/reference/lib/frame_detector_cc_impl.cc:54,55,63,77,90-93 */
#define LIQUID_FIRFILT_ARKAISER 7
typedef struct fxrx_qdet_s *qdetector_cccf;
/* Only the flexframe preamble configuration the reference uses is accepted:
 * 64 symbols, ARKAISER, k=2, m=7, beta=0.3; anything else returns NULL. */
qdetector_cccf qdetector_cccf_create_linear(fx_complex *sequence, unsigned int sequence_len, int ftype,
                                            unsigned int k, unsigned int m, float beta);
void   qdetector_cccf_destroy(qdetector_cccf q);
void   qdetector_cccf_set_threshold(qdetector_cccf q, float threshold);
/* Per-sample entry kept for link compatibility.  Samples are queued and searched on the GPU in
 * blocks; every detection is reported exactly once (non-NULL = the 512 aligned samples), possibly
 * some samples later than liquid would.  Block users should call fxrx_detect_* instead. */
void  *qdetector_cccf_execute(qdetector_cccf q, fx_complex x);
float  qdetector_cccf_get_tau(qdetector_cccf q);
float  qdetector_cccf_get_gamma(qdetector_cccf q);
float  qdetector_cccf_get_dphi(qdetector_cccf q);
float  qdetector_cccf_get_phi(qdetector_cccf q);
unsigned int qdetector_cccf_get_buf_len(qdetector_cccf q);
unsigned int fxrx_qdet_errors(qdetector_cccf q);     /* as fxrx_sync_errors */

/* frame generator (test / loopback source) -- /root/reference/lib/flex_tx_impl.cc:51,56,72,188,198-201 */
typedef struct { unsigned int check, fec0, fec1, mod_scheme; } flexframegenprops_s;
typedef struct fxrx_gen_s *flexframegen;
int          flexframegenprops_init_default(flexframegenprops_s *props);
flexframegen flexframegen_create(flexframegenprops_s *props);
void         flexframegen_destroy(flexframegen q);
int          flexframegen_setprops(flexframegen q, flexframegenprops_s *props);
int          flexframegen_assemble(flexframegen q, const unsigned char *header, const unsigned char *payload,
                                   unsigned int payload_len);
unsigned int flexframegen_getframelen(flexframegen q);
int          flexframegen_write_samples(flexframegen q, fx_complex *buffer, unsigned int buffer_len);
/* extension: design the pulse with a fractional-sample delay (channel emulation in tests/bench) */
void         fxrx_gen_set_delay(flexframegen q, float dt);

/* ------------------------------------------------------------------------------------------
 * (2) batched API
 * ------------------------------------------------------------------------------------------ */
enum { FXRX_MODE_FLEX_RX = 0, FXRX_MODE_DETECTOR = 1 };
enum { FXRX_OK = 0, FXRX_ERR_ARG = -1, FXRX_ERR_HIP = -2, FXRX_ERR_NODEVICE = -3, FXRX_ERR_STATE = -4 };

typedef struct {
    int          device;         /* HIP device ordinal */
    int          mode;           /* FXRX_MODE_* */
    unsigned int n_streams;
    float        threshold;      /* 0 -> default (0.5 flex_rx like liquid's flexframesync, 0.45 detector like
                                    /root/reference/lib/frame_detector_cc_impl.cc:55) */
    unsigned int segment_len;    /* speculation granularity in samples (0 -> auto) */
    int          want_framesyms; /* copy payload symbols back to the host with each result */
    int          equalizer;      /* 1: optional equaliser stage on (liquid: FLEXFRAMESYNC_ENABLE_EQ, compiled out of a stock libliquid):
                                    13-tap eqlms at 2 samples/symbol behind the matched filter, trained on the 64 p/n symbols, frozen
                                    afterwards; symbol instants move 3 symbols later.  0 (default): what flexframesync executes */
    int          soft_decision;  /* 1: decode the payload from per-bit soft values (liquid: flexframesync_decode_payload_soft, which the
                                    reference never calls): soft-input Viterbi for the convolutional stage(s) nearest the channel.
                                    With want_framesyms the soft values themselves come back too (fxrx_frame.soft_bits).  0: hard */
} fxrx_config;

typedef struct {
    unsigned int stream;
    int64_t      start;          /* absolute sample index (since create/reset) of aligned sample 0 */
    int          cfo_bin;        /* coarse CFO bin of the detector sweep */
    float        rxy, tau, gamma, dphi, phi;
    unsigned int pfb_index;
    float        pilot_dphi, pilot_phi, pilot_gain;
    int          header_valid, payload_valid;
    unsigned char header[20];    /* 14 user bytes + 6 protocol bytes */
    const unsigned char *payload; unsigned int payload_len;
    const fx_complex *framesyms; unsigned int num_framesyms;   /* host pointer or NULL */
    float        evm_db, rssi_db, cfo, evm_sum;
    unsigned int mod_scheme, mod_bps, check, fec0, fec1;
    const unsigned char *soft_bits; unsigned int num_soft_bits;   /* soft_decision + want_framesyms: one byte per coded bit in channel
                                                                     order, 0 = surely 0 ... 255 = surely 1; else NULL */
} fxrx_frame;

typedef struct fxrx_ctx_s fxrx_ctx;

const char *fxrx_last_error(void);
const char *fxrx_version(void);
int         fxrx_device_count(void);
/* page-locked host memory for input buffers: uploads from it run asynchronously (NULL + fxrx_last_error() on failure) */
void       *fxrx_pinned_alloc(size_t bytes);
void        fxrx_pinned_free(void *p);

fxrx_ctx *fxrx_create(const fxrx_config *cfg);
void      fxrx_destroy(fxrx_ctx *c);
void      fxrx_reset(fxrx_ctx *c);

/* Feed n_samples[s] new samples of every stream s.  iq[s] points at interleaved float32 (re,im);
 * on_device != 0 means the pointers are HIP device pointers on cfg.device (zero copy when the
 * stream has no carried-over tail).  Returns the number of results (frames, or detections in
 * detector mode) now readable through fxrx_result(), or a negative FXRX_ERR_*.  Results and
 * the buffers they point to stay valid until the next call on this context. */
int fxrx_process(fxrx_ctx *c, const void *const *iq, const uint64_t *n_samples, int on_device);
int fxrx_result(const fxrx_ctx *c, unsigned int i, fxrx_frame *out);

/* Pipelined form of fxrx_process (= submit + collect).  fxrx_submit enqueues the block's whole kernel chain on the block's
 * own HIP stream -- walkers, seek verification, chain (stitch / repair / resume state), plan, payload MF, PLL, packet decode,
 * results into pinned host memory -- and returns without waiting for any of it; what a stream carries from block to block
 * (resume state, unconsumed tail) stays on the device, so a block that continues the streams of the previous one simply
 * orders its first walkers behind that block's chain kernel.  fxrx_collect waits for the OLDEST submitted block and exposes
 * its results through fxrx_result / fxrx_last_timing.  With depth d (fxrx_set_depth, 1..16, default 1) up to d blocks may
 * be in flight.  After fxrx_reset the next block starts from a freshly reset synchroniser.  Blocks of one stream must be
 * submitted in order; input buffers, host or device, must stay valid until their block has been collected; results stay
 * valid until the next fxrx_collect / fxrx_process on the context (a submit in between does not touch them).  Returns 0
 * (submit) / result count (collect) or FXRX_ERR_*. */
int fxrx_set_depth(fxrx_ctx *c, unsigned int depth);
int fxrx_submit(fxrx_ctx *c, const void *const *iq, const uint64_t *n_samples, int on_device);
int fxrx_collect(fxrx_ctx *c);
/* Failure semantics.  A failing fxrx_submit (bad arguments, a batch too large for the arenas, an allocation that fails) leaves
 * the context exactly as it was: the same block, or another, can be submitted next and continues the streams from where they
 * stood.  A failing fxrx_collect drops every block in flight (their samples are never searched) and restarts all streams
 * from a freshly reset synchroniser with the next block submitted; sample positions keep counting.  Either way the context
 * stays usable -- unless the HIP runtime itself reports errors, which every later call will report again. */
/* 1: the oldest block in flight has finished (fxrx_collect will not wait), 0: not yet / nothing in flight, < 0: FXRX_ERR_* */
int fxrx_ready(const fxrx_ctx *c);
unsigned int fxrx_inflight(const fxrx_ctx *c);
/* tests: make the next `submits` calls of fxrx_submit / `collects` calls of fxrx_collect fail (FXRX_ERR_STATE) after they have
 * done their bookkeeping, to exercise the paths above */
int fxrx_debug_fail(fxrx_ctx *c, unsigned int submits, unsigned int collects);
/* diagnostic builds (-DFX_STAMPS) only: shader-clock deltas of the decode phases of payload job i */
int fxrx_debug_stamps(const fxrx_ctx *c, unsigned int i, uint32_t out[8]);
int fxrx_debug_chain_stamps(const fxrx_ctx *c, uint32_t out[8]);  /* chain kernel phase clocks (stream 0) of the last collected block */
int fxrx_debug_walk_stamps(const fxrx_ctx *c, uint64_t out[4]);   /* summed walker phase clocks: coarse, seek, align, header */
int fxrx_debug_walk_maxjob(const fxrx_ctx *c, uint64_t out[8]);   /* slowest walk job: 4 phase clocks, hops, coarse hops, frames, total */
int fxrx_debug_walk_jobs(const fxrx_ctx *c, uint32_t *out, unsigned int cap_jobs);   /* per walk job 6 words: 4 stamps, hops, frames; returns the job count */
/* device-resident payload symbols / hard decisions of the last call (NULL if none) */
const void *fxrx_device_framesyms(const fxrx_ctx *c, uint64_t *n_symbols);

typedef struct {
    double   walk_ms, paymf_ms, paypll_ms, paydec_ms, total_ms;   /* HIP-event times of the block's kernels, on the stream they ran on */
    uint64_t hops, walk_jobs, repairs, frames, payload_symbols, samples, hops_cheap;
    double   host_submit_ms, host_walkwait_ms;   /* wall time inside fxrx_submit (descriptor build + enqueue) / always 0: the host no longer waits for the walkers */
    double   seekverify_ms;                      /* fx_seekverify_kernel (full detector over the hops the walkers skipped) */
    uint64_t verify_hops, verify_failures;       /* hops re-checked / runs on which the exact detector fired (those spans are walked again, exactly) */
    double   host_collectwait_ms;                /* wall time fxrx_collect waited for the block's results */
    uint64_t walk_mode;                          /* 0: all streams started freshly reset, 1: some continued the previous block (true walkers ordered behind its chain kernel) */
    double   chain_ms;                           /* fx_chain_kernel + fx_plan_kernel */
    uint64_t replays;                            /* carry-buffer overflows / chain repairs handled so far (blocks behind were enqueued again) */
    uint64_t vb_blocks, vb_repairs;              /* batch Viterbi: trellis blocks (incl. padding slots) / blocks run again because a hand-over check failed */
    uint64_t late_decodes;                       /* blocks whose decode launches had to be completed at collect (more frames than the grids were sized for) */
    uint64_t vb_fallbacks;                       /* batch Viterbi: frames handed back to the wave-per-frame decoder (a hand-over that stayed unverified) */
} fxrx_timing;
int fxrx_last_timing(const fxrx_ctx *c, fxrx_timing *t);
/* Stage times come from HIP events recorded between the kernels of a block, and every event is one more packet in the block's
 * queue -- with blocks in flight the queues' packet rate is what limits throughput.  level 2: all stages;
 * 1: the PLL only (paypll_ms; what bench.py quotes its roofline line for); 0: none (the *_ms fields read 0); -1 (default):
 * 2 with one block at a time, 0 with depth > 1.  Counters are always there.  Environment: FXRX_TIMING. */
int fxrx_set_timing(fxrx_ctx *c, int level);
/* diagnostics: of the last collected block, ms since a common origin: host time at fxrx_submit, GPU time of its first and of its
 * last event, host time when fxrx_collect returned it (needs timing level 2 from the first submit on) */
int fxrx_debug_block_times(const fxrx_ctx *c, double out[4]);
/* the HIP stream (hipStream_t as void*) of the first slot of the ring of blocks in flight; every block runs its whole kernel
 * chain on its slot's stream.  To order external work against a block, use fxrx_collect (it returns when the block's
 * results are on the host). */
void *fxrx_stream(const fxrx_ctx *c);

/* batched frame generator: one frame -> samples (host) */
unsigned int fxrx_gen_frame_len(unsigned int mod_scheme, unsigned int check, unsigned int fec0, unsigned int fec1,
                                unsigned int payload_len);

/* block-API index maps of the reference (lib/flex_tx_impl.cc:75-181, lib/flex_rx_impl.cc:74-179); -1 = unsupported */
int fxrx_mod_from_index(int idx);   int fxrx_mod_to_index(unsigned int mod_scheme);
int fxrx_inner_from_index(int idx); int fxrx_inner_to_index(unsigned int fec);
int fxrx_outer_from_index(int idx); int fxrx_outer_to_index(unsigned int fec);

/* ------------------------------------------------------------------------------------------
 * (3) batched frame generator on the GPU -- the flex_tx counterpart
 *
 * What /root/reference/lib/flex_tx_impl.cc:191-209 (send_pkt) does per PDU -- flexframegen_assemble (:200) and
 * flexframegen_write_samples (:203-205) with the properties of :51-56 / :183-189 -- for many frames in one call,
 * straight into a device buffer.  Packet encoding of header and payload (CRC, whitening, both code stages,
 * interleavers, bit packing), modulation and pulse shaping run on the GPU; the host lays out descriptors.  Samples are
 * bit-identical to flexframegen_write_samples.
 * ------------------------------------------------------------------------------------------ */
typedef struct fxtx_ctx_s fxtx_ctx;
typedef struct {
    flexframegenprops_s props;          /* check, fec0, fec1, mod_scheme of this frame */
    const unsigned char *header;        /* 14 user bytes, or NULL for zeros */
    const unsigned char *payload;       /* payload_len bytes (host memory) */
    unsigned int        payload_len;    /* 0 .. 65535 */
    float               dt;             /* fractional-sample delay the pulse is designed with (0 = none) */
    unsigned long long  out_offset;     /* first sample of the frame in the output buffer */
} fxtx_frame;
fxtx_ctx    *fxtx_create(int device);                        /* NULL + fxrx_last_error() without a usable GPU */
void         fxtx_destroy(fxtx_ctx *c);
unsigned int fxtx_frame_len(const fxtx_frame *f);            /* samples the frame occupies */
/* Writes every frame's samples to out_device[out_offset ...] (interleaved float32 re,im; out_len samples long).  Frames
 * must not overlap; samples between frames are left untouched (zero the buffer first).  Returns 0 or FXRX_ERR_*;
 * synchronous (the frames are in the buffer when the call returns). */
int          fxtx_generate(fxtx_ctx *c, const fxtx_frame *frames, unsigned int n_frames, void *out_device,
                           unsigned long long out_len);
/* The synthetic channel of the test / bench source on the device, so that hundreds of distinct streams
 * never exist on the host: stream s = samples [s n_per_stream, (s+1) n_per_stream) of iq_device becomes
 * gain x[n] exp(j (phase + n cfo)) + noise, noise white Gaussian with standard deviation sigma per real dimension, drawn from
 * a counter-based generator keyed by `seed` (every sample is a function of (seed, n) alone).  n_per_stream must be even.
 * Synchronous; returns 0 or FXRX_ERR_*. */
typedef struct { float cfo, phase, gain, sigma; unsigned long long seed; } fxtx_channel;
int          fxtx_apply_channel(fxtx_ctx *c, void *iq_device, unsigned int n_streams, unsigned long long n_per_stream,
                          const fxtx_channel *ch);

#ifdef __cplusplus
}
#endif
#endif
