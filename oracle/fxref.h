/*
 * fxref.h -- CPU ORACLE for the flexframe receive path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link,
 * import or call anything under oracle/.  The product (gr-liquiddsp_amd/) never does.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in liquid-dsp (un-vendored, un-pinned:
 * /root/reference/lib/CMakeLists.txt:33, /root/reference/docs/where_is_liquid.txt:1), which is
 * absent from the build container, and the reference's own tests hold no golden vectors
 * (/root/reference/python/qa_flex_rx.py:34-37, /root/reference/lib/qa_liquiddsp.cc:30-36).
 * This file restates the published liquid-dsp v1.3.x algorithms behind the reference's call
 * sites; agreement with a real libliquid is unverified.
 *
 * Reference call sites restated here:
 *   flexframesync_create/execute/destroy   /root/reference/lib/flex_rx_impl.cc:49,213,71
 *   framesync callback + framesyncstats_s  /root/reference/lib/flex_rx_impl.cc:181-201, flex_rx_impl.h:48-55
 *   qdetector_cccf_create_linear/set_threshold/execute/destroy
 *                                          /root/reference/lib/frame_detector_cc_impl.cc:54,55,77,63
 *   msequence_create/advance/destroy       /root/reference/lib/frame_detector_cc_impl.cc:47-52
 *   flexframegen_* (test signal source)    /root/reference/lib/flex_tx_impl.cc:51-56,188,198-201
 *
 * CANONICAL ARITHMETIC (shared *specification*, two independent implementations: this C file set
 * and the HIP kernels).  All floating point is IEEE binary32, round-to-nearest-even, compiled
 * with -ffp-contract=off; a fused multiply-add happens only where fmaf() is written.  No libm
 * transcendental is applied to path data: sin/cos come from fxr_sincos_u32 (1024-entry table +
 * 3rd-order correction on a 32-bit phase), arg() from fxr_atan2 (Cephes-style polynomial).
 * Reductions use a fixed order: "tree" = balanced pairwise tree over the index, "seq" = ascending.
 */
#ifndef FXREF_H
#define FXREF_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } fxr_c32;

/* ---- numeric enums as stored in the frame header [RECALLED liquid.h v1.3.x values] ---- */
enum { FXR_CRC_UNKNOWN = 0, FXR_CRC_NONE, FXR_CRC_CHECKSUM, FXR_CRC_8, FXR_CRC_16, FXR_CRC_24, FXR_CRC_32 };
enum {
    FXR_FEC_UNKNOWN = 0, FXR_FEC_NONE = 1, FXR_FEC_REP3 = 2, FXR_FEC_REP5 = 3, FXR_FEC_HAMMING74 = 4,
    FXR_FEC_HAMMING84 = 5, FXR_FEC_HAMMING128 = 6, FXR_FEC_GOLAY2412 = 7, FXR_FEC_SECDED2216 = 8,
    FXR_FEC_SECDED3932 = 9, FXR_FEC_SECDED7264 = 10, FXR_FEC_CONV_V27 = 11, FXR_FEC_CONV_V29 = 12,
    FXR_FEC_CONV_V39 = 13, FXR_FEC_CONV_V615 = 14, FXR_FEC_CONV_V27P23 = 15, FXR_FEC_CONV_V27P34 = 16,
    FXR_FEC_CONV_V27P45 = 17, FXR_FEC_CONV_V27P56 = 18, FXR_FEC_CONV_V27P67 = 19, FXR_FEC_CONV_V27P78 = 20,
    FXR_FEC_RS_M8 = 27
};
enum {
    FXR_MODEM_UNKNOWN = 0, FXR_MODEM_PSK2 = 1, FXR_MODEM_PSK4 = 2, FXR_MODEM_PSK8 = 3, FXR_MODEM_PSK16 = 4,
    FXR_MODEM_DPSK2 = 9, FXR_MODEM_DPSK4 = 10, FXR_MODEM_DPSK8 = 11, FXR_MODEM_ASK4 = 18,
    FXR_MODEM_QAM16 = 27, FXR_MODEM_QAM32 = 28, FXR_MODEM_QAM64 = 29, FXR_MODEM_QPSK = 40
};

/* fixed frame geometry (reference: lib/frame_detector_cc_impl.h:34-36; liquid flexframe) */
#define FXR_K            2      /* samples / symbol */
#define FXR_M            7      /* filter semi-length, symbols */
#define FXR_BETA         0.3f
#define FXR_NPFB         32     /* matched-filter polyphase branches */
#define FXR_MF_TAPS      28     /* taps per branch: (2*NPFB*K*M+1)/NPFB */
#define FXR_PN_LEN       64
#define FXR_S_LEN        156    /* template samples: K*(64+2M) */
#define FXR_NFFT         512
#define FXR_RANGE        24     /* (int)(0.3*512/2pi) CFO bins each side */
#define FXR_HDR_USER     14     /* lib/flex_tx_impl.cc:58 */
#define FXR_HDR_DEC      20
#define FXR_HDR_ENC      54     /* CRC32 + SECDED7264 + HAMMING84 */
#define FXR_HDR_MOD      216    /* QPSK data symbols */
#define FXR_HDR_PILOTS   15
#define FXR_HDR_SYM      231
#define FXR_PILOT_SPACING 16
#define FXR_PROTOCOL     102    /* [RECALLED] FLEXFRAME_PROTOCOL = 101 + PACKETIZER_VERSION(1) */
#define FXR_PRE_DELAY    (2*FXR_M)            /* symbols before first p/n symbol leaves the MF */
#define FXR_SYM0_HDR     (FXR_PRE_DELAY + FXR_PN_LEN)          /* 78  */
#define FXR_SYM0_PAY     (FXR_SYM0_HDR + FXR_HDR_SYM)          /* 309 */
#define FXR_EQ_TAPS      13     /* optional equaliser: 2*k*p+1 taps, p = 3 */
#define FXR_EQ_DELAY     3      /* its delay in symbols */
#define FXR_EQ_MU        0.05f

/* ------------------------------------------------------------------ math (fxref_math.c) */
void     fxr_init(void);                        /* builds all shared tables once (idempotent) */
uint32_t fxr_rad2u32(float rad);                /* rintf(rad * 2^32/2pi) wrapped mod 2^32 */
uint32_t fxr_phase_inc(float units);            /* rintf(units) clamped below 2^31: PLL increments in phase units */
void     fxr_sincos_u32(uint32_t th, float *c, float *s);
float    fxr_phase_step(float units);            /* rintf(units) clamped below 2^31: the PLL's phase advance per symbol */
void     fxr_sincos_small(float step_units, float *c, float *s);   /* 5th-order series: cos/sin of that advance */
float    fxr_atan2(float y, float x);
float    fxr_sum_tree(const float *v, unsigned n);          /* n power of two */
fxr_c32  fxr_csum_tree(const fxr_c32 *v, unsigned n);       /* n power of two */
void     fxr_fft512(const fxr_c32 *in, fxr_c32 *out);       /* forward, unnormalised */
void     fxr_ifft512(const fxr_c32 *in, fxr_c32 *out);      /* inverse, unnormalised (swap trick) */
const fxr_c32 *fxr_twiddle512(void);            /* W^m = exp(-j 2 pi m/512), m=0..511 */
const fxr_c32 *fxr_sincos_table(void);          /* 1024 x (cos,sin) */

/* ------------------------------------------------------------------ filters / sequences */
typedef struct { unsigned m, g, a, n, v; } fxr_mseq;
void     fxr_mseq_init(fxr_mseq *q, unsigned m, unsigned g, unsigned a);
unsigned fxr_mseq_advance(fxr_mseq *q);
unsigned fxr_mseq_symbol(fxr_mseq *q, unsigned bps);
void     fxr_firdes_arkaiser(unsigned k, unsigned m, float beta, float dt, float *h /* 2km+1 */);
const float   *fxr_mf_proto(void);              /* 897 prototype taps (K=64) */
const float   *fxr_tx_taps(void);               /* 29 interpolator taps (dt = 0) */
const fxr_c32 *fxr_preamble_pn(void);           /* 64 */
const fxr_c32 *fxr_template(void);              /* 156 */
const fxr_c32 *fxr_template_fft(void);          /* 512 */
float          fxr_template_energy(void);       /* sum |s|^2, seq order */
const fxr_c32 *fxr_pilots(void);                /* 15 */
void           fxr_eq_init_taps(float *h /* FXR_EQ_TAPS */);    /* equaliser start: Kaiser low-pass, fc = 0.4, As = 40 dB, x 2 fc */

/* ------------------------------------------------------------------ FEC (fxref_fec.c) */
unsigned fxr_crc_len(int check);
uint32_t fxr_crc_key(int check, const uint8_t *msg, unsigned n);
void     fxr_scramble(uint8_t *x, unsigned n);
void     fxr_interleave(uint8_t *x, unsigned n, int decode);       /* in place, depth 4 */
unsigned fxr_fec_enc_len(int fs, unsigned dec_len);
void     fxr_fec_encode(int fs, unsigned dec_len, const uint8_t *dec, uint8_t *enc);
void     fxr_fec_decode(int fs, unsigned dec_len, const uint8_t *enc, uint8_t *dec);
unsigned fxr_packet_enc_len(unsigned n, int check, int fec0, int fec1);
void     fxr_packet_encode(unsigned n, int check, int fec0, int fec1, const uint8_t *msg, uint8_t *pkt);
int      fxr_packet_decode(unsigned n, int check, int fec0, int fec1, const uint8_t *pkt, uint8_t *msg);
int      fxr_packet_decode_soft(unsigned n, int check, int fec0, int fec1, uint8_t *soft /* 8 * enc_len, clobbered */, uint8_t *msg);
int      fxr_fec_supported(int fs);

/* ------------------------------------------------------------------ modem (fxref_modem.c) */
unsigned fxr_modem_bps(int ms);                 /* 0 = unsupported */
typedef struct { int ms; unsigned bps; float dpsk_phi; } fxr_modem;
void     fxr_modem_init(fxr_modem *q, int ms);
fxr_c32  fxr_modem_mod(fxr_modem *q, unsigned sym);
/* hard demod: returns symbol, writes remodulated point xhat and phase error imag(r conj(xhat)) */
unsigned fxr_modem_demod(fxr_modem *q, fxr_c32 r, fxr_c32 *xhat, float *phase_err);
/* soft decisions of one (carrier-recovered) symbol: bps bytes, MSB of the symbol first; 0 = surely 0 ... 255 = surely 1 */
void     fxr_modem_demod_soft(int ms, fxr_c32 r, unsigned hard_sym, uint8_t *soft);
unsigned fxr_qpm_sym_len(unsigned n, int check, int fec0, int fec1, int ms);

/* ------------------------------------------------------------------ frame generator */
typedef struct { int check, fec0, fec1, mod_scheme; } fxr_genprops;
/* returns number of samples written (= fxr_gen_frame_len) */
unsigned fxr_gen_frame_len(const fxr_genprops *p, unsigned payload_len);
unsigned fxr_gen_frame(const fxr_genprops *p, const uint8_t header[FXR_HDR_USER],
                       const uint8_t *payload, unsigned payload_len, float dt, fxr_c32 *out);

/* ------------------------------------------------------------------ detector (qdetector_cccf) */
typedef struct fxr_qdet fxr_qdet;
fxr_qdet *fxr_qdet_create_flexframe(void);      /* == create_linear(pn,64,ARKAISER,2,7,0.3) */
void      fxr_qdet_destroy(fxr_qdet *q);
void      fxr_qdet_reset(fxr_qdet *q);
void      fxr_qdet_set_threshold(fxr_qdet *q, float t);
/* per-sample: NULL or pointer to the 512 aligned samples (lib/frame_detector_cc_impl.cc:77) */
const fxr_c32 *fxr_qdet_execute(fxr_qdet *q, fxr_c32 x);
float     fxr_qdet_tau(const fxr_qdet *q);
float     fxr_qdet_gamma(const fxr_qdet *q);
float     fxr_qdet_dphi(const fxr_qdet *q);
float     fxr_qdet_phi(const fxr_qdet *q);
float     fxr_qdet_rxy(const fxr_qdet *q);
int       fxr_qdet_offset(const fxr_qdet *q);
uint64_t  fxr_qdet_num_hops(const fxr_qdet *q);
/* the per-sample loop of frame_detector_cc_impl::work (lib/frame_detector_cc_impl.cc:76-83):
 * feeds x[0..n) one sample at a time, records every non-NULL return.  pos = base + index of the
 * aligned window's sample 0.  Returns the number of detections (stored up to max). */
typedef struct { int64_t pos; float tau, gamma, dphi, phi, rxy; int offset; } fxr_detection;
unsigned  fxr_qdet_run(fxr_qdet *q, const fxr_c32 *x, uint64_t n, int64_t base, fxr_detection *out, unsigned max);

/* ------------------------------------------------------------------ frame synchroniser */
typedef struct {
    float evm, rssi, cfo;
    fxr_c32 *framesyms; unsigned num_framesyms;
    unsigned mod_scheme, mod_bps, check, fec0, fec1;
} fxr_stats;                                    /* mirrors framesyncstats_s */
typedef int (*fxr_callback)(unsigned char *header, int header_valid, unsigned char *payload,
                            unsigned payload_len, int payload_valid, fxr_stats stats, void *userdata);
typedef struct fxr_sync fxr_sync;
fxr_sync *fxr_sync_create(fxr_callback cb, void *userdata);
void      fxr_sync_destroy(fxr_sync *q);
void      fxr_sync_reset(fxr_sync *q);
void      fxr_sync_execute(fxr_sync *q, const fxr_c32 *x, unsigned n);
void      fxr_sync_execute_chunked(fxr_sync *q, const fxr_c32 *x, uint64_t n, unsigned chunk);   /* n samples in calls of `chunk` */
void      fxr_sync_set_threshold(fxr_sync *q, float t);
/* optional equaliser stage (liquid: FLEXFRAMESYNC_ENABLE_EQ, compiled out by default): 13-tap eqlms at 2 samples/symbol
 * behind the matched filter, trained on the 64 p/n symbols, frozen afterwards; every symbol instant moves 3 symbols later */
void      fxr_sync_set_equalizer(fxr_sync *q, int on);
/* soft-decision payload decoding (liquid: flexframesync_decode_payload_soft; off by default, as in the reference's use):
 * per-bit soft values from the carrier-recovered symbols, soft-input Viterbi */
void      fxr_sync_set_soft(fxr_sync *q, int on);
/* the soft values of the most recent frame decoded in soft mode (8 * coded bytes, channel order), for parity tests */
const uint8_t *fxr_sync_last_soft(const fxr_sync *q, unsigned *n);
/* introspection used by parity tests: estimates of the most recent frame */
typedef struct {
    uint64_t start;     /* absolute index (since create/reset_counters) of aligned sample 0 */
    int offset; float rxy, tau, gamma, dphi, phi; unsigned pfb_index; int mf_counter0;
    float pilot_dphi, pilot_phi, pilot_gain; float evm_sum;
} fxr_frameinfo;
void      fxr_sync_last_frame(const fxr_sync *q, fxr_frameinfo *fi);

#ifdef __cplusplus
}
#endif
#endif
