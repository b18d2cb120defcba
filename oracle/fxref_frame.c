/*
 * fxref_frame.c -- CPU ORACLE (test infrastructure; see fxref.h).  PARITY UNPINNED.
 *
 * Per-sample state machines of the receive path and the matching frame generator:
 *   fxr_qdet_*   <- qdetector_cccf_{create_linear,set_threshold,execute,destroy}
 *                   (/root/reference/lib/frame_detector_cc_impl.cc:54,55,77,63)
 *   fxr_sync_*   <- flexframesync_{create,execute,destroy} + framesync callback
 *                   (/root/reference/lib/flex_rx_impl.cc:49,213,71,181-201)
 *   fxr_gen_*    <- flexframegen_{assemble,getframelen,write_samples}
 *                   (/root/reference/lib/flex_tx_impl.cc:198-201), test signal source
 * Algorithms [RECALLED liquid-dsp v1.3.x qdetector_cccf.c, flexframesync.c, flexframegen.c,
 * qpilotsync.c, qpilotgen.c, qpacketmodem.c, nco.c, firpfb.c] as laid out in SURVEY.md 3.3/3.4,
 * deliberately kept as one state-machine step per input sample so that this file doubles as the
 * timed CPU baseline driven in 256-sample calls like /root/reference/lib/flex_rx_impl.cc:212-215.
 *
 * Restatement choices (inside "parity unpinned"):
 *  - NCOs carry a 32-bit integer phase (closed form theta_n = theta_0 + n*delta mod 2^32).
 *  - Peak search compares squared magnitudes; the first maximum in (offset, lag) order wins.
 *  - Half-window energies are balanced-tree sums of the 256 samples of that half.
 *  - After an ALIGN the detector returns to SEEK keeping the second half of the aligned window
 *    as its overlap half (liquid's exact post-detection bookkeeping is not recalled).
 *  - The equaliser (eqlms_cccf) is an option, off by default as liquid's FLEXFRAMESYNC_ENABLE_EQ 0 is (fxr_sync_set_equalizer):
 *    13 taps at 2 samples/symbol behind the matched filter (which is then evaluated at both sample phases), started as a
 *    Kaiser low-pass, trained by normalised LMS (mu = 0.05) on the 64 p/n symbols, frozen for header and payload; all symbol
 *    instants move FXR_EQ_DELAY = 3 symbols later [RECALLED flexframesync.c: delay = 2*m + 3].  Sums over the 13 taps run
 *    in balanced-tree order over 16 slots (what 16 lanes of a wavefront do).
 */
#include "fxref.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* a * conj(b) */
static inline fxr_c32 cmulc(fxr_c32 a, fxr_c32 b)
{
    float t = a.im * b.im, u = a.re * b.im;
    fxr_c32 r = { fmaf(a.re, b.re, t), fmaf(a.im, b.re, -u) };
    return r;
}
static inline float cm2(fxr_c32 a) { return fmaf(a.re, a.re, a.im * a.im); }
/* x * exp(-j theta) */
static inline fxr_c32 derot(fxr_c32 x, uint32_t th)
{
    float c, s; fxr_sincos_u32(th, &c, &s);
    fxr_c32 r = { fmaf(x.re, c, x.im * s), fmaf(x.im, c, -(x.re * s)) };
    return r;
}
static float half_energy(const fxr_c32 *x)
{
    float e[256];
    for (int i = 0; i < 256; i++) e[i] = cm2(x[i]);
    return fxr_sum_tree(e, 256);
}

/* =================================================================== frame generator */
static void pack_symbols(const uint8_t *enc, unsigned enc_len, unsigned bps, unsigned nsym, uint8_t *sym)
{
    unsigned nbits = 8 * enc_len;
    for (unsigned j = 0; j < nsym; j++) {
        unsigned s = 0;
        for (unsigned b = 0; b < bps; b++) {
            unsigned k = j * bps + b;
            unsigned bit = k < nbits ? (enc[k >> 3] >> (7 - (k & 7))) & 1u : 0u;
            s = (s << 1) | bit;
        }
        sym[j] = (uint8_t)s;
    }
}

unsigned fxr_gen_frame_len(const fxr_genprops *p, unsigned payload_len)
{
    unsigned n = fxr_qpm_sym_len(payload_len, p->check, p->fec0, p->fec1, p->mod_scheme);
    return FXR_K * (FXR_PN_LEN + FXR_HDR_SYM + n + 2 * FXR_M);
}

unsigned fxr_gen_frame(const fxr_genprops *p, const uint8_t header[FXR_HDR_USER],
                       const uint8_t *payload, unsigned payload_len, float dt, fxr_c32 *out)
{
    fxr_init();
    unsigned npay = fxr_qpm_sym_len(payload_len, p->check, p->fec0, p->fec1, p->mod_scheme);
    unsigned nsym = FXR_PN_LEN + FXR_HDR_SYM + npay + 2 * FXR_M;
    fxr_c32 *x = (fxr_c32 *)calloc(nsym, sizeof(fxr_c32));

    memcpy(x, fxr_preamble_pn(), FXR_PN_LEN * sizeof(fxr_c32));

    /* header: 14 user bytes + 6 protocol bytes, CRC32 / SECDED7264 / HAMMING84, QPSK, pilots */
    uint8_t hd[FXR_HDR_DEC], he[FXR_HDR_ENC], hs[FXR_HDR_MOD];
    memcpy(hd, header, FXR_HDR_USER);
    hd[14] = FXR_PROTOCOL;
    hd[15] = (uint8_t)((payload_len >> 8) & 0xff);
    hd[16] = (uint8_t)(payload_len & 0xff);
    hd[17] = (uint8_t)p->mod_scheme;
    hd[18] = (uint8_t)(((p->check & 0x07) << 5) | (p->fec0 & 0x1f));
    hd[19] = (uint8_t)(p->fec1 & 0x1f);
    fxr_packet_encode(FXR_HDR_DEC, FXR_CRC_32, FXR_FEC_SECDED7264, FXR_FEC_HAMMING84, hd, he);
    pack_symbols(he, FXR_HDR_ENC, 2, FXR_HDR_MOD, hs);
    fxr_modem qm; fxr_modem_init(&qm, FXR_MODEM_QPSK);
    const fxr_c32 *pil = fxr_pilots();
    for (unsigned i = 0, n = 0, pp = 0; i < FXR_HDR_SYM; i++)
        x[FXR_PN_LEN + i] = (i % FXR_PILOT_SPACING) == 0 ? pil[pp++] : fxr_modem_mod(&qm, hs[n++]);

    /* payload */
    unsigned el = fxr_packet_enc_len(payload_len, p->check, p->fec0, p->fec1);
    uint8_t *pe = (uint8_t *)malloc(el + 1), *ps = (uint8_t *)malloc(npay + 1);
    fxr_packet_encode(payload_len, p->check, p->fec0, p->fec1, payload, pe);
    pack_symbols(pe, el, fxr_modem_bps(p->mod_scheme), npay, ps);
    fxr_modem pm; fxr_modem_init(&pm, p->mod_scheme);
    for (unsigned i = 0; i < npay; i++) x[FXR_PN_LEN + FXR_HDR_SYM + i] = fxr_modem_mod(&pm, ps[i]);
    free(pe); free(ps);

    /* interpolate by k=2: y[2n+i] = sum_t h[i+2t] x[n-t] */
    float hdt[2 * FXR_K * FXR_M + 1];
    const float *h = fxr_tx_taps();
    if (dt != 0.0f) { fxr_firdes_arkaiser(FXR_K, FXR_M, FXR_BETA, dt, hdt); h = hdt; }
    for (unsigned n = 0; n < nsym; n++)
        for (unsigned i = 0; i < FXR_K; i++) {
            float ar = 0, ai = 0;
            for (unsigned t = 0; t < 15; t++) {
                unsigned hi = i + FXR_K * t;
                if (hi > 2 * FXR_K * FXR_M || t > n) continue;
                ar = fmaf(h[hi], x[n - t].re, ar);
                ai = fmaf(h[hi], x[n - t].im, ai);
            }
            out[FXR_K * n + i].re = ar; out[FXR_K * n + i].im = ai;
        }
    free(x);
    return FXR_K * nsym;
}

/* =================================================================== detector */
enum { QD_SEEK = 0, QD_ALIGN = 1 };
struct fxr_qdet {
    fxr_c32 buf0[FXR_NFFT], buf1[FXR_NFFT];
    unsigned counter; int state; float threshold;
    float x2_0, x2_1;
    int offset; float rxy, tau, gamma, dphi, phi;
    uint64_t hops;
};

fxr_qdet *fxr_qdet_create_flexframe(void)
{
    fxr_init();
    fxr_qdet *q = (fxr_qdet *)calloc(1, sizeof *q);
    q->threshold = 0.5f;
    fxr_qdet_reset(q);
    return q;
}
void fxr_qdet_destroy(fxr_qdet *q) { free(q); }
void fxr_qdet_reset(fxr_qdet *q)
{
    memset(q->buf0, 0, sizeof q->buf0);
    q->counter = FXR_NFFT / 2; q->state = QD_SEEK; q->x2_0 = q->x2_1 = 0.0f;
}
void fxr_qdet_set_threshold(fxr_qdet *q, float t) { q->threshold = t; }
float fxr_qdet_tau(const fxr_qdet *q) { return q->tau; }
float fxr_qdet_gamma(const fxr_qdet *q) { return q->gamma; }
float fxr_qdet_dphi(const fxr_qdet *q) { return q->dphi; }
float fxr_qdet_phi(const fxr_qdet *q) { return q->phi; }
float fxr_qdet_rxy(const fxr_qdet *q) { return q->rxy; }
int fxr_qdet_offset(const fxr_qdet *q) { return q->offset; }
uint64_t fxr_qdet_num_hops(const fxr_qdet *q) { return q->hops; }

/* R = IFFT( X[i] conj(S[(i-offset) mod N]) ), unnormalised */
static void xcorr_offset(const fxr_c32 *X, int offset, fxr_c32 *R)
{
    const fxr_c32 *S = fxr_template_fft();
    fxr_c32 Y[FXR_NFFT];
    for (int i = 0; i < FXR_NFFT; i++) Y[i] = cmulc(X[i], S[(i + FXR_NFFT - offset) % FXR_NFFT]);
    fxr_ifft512(Y, R);
}

/* returns 1 when the hop detected a frame already aligned to lag 0 (window complete) */
static int qdet_seek(fxr_qdet *q)
{
    fxr_c32 X[FXR_NFFT], R[FXR_NFFT];
    q->hops++;
    q->x2_1 = half_energy(q->buf0 + FXR_NFFT / 2);
    float g0 = sqrtf(q->x2_0 + q->x2_1) * sqrtf((float)FXR_S_LEN / (float)FXR_NFFT);
    if (!(g0 < 1e-10f)) {
        fxr_fft512(q->buf0, X);
        float g = 1.0f / ((float)FXR_NFFT * g0 * sqrtf(fxr_template_energy()));
        float best = -1.0f; unsigned bi = 0; int bo = 0;
        for (int off = -FXR_RANGE; off <= FXR_RANGE; off++) {
            xcorr_offset(X, off, R);
            for (unsigned i = 0; i < FXR_NFFT; i++) {
                float m = cm2(R[i]);
                if (m > best) { best = m; bi = i; bo = off; }
            }
        }
        float peak = sqrtf(best) * g;
        if (peak > q->threshold && bi < FXR_NFFT - FXR_S_LEN) {
            q->state = QD_ALIGN; q->offset = bo; q->rxy = peak;
            memmove(q->buf0, q->buf0 + bi, (FXR_NFFT - bi) * sizeof(fxr_c32));
            q->counter = FXR_NFFT - bi;
            return bi == 0;
        }
    }
    memmove(q->buf0, q->buf0 + FXR_NFFT / 2, (FXR_NFFT / 2) * sizeof(fxr_c32));
    q->x2_0 = q->x2_1; q->x2_1 = 0.0f;
    q->counter = FXR_NFFT / 2;
    return 0;
}

static void qdet_align(fxr_qdet *q)
{
    fxr_c32 X[FXR_NFFT], R[FXR_NFFT], P[FXR_NFFT];
    const fxr_c32 *s = fxr_template();
    fxr_fft512(q->buf0, X);
    xcorr_offset(X, q->offset, R);
    /* timing: quadratic through sqrt|rxy| at lags -1, 0, +1 */
    float yneg = sqrtf(sqrtf(cm2(R[FXR_NFFT - 1])));
    float y0   = sqrtf(sqrtf(cm2(R[0])));
    float ypos = sqrtf(sqrtf(cm2(R[1])));
    float a = 0.5f * (ypos + yneg) - y0;
    float b = 0.5f * (ypos - yneg);
    float tau = a == 0.0f ? 0.0f : -b / (2.0f * a);
    if (!(fabsf(tau) < 1.0f)) tau = 0.0f;
    float gh = fmaf(fmaf(a, tau, b), tau, y0);
    q->tau = tau;
    q->gamma = gh * gh / ((float)FXR_NFFT * fxr_template_energy());
    /* carrier frequency: peak of FFT( x conj(s) ) */
    for (int i = 0; i < FXR_NFFT; i++) {
        if (i < FXR_S_LEN) P[i] = cmulc(q->buf0[i], s[i]); else { P[i].re = 0; P[i].im = 0; }
    }
    fxr_fft512(P, X);
    float v0 = -1.0f; unsigned i0 = 0;
    for (unsigned i = 0; i < FXR_NFFT; i++) { float m = cm2(X[i]); if (m > v0) { v0 = m; i0 = i; } }
    v0 = sqrtf(v0);
    float vneg = sqrtf(cm2(X[(i0 + FXR_NFFT - 1) % FXR_NFFT]));
    float vpos = sqrtf(cm2(X[(i0 + 1) % FXR_NFFT]));
    a = 0.5f * (vpos + vneg) - v0;
    b = 0.5f * (vpos - vneg);
    float idx = a == 0.0f ? 0.0f : -b / (2.0f * a);
    float index = (float)i0 + idx;
    q->dphi = (i0 > FXR_NFFT / 2 ? index - (float)FXR_NFFT : index) * (6.28318531f / (float)FXR_NFFT);
    /* carrier phase: arg sum_i p[i] exp(-j dphi i), tree order over 256 slots */
    uint32_t dl = fxr_rad2u32(q->dphi);
    fxr_c32 term[256];
    for (unsigned i = 0; i < 256; i++) {
        if (i < FXR_S_LEN) term[i] = derot(P[i], dl * i); else { term[i].re = 0; term[i].im = 0; }
    }
    fxr_c32 metric = fxr_csum_tree(term, 256);
    q->phi = fxr_atan2(metric.im, metric.re);
    /* hand the aligned window out, go back to seeking with its second half as overlap */
    memcpy(q->buf1, q->buf0, sizeof q->buf1);
    memmove(q->buf0, q->buf0 + FXR_NFFT / 2, (FXR_NFFT / 2) * sizeof(fxr_c32));
    q->x2_0 = half_energy(q->buf0); q->x2_1 = 0.0f;
    q->counter = FXR_NFFT / 2; q->state = QD_SEEK;
}

const fxr_c32 *fxr_qdet_execute(fxr_qdet *q, fxr_c32 x)
{
    q->buf0[q->counter++] = x;
    if (q->counter < FXR_NFFT) return NULL;
    if (q->state == QD_SEEK && !qdet_seek(q)) return NULL;
    qdet_align(q);
    return q->buf1;
}

unsigned fxr_qdet_run(fxr_qdet *q, const fxr_c32 *x, uint64_t n, int64_t base, fxr_detection *out, unsigned max)
{
    unsigned nd = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (!fxr_qdet_execute(q, x[i])) continue;
        if (nd < max) {
            fxr_detection d = { base + (int64_t)i - (FXR_NFFT - 1), q->tau, q->gamma, q->dphi, q->phi, q->rxy, q->offset };
            out[nd] = d;
        }
        nd++;
    }
    return nd;
}

/* =================================================================== frame synchroniser */
enum { FS_DETECT = 0, FS_PREAMBLE, FS_HEADER, FS_PAYLOAD };
struct fxr_sync {
    fxr_callback cb; void *ud;
    fxr_qdet *det;
    int state;
    uint64_t abs_in;            /* top-level input samples consumed so far */
    /* per-frame */
    fxr_frameinfo fi;
    uint32_t mix_th, mix_dl; float mf_scale;
    unsigned pfb; int mf_counter;
    fxr_c32 win[FXR_MF_TAPS]; unsigned wpos;
    unsigned sym_counter;       /* MF output symbols seen in this frame */
    fxr_c32 hdr_sym[FXR_HDR_SYM], hdr_mod[FXR_HDR_MOD];
    uint8_t hdr_dec[FXR_HDR_DEC]; int hdr_valid;
    /* payload */
    unsigned pay_len, pay_sym_len; int check, fec0, fec1, ms;
    fxr_modem demod; uint32_t pll_th; float pll_f; float evm_sum;
    float pll_c, pll_s;         /* carrier phasor exp(j pll_th), re-read from the table every 8th symbol, turned incrementally in between */
    fxr_c32 *pay_sym; uint8_t *pay_hard; uint8_t *pay_dec; unsigned pay_cap;
    unsigned pay_counter;
    /* optional equaliser */
    int eq_on; fxr_c32 eq_w[FXR_EQ_TAPS], eq_buf[FXR_EQ_TAPS];   /* eq_buf[12] is the newest matched-filter output */
    /* optional soft-decision decoding */
    int soft_on; uint8_t *soft, *soft_keep; unsigned soft_n;
};

fxr_sync *fxr_sync_create(fxr_callback cb, void *ud)
{
    fxr_init();
    fxr_sync *q = (fxr_sync *)calloc(1, sizeof *q);
    q->cb = cb; q->ud = ud;
    q->det = fxr_qdet_create_flexframe();
    fxr_sync_reset(q);
    return q;
}
void fxr_sync_destroy(fxr_sync *q)
{
    if (!q) return;
    fxr_qdet_destroy(q->det); free(q->pay_sym); free(q->pay_hard); free(q->pay_dec); free(q->soft); free(q->soft_keep); free(q);
}
void fxr_sync_set_threshold(fxr_sync *q, float t) { fxr_qdet_set_threshold(q->det, t); }
void fxr_sync_set_equalizer(fxr_sync *q, int on) { q->eq_on = on != 0; }
void fxr_sync_set_soft(fxr_sync *q, int on) { q->soft_on = on != 0; }
const uint8_t *fxr_sync_last_soft(const fxr_sync *q, unsigned *n) { if (n) *n = q->soft_n; return q->soft_keep; }
void fxr_sync_reset(fxr_sync *q)
{
    fxr_qdet_reset(q->det);
    q->state = FS_DETECT; q->sym_counter = 0; q->pay_counter = 0;
    memset(q->win, 0, sizeof q->win); q->wpos = 0;
}
void fxr_sync_last_frame(const fxr_sync *q, fxr_frameinfo *fi) { *fi = q->fi; }

static void sync_run(fxr_sync *q, const fxr_c32 *x, unsigned n, int top);

/* equaliser output: sum_i conj(w[i]) buf[i], balanced tree over 16 slots */
static fxr_c32 eq_output(const fxr_sync *q)
{
    fxr_c32 p[16];
    for (int i = 0; i < 16; i++) { if (i < FXR_EQ_TAPS) p[i] = cmulc(q->eq_buf[i], q->eq_w[i]); else { p[i].re = 0; p[i].im = 0; } }
    return fxr_csum_tree(p, 16);
}
/* one normalised-LMS step towards the known symbol d, given the output y just produced */
static void eq_train(fxr_sync *q, fxr_c32 d, fxr_c32 y)
{
    float e2[16];
    for (int i = 0; i < 16; i++) e2[i] = i < FXR_EQ_TAPS ? cm2(q->eq_buf[i]) : 0.0f;
    float x2 = fxr_sum_tree(e2, 16);
    if (!(x2 > 0.0f)) return;
    fxr_c32 e = { d.re - y.re, d.im - y.im };
    float g = FXR_EQ_MU / x2;
    for (int i = 0; i < FXR_EQ_TAPS; i++) {
        fxr_c32 c = cmulc(q->eq_buf[i], e);                   /* buf conj(e) */
        q->eq_w[i].re = fmaf(g, c.re, q->eq_w[i].re); q->eq_w[i].im = fmaf(g, c.im, q->eq_w[i].im);
    }
}

/* mix down, push into the polyphase MF, emit a symbol every second sample */
static int sync_step(fxr_sync *q, fxr_c32 x, fxr_c32 *y)
{
    fxr_c32 v = derot(x, q->mix_th);
    q->mix_th += q->mix_dl;
    q->wpos = (q->wpos + FXR_MF_TAPS - 1) % FXR_MF_TAPS;       /* newest sample sits at wpos */
    q->win[q->wpos] = v;
    const float *H = fxr_mf_proto();
    float ar = 0.0f, ai = 0.0f;
    for (unsigned t = 0; t < FXR_MF_TAPS; t++) {                /* t = 0 is the newest sample */
        fxr_c32 w = q->win[(q->wpos + t) % FXR_MF_TAPS];
        float h = H[q->pfb + FXR_NPFB * t];
        ar = fmaf(h, w.re, ar); ai = fmaf(h, w.im, ai);
    }
    if (q->eq_on) {                                             /* every matched-filter output goes through the equaliser's window */
        memmove(q->eq_buf, q->eq_buf + 1, (FXR_EQ_TAPS - 1) * sizeof(fxr_c32));
        q->eq_buf[FXR_EQ_TAPS - 1].re = ar * q->mf_scale; q->eq_buf[FXR_EQ_TAPS - 1].im = ai * q->mf_scale;
    }
    q->mf_counter++;
    if (q->mf_counter < 1) return 0;
    q->mf_counter -= FXR_K;
    if (q->eq_on) { *y = eq_output(q); return 1; }
    y->re = ar * q->mf_scale; y->im = ai * q->mf_scale;
    return 1;
}

static void sync_deliver(fxr_sync *q, int payload_valid)
{
    fxr_stats st; memset(&st, 0, sizeof st);
    st.rssi = 20.0f * log10f(q->fi.gamma);
    st.cfo = q->fi.dphi;
    if (q->hdr_valid) {
        st.evm = 10.0f * log10f(q->evm_sum / (float)(q->pay_sym_len ? q->pay_sym_len : 1));
        st.framesyms = q->pay_sym; st.num_framesyms = q->pay_sym_len;
        st.mod_scheme = (unsigned)q->ms; st.mod_bps = fxr_modem_bps(q->ms);
        st.check = (unsigned)q->check; st.fec0 = (unsigned)q->fec0; st.fec1 = (unsigned)q->fec1;
    }
    q->fi.evm_sum = q->evm_sum;
    if (q->cb)
        q->cb(q->hdr_dec, q->hdr_valid, q->hdr_valid ? q->pay_dec : NULL,
              q->hdr_valid ? q->pay_len : 0, payload_valid, st, q->ud);
    fxr_sync_reset(q);
}

static void sync_decode_payload(fxr_sync *q)
{
    unsigned el = fxr_packet_enc_len(q->pay_len, q->check, q->fec0, q->fec1);
    unsigned bps = q->demod.bps;
    if (q->soft_on) {
        /* per-bit soft values from the carrier-recovered symbols, in channel order; bits beyond the last symbol do not exist */
        q->soft = (uint8_t *)realloc(q->soft, 8 * (size_t)el + 8); q->soft_keep = (uint8_t *)realloc(q->soft_keep, 8 * (size_t)el + 8);
        memset(q->soft, 0, 8 * (size_t)el + 8);
        for (unsigned j = 0; j < q->pay_sym_len; j++) {
            uint8_t sb[8];
            fxr_modem_demod_soft(q->ms, q->pay_sym[j], q->pay_hard[j], sb);
            for (unsigned b = 0; b < bps; b++) if (j * bps + b < 8 * el) q->soft[j * bps + b] = sb[b];
        }
        memcpy(q->soft_keep, q->soft, 8 * (size_t)el); q->soft_n = 8 * el;
        int oks = fxr_packet_decode_soft(q->pay_len, q->check, q->fec0, q->fec1, q->soft, q->pay_dec);
        sync_deliver(q, oks);
        return;
    }
    uint8_t *pkt = (uint8_t *)calloc(el + 1, 1);
    for (unsigned j = 0; j < q->pay_sym_len; j++)
        for (unsigned b = 0; b < bps; b++) {
            unsigned k = j * bps + b;
            if (k < 8 * el && ((q->pay_hard[j] >> (bps - 1 - b)) & 1u)) pkt[k >> 3] |= (uint8_t)(0x80u >> (k & 7));
        }
    int ok = fxr_packet_decode(q->pay_len, q->check, q->fec0, q->fec1, pkt, q->pay_dec);
    free(pkt);
    sync_deliver(q, ok);
}

static void sync_decode_header(fxr_sync *q)
{
    const fxr_c32 *pil = fxr_pilots(), *tw = fxr_twiddle512();
    fxr_c32 b[FXR_HDR_PILOTS];
    for (unsigned p = 0; p < FXR_HDR_PILOTS; p++) b[p] = cmulc(q->hdr_sym[FXR_PILOT_SPACING * p], pil[p]);
    /* 32-point DFT of the 15 de-rotated pilots, sequential accumulation */
    float best = -1.0f; unsigned i0 = 0; float m2[32];
    for (unsigned k = 0; k < 32; k++) {
        float ar = 0.0f, ai = 0.0f;
        for (unsigned p = 0; p < FXR_HDR_PILOTS; p++) {
            fxr_c32 w = tw[16 * ((k * p) & 31)];
            ar = fmaf(b[p].re, w.re, ar); ar = fmaf(-b[p].im, w.im, ar);
            ai = fmaf(b[p].re, w.im, ai); ai = fmaf(b[p].im, w.re, ai);
        }
        m2[k] = fmaf(ar, ar, ai * ai);
        if (m2[k] > best) { best = m2[k]; i0 = k; }
    }
    float y0 = sqrtf(m2[i0]), yneg = sqrtf(m2[(i0 + 31) & 31]), ypos = sqrtf(m2[(i0 + 1) & 31]);
    float a = 0.5f * (ypos + yneg) - y0, bb = 0.5f * (ypos - yneg);
    float idx = a == 0.0f ? 0.0f : -bb / (2.0f * a);
    float index = i0 < 16 ? (float)i0 : (float)i0 - 32.0f;
    float dphi = (index + idx) * (6.28318531f / 512.0f);        /* 2pi / (nfft * pilot_spacing) */
    uint32_t dl = fxr_rad2u32(dphi);
    float mr = 0.0f, mi = 0.0f;
    for (unsigned p = 0; p < FXR_HDR_PILOTS; p++) {
        fxr_c32 t = derot(b[p], dl * (FXR_PILOT_SPACING * p));
        mr += t.re; mi += t.im;
    }
    float phi = fxr_atan2(mi, mr);
    float ghat = sqrtf(fmaf(mr, mr, mi * mi)) / (float)FXR_HDR_PILOTS;
    float g = 1.0f / ghat;
    uint32_t ph = fxr_rad2u32(phi);
    q->fi.pilot_dphi = dphi; q->fi.pilot_phi = phi; q->fi.pilot_gain = ghat;
    uint8_t he[FXR_HDR_ENC]; memset(he, 0, sizeof he);
    for (unsigned i = 0, n = 0; i < FXR_HDR_SYM; i++) {
        if ((i % FXR_PILOT_SPACING) == 0) continue;
        fxr_c32 y = derot(q->hdr_sym[i], ph + dl * i);
        y.re *= g; y.im *= g;
        q->hdr_mod[n] = y;
        unsigned s = (y.re > 0.0f ? 0u : 1u) | (y.im > 0.0f ? 0u : 2u);
        /* 2 bits per symbol, MSB first */
        if (s & 2) he[(2 * n) >> 3] |= (uint8_t)(0x80u >> ((2 * n) & 7));
        if (s & 1) he[(2 * n + 1) >> 3] |= (uint8_t)(0x80u >> ((2 * n + 1) & 7));
        n++;
    }
    q->hdr_valid = fxr_packet_decode(FXR_HDR_DEC, FXR_CRC_32, FXR_FEC_SECDED7264, FXR_FEC_HAMMING84, he, q->hdr_dec);
    if (q->hdr_valid) {
        const uint8_t *h = q->hdr_dec + FXR_HDR_USER;
        q->pay_len = ((unsigned)h[1] << 8) | h[2];
        q->ms = h[3]; q->check = (h[4] >> 5) & 7; q->fec0 = h[4] & 0x1f; q->fec1 = h[5] & 0x1f;
        if (h[0] != FXR_PROTOCOL || fxr_modem_bps(q->ms) == 0 ||
            q->check == FXR_CRC_UNKNOWN || q->check > FXR_CRC_32 ||
            !fxr_fec_supported(q->fec0) || !fxr_fec_supported(q->fec1))
            q->hdr_valid = 0;
    }
    if (!q->hdr_valid) { q->evm_sum = 0.0f; sync_deliver(q, 0); return; }
    q->pay_sym_len = fxr_qpm_sym_len(q->pay_len, q->check, q->fec0, q->fec1, q->ms);
    if (q->pay_sym_len + 1 > q->pay_cap) {
        q->pay_cap = q->pay_sym_len + 1;
        q->pay_sym = (fxr_c32 *)realloc(q->pay_sym, q->pay_cap * sizeof(fxr_c32));
        q->pay_hard = (uint8_t *)realloc(q->pay_hard, q->pay_cap);
    }
    q->pay_dec = (uint8_t *)realloc(q->pay_dec, q->pay_len + 8);
    fxr_modem_init(&q->demod, q->ms);
    q->pll_f = dphi * 683565248.0f;             /* rad/symbol -> phase units/symbol */
    q->pll_th = ph + dl * FXR_HDR_SYM;
    q->evm_sum = 0.0f; q->pay_counter = 0;
    q->state = FS_PAYLOAD;
    if (q->pay_sym_len == 0) sync_decode_payload(q);
}

static void sync_on_symbol(fxr_sync *q, fxr_c32 y)
{
    unsigned c = q->sym_counter++;
    const unsigned dly = q->eq_on ? FXR_EQ_DELAY : 0u;  /* the equaliser delays every symbol instant by 3 */
    if (c < FXR_SYM0_HDR + dly) {                       /* MF (+ equaliser) delay, then the p/n symbols: they train the equaliser */
        if (q->eq_on && c >= FXR_PRE_DELAY + dly) eq_train(q, fxr_preamble_pn()[c - FXR_PRE_DELAY - dly], y);
        return;
    }
    if (c < FXR_SYM0_PAY + dly) {
        q->hdr_sym[c - FXR_SYM0_HDR - dly] = y;
        if (c == FXR_SYM0_PAY + dly - 1) sync_decode_header(q);
        return;
    }
    /* payload: decision-directed 2nd-order PLL, alpha = 1e-4, beta = sqrt(alpha) [RECALLED nco_crcf_pll_step: frequency +=
     * alpha * error, phase += beta * error, then nco step], error = imag(r conj(xhat)).  The carrier phasor is read from the
     * sin/cos table at every 8th symbol and turned by the (integer) phase increment in between -- a GPU-friendly canonical
     * form: the table look-up leaves the symbol-to-symbol recurrence. */
    if ((q->pay_counter & 7u) == 0) fxr_sincos_u32(q->pll_th, &q->pll_c, &q->pll_s);
    fxr_c32 r = { fmaf(y.re, q->pll_c, y.im * q->pll_s), fmaf(y.im, q->pll_c, -(y.re * q->pll_s)) }, xh; float pe;
    unsigned s = fxr_modem_demod(&q->demod, r, &xh, &pe);
    float dr = r.re - xh.re, di = r.im - xh.im;
    q->evm_sum += fmaf(dr, dr, di * di);
    /* loop filter kept in phase units: alpha = 1e-4 and beta = 1e-2 pre-multiplied by 2^32/2pi */
    q->pll_f = fmaf(pe, 68356.5248f, q->pll_f);
    float step = fxr_phase_step(fmaf(pe, 6835652.5f, q->pll_f));
    q->pll_th += (uint32_t)(int32_t)step;
    {
        float cd, sd; fxr_sincos_small(step, &cd, &sd);
        float c2 = fmaf(q->pll_c, cd, -(q->pll_s * sd)), s2 = fmaf(q->pll_s, cd, q->pll_c * sd);
        q->pll_c = c2; q->pll_s = s2;
    }
    q->pay_sym[q->pay_counter] = r; q->pay_hard[q->pay_counter] = (uint8_t)s;
    if (++q->pay_counter == q->pay_sym_len) sync_decode_payload(q);
}

static void sync_run(fxr_sync *q, const fxr_c32 *x, unsigned n, int top)
{
    for (unsigned i = 0; i < n; i++) {
        if (top) q->abs_in++;
        if (q->state == FS_DETECT) {
            const fxr_c32 *v = fxr_qdet_execute(q->det, x[i]);
            if (!v) continue;
            memset(&q->fi, 0, sizeof q->fi);
            q->fi.start = q->abs_in - FXR_NFFT;         /* abs index of aligned sample 0 */
            q->fi.offset = fxr_qdet_offset(q->det); q->fi.rxy = fxr_qdet_rxy(q->det);
            q->fi.tau = fxr_qdet_tau(q->det); q->fi.gamma = fxr_qdet_gamma(q->det);
            q->fi.dphi = fxr_qdet_dphi(q->det); q->fi.phi = fxr_qdet_phi(q->det);
            if (q->fi.tau > 0.0f) { q->pfb = (unsigned)(q->fi.tau * (float)FXR_NPFB) % FXR_NPFB; q->mf_counter = 0; }
            else { q->pfb = (unsigned)((1.0f + q->fi.tau) * (float)FXR_NPFB) % FXR_NPFB; q->mf_counter = 1; }
            q->fi.pfb_index = q->pfb; q->fi.mf_counter0 = q->mf_counter;
            q->mf_scale = 0.5f / q->fi.gamma;
            q->mix_dl = fxr_rad2u32(q->fi.dphi); q->mix_th = fxr_rad2u32(q->fi.phi);
            q->state = FS_PREAMBLE; q->sym_counter = 0;
            memset(q->win, 0, sizeof q->win); q->wpos = 0;
            if (q->eq_on) {
                float h[FXR_EQ_TAPS]; fxr_eq_init_taps(h);
                for (int t = 0; t < FXR_EQ_TAPS; t++) { q->eq_w[t].re = h[t]; q->eq_w[t].im = 0.0f; }
                memset(q->eq_buf, 0, sizeof q->eq_buf);
            }
            fxr_c32 held[FXR_NFFT]; memcpy(held, v, sizeof held);
            sync_run(q, held, FXR_NFFT, 0);             /* re-feed the aligned window */
            continue;
        }
        fxr_c32 y;
        if (sync_step(q, x[i], &y)) sync_on_symbol(q, y);
    }
}

void fxr_sync_execute(fxr_sync *q, const fxr_c32 *x, unsigned n) { sync_run(q, x, n, 1); }

/* the caller's loop of /root/reference/lib/flex_rx_impl.cc:212-215: `chunk` samples per execute call */
void fxr_sync_execute_chunked(fxr_sync *q, const fxr_c32 *x, uint64_t n, unsigned chunk)
{
    for (uint64_t i = 0; i < n; i += chunk) sync_run(q, x + i, (unsigned)((n - i) < chunk ? (n - i) : chunk), 1);
}
