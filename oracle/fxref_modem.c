/*
 * fxref_modem.c -- CPU ORACLE (test infrastructure; see fxref.h).  PARITY UNPINNED.
 *
 * Linear modems for the 11 schemes the reference exposes
 * (/root/reference/lib/flex_tx_impl.cc:75-116, /root/reference/lib/flex_rx_impl.cc:138-179) plus the
 * QPSK used by the flexframe header.  [RECALLED liquid-dsp v1.3.x modem_psk.c, modem_dpsk.c,
 * modem_ask.c, modem_qam.c, modem_qpsk.c]: Gray-coded; PSK points at 2 pi i/M, square/rectangular
 * QAM with unit average energy, ASK4 scaled by 1/sqrt(5).  Hard decisions are taken geometrically
 * (nearest point) so that they need no libm call; the phase error handed to the payload PLL is
 * imag(r conj(xhat)).  Soft decisions (optional; liquid's modem_demodulate_soft): per coded bit an unsigned byte,
 * 0 = surely 0, 255 = surely 1, from the difference of the squared distances to the nearest constellation point whose
 * label carries a 0 / a 1 at that bit.
 */
#include "fxref.h"
#include <math.h>

static inline unsigned gray_enc(unsigned x) { return x ^ (x >> 1); }
static inline unsigned gray_dec(unsigned x) { unsigned y = x; while (x >>= 1) y ^= x; return y; }

#define TWO_PI_F 6.28318531f
#define PI_F     3.14159265f

unsigned fxr_modem_bps(int ms)
{
    switch (ms) {
    case FXR_MODEM_PSK2: case FXR_MODEM_DPSK2: return 1;
    case FXR_MODEM_PSK4: case FXR_MODEM_DPSK4: case FXR_MODEM_ASK4: case FXR_MODEM_QPSK: return 2;
    case FXR_MODEM_PSK8: case FXR_MODEM_DPSK8: return 3;
    case FXR_MODEM_PSK16: case FXR_MODEM_QAM16: return 4;
    case FXR_MODEM_QAM32: return 5;
    case FXR_MODEM_QAM64: return 6;
    default: return 0;
    }
}

void fxr_modem_init(fxr_modem *q, int ms) { q->ms = ms; q->bps = fxr_modem_bps(ms); q->dpsk_phi = 0.0f; }

static void qam_dims(int ms, unsigned *mi, unsigned *mq, float *alpha)
{
    switch (ms) {
    case FXR_MODEM_QAM16: *mi = 2; *mq = 2; *alpha = 0.316227766f; break;   /* 1/sqrt(10) */
    case FXR_MODEM_QAM32: *mi = 3; *mq = 2; *alpha = 0.196116135f; break;   /* 1/sqrt(26) */
    default:              *mi = 3; *mq = 3; *alpha = 0.154303350f; break;   /* 1/sqrt(42) */
    }
}

/* unit phasor at phase index i of M: exact axis points for M <= 4, the shared sincos table beyond */
static fxr_c32 psk_point(unsigned i, unsigned bps)
{
    fxr_c32 p;
    if (bps <= 2) {
        static const fxr_c32 axis[4] = { { 1.0f, 0.0f }, { 0.0f, 1.0f }, { -1.0f, 0.0f }, { 0.0f, -1.0f } };
        return axis[bps == 1 ? 2 * (i & 1) : (i & 3)];
    }
    fxr_sincos_u32((uint32_t)i << (32 - bps), &p.re, &p.im);
    return p;
}

fxr_c32 fxr_modem_mod(fxr_modem *q, unsigned sym)
{
    fxr_c32 y = { 0, 0 };
    switch (q->ms) {
    case FXR_MODEM_QPSK:
        y.re = (sym & 1) ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
        y.im = (sym & 2) ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
        return y;
    case FXR_MODEM_PSK2: case FXR_MODEM_PSK4: case FXR_MODEM_PSK8: case FXR_MODEM_PSK16:
        return psk_point(gray_dec(sym), q->bps);
    case FXR_MODEM_DPSK2: case FXR_MODEM_DPSK4: case FXR_MODEM_DPSK8: {
        /* dpsk_phi holds the running phase index as an integer-valued float */
        unsigned idx = ((unsigned)q->dpsk_phi + gray_dec(sym)) & ((1u << q->bps) - 1u);
        q->dpsk_phi = (float)idx;
        return psk_point(idx, q->bps); }
    case FXR_MODEM_ASK4:
        y.re = (2.0f * (float)gray_dec(sym) - 3.0f) * 0.447213595f;         /* 1/sqrt(5) */
        return y;
    default: {
        unsigned mi, mq; float al; qam_dims(q->ms, &mi, &mq, &al);
        unsigned si = gray_dec(sym >> mq), sq = gray_dec(sym & ((1u << mq) - 1u));
        y.re = (2.0f * (float)si - (float)((1u << mi) - 1u)) * al;
        y.im = (2.0f * (float)sq - (float)((1u << mq) - 1u)) * al;
        return y; }
    }
}

/* nearest level index on a uniform grid of L points spaced 2*al, centred on 0 */
static inline unsigned pam_index(float v, float inv2al, unsigned L)
{
    float t = floorf(fmaf(v, inv2al, 0.5f * (float)L));
    if (t < 0.0f) t = 0.0f;
    if (t > (float)(L - 1)) t = (float)(L - 1);
    return (unsigned)t;
}

/* nearest of M equally spaced phases: index = round(theta * M / 2pi) mod M */
static inline unsigned psk_index(fxr_c32 r, unsigned bps)
{
    if (bps == 1) return r.re > 0.0f ? 0u : 1u;
    if (bps == 2) {
        if (fabsf(r.re) >= fabsf(r.im)) return r.re > 0.0f ? 0u : 2u;
        return r.im > 0.0f ? 1u : 3u;
    }
    float th = fxr_atan2(r.im, r.re);
    float t = rintf(th * ((float)(1u << bps) * 0.159154943f));              /* M / 2pi */
    return (unsigned)((int)t) & ((1u << bps) - 1u);
}

unsigned fxr_modem_demod(fxr_modem *q, fxr_c32 r, fxr_c32 *xhat, float *phase_err)
{
    unsigned sym; fxr_c32 xh = { 0, 0 };
    switch (q->ms) {
    case FXR_MODEM_QPSK:
        sym = (r.re > 0.0f ? 0u : 1u) | (r.im > 0.0f ? 0u : 2u);
        xh.re = (sym & 1) ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
        xh.im = (sym & 2) ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
        break;
    case FXR_MODEM_PSK2: case FXR_MODEM_PSK4: case FXR_MODEM_PSK8: case FXR_MODEM_PSK16: {
        unsigned idx = psk_index(r, q->bps);
        sym = gray_enc(idx); xh = psk_point(idx, q->bps);
        break; }
    case FXR_MODEM_DPSK2: case FXR_MODEM_DPSK4: case FXR_MODEM_DPSK8: {
        unsigned M1 = (1u << q->bps) - 1u;
        unsigned idx = psk_index(r, q->bps), prev = (unsigned)q->dpsk_phi;
        sym = gray_enc((idx - prev) & M1);
        q->dpsk_phi = (float)idx; xh = psk_point(idx, q->bps);
        break; }
    case FXR_MODEM_ASK4: {
        unsigned idx = pam_index(r.re, 1.11803399f, 4);                     /* sqrt(5)/2 */
        sym = gray_enc(idx); xh.re = (2.0f * (float)idx - 3.0f) * 0.447213595f;
        break; }
    default: {
        unsigned mi, mq; float al; qam_dims(q->ms, &mi, &mq, &al);
        float inv = 0.5f / al;
        unsigned ii = pam_index(r.re, inv, 1u << mi), iq = pam_index(r.im, inv, 1u << mq);
        sym = (gray_enc(ii) << mq) | gray_enc(iq);
        xh.re = (2.0f * (float)ii - (float)((1u << mi) - 1u)) * al;
        xh.im = (2.0f * (float)iq - (float)((1u << mq) - 1u)) * al;
        break; }
    }
    if (xhat) *xhat = xh;
    if (phase_err) {
        /* [RECALLED liquid modem_get_demodulator_phase_error]: imag(r * conj(xhat)), not its argument */
        *phase_err = fmaf(r.im, xh.re, -(r.re * xh.im));
    }
    return sym;
}

/* ---------------------------------------------------------------- soft decisions
 * soft[b], b = 0 .. bps-1 (MSB of the symbol first) = clamp(rint(127 + 16 gamma (d0 - d1)), 0, 255), gamma = 1.2 M
 * [RECALLED liquid modem_demodulate_soft: gamma = 1.2 M, soft_bit = llr * 16 + 127], d0 / d1 = squared distance from r to
 * the nearest point with a 0 / a 1 at that bit.  Restatement choices: the search is exhaustive over the M phases for PSK;
 * ASK and the square / rectangular QAMs are separable, so each axis is searched over its own levels; differential PSK has
 * no soft form here (as liquid falls back to its hard decision): the hard symbol's bits, as 0 / 255. */
static inline uint8_t soft_byte(float d0, float d1, float gamma16)
{
    float t = rintf(fmaf(d0 - d1, gamma16, 127.0f));
    if (t < 0.0f) t = 0.0f;
    if (t > 255.0f) t = 255.0f;
    return (uint8_t)t;
}
/* one axis of an ASK / QAM constellation: L levels (2 i - (L-1)) al, label gray(i), nb bits */
static void soft_axis(float v, unsigned nb, float al, float gamma16, uint8_t *soft)
{
    unsigned L = 1u << nb;
    float d0[3], d1[3];
    for (unsigned b = 0; b < nb; b++) { d0[b] = 1e30f; d1[b] = 1e30f; }
    for (unsigned i = 0; i < L; i++) {
        float dx = v - (2.0f * (float)i - (float)(L - 1)) * al, d = dx * dx;
        unsigned g = gray_enc(i);
        for (unsigned b = 0; b < nb; b++) {
            if ((g >> (nb - 1 - b)) & 1u) { if (d < d1[b]) d1[b] = d; } else { if (d < d0[b]) d0[b] = d; }
        }
    }
    for (unsigned b = 0; b < nb; b++) soft[b] = soft_byte(d0[b], d1[b], gamma16);
}

void fxr_modem_demod_soft(int ms, fxr_c32 r, unsigned hard_sym, uint8_t *soft)
{
    unsigned bps = fxr_modem_bps(ms);
    float gamma16 = 1.2f * (float)(1u << bps) * 16.0f;
    switch (ms) {
    case FXR_MODEM_DPSK2: case FXR_MODEM_DPSK4: case FXR_MODEM_DPSK8:
        for (unsigned b = 0; b < bps; b++) soft[b] = ((hard_sym >> (bps - 1 - b)) & 1u) ? 255 : 0;
        return;
    case FXR_MODEM_PSK2: case FXR_MODEM_PSK4: case FXR_MODEM_PSK8: case FXR_MODEM_PSK16: {
        float d0[4], d1[4];
        for (unsigned b = 0; b < bps; b++) { d0[b] = 1e30f; d1[b] = 1e30f; }
        for (unsigned i = 0; i < (1u << bps); i++) {
            fxr_c32 p = psk_point(i, bps);
            float dx = r.re - p.re, dy = r.im - p.im, d = fmaf(dx, dx, dy * dy);
            unsigned g = gray_enc(i);
            for (unsigned b = 0; b < bps; b++) {
                if ((g >> (bps - 1 - b)) & 1u) { if (d < d1[b]) d1[b] = d; } else { if (d < d0[b]) d0[b] = d; }
            }
        }
        for (unsigned b = 0; b < bps; b++) soft[b] = soft_byte(d0[b], d1[b], gamma16);
        return; }
    case FXR_MODEM_ASK4: soft_axis(r.re, 2, 0.447213595f, gamma16, soft); return;
    case FXR_MODEM_QPSK:                            /* symbol = (re < 0) | (im < 0) << 1: MSB is the imaginary axis */
        soft_axis(-r.im, 1, (float)M_SQRT1_2, gamma16, soft); soft_axis(-r.re, 1, (float)M_SQRT1_2, gamma16, soft + 1); return;
    default: {
        unsigned mi, mq; float al; qam_dims(ms, &mi, &mq, &al);
        soft_axis(r.re, mi, al, gamma16, soft); soft_axis(r.im, mq, al, gamma16, soft + mi);
        return; }
    }
}

unsigned fxr_qpm_sym_len(unsigned n, int check, int fec0, int fec1, int ms)
{
    unsigned bps = fxr_modem_bps(ms);
    if (!bps) return 0;
    unsigned bits = 8 * fxr_packet_enc_len(n, check, fec0, fec1);
    return (bits + bps - 1) / bps;
}
