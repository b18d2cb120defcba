/*
 * fxref_filt.c -- CPU ORACLE (test infrastructure; see fxref.h).  PARITY UNPINNED.
 *
 * m-sequence generator, approximate r-Kaiser ("ARKAISER") square-root Nyquist design and the
 * constant tables derived from them (p/n preamble, detector template, pilots, MF prototype).
 *
 * Reference parameters: msequence_create(7, 0x0089, 1), 64 symbols, bit 2i -> re, 2i+1 -> im,
 * +-sqrt(1/2)  (/root/reference/lib/frame_detector_cc_impl.cc:46-52);
 * qdetector_cccf_create_linear(pn, 64, LIQUID_FIRFILT_ARKAISER, k=2, m=7, beta=0.3)
 * (/root/reference/lib/frame_detector_cc_impl.cc:54, frame_detector_cc_impl.h:34-36).
 * Algorithms [RECALLED liquid-dsp v1.3.x: msequence.c, rkaiser.c (liquid_firdes_arkaiser),
 * firdes.c (liquid_firdes_kaiser, kaiser_beta_As), firinterp.c, firpfb.c, qpilotgen.c].
 */
#include "fxref.h"
#include <math.h>
#include <string.h>

void fxr_math_init_(void);

/* ---------------------------------------------------------------- m-sequence (LFSR) */
void fxr_mseq_init(fxr_mseq *q, unsigned m, unsigned g, unsigned a)
{
    q->m = m; q->g = g >> 1; q->a = a; q->n = (1u << m) - 1u; q->v = a;
}

unsigned fxr_mseq_advance(fxr_mseq *q)
{
    unsigned b = (unsigned)__builtin_popcount(q->v & q->g) & 1u;
    q->v = ((q->v << 1) | b) & q->n;
    return b;
}

unsigned fxr_mseq_symbol(fxr_mseq *q, unsigned bps)
{
    unsigned s = 0;
    for (unsigned i = 0; i < bps; i++) s = (s << 1) | fxr_mseq_advance(q);
    return s;
}

/* ---------------------------------------------------------------- filter design (double) */
static double bessel_i0(double z)
{
    double t = 1.0, s = 1.0, h = 0.5 * z;
    for (int k = 1; k < 64; k++) { t *= (h / k) * (h / k); s += t; if (t < 1e-18 * s) break; }
    return s;
}

static double sinc(double x) { return fabs(x) < 1e-12 ? 1.0 : sin(M_PI * x) / (M_PI * x); }

static double kaiser_beta_As(double As)
{
    As = fabs(As);
    if (As > 50.0) return 0.1102 * (As - 8.7);
    if (As > 21.0) return 0.5842 * pow(As - 21.0, 0.4) + 0.07886 * (As - 21.0);
    return 0.0;
}

static double kaiser_w(unsigned i, unsigned n, double beta, double mu)
{
    double t = (double)i - (double)(n - 1) / 2.0 + mu;
    double r = 2.0 * t / (double)n;
    double a = 1.0 - r * r;
    return bessel_i0(beta * sqrt(a > 0 ? a : 0)) / bessel_i0(beta);
}

/* approximate r-Kaiser: Kaiser-windowed sinc whose cut-off is nudged by rho_hat(m, beta) */
void fxr_firdes_arkaiser(unsigned k, unsigned m, float beta_f, float dt, float *h)
{
    double beta = beta_f, lm = log((double)m), lb = log(beta);
    double c0 = 0.762886 + 0.067663 * lm;
    double c1 = 0.065515;
    double c2 = log(1.0 - 0.088 * pow((double)m, -1.6));
    double rho = c0 + c1 * lb + c2 * lb * lb;
    if (rho <= 0.0 || rho >= 1.0) rho = 0.5;
    unsigned n = 2 * k * m + 1;
    double del = beta * rho / (double)k;                 /* transition width */
    double As = 14.26 * del * (double)n + 7.95;          /* Kaiser's length formula, solved for As */
    double fc = 0.5 * (1.0 + beta * (1.0 - rho)) / (double)k;
    double kb = kaiser_beta_As(As);
    double hd[2 * 64 * 16 + 1];
    double e2 = 0;
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0 + dt;
        hd[i] = sinc(2.0 * fc * t) * kaiser_w(i, n, kb, dt);
        e2 += hd[i] * hd[i];
    }
    double g = sqrt((double)k / e2);
    for (unsigned i = 0; i < n; i++) h[i] = (float)(hd[i] * g);
}

/* initial taps of the optional equaliser [RECALLED eqlms_cccf_create_lowpass(2*k*p+1 = 13, fc = 0.4): a Kaiser-windowed
 * sinc (As = 40 dB) scaled by 2 fc] */
void fxr_eq_init_taps(float *h)
{
    const unsigned n = FXR_EQ_TAPS; const double fc = 0.4, kb = kaiser_beta_As(40.0);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0;
        h[i] = (float)(sinc(2.0 * fc * t) * kaiser_w(i, n, kb, 0.0) * 2.0 * fc);
    }
}

/* ---------------------------------------------------------------- constant tables */
static float   g_proto[2 * FXR_NPFB * FXR_K * FXR_M + 1];
static float   g_tx[2 * FXR_K * FXR_M + 1];
static fxr_c32 g_pn[FXR_PN_LEN];
static fxr_c32 g_tmpl[FXR_S_LEN];
static fxr_c32 g_tmpl_fft[FXR_NFFT];
static float   g_tmpl_e;
static fxr_c32 g_pil[FXR_HDR_PILOTS];
static int     g_ready = 0;

void fxr_init(void)
{
    if (g_ready) return;
    fxr_math_init_();
    fxr_firdes_arkaiser(FXR_NPFB * FXR_K, FXR_M, FXR_BETA, 0.0f, g_proto);
    fxr_firdes_arkaiser(FXR_K, FXR_M, FXR_BETA, 0.0f, g_tx);

    fxr_mseq ms; fxr_mseq_init(&ms, 7, 0x0089, 1);
    for (int i = 0; i < FXR_PN_LEN; i++) {
        g_pn[i].re = fxr_mseq_advance(&ms) ? (float)M_SQRT1_2 : -(float)M_SQRT1_2;
        g_pn[i].im = fxr_mseq_advance(&ms) ? (float)M_SQRT1_2 : -(float)M_SQRT1_2;
    }
    /* template = interp(pn, k=2) flushed with 2m zeros: y[2n+i] = sum_t h[i+2t] x[n-t] */
    for (int n = 0; n < FXR_PN_LEN + 2 * FXR_M; n++)
        for (int i = 0; i < FXR_K; i++) {
            float ar = 0, ai = 0;
            for (int t = 0; t < 15; t++) {
                int hi = i + FXR_K * t, xi = n - t;
                if (hi > 2 * FXR_K * FXR_M || xi < 0 || xi >= FXR_PN_LEN) continue;
                ar = fmaf(g_tx[hi], g_pn[xi].re, ar);
                ai = fmaf(g_tx[hi], g_pn[xi].im, ai);
            }
            g_tmpl[FXR_K * n + i].re = ar; g_tmpl[FXR_K * n + i].im = ai;
        }
    fxr_c32 buf[FXR_NFFT]; memset(buf, 0, sizeof buf);
    memcpy(buf, g_tmpl, sizeof g_tmpl);
    fxr_fft512(buf, g_tmpl_fft);
    g_tmpl_e = 0;
    for (int i = 0; i < FXR_S_LEN; i++) g_tmpl_e += fmaf(g_tmpl[i].re, g_tmpl[i].re, g_tmpl[i].im * g_tmpl[i].im);

    /* pilots: m-sequence m=4 (g=0x13), 2 bits/symbol, QPSK at pi/4 + s pi/2 */
    fxr_mseq_init(&ms, 4, 0x0013, 1);
    static const float pr[4] = { (float)M_SQRT1_2, -(float)M_SQRT1_2, -(float)M_SQRT1_2, (float)M_SQRT1_2 };
    static const float pi_[4] = { (float)M_SQRT1_2, (float)M_SQRT1_2, -(float)M_SQRT1_2, -(float)M_SQRT1_2 };
    for (int i = 0; i < FXR_HDR_PILOTS; i++) {
        unsigned s = fxr_mseq_symbol(&ms, 2);
        g_pil[i].re = pr[s]; g_pil[i].im = pi_[s];
    }
    g_ready = 1;
}

const float   *fxr_mf_proto(void)        { fxr_init(); return g_proto; }
const float   *fxr_tx_taps(void)         { fxr_init(); return g_tx; }
const fxr_c32 *fxr_preamble_pn(void)     { fxr_init(); return g_pn; }
const fxr_c32 *fxr_template(void)        { fxr_init(); return g_tmpl; }
const fxr_c32 *fxr_template_fft(void)    { fxr_init(); return g_tmpl_fft; }
float          fxr_template_energy(void) { fxr_init(); return g_tmpl_e; }
const fxr_c32 *fxr_pilots(void)          { fxr_init(); return g_pil; }
