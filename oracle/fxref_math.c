/*
 * fxref_math.c -- CPU ORACLE (test infrastructure; see fxref.h header).  PARITY UNPINNED.
 *
 * Canonical scalar arithmetic: 32-bit-phase sin/cos, polynomial atan2, balanced-tree sums and
 * the 8x8x8 decimation-in-frequency FFT-512 that qdetector_cccf's correlator uses
 * ([RECALLED liquid-dsp qdetector_cccf.c: fft_create_plan(nfft,...); call site
 * /root/reference/lib/frame_detector_cc_impl.cc:77]).  liquid calls FFTW or its own radix
 * code; the transform is the same DFT, the butterfly network below is this repo's choice.
 */
#include "fxref.h"
#include <math.h>
#include <string.h>

static fxr_c32 g_tw[512];
static fxr_c32 g_sc[1024];
static int     g_math_ready = 0;

void fxr_math_init_(void)
{
    if (g_math_ready) return;
    for (int m = 0; m < 512; m++) {
        double a = 2.0 * M_PI * (double)m / 512.0;
        g_tw[m].re = (float)cos(a);
        g_tw[m].im = (float)(-sin(a));
    }
    for (int k = 0; k < 1024; k++) {
        double a = 2.0 * M_PI * (double)k / 1024.0;
        g_sc[k].re = (float)cos(a);
        g_sc[k].im = (float)sin(a);
    }
    g_math_ready = 1;
}

const fxr_c32 *fxr_twiddle512(void) { fxr_math_init_(); return g_tw; }
const fxr_c32 *fxr_sincos_table(void) { fxr_math_init_(); return g_sc; }

/* radians -> 32-bit phase (2^32 == one turn), wrapping */
uint32_t fxr_rad2u32(float rad)
{
    float t = rintf(rad * 683565248.0f);        /* 2^32/(2 pi) rounded to binary32 */
    return (uint32_t)(int64_t)t;
}

/* PLL increments, already in phase units (2^32 = one turn); clamp so that a 32-bit convert is exact on any input */
uint32_t fxr_phase_inc(float units)
{
    float t = rintf(units);
    t = fminf(fmaxf(t, -2147483520.0f), 2147483520.0f);
    return (uint32_t)(int32_t)t;
}

/* cos/sin of a 32-bit phase: table on the top 10 bits, series on the remaining 22 */
void fxr_sincos_u32(uint32_t th, float *c, float *s)
{
    const fxr_c32 t = g_sc[th >> 22];
    float d  = (float)(th & 0x3FFFFFu) * 1.4629180792671596e-9f;   /* 2 pi / 2^32 */
    float d2 = d * d;
    float cd = fmaf(d2, -0.5f, 1.0f);
    float sd = fmaf(d2 * d, -0.16666667f, d);
    *c = fmaf(t.re, cd, -(t.im * sd));
    *s = fmaf(t.im, cd, t.re * sd);
}

/* payload PLL: the phase advance of one symbol in whole phase units (2^32 = one turn; rounded, clamped below 2^31 so that
 * the conversion to an integer is the same everywhere), and cos / sin of that advance by a 5th-order series: the loop turns
 * its carrier phasor by this per symbol instead of looking the whole phase up again (fxref_frame.c: sync_on_symbol) */
float fxr_phase_step(float units)
{
    float t = rintf(units);
    if (t < -2147483520.0f) t = -2147483520.0f;
    if (t > 2147483520.0f) t = 2147483520.0f;
    return t;
}
void fxr_sincos_small(float step_units, float *c, float *s)
{
    float x  = step_units * 1.4629180792671596e-9f;                  /* 2 pi / 2^32 */
    float x2 = x * x;
    *c = fmaf(x2, fmaf(x2, 4.16666679e-2f, -0.5f), 1.0f);
    *s = fmaf(x * x2, fmaf(x2, 8.33333377e-3f, -0.16666667f), x);
}

/* atan2 from +,*,/ only (Cephes atanf kernel, one division); returns angle in (-pi, pi] */
float fxr_atan2(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    int big = mn > 0.41421356f * mx;            /* ratio above tan(pi/8): reduce around pi/4 */
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float base = big ? 0.78539816f : 0.0f;
    float a = num / den;
    float z = a * a;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float r = fmaf(p * z, a, a) + base;         /* atan(mn/mx) in [0, pi/4] */
    if (ay > ax) r = 1.57079633f - r;
    if (x < 0.0f) r = 3.14159265f - r;
    r = y < 0.0f ? -r : r;
    return mx == 0.0f ? 0.0f : r;
}

float fxr_sum_tree(const float *v, unsigned n)
{
    if (n == 1) return v[0];
    return fxr_sum_tree(v, n / 2) + fxr_sum_tree(v + n / 2, n / 2);
}

fxr_c32 fxr_csum_tree(const fxr_c32 *v, unsigned n)
{
    if (n == 1) return v[0];
    fxr_c32 a = fxr_csum_tree(v, n / 2), b = fxr_csum_tree(v + n / 2, n / 2);
    fxr_c32 r = { a.re + b.re, a.im + b.im };
    return r;
}

/* ---------------------------------------------------------------- FFT-512 = 8 x 8 x 8 */
static inline fxr_c32 cadd(fxr_c32 a, fxr_c32 b) { fxr_c32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline fxr_c32 csub(fxr_c32 a, fxr_c32 b) { fxr_c32 r = { a.re - b.re, a.im - b.im }; return r; }
/* a * w */
static inline fxr_c32 cmul(fxr_c32 a, fxr_c32 w)
{
    float t = a.im * w.im, u = a.im * w.re;
    fxr_c32 r = { fmaf(a.re, w.re, -t), fmaf(a.re, w.im, u) };
    return r;
}

#define FXR_C8 0.70710678118654752f

/* DFT4 with sign -: out[k] = sum c[n] (-j)^(nk) */
static inline void dft4(fxr_c32 c0, fxr_c32 c1, fxr_c32 c2, fxr_c32 c3, fxr_c32 *o0, fxr_c32 *o1, fxr_c32 *o2, fxr_c32 *o3)
{
    fxr_c32 d0 = cadd(c0, c2), d1 = cadd(c1, c3), d2 = csub(c0, c2), e = csub(c1, c3);
    fxr_c32 d3 = { e.im, -e.re };               /* (c1-c3) * (-j) */
    *o0 = cadd(d0, d1); *o2 = csub(d0, d1);
    *o1 = cadd(d2, d3); *o3 = csub(d2, d3);
}

/* in-place 8-point forward DFT, natural order in, natural order out */
static inline void dft8(fxr_c32 a[8])
{
    fxr_c32 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    fxr_c32 b1 = cadd(a[1], a[5]), t5 = csub(a[1], a[5]);
    fxr_c32 b2 = cadd(a[2], a[6]), t6 = csub(a[2], a[6]);
    fxr_c32 b3 = cadd(a[3], a[7]), t7 = csub(a[3], a[7]);
    fxr_c32 b5 = { (t5.re + t5.im) * FXR_C8, (t5.im - t5.re) * FXR_C8 };     /* * W8^1 */
    fxr_c32 b6 = { t6.im, -t6.re };                                          /* * W8^2 */
    fxr_c32 b7 = { (t7.im - t7.re) * FXR_C8, -((t7.re + t7.im) * FXR_C8) };  /* * W8^3 */
    dft4(b0, b1, b2, b3, &a[0], &a[2], &a[4], &a[6]);
    dft4(b4, b5, b6, b7, &a[1], &a[3], &a[5], &a[7]);
}

void fxr_fft512(const fxr_c32 *in, fxr_c32 *out)
{
    fxr_c32 A[8][64], B[8][8][8], a[8];
    fxr_math_init_();
    for (int j = 0; j < 64; j++) {              /* stage A: DFT8 over q, twiddle W512^(j r) */
        for (int q = 0; q < 8; q++) a[q] = in[j + 64 * q];
        dft8(a);
        A[0][j] = a[0];
        for (int r = 1; r < 8; r++) A[r][j] = cmul(a[r], g_tw[j * r]);
    }
    for (int r = 0; r < 8; r++)                 /* stage B: DFT8 over p, twiddle W64^(j0 s) */
        for (int j0 = 0; j0 < 8; j0++) {
            for (int p = 0; p < 8; p++) a[p] = A[r][j0 + 8 * p];
            dft8(a);
            B[r][0][j0] = a[0];
            for (int s = 1; s < 8; s++) B[r][s][j0] = cmul(a[s], g_tw[8 * j0 * s]);
        }
    for (int r = 0; r < 8; r++)                 /* stage C: DFT8 over j0 */
        for (int s = 0; s < 8; s++) {
            for (int j0 = 0; j0 < 8; j0++) a[j0] = B[r][s][j0];
            dft8(a);
            for (int t = 0; t < 8; t++) out[r + 8 * s + 64 * t] = a[t];
        }
}

void fxr_ifft512(const fxr_c32 *in, fxr_c32 *out)
{
    fxr_c32 t[512];
    for (int i = 0; i < 512; i++) { t[i].re = in[i].im; t[i].im = in[i].re; }
    fxr_fft512(t, out);
    for (int i = 0; i < 512; i++) { float r = out[i].re; out[i].re = out[i].im; out[i].im = r; }
}
