/* oracle_sanitize.c -- AddressSanitizer / UndefinedBehaviorSanitizer run of the CPU oracle (tests/test_sanitizers.py builds it
 * together with oracle/fxref_*.c under -fsanitize=address,undefined and runs it; no GPU sanitizers exist on this pool).
 * Exercises every modulation, every FEC in both positions, every CRC, empty and maximum payloads, the detector, the
 * synchroniser in odd chunkings, the equaliser and soft-decision options -- on clean and on noisy input. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../oracle/fxref.h"

static unsigned long long rng_s = 88172645463325252ull;
static unsigned rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (unsigned)(rng_s >> 32); }
static float gauss(void) { float u1 = ((rnd() >> 8) + 1.0f) / 16777216.0f, u2 = (rnd() >> 8) / 16777216.0f; return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); }

static unsigned n_frames, n_valid, n_bytes_ok;
static const unsigned char *want; static unsigned want_len;
static int cb(unsigned char *h, int hv, unsigned char *p, unsigned n, int pv, fxr_stats st, void *ud)
{
    (void)h; (void)ud;
    n_frames++; n_valid += (unsigned)(hv && pv);
    if (hv && pv && n == want_len && (n == 0 || memcmp(p, want, n) == 0)) n_bytes_ok++;
    if (hv && st.num_framesyms) { volatile float t = st.framesyms[st.num_framesyms - 1].re; (void)t; }
    return 0;
}

static int one(int ms, int check, int fec0, int fec1, unsigned plen, float sigma, unsigned chunk, int eq, int soft)
{
    fxr_genprops pr = { check, fec0, fec1, ms };
    unsigned char *pl = (unsigned char *)malloc(plen + 1), hd[FXR_HDR_USER];
    for (unsigned i = 0; i < plen; i++) pl[i] = (unsigned char)rnd();
    for (unsigned i = 0; i < FXR_HDR_USER; i++) hd[i] = (unsigned char)rnd();
    const unsigned fl = fxr_gen_frame_len(&pr, plen), lead = 300 + rnd() % 700, tail = 1200;
    const unsigned n = lead + fl + tail;
    fxr_c32 *x = (fxr_c32 *)calloc(n, sizeof(fxr_c32));
    if (fxr_gen_frame(&pr, hd, pl, plen, 0.37f * ((int)(rnd() % 200) - 100) / 100.0f, x + lead) != fl) { printf("frame length mismatch\n"); return 1; }
    const float cfo = 0.04f * ((int)(rnd() % 200) - 100) / 100.0f, ph = 0.01f * (rnd() % 600);
    for (unsigned i = 0; i < n; i++) {
        const float c = cosf(cfo * i + ph), s = sinf(cfo * i + ph), re = x[i].re, im = x[i].im;
        x[i].re = re * c - im * s + sigma * gauss(); x[i].im = re * s + im * c + sigma * gauss();
    }
    n_frames = n_valid = n_bytes_ok = 0; want = pl; want_len = plen;
    fxr_sync *q = fxr_sync_create(cb, NULL);
    if (eq) fxr_sync_set_equalizer(q, 1);
    if (soft) fxr_sync_set_soft(q, 1);
    fxr_sync_execute_chunked(q, x, n, chunk);
    fxr_sync_reset(q);
    fxr_sync_execute(q, x, n / 2);                       /* a frame cut short, then destroyed mid-frame */
    fxr_sync_destroy(q);
    int bad = sigma < 0.05f && n_bytes_ok < 1;
    if (bad) printf("FAIL ms %d check %d fec %d/%d len %u chunk %u eq %d soft %d: frames %u valid %u\n", ms, check, fec0, fec1, plen, chunk, eq, soft, n_frames, n_valid);
    free(x); free(pl);
    return bad;
}

int main(void)
{
    fxr_init();
    static const int mods[] = { 1, 2, 3, 4, 9, 10, 11, 18, 27, 28, 29, 40 };
    static const int fecs[] = { 1, 4, 5, 6, 7, 8, 9, 10, 11, 15, 16, 17, 18, 19, 20, 27 };
    static const unsigned chunks[] = { 256, 1, 7, 255, 257, 4096, 100000 };
    int bad = 0, runs = 0;
    for (unsigned m = 0; m < sizeof mods / sizeof *mods; m++)                       /* every modulation */
        { bad += one(mods[m], FXR_CRC_24, FXR_FEC_CONV_V27, FXR_FEC_NONE, 64 + m, 0.01f, chunks[m % 7], 0, 0); runs++; }
    for (unsigned f = 0; f < sizeof fecs / sizeof *fecs; f++) {                     /* every code, inner and outer position */
        bad += one(FXR_MODEM_PSK4, FXR_CRC_32, fecs[f], FXR_FEC_NONE, 100 + 3 * f, 0.01f, 256, 0, 0);
        bad += one(FXR_MODEM_QAM16, FXR_CRC_16, FXR_FEC_CONV_V27P23, fecs[f] == 11 ? FXR_FEC_HAMMING128 : fecs[f], 77 + f, 0.01f, 512, 0, 0); runs += 2;
    }
    for (int c = FXR_CRC_NONE; c <= FXR_CRC_32; c++) { bad += one(FXR_MODEM_PSK8, c, FXR_FEC_GOLAY2412, FXR_FEC_NONE, 31, 0.01f, 256, 0, 0); runs++; }
    bad += one(FXR_MODEM_PSK4, FXR_CRC_24, FXR_FEC_CONV_V27, FXR_FEC_NONE, 0, 0.01f, 256, 0, 0);           /* empty payload */
    bad += one(FXR_MODEM_QAM64, FXR_CRC_32, FXR_FEC_NONE, FXR_FEC_NONE, 65535, 0.002f, 256, 0, 0);          /* maximum payload */
    bad += one(FXR_MODEM_PSK2, FXR_CRC_24, FXR_FEC_CONV_V27P78, FXR_FEC_RS_M8, 4000, 0.01f, 256, 0, 0);
    bad += one(FXR_MODEM_PSK4, FXR_CRC_24, FXR_FEC_CONV_V27, FXR_FEC_NONE, 500, 0.01f, 256, 1, 0);          /* equaliser */
    bad += one(FXR_MODEM_QAM16, FXR_CRC_24, FXR_FEC_CONV_V27P23, FXR_FEC_CONV_V27, 300, 0.01f, 256, 0, 1);  /* soft decisions */
    bad += one(FXR_MODEM_QAM32, FXR_CRC_24, FXR_FEC_CONV_V27P56, FXR_FEC_SECDED3932, 300, 0.01f, 256, 1, 1);
    runs += 6;
    for (int k = 0; k < 12; k++) { (void)one(mods[rnd() % 12], FXR_CRC_24, fecs[rnd() % 16], fecs[rnd() % 16], rnd() % 700, 0.5f, chunks[k % 7], k & 1, (k >> 1) & 1); runs++; }   /* noise: broken headers, failing payloads */
    /* the bare detector, sample by sample, with resets in between */
    {
        fxr_genprops pr = { FXR_CRC_24, FXR_FEC_CONV_V27, FXR_FEC_NONE, FXR_MODEM_PSK4 };
        unsigned char pl[40] = { 0 }, hd[FXR_HDR_USER] = { 0 };
        const unsigned fl = fxr_gen_frame_len(&pr, 40), n = 3 * (fl + 400) + 1000;
        fxr_c32 *x = (fxr_c32 *)calloc(n, sizeof(fxr_c32));
        for (int k = 0; k < 3; k++) fxr_gen_frame(&pr, hd, pl, 40, 0.0f, x + 200 + k * (fl + 400));
        for (unsigned i = 0; i < n; i++) { x[i].re += 0.02f * gauss(); x[i].im += 0.02f * gauss(); }
        fxr_qdet *d = fxr_qdet_create_flexframe(); fxr_qdet_set_threshold(d, 0.45f);
        fxr_detection det[16];
        unsigned nd = fxr_qdet_run(d, x, n, 0, det, 16);
        fxr_qdet_reset(d); nd += fxr_qdet_run(d, x + 100, n - 100, 100, det, 2);   /* output capacity smaller than the detections */
        if (nd < 6) { printf("FAIL detector: %u detections\n", nd); bad++; }
        fxr_qdet_destroy(d); free(x); runs++;
    }
    printf("oracle under ASan/UBSan: %d runs, %d failures\n", runs, bad);
    return bad ? 1 : 0;
}
