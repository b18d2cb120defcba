/*
 * fxref_fec.c -- CPU ORACLE (test infrastructure; see fxref.h).  PARITY UNPINNED.
 *
 * Packet coding chain behind qpacketmodem_decode / packetizer_decode, which liquid's
 * flexframesync runs on the header and on the payload ([RECALLED liquid-dsp v1.3.x packetizer.c,
 * interleaver.c, crc.c, scramble.c, fec_conv.c + libfec viterbi27, fec_hamming84.c,
 * fec_secded7264.c]; reached from /root/reference/lib/flex_rx_impl.cc:213).  The FEC / CRC menu
 * the reference selects from is at /root/reference/lib/flex_tx_impl.cc:52,118-181 and
 * /root/reference/lib/flex_rx_impl.cc:74-136.
 *
 * Restatement notes (all inside "parity unpinned"):
 *  - CRCs are the clean reflected forms (register masked to its width); liquid keeps a 32-bit
 *    register for the 8/16/24-bit keys.
 *  - Hamming(7,4)/(8,4)/(12,8), Golay(24,12) and SECDED(22,16)/(39,32)/(72,64) use textbook constructions
 *    (systematic / positional Hamming, cyclic Golay with g = 0xC75 + overall parity, Hsiao odd-weight columns),
 *    not liquid's literal generator tables, which are not recalled.  Reed-Solomon RS_M8 = RS(255,223), GF(2^8)/0x11d,
 *    roots alpha^1..alpha^32, shortened per block as liquid's fec_rs does.
 *  - Viterbi: K=7 (0x6d,0x4f), hard decisions, Hamming branch metric, punctured positions cost 0,
 *    ties keep the lower predecessor, traceback from state 0.  libfec's 8-bit soft metric on
 *    0/255 inputs selects the same path except on exact metric ties.
 */
#include "fxref.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---------------------------------------------------------------- CRC / checksum */
unsigned fxr_crc_len(int check)
{
    switch (check) {
    case FXR_CRC_NONE: return 0;
    case FXR_CRC_CHECKSUM: case FXR_CRC_8: return 1;
    case FXR_CRC_16: return 2;
    case FXR_CRC_24: return 3;
    case FXR_CRC_32: return 4;
    default: return 0;
    }
}

static uint32_t bitrev(uint32_t v, unsigned w)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < w; i++) if (v & (1u << i)) r |= 1u << (w - 1 - i);
    return r;
}

static uint32_t crc_reflected(uint32_t poly, unsigned w, const uint8_t *msg, unsigned n)
{
    uint32_t mask = w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
    uint32_t p = bitrev(poly, w), key = mask;
    for (unsigned i = 0; i < n; i++) {
        key ^= msg[i];
        for (int j = 0; j < 8; j++) key = (key >> 1) ^ (p & (0u - (key & 1u)));
    }
    return (~key) & mask;
}

uint32_t fxr_crc_key(int check, const uint8_t *msg, unsigned n)
{
    switch (check) {
    case FXR_CRC_CHECKSUM: { uint32_t s = 0; for (unsigned i = 0; i < n; i++) s += msg[i]; return (~s + 1u) & 0xff; }
    case FXR_CRC_8:  return crc_reflected(0x07, 8, msg, n);
    case FXR_CRC_16: return crc_reflected(0x8005, 16, msg, n);
    case FXR_CRC_24: return crc_reflected(0x5D6DCB, 24, msg, n);
    case FXR_CRC_32: return crc_reflected(0x04C11DB7, 32, msg, n);
    default: return 0;
    }
}

/* ---------------------------------------------------------------- whitening */
void fxr_scramble(uint8_t *x, unsigned n)
{
    static const uint8_t m[4] = { 0xb4, 0x6a, 0x8b, 0xc5 };
    for (unsigned i = 0; i < n; i++) x[i] ^= m[i & 3];
}

/* ---------------------------------------------------------------- interleaver (depth 4) */
static void ilv_dims(unsigned n, unsigned *M, unsigned *N)
{
    *M = 1 + (unsigned)floorf(sqrtf((float)n));
    *N = n / *M;
    while (n >= (*M) * (*N)) (*N)++;
}

/* one pass of disjoint swaps between even bytes 2i and odd bytes 2j+1 under a bit mask */
static void ilv_pass(uint8_t *x, unsigned n, unsigned M, unsigned N, uint8_t mask)
{
    unsigned m = 0, nn = n / 3, n2 = n / 2, j;
    for (unsigned i = 0; i < n2; i++) {
        do {
            j = m * N + nn;
            m++;
            if (m == M) { nn = (nn + 1) % N; m = 0; }
        } while (j >= n2);
        uint8_t a = x[2 * i], b = x[2 * j + 1];
        x[2 * i]     = (uint8_t)((a & ~mask) | (b & mask));
        x[2 * j + 1] = (uint8_t)((a & mask) | (b & ~mask));
    }
}

void fxr_interleave(uint8_t *x, unsigned n, int decode)
{
    unsigned M, N; ilv_dims(n, &M, &N);
    if (!decode) {
        ilv_pass(x, n, M, N, 0xff);
        ilv_pass(x, n, M, N + 2, 0x0f);
        ilv_pass(x, n, M, N + 4, 0x55);
        ilv_pass(x, n, M, N + 8, 0x33);
    } else {
        ilv_pass(x, n, M, N + 8, 0x33);
        ilv_pass(x, n, M, N + 4, 0x55);
        ilv_pass(x, n, M, N + 2, 0x0f);
        ilv_pass(x, n, M, N, 0xff);
    }
}

/* ---------------------------------------------------------------- Hamming(8,4) */
static uint8_t h84_enc[16], h84_dec[256];
static int h84_ready = 0;
static void h84_init(void)
{
    if (h84_ready) return;
    /* [RECALLED -- confident] liquid's fec_hamming84.c generator table: bits [p1 p2 d1 p4 d2 d3 d4 p8], MSB first, d1 the
     * nibble's MSB; p1 = d1^d2^d4, p2 = d1^d3^d4, p4 = d2^d3^d4, p8 = overall (even) parity.  The table below is that rule
     * evaluated, and equals the sixteen literals as recalled (00 d2 55 87 99 4b cc 1e e1 33 b4 66 78 aa 2d ff). */
    for (unsigned d = 0; d < 16; d++) {
        unsigned d1 = (d >> 3) & 1, d2 = (d >> 2) & 1, d3 = (d >> 1) & 1, d4 = d & 1;
        unsigned p1 = d1 ^ d2 ^ d4, p2 = d1 ^ d3 ^ d4, p4 = d2 ^ d3 ^ d4;
        unsigned c = (p1 << 7) | (p2 << 6) | (d1 << 5) | (p4 << 4) | (d2 << 3) | (d3 << 2) | (d4 << 1);
        c |= (unsigned)__builtin_popcount(c) & 1u;
        h84_enc[d] = (uint8_t)c;
    }
    for (unsigned r = 0; r < 256; r++) {
        unsigned best = 0, bd = 9;
        for (unsigned d = 0; d < 16; d++) {
            unsigned dist = (unsigned)__builtin_popcount(r ^ h84_enc[d]);
            if (dist < bd) { bd = dist; best = d; }
        }
        h84_dec[r] = (uint8_t)best;
    }
    h84_ready = 1;
}

/* ---------------------------------------------------------------- SECDED(72,64), Hsiao columns */
static uint8_t sd_col[64];
static int sd_ready = 0;
static void sd_init(void)
{
    if (sd_ready) return;
    unsigned n = 0;
    for (unsigned v = 1; v < 256 && n < 56; v++) if (__builtin_popcount(v) == 3) sd_col[n++] = (uint8_t)v;
    for (unsigned v = 1; v < 256 && n < 64; v++) if (__builtin_popcount(v) == 5) sd_col[n++] = (uint8_t)v;
    sd_ready = 1;
}
static uint8_t sd_parity(const uint8_t d[8])
{
    uint8_t p = 0;
    for (unsigned j = 0; j < 64; j++) if (d[j >> 3] & (0x80u >> (j & 7))) p ^= sd_col[j];
    return p;
}
static void sd_decode_block(const uint8_t e[9], uint8_t d[8])
{
    memcpy(d, e + 1, 8);
    uint8_t s = (uint8_t)(e[0] ^ sd_parity(d));
    if (s == 0 || __builtin_popcount(s) == 1) return;   /* clean, or the parity byte took the hit */
    for (unsigned j = 0; j < 64; j++)
        if (sd_col[j] == s) { d[j >> 3] ^= (uint8_t)(0x80u >> (j & 7)); return; }
    /* double error: detected, left uncorrected */
}

/* ---------------------------------------------------------------- SECDED(22,16) and SECDED(39,32): same shape as (72,64) */
static uint8_t sd22_col[16], sd39_col[32];
static int sdx_ready = 0;
static void sdx_init(void)
{
    if (sdx_ready) return;
    unsigned n = 0;
    for (unsigned v = 1; v < 64 && n < 16; v++) if (__builtin_popcount(v) == 3) sd22_col[n++] = (uint8_t)v;
    n = 0;
    for (unsigned v = 1; v < 128 && n < 32; v++) if (__builtin_popcount(v) == 3) sd39_col[n++] = (uint8_t)v;
    sdx_ready = 1;
}
/* nd data bytes, one leading parity byte holding a (cols-wide) Hsiao parity */
static uint8_t sdx_parity(const uint8_t *col, const uint8_t *d, unsigned nd)
{
    uint8_t p = 0;
    for (unsigned j = 0; j < 8 * nd; j++) if (d[j >> 3] & (0x80u >> (j & 7))) p ^= col[j];
    return p;
}
static void sdx_encode(const uint8_t *col, unsigned nd, unsigned n, const uint8_t *dec, uint8_t *enc)
{
    unsigned i = 0, j = 0;
    for (; i + nd <= n; i += nd, j += nd + 1) { enc[j] = sdx_parity(col, dec + i, nd); memcpy(enc + j + 1, dec + i, nd); }
    if (n % nd) { uint8_t d[8] = { 0 }; memcpy(d, dec + i, n % nd); enc[j] = sdx_parity(col, d, nd); memcpy(enc + j + 1, d, n % nd); }
}
static void sdx_decode_block(const uint8_t *col, unsigned nd, const uint8_t *e, uint8_t *d)
{
    memcpy(d, e + 1, nd);
    uint8_t s = (uint8_t)(e[0] ^ sdx_parity(col, d, nd));
    if (s == 0 || __builtin_popcount(s) == 1) return;
    for (unsigned j = 0; j < 8 * nd; j++) if (col[j] == s) { d[j >> 3] ^= (uint8_t)(0x80u >> (j & 7)); return; }
}
static void sdx_decode(const uint8_t *col, unsigned nd, unsigned n, const uint8_t *enc, uint8_t *dec)
{
    unsigned i = 0, j = 0;
    for (; i + nd <= n; i += nd, j += nd + 1) sdx_decode_block(col, nd, enc + j, dec + i);
    if (n % nd) { uint8_t e[9] = { 0 }, d[8]; memcpy(e, enc + j, n % nd + 1); sdx_decode_block(col, nd, e, d); memcpy(dec + i, d, n % nd); }
}

/* ---------------------------------------------------------------- bit-packed block codes: Hamming(7,4), Hamming(12,8), Golay(24,12)
 * k-bit blocks are taken MSB first from the byte stream (last one zero padded), each becomes an n-bit codeword,
 * codewords are written MSB first back to back.  Decoding is nearest-codeword through syndrome tables. */
static uint8_t  h74_enc[16], h74_dec[128];
static uint16_t h128_enc[256]; static uint8_t h128_dec[4096];
static uint32_t gol_enc[4096], gol_err[4096];            /* codeword of a 12-bit word; error pattern of a syndrome */
static int blk_ready = 0;
static unsigned gol_syndrome(uint32_t cw)                 /* parity part recomputed from the data part, xor received parity */
{
    return (unsigned)((gol_enc[(cw >> 12) & 0xfff] ^ cw) & 0xfff);
}
static void blk_init(void)
{
    if (blk_ready) return;
    for (unsigned d = 0; d < 16; d++) {                    /* [RECALLED -- confident] liquid's fec_hamming74.c table: [p1 p2 d1 p4 d2 d3 d4] (00 69 2a 43 4c 25 66 0f 70 19 5a 33 3c 55 16 7f) */
        unsigned d1 = (d >> 3) & 1, d2 = (d >> 2) & 1, d3 = (d >> 1) & 1, d4 = d & 1;
        h74_enc[d] = (uint8_t)(((d1 ^ d2 ^ d4) << 6) | ((d1 ^ d3 ^ d4) << 5) | (d1 << 4) | ((d2 ^ d3 ^ d4) << 3) | (d2 << 2) | (d3 << 1) | d4);
    }
    for (unsigned r = 0; r < 128; r++) { unsigned best = 0, bd = 99; for (unsigned d = 0; d < 16; d++) { unsigned w = (unsigned)__builtin_popcount(r ^ h74_enc[d]); if (w < bd) { bd = w; best = d; } } h74_dec[r] = (uint8_t)best; }
    for (unsigned d = 0; d < 256; d++) {                   /* (12,8): data bits at positions 3,5,6,7,9,10,11,12 (1-based), parity at 1,2,4,8 */
        static const unsigned pos[8] = { 3, 5, 6, 7, 9, 10, 11, 12 };
        unsigned cw = 0;
        for (unsigned i = 0; i < 8; i++) if (d & (0x80u >> i)) cw |= 1u << (12 - pos[i]);
        for (unsigned pb = 1; pb <= 8; pb <<= 1) {
            unsigned par = 0;
            for (unsigned q = 1; q <= 12; q++) if ((q & pb) && (cw & (1u << (12 - q)))) par ^= 1;
            if (par) cw |= 1u << (12 - pb);
        }
        h128_enc[d] = (uint16_t)cw;
    }
    for (unsigned r = 0; r < 4096; r++) { unsigned best = 0, bd = 99; for (unsigned d = 0; d < 256; d++) { unsigned w = (unsigned)__builtin_popcount(r ^ h128_enc[d]); if (w < bd) { bd = w; best = d; } } h128_dec[r] = (uint8_t)best; }
    for (unsigned d = 0; d < 4096; d++) {                  /* extended Golay: 12 data bits, 11 CRC-like parity bits (g = 0xC75), overall parity */
        uint32_t reg = d << 11;
        for (int i = 22; i >= 11; i--) if (reg & (1u << i)) reg ^= 0xC75u << (i - 11);
        uint32_t cw23 = (d << 11) | (reg & 0x7ff);
        gol_enc[d] = (cw23 << 1) | ((uint32_t)__builtin_popcount(cw23) & 1u);
    }
    for (unsigned i = 0; i < 4096; i++) gol_err[i] = 0xFFFFFFFFu;
    /* all error patterns of weight <= 3 over 24 bits, lowest weight first; first writer of a syndrome wins */
    gol_err[0] = 0;
    for (int w = 1; w <= 3; w++)
        for (int a = 0; a < 24; a++) for (int b = (w >= 2 ? a + 1 : 24); b <= 24; b++) for (int c = (w >= 3 ? b + 1 : 24); c <= 24; c++) {
            if ((w >= 2 && b >= 24) || (w >= 3 && c >= 24)) continue;
            uint32_t e = 1u << a; if (w >= 2) e |= 1u << b; if (w >= 3) e |= 1u << c;
            unsigned syn = gol_syndrome(e);
            if (gol_err[syn] == 0xFFFFFFFFu) gol_err[syn] = e;
            if (w < 3) break;
        }
    blk_ready = 1;
}
static inline unsigned take_bits(const uint8_t *b, unsigned nbytes, unsigned pos, unsigned n)
{
    unsigned v = 0;
    for (unsigned i = 0; i < n; i++) { unsigned k = pos + i; v = (v << 1) | (k < 8 * nbytes ? (b[k >> 3] >> (7 - (k & 7))) & 1u : 0u); }
    return v;
}
static inline void put_bits(uint8_t *b, unsigned nbytes, unsigned pos, unsigned n, unsigned v)
{
    for (unsigned i = 0; i < n; i++) { unsigned k = pos + i; if (k < 8 * nbytes && ((v >> (n - 1 - i)) & 1u)) b[k >> 3] |= (uint8_t)(0x80u >> (k & 7)); }
}
static int blk_spec(int fs, unsigned *k, unsigned *n)
{
    switch (fs) {
    case FXR_FEC_HAMMING74: *k = 4; *n = 7; return 1;
    case FXR_FEC_HAMMING128: *k = 8; *n = 12; return 1;
    case FXR_FEC_GOLAY2412: *k = 12; *n = 24; return 1;
    default: return 0;
    }
}
static unsigned blk_enc_len(unsigned k, unsigned n, unsigned dec_len) { unsigned nb = (8 * dec_len + k - 1) / k; return (nb * n + 7) / 8; }
static void blk_encode(int fs, unsigned dec_len, const uint8_t *dec, uint8_t *enc)
{
    unsigned k, n; blk_spec(fs, &k, &n); blk_init();
    unsigned nb = (8 * dec_len + k - 1) / k, el = blk_enc_len(k, n, dec_len);
    memset(enc, 0, el);
    for (unsigned j = 0; j < nb; j++) {
        unsigned d = take_bits(dec, dec_len, j * k, k);
        unsigned cw = fs == FXR_FEC_HAMMING74 ? h74_enc[d] : fs == FXR_FEC_HAMMING128 ? h128_enc[d] : gol_enc[d];
        put_bits(enc, el, j * n, n, cw);
    }
}
static void blk_decode(int fs, unsigned dec_len, const uint8_t *enc, uint8_t *dec)
{
    unsigned k, n; blk_spec(fs, &k, &n); blk_init();
    unsigned nb = (8 * dec_len + k - 1) / k, el = blk_enc_len(k, n, dec_len);
    memset(dec, 0, dec_len);
    for (unsigned j = 0; j < nb; j++) {
        unsigned r = take_bits(enc, el, j * n, n), d;
        if (fs == FXR_FEC_HAMMING74) d = h74_dec[r];
        else if (fs == FXR_FEC_HAMMING128) d = h128_dec[r];
        else { uint32_t e = gol_err[gol_syndrome(r)]; if (e != 0xFFFFFFFFu) r ^= e; d = (r >> 12) & 0xfff; }
        put_bits(dec, dec_len, j * k, k, d);
    }
}

/* ---------------------------------------------------------------- Reed-Solomon RS(255,223) over GF(2^8), shortened per block
 * [RECALLED liquid fec_rs.c + libfec init_rs_char(8, 0x11d, fcr=1, prim=1, nroots=32)]: the message is cut into
 * nb = ceil(n/223) blocks of dl = ceil(n/nb) bytes (last one zero padded), each followed by 32 parity bytes.
 * Decoder: syndromes, Berlekamp-Massey, Chien search, Forney; a block with more than 16 byte errors is left as is. */
static uint8_t rs_exp[512], rs_log[256], rs_gen[33];
static int rs_ready = 0;
static inline uint8_t gmul(uint8_t a, uint8_t b) { return (a && b) ? rs_exp[rs_log[a] + rs_log[b]] : 0; }
static inline uint8_t gdiv(uint8_t a, uint8_t b) { return a ? rs_exp[rs_log[a] + 255 - rs_log[b]] : 0; }
static void rs_init(void)
{
    if (rs_ready) return;
    unsigned x = 1;
    for (unsigned i = 0; i < 255; i++) { rs_exp[i] = (uint8_t)x; rs_log[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11d; }
    for (unsigned i = 255; i < 512; i++) rs_exp[i] = rs_exp[i - 255];
    rs_log[0] = 0;
    memset(rs_gen, 0, sizeof rs_gen); rs_gen[0] = 1;                     /* g(x) = prod (x - alpha^i), i = 1..32; rs_gen[k] = coeff of x^k */
    for (unsigned i = 1; i <= 32; i++) {
        uint8_t root = rs_exp[i];
        for (int k = (int)i; k > 0; k--) rs_gen[k] = (uint8_t)(rs_gen[k - 1] ^ gmul(rs_gen[k], root));
        rs_gen[0] = gmul(rs_gen[0], root);
    }
    rs_ready = 1;
}
static void rs_dims(unsigned n, unsigned *nb, unsigned *dl)
{
    *nb = (n + 222) / 223; if (*nb == 0) *nb = 1;
    *dl = (n + *nb - 1) / *nb;
}
static void rs_encode_block(const uint8_t *d, unsigned dl, uint8_t *out /* dl + 32 */)
{
    uint8_t par[32]; memset(par, 0, 32);                                 /* par[31] is the highest-order remainder term */
    for (unsigned i = 0; i < dl; i++) {
        uint8_t fb = (uint8_t)(d[i] ^ par[31]);
        for (int k = 31; k > 0; k--) par[k] = (uint8_t)(par[k - 1] ^ gmul(fb, rs_gen[k]));
        par[0] = gmul(fb, rs_gen[0]);
    }
    memcpy(out, d, dl);
    for (unsigned k = 0; k < 32; k++) out[dl + k] = par[31 - k];
}
static void rs_decode_block(uint8_t *r, unsigned N /* dl + 32 */)
{
    uint8_t S[32]; int nz = 0;
    for (unsigned i = 0; i < 32; i++) {
        uint8_t a = rs_exp[i + 1], s = 0;
        for (unsigned j = 0; j < N; j++) s = (uint8_t)(gmul(s, a) ^ r[j]);
        S[i] = s; nz |= s;
    }
    if (!nz) return;
    uint8_t L_[33], B[33], T[33]; memset(L_, 0, 33); memset(B, 0, 33); L_[0] = B[0] = 1;
    unsigned L = 0, m = 1; uint8_t b = 1;
    for (unsigned k = 0; k < 32; k++) {                                   /* Berlekamp-Massey */
        uint8_t d = S[k];
        for (unsigned i = 1; i <= L; i++) d ^= gmul(L_[i], S[k - i]);
        if (d == 0) { m++; continue; }
        memcpy(T, L_, 33);
        uint8_t coef = gdiv(d, b);
        for (unsigned i = 0; i + m <= 32; i++) L_[i + m] ^= gmul(coef, B[i]);
        if (2 * L <= k) { L = k + 1 - L; memcpy(B, T, 33); b = d; m = 1; } else m++;
    }
    if (L > 16) return;
    uint8_t Om[32];                                                       /* Omega = S * Lambda mod x^32 */
    for (unsigned i = 0; i < 32; i++) { uint8_t v = 0; for (unsigned j = 0; j <= i && j <= L; j++) v ^= gmul(L_[j], S[i - j]); Om[i] = v; }
    unsigned pos[16], np = 0; uint8_t val[16];
    for (unsigned j = 0; j < N; j++) {                                    /* Chien: symbol j has locator X = alpha^(N-1-j) */
        unsigned e = (N - 1 - j) % 255, inv = (255 - e) % 255;            /* X^-1 = alpha^inv */
        uint8_t v = 0;
        for (unsigned i = 0; i <= L; i++) v ^= gmul(L_[i], rs_exp[(inv * i) % 255]);
        if (v) continue;
        if (np == 16) return;
        uint8_t num = 0, den = 0;
        for (unsigned i = 0; i < 32; i++) num ^= gmul(Om[i], rs_exp[(inv * i) % 255]);
        for (unsigned i = 1; i <= L; i += 2) den ^= gmul(L_[i], rs_exp[(inv * (i - 1)) % 255]);   /* formal derivative */
        if (den == 0) return;
        pos[np] = j; val[np] = gdiv(num, den); np++;
    }
    if (np != L) return;                                                  /* locator does not split: uncorrectable */
    for (unsigned i = 0; i < np; i++) r[pos[i]] ^= val[i];
}
static unsigned rs_enc_len(unsigned n) { unsigned nb, dl; rs_dims(n, &nb, &dl); return nb * (dl + 32); }
static void rs_encode(unsigned n, const uint8_t *dec, uint8_t *enc)
{
    unsigned nb, dl; rs_dims(n, &nb, &dl); rs_init();
    for (unsigned b = 0; b < nb; b++) {
        uint8_t d[223]; memset(d, 0, sizeof d);
        unsigned off = b * dl, len = off < n ? (n - off < dl ? n - off : dl) : 0;
        memcpy(d, dec + off, len);
        rs_encode_block(d, dl, enc + b * (dl + 32));
    }
}
static void rs_decode(unsigned n, const uint8_t *enc, uint8_t *dec)
{
    unsigned nb, dl; rs_dims(n, &nb, &dl); rs_init();
    for (unsigned b = 0; b < nb; b++) {
        uint8_t r[255]; memcpy(r, enc + b * (dl + 32), dl + 32);
        rs_decode_block(r, dl + 32);
        unsigned off = b * dl, len = off < n ? (n - off < dl ? n - off : dl) : 0;
        memcpy(dec + off, r, len);
    }
}

/* ---------------------------------------------------------------- convolutional K=7 r=1/2 (+puncturing) */
#define V27_A 0x6d
#define V27_B 0x4f
typedef struct { unsigned p; uint8_t a[8], b[8]; } punc_t;
static const punc_t *punc_of(int fs)
{
    static const punc_t none = { 1, {1}, {1} };
    static const punc_t p23  = { 2, {1,1}, {1,0} };
    static const punc_t p34  = { 3, {1,1,0}, {1,0,1} };
    static const punc_t p45  = { 4, {1,1,1,1}, {1,0,0,0} };
    static const punc_t p56  = { 5, {1,1,0,1,0}, {1,0,1,0,1} };
    static const punc_t p67  = { 6, {1,1,1,0,1,0}, {1,0,0,1,0,1} };
    static const punc_t p78  = { 7, {1,1,1,1,0,1,0}, {1,0,0,0,1,0,1} };
    switch (fs) {
    case FXR_FEC_CONV_V27: return &none;
    case FXR_FEC_CONV_V27P23: return &p23;
    case FXR_FEC_CONV_V27P34: return &p34;
    case FXR_FEC_CONV_V27P45: return &p45;
    case FXR_FEC_CONV_V27P56: return &p56;
    case FXR_FEC_CONV_V27P67: return &p67;
    case FXR_FEC_CONV_V27P78: return &p78;
    default: return NULL;
    }
}
static inline unsigned par7(unsigned v) { return (unsigned)__builtin_popcount(v) & 1u; }

static unsigned conv_enc_bits(const punc_t *pp, unsigned dec_len)
{
    unsigned n = 8 * dec_len + 6, bits = 0;
    for (unsigned t = 0; t < n; t++) bits += pp->a[t % pp->p] + pp->b[t % pp->p];
    return bits;
}

static void conv_encode(const punc_t *pp, unsigned dec_len, const uint8_t *dec, uint8_t *enc)
{
    unsigned n = 8 * dec_len + 6, sr = 0, nb = 0;
    unsigned enc_len = (conv_enc_bits(pp, dec_len) + 7) / 8;
    memset(enc, 0, enc_len);
    for (unsigned t = 0; t < n; t++) {
        unsigned bit = t < 8 * dec_len ? (dec[t >> 3] >> (7 - (t & 7))) & 1u : 0u;
        sr = ((sr << 1) | bit) & 0x7f;
        unsigned c = t % pp->p;
        if (pp->a[c]) { if (par7(sr & V27_A)) enc[nb >> 3] |= (uint8_t)(0x80u >> (nb & 7)); nb++; }
        if (pp->b[c]) { if (par7(sr & V27_B)) enc[nb >> 3] |= (uint8_t)(0x80u >> (nb & 7)); nb++; }
    }
}

static void conv_decode(const punc_t *pp, unsigned dec_len, const uint8_t *enc, uint8_t *dec)
{
    unsigned n = 8 * dec_len + 6, nb = 0;
    uint64_t *dw = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
    uint32_t pm[64], nm[64];
    for (int s = 0; s < 64; s++) pm[s] = 1u << 24;
    pm[0] = 0;
    for (unsigned t = 0; t < n; t++) {
        unsigned c = t % pp->p;
        int ra = -1, rb = -1;                   /* -1 = punctured (erasure) */
        if (pp->a[c]) { ra = (enc[nb >> 3] >> (7 - (nb & 7))) & 1; nb++; }
        if (pp->b[c]) { rb = (enc[nb >> 3] >> (7 - (nb & 7))) & 1; nb++; }
        uint64_t d = 0;
        for (unsigned s = 0; s < 64; s++) {
            unsigned b = s & 1, p0 = s >> 1, p1 = p0 | 32;
            unsigned sr0 = ((p0 << 1) | b) & 0x7f, sr1 = ((p1 << 1) | b) & 0x7f;
            uint32_t m0 = pm[p0], m1 = pm[p1];
            if (ra >= 0) { m0 += par7(sr0 & V27_A) != (unsigned)ra; m1 += par7(sr1 & V27_A) != (unsigned)ra; }
            if (rb >= 0) { m0 += par7(sr0 & V27_B) != (unsigned)rb; m1 += par7(sr1 & V27_B) != (unsigned)rb; }
            if (m1 < m0) { nm[s] = m1; d |= 1ull << s; } else nm[s] = m0;
        }
        memcpy(pm, nm, sizeof pm);
        dw[t] = d;
    }
    memset(dec, 0, dec_len);
    unsigned s = 0;
    for (unsigned t = n; t-- > 0;) {
        unsigned bit = s & 1;
        if (t < 8 * dec_len && bit) dec[t >> 3] |= (uint8_t)(0x80u >> (t & 7));
        s = (s >> 1) | ((unsigned)((dw[t] >> s) & 1ull) << 5);
    }
    free(dw);
}

/* soft-decision form: one byte per coded bit (0 = surely 0 ... 255 = surely 1).  Branch cost of an expected bit e against
 * a received soft value v: e ? 255 - v : v; punctured positions cost nothing.  Same trellis, same tie rule, same traceback. */
static void conv_decode_soft(const punc_t *pp, unsigned dec_len, const uint8_t *soft, uint8_t *dec)
{
    unsigned n = 8 * dec_len + 6, nb = 0;
    uint64_t *dw = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
    uint32_t pm[64], nm[64];
    for (int s = 0; s < 64; s++) pm[s] = 1u << 28;
    pm[0] = 0;
    for (unsigned t = 0; t < n; t++) {
        unsigned c = t % pp->p;
        int ra = -1, rb = -1;
        if (pp->a[c]) ra = soft[nb++];
        if (pp->b[c]) rb = soft[nb++];
        uint64_t d = 0;
        for (unsigned s = 0; s < 64; s++) {
            unsigned b = s & 1, p0 = s >> 1, p1 = p0 | 32;
            unsigned sr0 = ((p0 << 1) | b) & 0x7f, sr1 = ((p1 << 1) | b) & 0x7f;
            uint32_t m0 = pm[p0], m1 = pm[p1];
            if (ra >= 0) { m0 += par7(sr0 & V27_A) ? 255u - (unsigned)ra : (unsigned)ra; m1 += par7(sr1 & V27_A) ? 255u - (unsigned)ra : (unsigned)ra; }
            if (rb >= 0) { m0 += par7(sr0 & V27_B) ? 255u - (unsigned)rb : (unsigned)rb; m1 += par7(sr1 & V27_B) ? 255u - (unsigned)rb : (unsigned)rb; }
            if (m1 < m0) { nm[s] = m1; d |= 1ull << s; } else nm[s] = m0;
        }
        memcpy(pm, nm, sizeof pm);
        dw[t] = d;
    }
    memset(dec, 0, dec_len);
    unsigned s = 0;
    for (unsigned t = n; t-- > 0;) {
        unsigned bit = s & 1;
        if (t < 8 * dec_len && bit) dec[t >> 3] |= (uint8_t)(0x80u >> (t & 7));
        s = (s >> 1) | ((unsigned)((dw[t] >> s) & 1ull) << 5);
    }
    free(dw);
}

/* the interleaver's swaps on an array of soft bits (8 per byte, MSB first) */
static void ilv_pass_soft(uint8_t *x, unsigned n, unsigned M, unsigned N, uint8_t mask)
{
    unsigned m = 0, nn = n / 3, n2 = n / 2, j;
    for (unsigned i = 0; i < n2; i++) {
        do {
            j = m * N + nn;
            m++;
            if (m == M) { nn = (nn + 1) % N; m = 0; }
        } while (j >= n2);
        for (unsigned k = 0; k < 8; k++)
            if (mask & (0x80u >> k)) { uint8_t t = x[8 * (2 * i) + k]; x[8 * (2 * i) + k] = x[8 * (2 * j + 1) + k]; x[8 * (2 * j + 1) + k] = t; }
    }
}
static void deinterleave_soft(uint8_t *x, unsigned n)
{
    unsigned M, N; ilv_dims(n, &M, &N);
    ilv_pass_soft(x, n, M, N + 8, 0x33); ilv_pass_soft(x, n, M, N + 4, 0x55);
    ilv_pass_soft(x, n, M, N + 2, 0x0f); ilv_pass_soft(x, n, M, N, 0xff);
}
static void soft_to_hard(const uint8_t *soft, unsigned n, uint8_t *out)
{
    for (unsigned j = 0; j < n; j++) {
        unsigned v = 0;
        for (unsigned b = 0; b < 8; b++) v = (v << 1) | (soft[8 * j + b] > 127 ? 1u : 0u);
        out[j] = (uint8_t)v;
    }
}

/* Soft-decision packet decoder: soft holds 8 * l1 values (modified in place).  A convolutional stage decodes from soft
 * values as long as nothing before it made hard decisions: the stage nearest the channel (fec1), and fec0 too when fec1 is
 * FEC_NONE; every other stage takes hard decisions (value > 127) and decodes as fxr_packet_decode does. */
int fxr_packet_decode_soft(unsigned n, int check, int fec0, int fec1, uint8_t *soft, uint8_t *msg)
{
    unsigned cl = fxr_crc_len(check), k = n + cl;
    unsigned l0 = fxr_fec_enc_len(fec0, k), l1 = fxr_fec_enc_len(fec1, l0);
    uint8_t *b0 = (uint8_t *)calloc(l1 + 16, 1), *b1 = (uint8_t *)calloc(l1 + 16, 1);
    const punc_t *p1 = punc_of(fec1), *p0 = punc_of(fec0);
    int still_soft = 0;
    deinterleave_soft(soft, l1);
    if (p1) conv_decode_soft(p1, l0, soft, b1);
    else if (fec1 == FXR_FEC_NONE) still_soft = 1;
    else { soft_to_hard(soft, l1, b0); fxr_fec_decode(fec1, l0, b0, b1); }
    if (still_soft) {
        deinterleave_soft(soft, l0);
        if (p0) conv_decode_soft(p0, k, soft, b0);
        else { soft_to_hard(soft, l0, b1); fxr_fec_decode(fec0, k, b1, b0); }
    } else {
        fxr_interleave(b1, l0, 1); fxr_fec_decode(fec0, k, b1, b0);
    }
    fxr_scramble(b0, k);
    uint32_t key = 0;
    for (unsigned i = 0; i < cl; i++) key = (key << 8) | b0[n + i];
    memcpy(msg, b0, n);
    int ok = (fxr_crc_key(check, b0, n) == key);
    free(b0); free(b1);
    return ok;
}

/* ---------------------------------------------------------------- FEC dispatch */
int fxr_fec_supported(int fs)
{
    unsigned k, n;
    return fs == FXR_FEC_NONE || fs == FXR_FEC_HAMMING84 || fs == FXR_FEC_SECDED7264 || fs == FXR_FEC_SECDED2216 ||
           fs == FXR_FEC_SECDED3932 || fs == FXR_FEC_RS_M8 || blk_spec(fs, &k, &n) || punc_of(fs) != NULL;
}

unsigned fxr_fec_enc_len(int fs, unsigned n)
{
    const punc_t *pp = punc_of(fs);
    if (pp) return (conv_enc_bits(pp, n) + 7) / 8;
    unsigned bk, bn;
    if (blk_spec(fs, &bk, &bn)) return blk_enc_len(bk, bn, n);
    switch (fs) {
    case FXR_FEC_RS_M8: return rs_enc_len(n);
    case FXR_FEC_SECDED2216: return 3 * (n / 2) + ((n % 2) ? (n % 2) + 1 : 0);
    case FXR_FEC_SECDED3932: return 5 * (n / 4) + ((n % 4) ? (n % 4) + 1 : 0);
    case FXR_FEC_HAMMING84: return 2 * n;
    case FXR_FEC_SECDED7264: return 9 * (n / 8) + ((n % 8) ? (n % 8) + 1 : 0);
    default: return n;
    }
}

void fxr_fec_encode(int fs, unsigned n, const uint8_t *dec, uint8_t *enc)
{
    const punc_t *pp = punc_of(fs);
    if (pp) { conv_encode(pp, n, dec, enc); return; }
    unsigned bk, bn;
    if (blk_spec(fs, &bk, &bn)) { blk_encode(fs, n, dec, enc); return; }
    switch (fs) {
    case FXR_FEC_RS_M8: rs_encode(n, dec, enc); return;
    case FXR_FEC_SECDED2216: sdx_init(); sdx_encode(sd22_col, 2, n, dec, enc); return;
    case FXR_FEC_SECDED3932: sdx_init(); sdx_encode(sd39_col, 4, n, dec, enc); return;
    case FXR_FEC_HAMMING84:
        h84_init();
        for (unsigned i = 0; i < n; i++) { enc[2 * i] = h84_enc[dec[i] >> 4]; enc[2 * i + 1] = h84_enc[dec[i] & 15]; }
        return;
    case FXR_FEC_SECDED7264: {
        sd_init();
        unsigned i = 0, j = 0;
        for (; i + 8 <= n; i += 8, j += 9) { enc[j] = sd_parity(dec + i); memcpy(enc + j + 1, dec + i, 8); }
        if (n % 8) {
            uint8_t d[8] = { 0 }; memcpy(d, dec + i, n % 8);
            enc[j] = sd_parity(d); memcpy(enc + j + 1, d, n % 8);
        }
        return; }
    default: memcpy(enc, dec, n); return;
    }
}

void fxr_fec_decode(int fs, unsigned n, const uint8_t *enc, uint8_t *dec)
{
    const punc_t *pp = punc_of(fs);
    if (pp) { conv_decode(pp, n, enc, dec); return; }
    unsigned bk, bn;
    if (blk_spec(fs, &bk, &bn)) { blk_decode(fs, n, enc, dec); return; }
    switch (fs) {
    case FXR_FEC_RS_M8: rs_decode(n, enc, dec); return;
    case FXR_FEC_SECDED2216: sdx_init(); sdx_decode(sd22_col, 2, n, enc, dec); return;
    case FXR_FEC_SECDED3932: sdx_init(); sdx_decode(sd39_col, 4, n, enc, dec); return;
    case FXR_FEC_HAMMING84:
        h84_init();
        for (unsigned i = 0; i < n; i++) dec[i] = (uint8_t)((h84_dec[enc[2 * i]] << 4) | h84_dec[enc[2 * i + 1]]);
        return;
    case FXR_FEC_SECDED7264: {
        sd_init();
        unsigned i = 0, j = 0;
        for (; i + 8 <= n; i += 8, j += 9) sd_decode_block(enc + j, dec + i);
        if (n % 8) {
            uint8_t e[9] = { 0 }, d[8];
            memcpy(e, enc + j, n % 8 + 1);
            sd_decode_block(e, d);
            memcpy(dec + i, d, n % 8);
        }
        return; }
    default: memcpy(dec, enc, n); return;
    }
}

/* ---------------------------------------------------------------- packetizer */
unsigned fxr_packet_enc_len(unsigned n, int check, int fec0, int fec1)
{
    unsigned k = n + fxr_crc_len(check);
    return fxr_fec_enc_len(fec1, fxr_fec_enc_len(fec0, k));
}

void fxr_packet_encode(unsigned n, int check, int fec0, int fec1, const uint8_t *msg, uint8_t *pkt)
{
    unsigned cl = fxr_crc_len(check), k = n + cl;
    unsigned l0 = fxr_fec_enc_len(fec0, k), l1 = fxr_fec_enc_len(fec1, l0);
    uint8_t *b0 = (uint8_t *)calloc(l1 + 16, 1), *b1 = (uint8_t *)calloc(l1 + 16, 1);
    memcpy(b0, msg, n);
    uint32_t key = fxr_crc_key(check, b0, n);
    for (unsigned i = 0; i < cl; i++) { b0[n + cl - i - 1] = (uint8_t)(key & 0xff); key >>= 8; }
    fxr_scramble(b0, k);
    fxr_fec_encode(fec0, k, b0, b1);  fxr_interleave(b1, l0, 0);
    fxr_fec_encode(fec1, l0, b1, b0); fxr_interleave(b0, l1, 0);
    memcpy(pkt, b0, l1);
    free(b0); free(b1);
}

int fxr_packet_decode(unsigned n, int check, int fec0, int fec1, const uint8_t *pkt, uint8_t *msg)
{
    unsigned cl = fxr_crc_len(check), k = n + cl;
    unsigned l0 = fxr_fec_enc_len(fec0, k), l1 = fxr_fec_enc_len(fec1, l0);
    uint8_t *b0 = (uint8_t *)calloc(l1 + 16, 1), *b1 = (uint8_t *)calloc(l1 + 16, 1);
    memcpy(b0, pkt, l1);
    fxr_interleave(b0, l1, 1); fxr_fec_decode(fec1, l0, b0, b1);
    fxr_interleave(b1, l0, 1); fxr_fec_decode(fec0, k, b1, b0);
    fxr_scramble(b0, k);
    uint32_t key = 0;
    for (unsigned i = 0; i < cl; i++) key = (key << 8) | b0[n + i];
    memcpy(msg, b0, n);
    int ok = (fxr_crc_key(check, b0, n) == key);
    free(b0); free(b1);
    return ok;
}
