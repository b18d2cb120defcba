// fx_codec.hpp -- host-side (C++) pieces of the product: constant-table generation for the kernels,
// the packet coding chain in the *encode* direction plus the bit-permutation tables the decode
// kernel gathers through, linear modulators, and the flexframe generator (flex_tx counterpart,
// /root/reference/lib/flex_tx_impl.cc:191-209) that tests and bench.py use as signal source.
//
// Everything here is setup / transmit-side work on the host.  The receive arithmetic itself lives in
// fx_kernels.hip and has no host implementation in this library.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <map>
#include <tuple>
#include "fx_common.h"

namespace fx {

struct cf { float re, im; };

// ------------------------------------------------------------------ m-sequence
struct MSeq {
    unsigned m, g, a, n, v;
    MSeq(unsigned m_, unsigned g_, unsigned a_) : m(m_), g(g_ >> 1), a(a_), n((1u << m_) - 1u), v(a_) {}
    unsigned advance() { unsigned b = (unsigned)__builtin_popcount(v & g) & 1u; v = ((v << 1) | b) & n; return b; }
    unsigned symbol(unsigned bps) { unsigned s = 0; for (unsigned i = 0; i < bps; i++) s = (s << 1) | advance(); return s; }
};

// ------------------------------------------------------------------ ARKAISER square-root Nyquist design
inline double bessel_i0(double z)
{
    double t = 1.0, s = 1.0, h = 0.5 * z;
    for (int k = 1; k < 64; k++) { t *= (h / k) * (h / k); s += t; if (t < 1e-18 * s) break; }
    return s;
}
inline void design_arkaiser(unsigned k, unsigned m, float beta_f, float dt, float *h)
{
    double beta = beta_f, lm = std::log((double)m), lb = std::log(beta);
    double c0 = 0.762886 + 0.067663 * lm;
    double c1 = 0.065515;
    double c2 = std::log(1.0 - 0.088 * std::pow((double)m, -1.6));
    double rho = c0 + c1 * lb + c2 * lb * lb;
    if (rho <= 0.0 || rho >= 1.0) rho = 0.5;
    unsigned n = 2 * k * m + 1;
    double del = beta * rho / (double)k;
    double As = 14.26 * del * (double)n + 7.95;
    double fc = 0.5 * (1.0 + beta * (1.0 - rho)) / (double)k;
    double aAs = std::fabs(As), kb;
    if (aAs > 50.0) kb = 0.1102 * (aAs - 8.7);
    else if (aAs > 21.0) kb = 0.5842 * std::pow(aAs - 21.0, 0.4) + 0.07886 * (aAs - 21.0);
    else kb = 0.0;
    std::vector<double> hd(n);
    double e2 = 0;
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0 + dt;
        double xs = 2.0 * fc * t;
        double sn = std::fabs(xs) < 1e-12 ? 1.0 : std::sin(M_PI * xs) / (M_PI * xs);
        double r = 2.0 * t / (double)n;
        double a = 1.0 - r * r;
        double w = bessel_i0(kb * std::sqrt(a > 0 ? a : 0)) / bessel_i0(kb);
        hd[i] = sn * w;
        e2 += hd[i] * hd[i];
    }
    double g = std::sqrt((double)k / e2);
    for (unsigned i = 0; i < n; i++) h[i] = (float)(hd[i] * g);
}

// initial taps of the optional equaliser: Kaiser-windowed sinc low-pass (13 taps, fc = 0.4, As = 40 dB) scaled by 2 fc
inline void design_eq_init(float *h /* 13 */)
{
    const unsigned n = 13; const double fc = 0.4, kb = 0.5842 * std::pow(40.0 - 21.0, 0.4) + 0.07886 * (40.0 - 21.0);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0, xs = 2.0 * fc * t;
        double sn = std::fabs(xs) < 1e-12 ? 1.0 : std::sin(M_PI * xs) / (M_PI * xs);
        double r = 2.0 * t / (double)n, a = 1.0 - r * r;
        h[i] = (float)(sn * (bessel_i0(kb * std::sqrt(a > 0 ? a : 0)) / bessel_i0(kb)) * 2.0 * fc);
    }
}

// ------------------------------------------------------------------ host FFT-512 (same 8x8x8 network as the kernel)
namespace hfft {
inline cf add(cf a, cf b) { return { a.re + b.re, a.im + b.im }; }
inline cf sub(cf a, cf b) { return { a.re - b.re, a.im - b.im }; }
inline cf mul(cf a, cf w) { float t = a.im * w.im, u = a.im * w.re; return { std::fmaf(a.re, w.re, -t), std::fmaf(a.re, w.im, u) }; }
inline void dft4(cf c0, cf c1, cf c2, cf c3, cf &o0, cf &o1, cf &o2, cf &o3)
{
    cf d0 = add(c0, c2), d1 = add(c1, c3), d2 = sub(c0, c2), e = sub(c1, c3), d3 = { e.im, -e.re };
    o0 = add(d0, d1); o2 = sub(d0, d1); o1 = add(d2, d3); o3 = sub(d2, d3);
}
inline void dft8(cf a[8])
{
    const float c8 = 0.70710678118654752f;
    cf b0 = add(a[0], a[4]), b4 = sub(a[0], a[4]), b1 = add(a[1], a[5]), t5 = sub(a[1], a[5]);
    cf b2 = add(a[2], a[6]), t6 = sub(a[2], a[6]), b3 = add(a[3], a[7]), t7 = sub(a[3], a[7]);
    cf b5 = { (t5.re + t5.im) * c8, (t5.im - t5.re) * c8 };
    cf b6 = { t6.im, -t6.re };
    cf b7 = { (t7.im - t7.re) * c8, -((t7.re + t7.im) * c8) };
    dft4(b0, b1, b2, b3, a[0], a[2], a[4], a[6]);
    dft4(b4, b5, b6, b7, a[1], a[3], a[5], a[7]);
}
inline void fft512(const cf *in, cf *out, const cf *tw)
{
    static thread_local cf A[8][64], B[8][8][8];
    cf a[8];
    for (int j = 0; j < 64; j++) {
        for (int q = 0; q < 8; q++) a[q] = in[j + 64 * q];
        dft8(a);
        A[0][j] = a[0];
        for (int r = 1; r < 8; r++) A[r][j] = mul(a[r], tw[j * r]);
    }
    for (int r = 0; r < 8; r++)
        for (int j0 = 0; j0 < 8; j0++) {
            for (int p = 0; p < 8; p++) a[p] = A[r][j0 + 8 * p];
            dft8(a);
            B[r][0][j0] = a[0];
            for (int s = 1; s < 8; s++) B[r][s][j0] = mul(a[s], tw[8 * j0 * s]);
        }
    for (int r = 0; r < 8; r++)
        for (int s = 0; s < 8; s++) {
            for (int j0 = 0; j0 < 8; j0++) a[j0] = B[r][s][j0];
            dft8(a);
            for (int t = 0; t < 8; t++) out[r + 8 * s + 64 * t] = a[t];
        }
}
}  // namespace hfft

// ------------------------------------------------------------------ CRC / whitening / interleaver
inline unsigned crc_len(unsigned check)
{
    switch (check) { case FX_CRC_CHECKSUM: case FX_CRC_8: return 1; case FX_CRC_16: return 2; case FX_CRC_24: return 3; case FX_CRC_32: return 4; default: return 0; }
}
inline uint32_t crc_key(unsigned check, const uint8_t *msg, unsigned n)
{
    uint32_t prev, mask;
    switch (check) {
    case FX_CRC_CHECKSUM: { uint32_t s = 0; for (unsigned i = 0; i < n; i++) s += msg[i]; return (~s + 1u) & 0xff; }
    case FX_CRC_8:  prev = 0xE0u; mask = 0xFFu; break;
    case FX_CRC_16: prev = 0xA001u; mask = 0xFFFFu; break;
    case FX_CRC_24: prev = 0xD3B6BAu; mask = 0xFFFFFFu; break;
    case FX_CRC_32: prev = 0xEDB88320u; mask = 0xFFFFFFFFu; break;
    default: return 0;
    }
    uint32_t key = mask;
    for (unsigned i = 0; i < n; i++) { key ^= msg[i]; for (int j = 0; j < 8; j++) key = (key >> 1) ^ (prev & (0u - (key & 1u))); }
    return (~key) & mask;
}
inline void scramble(uint8_t *x, unsigned n) { static const uint8_t m[4] = { 0xb4, 0x6a, 0x8b, 0xc5 }; for (unsigned i = 0; i < n; i++) x[i] ^= m[i & 3]; }

// The byte interleaver is four passes of disjoint masked swaps between byte 2i and byte 2j+1, j walking
// an M x N grid column-wise.  `visit(pass, i, j)` lets callers apply it to bytes or to bit labels.
struct Interleaver {
    unsigned n, M, N;
    explicit Interleaver(unsigned n_) : n(n_)
    {
        M = 1 + (unsigned)std::floor(std::sqrt((float)n));
        N = n / M; while (n >= M * N) N++;
    }
    template <class F> void pass(unsigned Np, F &&f) const
    {
        unsigned m = 0, nn = n / 3, n2 = n / 2, j;
        for (unsigned i = 0; i < n2; i++) {
            do { j = m * Np + nn; m++; if (m == M) { nn = (nn + 1) % Np; m = 0; } } while (j >= n2);
            f(2 * i, 2 * j + 1);
        }
    }
    static constexpr uint8_t masks[4] = { 0xff, 0x0f, 0x55, 0x33 };
    static constexpr unsigned grow[4] = { 0, 2, 4, 8 };
    void encode(uint8_t *x) const
    {
        for (int p = 0; p < 4; p++) {
            const uint8_t mk = masks[p];
            pass(N + grow[p], [&](unsigned a, unsigned b) {
                uint8_t va = x[a], vb = x[b];
                x[a] = (uint8_t)((va & ~mk) | (vb & mk)); x[b] = (uint8_t)((va & mk) | (vb & ~mk));
            });
        }
    }
    // gather table of the DE-interleaver: out bit i (byte i>>3, mask 0x80>>(i&7)) = in bit table[i]
    std::vector<uint32_t> decode_gather() const
    {
        std::vector<uint32_t> lab(8 * (size_t)n);
        for (size_t i = 0; i < lab.size(); i++) lab[i] = (uint32_t)i;
        for (int p = 3; p >= 0; p--) {
            const uint8_t mk = masks[p];
            pass(N + grow[p], [&](unsigned a, unsigned b) {
                for (int bit = 0; bit < 8; bit++)
                    if (mk & (0x80u >> bit)) std::swap(lab[8 * (size_t)a + bit], lab[8 * (size_t)b + bit]);
            });
        }
        return lab;
    }
};

// ------------------------------------------------------------------ block codes
struct BlockCodes {
    uint8_t h84_enc[16], h84_dec[256], sd_col[64];
    uint8_t sd22_col[16], sd39_col[32];
    uint8_t h74_enc[16], h74_dec[128]; uint16_t h128_enc[256]; uint8_t h128_dec[4096];
    uint32_t gol_enc[4096], gol_err[4096];
    uint8_t rs_exp[512], rs_log[256], rs_gen[33];      // GF(2^8)/0x11d tables, RS(255,223) generator (roots alpha^1..alpha^32)
    uint8_t gmul(uint8_t a, uint8_t b) const { return (a && b) ? rs_exp[rs_log[a] + rs_log[b]] : 0; }
    unsigned gol_syndrome(uint32_t cw) const { return (unsigned)((gol_enc[(cw >> 12) & 0xfff] ^ cw) & 0xfff); }
    BlockCodes()
    {
        {   // GF(256) and the Reed-Solomon generator polynomial
            unsigned x = 1;
            for (unsigned i = 0; i < 255; i++) { rs_exp[i] = (uint8_t)x; rs_log[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11d; }
            for (unsigned i = 255; i < 512; i++) rs_exp[i] = rs_exp[i - 255];
            rs_log[0] = 0;
            std::memset(rs_gen, 0, sizeof rs_gen); rs_gen[0] = 1;
            for (unsigned i = 1; i <= 32; i++) {
                const uint8_t root = rs_exp[i];
                for (int k = (int)i; k > 0; k--) rs_gen[k] = (uint8_t)(rs_gen[k - 1] ^ gmul(rs_gen[k], root));
                rs_gen[0] = gmul(rs_gen[0], root);
            }
        }
        {   // Hsiao columns for the two short SECDED codes: the first 16 / 32 weight-3 words of 6 / 7 bits
            unsigned k = 0;
            for (unsigned v = 1; v < 64 && k < 16; v++) if (__builtin_popcount(v) == 3) sd22_col[k++] = (uint8_t)v;
            k = 0;
            for (unsigned v = 1; v < 128 && k < 32; v++) if (__builtin_popcount(v) == 3) sd39_col[k++] = (uint8_t)v;
        }
        for (unsigned d = 0; d < 16; d++) {     // Hamming(7,4), systematic
            // liquid's fec_hamming74.c table [recalled]: [p1 p2 d1 p4 d2 d3 d4], d1 = the nibble's MSB
            const unsigned d1 = (d >> 3) & 1, d2 = (d >> 2) & 1, d3 = (d >> 1) & 1, d4 = d & 1;
            h74_enc[d] = (uint8_t)(((d1 ^ d2 ^ d4) << 6) | ((d1 ^ d3 ^ d4) << 5) | (d1 << 4) | ((d2 ^ d3 ^ d4) << 3) | (d2 << 2) | (d3 << 1) | d4);
        }
        for (unsigned r = 0; r < 128; r++) {
            unsigned best = 0, bd = 99;
            for (unsigned d = 0; d < 16; d++) { unsigned w = (unsigned)__builtin_popcount(r ^ h74_enc[d]); if (w < bd) { bd = w; best = d; } }
            h74_dec[r] = (uint8_t)best;
        }
        for (unsigned d = 0; d < 256; d++) {    // Hamming(12,8), parity at positions 1,2,4,8 (1-based, MSB first)
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
        for (unsigned r = 0; r < 4096; r++) {
            unsigned best = 0, bd = 99;
            for (unsigned d = 0; d < 256; d++) { unsigned w = (unsigned)__builtin_popcount(r ^ h128_enc[d]); if (w < bd) { bd = w; best = d; } }
            h128_dec[r] = (uint8_t)best;
        }
        for (unsigned d = 0; d < 4096; d++) {   // extended Golay(24,12): cyclic (23,12) with g = 0xC75, plus overall parity
            uint32_t reg = d << 11;
            for (int i = 22; i >= 11; i--) if (reg & (1u << i)) reg ^= 0xC75u << (i - 11);
            uint32_t cw23 = (d << 11) | (reg & 0x7ff);
            gol_enc[d] = (cw23 << 1) | ((uint32_t)__builtin_popcount(cw23) & 1u);
        }
        for (auto &e : gol_err) e = 0xFFFFFFFFu;
        gol_err[0] = 0;
        auto note = [&](uint32_t e) { unsigned sy = gol_syndrome(e); if (gol_err[sy] == 0xFFFFFFFFu) gol_err[sy] = e; };
        for (int a = 0; a < 24; a++) note(1u << a);
        for (int a = 0; a < 24; a++) for (int b = a + 1; b < 24; b++) note((1u << a) | (1u << b));
        for (int a = 0; a < 24; a++) for (int b = a + 1; b < 24; b++) for (int c2 = b + 1; c2 < 24; c2++) note((1u << a) | (1u << b) | (1u << c2));
        for (unsigned d = 0; d < 16; d++) {
            // liquid's fec_hamming84.c table [recalled]: [p1 p2 d1 p4 d2 d3 d4 p8], p8 = overall parity (00 d2 55 87 99 4b cc 1e e1 33 b4 66 78 aa 2d ff)
            const unsigned d1 = (d >> 3) & 1, d2 = (d >> 2) & 1, d3 = (d >> 1) & 1, d4 = d & 1;
            unsigned c = ((d1 ^ d2 ^ d4) << 7) | ((d1 ^ d3 ^ d4) << 6) | (d1 << 5) | ((d2 ^ d3 ^ d4) << 4) | (d2 << 3) | (d3 << 2) | (d4 << 1);
            c |= (unsigned)__builtin_popcount(c) & 1u;
            h84_enc[d] = (uint8_t)c;
        }
        for (unsigned r = 0; r < 256; r++) {
            unsigned best = 0, bd = 9;
            for (unsigned d = 0; d < 16; d++) { unsigned dist = (unsigned)__builtin_popcount(r ^ h84_enc[d]); if (dist < bd) { bd = dist; best = d; } }
            h84_dec[r] = (uint8_t)best;
        }
        unsigned n = 0;
        for (unsigned v = 1; v < 256 && n < 56; v++) if (__builtin_popcount(v) == 3) sd_col[n++] = (uint8_t)v;
        for (unsigned v = 1; v < 256 && n < 64; v++) if (__builtin_popcount(v) == 5) sd_col[n++] = (uint8_t)v;
    }
    uint8_t sd_parity(const uint8_t d[8]) const
    {
        uint8_t p = 0;
        for (unsigned j = 0; j < 64; j++) if (d[j >> 3] & (0x80u >> (j & 7))) p ^= sd_col[j];
        return p;
    }
};
inline const BlockCodes &block_codes() { static BlockCodes bc; return bc; }

inline int conv_period(unsigned fs)
{
    switch (fs) {
    case FX_FEC_CONV_V27: return 1; case FX_FEC_CONV_V27P23: return 2; case FX_FEC_CONV_V27P34: return 3;
    case FX_FEC_CONV_V27P45: return 4; case FX_FEC_CONV_V27P56: return 5; case FX_FEC_CONV_V27P67: return 6;
    case FX_FEC_CONV_V27P78: return 7; default: return 0;
    }
}
inline void conv_masks(int p, unsigned &pa, unsigned &pb)
{
    switch (p) {
    case 2: pa = 0x3; pb = 0x1; break;   case 3: pa = 0x3; pb = 0x5; break;   case 4: pa = 0xf; pb = 0x1; break;
    case 5: pa = 0xb; pb = 0x15; break;  case 6: pa = 0x17; pb = 0x29; break; case 7: pa = 0x2f; pb = 0x51; break;
    default: pa = 1; pb = 1; break;
    }
}
inline bool blk_spec(unsigned fs, unsigned &k, unsigned &n)
{
    switch (fs) {
    case FX_FEC_HAMMING74: k = 4; n = 7; return true;
    case FX_FEC_HAMMING128: k = 8; n = 12; return true;
    case FX_FEC_GOLAY2412: k = 12; n = 24; return true;
    default: return false;
    }
}
inline bool fec_supported(unsigned fs)
{
    unsigned k, n;
    return fs == FX_FEC_NONE || fs == FX_FEC_HAMMING84 || fs == FX_FEC_SECDED7264 || fs == FX_FEC_SECDED2216 ||
           fs == FX_FEC_SECDED3932 || fs == FX_FEC_RS_M8 || blk_spec(fs, k, n) || conv_period(fs) != 0;
}
inline void rs_dims(unsigned n, unsigned &nb, unsigned &dl) { nb = (n + 222) / 223; if (nb == 0) nb = 1; dl = (n + nb - 1) / nb; }
inline unsigned fec_enc_len(unsigned fs, unsigned n)
{
    int p = conv_period(fs);
    if (p) { unsigned T = 8 * n + 6; unsigned bits = p == 1 ? 2 * T : T + (T + (unsigned)p - 1) / (unsigned)p; return (bits + 7) / 8; }
    unsigned bk, bn;
    if (blk_spec(fs, bk, bn)) { unsigned nb = (8 * n + bk - 1) / bk; return (nb * bn + 7) / 8; }
    if (fs == FX_FEC_RS_M8) { unsigned nb, dl; rs_dims(n, nb, dl); return nb * (dl + 32); }
    if (fs == FX_FEC_SECDED2216) return 3 * (n / 2) + ((n % 2) ? (n % 2) + 1 : 0);
    if (fs == FX_FEC_SECDED3932) return 5 * (n / 4) + ((n % 4) ? (n % 4) + 1 : 0);
    if (fs == FX_FEC_HAMMING84) return 2 * n;
    if (fs == FX_FEC_SECDED7264) return 9 * (n / 8) + ((n % 8) ? (n % 8) + 1 : 0);
    return n;
}
inline void fec_encode(unsigned fs, unsigned n, const uint8_t *dec, uint8_t *enc)
{
    const BlockCodes &bc = block_codes();
    int p = conv_period(fs);
    if (p) {
        unsigned pa, pb; conv_masks(p, pa, pb);
        unsigned T = 8 * n + 6, sr = 0, nb = 0, el = fec_enc_len(fs, n);
        std::memset(enc, 0, el);
        for (unsigned t = 0; t < T; t++) {
            unsigned bit = t < 8 * n ? (dec[t >> 3] >> (7 - (t & 7))) & 1u : 0u;
            sr = ((sr << 1) | bit) & 0x7f;
            unsigned col = t % (unsigned)p;
            if ((pa >> col) & 1) { if (__builtin_popcount(sr & 0x6d) & 1) enc[nb >> 3] |= (uint8_t)(0x80u >> (nb & 7)); nb++; }
            if ((pb >> col) & 1) { if (__builtin_popcount(sr & 0x4f) & 1) enc[nb >> 3] |= (uint8_t)(0x80u >> (nb & 7)); nb++; }
        }
        return;
    }
    unsigned bk, bn;
    if (blk_spec(fs, bk, bn)) {      // bit-packed block codes: k-bit blocks MSB first -> n-bit codewords back to back
        const unsigned nb = (8 * n + bk - 1) / bk, el = fec_enc_len(fs, n);
        std::memset(enc, 0, el);
        for (unsigned j = 0; j < nb; j++) {
            unsigned d = 0;
            for (unsigned i = 0; i < bk; i++) { unsigned q = j * bk + i; d = (d << 1) | (q < 8 * n ? (dec[q >> 3] >> (7 - (q & 7))) & 1u : 0u); }
            const uint32_t cw = fs == FX_FEC_HAMMING74 ? bc.h74_enc[d] : fs == FX_FEC_HAMMING128 ? bc.h128_enc[d] : bc.gol_enc[d];
            for (unsigned i = 0; i < bn; i++) { unsigned q = j * bn + i; if ((cw >> (bn - 1 - i)) & 1u) enc[q >> 3] |= (uint8_t)(0x80u >> (q & 7)); }
        }
        return;
    }
    if (fs == FX_FEC_RS_M8) {        // RS(255,223), message cut into equal shortened blocks, 32 parity bytes each
        unsigned nb, dl; rs_dims(n, nb, dl);
        for (unsigned b = 0; b < nb; b++) {
            uint8_t d[223] = { 0 }, par[32] = { 0 };
            const unsigned off = b * dl, len = off < n ? std::min(n - off, dl) : 0;
            std::memcpy(d, dec + off, len);
            for (unsigned i = 0; i < dl; i++) {
                const uint8_t fb = (uint8_t)(d[i] ^ par[31]);
                for (int k = 31; k > 0; k--) par[k] = (uint8_t)(par[k - 1] ^ bc.gmul(fb, bc.rs_gen[k]));
                par[0] = bc.gmul(fb, bc.rs_gen[0]);
            }
            uint8_t *o = enc + b * (dl + 32);
            std::memcpy(o, d, dl);
            for (unsigned k = 0; k < 32; k++) o[dl + k] = par[31 - k];
        }
        return;
    }
    if (fs == FX_FEC_SECDED2216 || fs == FX_FEC_SECDED3932) {
        const unsigned nd = fs == FX_FEC_SECDED2216 ? 2 : 4; const uint8_t *col = fs == FX_FEC_SECDED2216 ? bc.sd22_col : bc.sd39_col;
        auto par = [&](const uint8_t *d) { uint8_t pp = 0; for (unsigned j = 0; j < 8 * nd; j++) if (d[j >> 3] & (0x80u >> (j & 7))) pp ^= col[j]; return pp; };
        unsigned i = 0, j = 0;
        for (; i + nd <= n; i += nd, j += nd + 1) { enc[j] = par(dec + i); std::memcpy(enc + j + 1, dec + i, nd); }
        if (n % nd) { uint8_t d[8] = { 0 }; std::memcpy(d, dec + i, n % nd); enc[j] = par(d); std::memcpy(enc + j + 1, d, n % nd); }
        return;
    }
    if (fs == FX_FEC_HAMMING84) { for (unsigned i = 0; i < n; i++) { enc[2 * i] = bc.h84_enc[dec[i] >> 4]; enc[2 * i + 1] = bc.h84_enc[dec[i] & 15]; } return; }
    if (fs == FX_FEC_SECDED7264) {
        unsigned i = 0, j = 0;
        for (; i + 8 <= n; i += 8, j += 9) { enc[j] = bc.sd_parity(dec + i); std::memcpy(enc + j + 1, dec + i, 8); }
        if (n % 8) { uint8_t d[8] = { 0 }; std::memcpy(d, dec + i, n % 8); enc[j] = bc.sd_parity(d); std::memcpy(enc + j + 1, d, n % 8); }
        return;
    }
    std::memcpy(enc, dec, n);
}

struct PacketPlan { unsigned n, check, fec0, fec1, k, l0, l1; };
inline PacketPlan packet_plan(unsigned n, unsigned check, unsigned fec0, unsigned fec1)
{
    PacketPlan p{ n, check, fec0, fec1, 0, 0, 0 };
    p.k = n + crc_len(check); p.l0 = fec_enc_len(fec0, p.k); p.l1 = fec_enc_len(fec1, p.l0);
    return p;
}
inline std::vector<uint8_t> packet_encode(const PacketPlan &p, const uint8_t *msg)
{
    std::vector<uint8_t> b0(p.l1 + 16, 0), b1(p.l1 + 16, 0);
    std::memcpy(b0.data(), msg, p.n);
    uint32_t key = crc_key(p.check, b0.data(), p.n);
    unsigned cl = p.k - p.n;
    for (unsigned i = 0; i < cl; i++) { b0[p.n + cl - i - 1] = (uint8_t)(key & 0xff); key >>= 8; }
    scramble(b0.data(), p.k);
    fec_encode(p.fec0, p.k, b0.data(), b1.data());  Interleaver(p.l0).encode(b1.data());
    fec_encode(p.fec1, p.l0, b1.data(), b0.data()); Interleaver(p.l1).encode(b0.data());
    b0.resize(p.l1);
    return b0;
}

// ------------------------------------------------------------------ linear modulators (transmit side)
inline unsigned gray_dec(unsigned x) { unsigned y = x; while (x >>= 1) y ^= x; return y; }
inline unsigned modem_bps(unsigned ms)
{
    switch (ms) {
    case FX_MODEM_PSK2: case FX_MODEM_DPSK2: return 1;
    case FX_MODEM_PSK4: case FX_MODEM_DPSK4: case FX_MODEM_ASK4: case FX_MODEM_QPSK: return 2;
    case FX_MODEM_PSK8: case FX_MODEM_DPSK8: return 3;
    case FX_MODEM_PSK16: case FX_MODEM_QAM16: return 4;
    case FX_MODEM_QAM32: return 5; case FX_MODEM_QAM64: return 6;
    default: return 0;
    }
}
struct Modulator {
    unsigned ms, bps, dpsk_idx = 0; const cf *sc;
    Modulator(unsigned ms_, const cf *sincos1024) : ms(ms_), bps(modem_bps(ms_)), sc(sincos1024) {}
    cf psk(unsigned i) const
    {
        static const cf axis[4] = { { 1.0f, 0.0f }, { 0.0f, 1.0f }, { -1.0f, 0.0f }, { 0.0f, -1.0f } };   // exact for M <= 4
        if (bps <= 2) return axis[bps == 1 ? 2 * (i & 1) : (i & 3)];
        return sc[(i << (32 - bps)) >> 22];
    }
    cf mod(unsigned sym)
    {
        const float h = 0.70710678118654752f;
        switch (ms) {
        case FX_MODEM_QPSK: return { (sym & 1) ? -h : h, (sym & 2) ? -h : h };
        case FX_MODEM_PSK2: case FX_MODEM_PSK4: case FX_MODEM_PSK8: case FX_MODEM_PSK16: return psk(gray_dec(sym));
        case FX_MODEM_DPSK2: case FX_MODEM_DPSK4: case FX_MODEM_DPSK8:
            dpsk_idx = (dpsk_idx + gray_dec(sym)) & ((1u << bps) - 1u); return psk(dpsk_idx);
        case FX_MODEM_ASK4: return { (2.0f * (float)gray_dec(sym) - 3.0f) * 0.447213595f, 0.0f };
        default: {
            unsigned mi, mq; float al;
            if (ms == FX_MODEM_QAM16) { mi = 2; mq = 2; al = 0.316227766f; }
            else if (ms == FX_MODEM_QAM32) { mi = 3; mq = 2; al = 0.196116135f; }
            else { mi = 3; mq = 3; al = 0.154303350f; }
            unsigned si = gray_dec(sym >> mq), sq = gray_dec(sym & ((1u << mq) - 1u));
            return { (2.0f * (float)si - (float)((1u << mi) - 1u)) * al, (2.0f * (float)sq - (float)((1u << mq) - 1u)) * al };
        }
        }
    }
};

// ------------------------------------------------------------------ constant tables for kernels and generator
struct HostTables {
    cf tw[512], sc[1024], S[512], s[FX_S_LEN], pn[FX_PN_LEN], pilots[FX_HDR_PILOTS];
    float proto[FX_PROTO_LEN], txh[2 * FX_K * FX_M + 1], s2sum;
    std::vector<uint32_t> perm54, perm27;
    HostTables()
    {
        for (int m = 0; m < 512; m++) { double a = 2.0 * M_PI * (double)m / 512.0; tw[m] = { (float)std::cos(a), (float)(-std::sin(a)) }; }
        for (int k = 0; k < 1024; k++) { double a = 2.0 * M_PI * (double)k / 1024.0; sc[k] = { (float)std::cos(a), (float)std::sin(a) }; }
        design_arkaiser(FX_NPFB * FX_K, FX_M, FX_BETA, 0.0f, proto);
        design_arkaiser(FX_K, FX_M, FX_BETA, 0.0f, txh);
        MSeq ms(7, 0x0089, 1);
        const float h = (float)M_SQRT1_2;
        for (int i = 0; i < FX_PN_LEN; i++) { pn[i].re = ms.advance() ? h : -h; pn[i].im = ms.advance() ? h : -h; }
        std::vector<cf> sy(FX_PN_LEN + 2 * FX_M, cf{ 0, 0 });
        std::memcpy(sy.data(), pn, sizeof pn);
        interp(sy.data(), (unsigned)sy.size(), txh, s);
        cf buf[512]; std::memset(buf, 0, sizeof buf); std::memcpy(buf, s, sizeof s);
        hfft::fft512(buf, S, tw);
        s2sum = 0; for (int i = 0; i < FX_S_LEN; i++) s2sum += std::fmaf(s[i].re, s[i].re, s[i].im * s[i].im);
        MSeq mp(4, 0x0013, 1);
        for (int i = 0; i < FX_HDR_PILOTS; i++) {
            unsigned q = mp.symbol(2);
            pilots[i] = { (q == 0 || q == 3) ? h : -h, (q < 2) ? h : -h };
        }
        perm54 = Interleaver(FX_HDR_ENC).decode_gather();
        perm27 = Interleaver(FX_HDR_E0).decode_gather();
    }
    // y[2n+i] = sum_t h[i+2t] x[n-t], t ascending (29-tap pulse, k = 2)
    static void interp(const cf *x, unsigned nsym, const float *h, cf *y)
    {
        for (unsigned n = 0; n < nsym; n++)
            for (unsigned i = 0; i < FX_K; i++) {
                float ar = 0, ai = 0;
                for (unsigned t = 0; t < 15; t++) {
                    unsigned hi = i + FX_K * t;
                    if (hi > 2 * FX_K * FX_M || t > n) continue;
                    ar = std::fmaf(h[hi], x[n - t].re, ar); ai = std::fmaf(h[hi], x[n - t].im, ai);
                }
                y[FX_K * n + i] = { ar, ai };
            }
    }
};
inline const HostTables &host_tables() { static HostTables t; return t; }

// ------------------------------------------------------------------ frame generator
inline void pack_symbols(const uint8_t *enc, unsigned enc_len, unsigned bps, unsigned nsym, uint8_t *sym)
{
    unsigned nbits = 8 * enc_len;
    for (unsigned j = 0; j < nsym; j++) {
        unsigned s = 0;
        for (unsigned b = 0; b < bps; b++) { unsigned k = j * bps + b; s = (s << 1) | (k < nbits ? (enc[k >> 3] >> (7 - (k & 7))) & 1u : 0u); }
        sym[j] = (uint8_t)s;
    }
}
struct FrameGen {
    unsigned check = FX_CRC_32, fec0 = FX_FEC_NONE, fec1 = FX_FEC_NONE, ms = FX_MODEM_QPSK;
    float dt = 0.0f;
    std::vector<cf> syms;        // assembled symbols incl. 2m flush zeros
    unsigned payload_syms(unsigned payload_len) const
    {
        PacketPlan p = packet_plan(payload_len, check, fec0, fec1);
        unsigned bps = modem_bps(ms);
        return bps ? (8 * p.l1 + bps - 1) / bps : 0;
    }
    unsigned frame_len(unsigned payload_len) const { return FX_K * (FX_PN_LEN + FX_HDR_SYM + payload_syms(payload_len) + 2 * FX_M); }
    // the 20 header bytes: 14 user bytes + protocol id, payload length, modulation, CRC / FEC schemes
    void head_bytes(const uint8_t *header14, unsigned payload_len, uint8_t *hd) const
    {
        if (header14) std::memcpy(hd, header14, FX_HDR_USER); else std::memset(hd, 0, FX_HDR_USER);
        hd[14] = FX_PROTOCOL; hd[15] = (uint8_t)((payload_len >> 8) & 0xff); hd[16] = (uint8_t)(payload_len & 0xff);
        hd[17] = (uint8_t)ms; hd[18] = (uint8_t)(((check & 7) << 5) | (fec0 & 0x1f)); hd[19] = (uint8_t)(fec1 & 0x1f);
    }
    // ... encoded (CRC-32, SECDED(72,64), Hamming(8,4)) and cut into the 216 QPSK words of the header symbols
    void head_words(const uint8_t *header14, unsigned payload_len, uint8_t *hs) const
    {
        uint8_t hd[FX_HDR_DEC];
        head_bytes(header14, payload_len, hd);
        PacketPlan hp = packet_plan(FX_HDR_DEC, FX_CRC_32, FX_FEC_SECDED7264, FX_FEC_HAMMING84);
        std::vector<uint8_t> he = packet_encode(hp, hd);
        pack_symbols(he.data(), FX_HDR_ENC, 2, FX_HDR_MOD, hs);
    }
    // preamble + header symbols (FX_PN_LEN + FX_HDR_SYM of them)
    void head(const uint8_t *header14, unsigned payload_len, cf *out) const
    {
        const HostTables &T = host_tables();
        std::memcpy(out, T.pn, sizeof T.pn);
        uint8_t hs[FX_HDR_MOD]; head_words(header14, payload_len, hs);
        Modulator qm(FX_MODEM_QPSK, T.sc);
        for (unsigned i = 0, n = 0, pp = 0; i < FX_HDR_SYM; i++)
            out[FX_PN_LEN + i] = (i % FX_PILOT_SPACING) == 0 ? T.pilots[pp++] : qm.mod(hs[n++]);
    }
    // payload symbols as the modem's input words (what pack_symbols hands to Modulator::mod)
    std::vector<uint8_t> payload_words(const uint8_t *payload, unsigned payload_len) const
    {
        const unsigned npay = payload_syms(payload_len);
        PacketPlan pl = packet_plan(payload_len, check, fec0, fec1);
        std::vector<uint8_t> pe = packet_encode(pl, payload);
        std::vector<uint8_t> ps(npay + 1);
        pack_symbols(pe.data(), pl.l1, modem_bps(ms), npay, ps.data());
        ps.resize(npay);
        return ps;
    }
    // ... and as constellation indices for fx_txgen_kernel's tx_point(): Gray decoding and the DPSK phase accumulation,
    // i.e. all the integer work of Modulator::mod, done here; the kernel does the float work
    std::vector<uint8_t> payload_indices(const uint8_t *payload, unsigned payload_len) const
    {
        std::vector<uint8_t> w = payload_words(payload, payload_len);
        const unsigned bps = modem_bps(ms);
        unsigned acc = 0;
        for (auto &v : w) {
            const unsigned sym = v;
            switch (ms) {
            case FX_MODEM_QPSK: break;
            case FX_MODEM_PSK2: case FX_MODEM_PSK4: case FX_MODEM_PSK8: case FX_MODEM_PSK16: case FX_MODEM_ASK4: v = (uint8_t)gray_dec(sym); break;
            case FX_MODEM_DPSK2: case FX_MODEM_DPSK4: case FX_MODEM_DPSK8: acc = (acc + gray_dec(sym)) & ((1u << bps) - 1u); v = (uint8_t)acc; break;
            default: {
                const unsigned mq = ms == FX_MODEM_QAM64 ? 3u : 2u;
                v = (uint8_t)((gray_dec(sym >> mq) << mq) | gray_dec(sym & ((1u << mq) - 1u)));
            }
            }
        }
        return w;
    }
    void assemble(const uint8_t *header14, const uint8_t *payload, unsigned payload_len)
    {
        const HostTables &T = host_tables();
        unsigned npay = payload_syms(payload_len);
        syms.assign(FX_PN_LEN + FX_HDR_SYM + npay + 2 * FX_M, cf{ 0, 0 });
        head(header14, payload_len, syms.data());
        std::vector<uint8_t> ps = payload_words(payload, payload_len);
        Modulator pm(ms, T.sc);
        for (unsigned i = 0; i < npay; i++) syms[FX_PN_LEN + FX_HDR_SYM + i] = pm.mod(ps[i]);
    }
    unsigned write(cf *out) const
    {
        const HostTables &T = host_tables();
        float hdt[2 * FX_K * FX_M + 1];
        const float *h = T.txh;
        if (dt != 0.0f) { design_arkaiser(FX_K, FX_M, FX_BETA, dt, hdt); h = hdt; }
        HostTables::interp(syms.data(), (unsigned)syms.size(), h, out);
        return FX_K * (unsigned)syms.size();
    }
};

}  // namespace fx
