// fx_device.h -- device-side arithmetic primitives for the flexframe RX kernels (gfx950, wave64).
//
// Every function here follows the canonical arithmetic written down in DESIGN.md §4: IEEE binary32,
// no contraction (-ffp-contract=off) except where fmaf() is spelled out, no libm transcendental on
// path data, fixed reduction orders.  That is what lets detection indices, polyphase-branch picks and
// hard decisions come out identical to a scalar CPU run of the same algorithm.
#pragma once
#include <hip/hip_runtime.h>
#include "fx_common.h"

struct FxTables {
    float2   tw[512];        // exp(-j 2 pi m / 512)
    float2   sc[1024];       // (cos, sin)(2 pi k / 1024)
    float2   S[512];         // FFT of the zero-padded detector template
    float2   s[FX_S_LEN];    // detector template (RRC-shaped p/n preamble)
    float2   TD[512];        // FFT of td[k] = s[k+1] conj(s[k]) (coarse pre-lock scan only)
    float2   pilots[16];
    float2   pn[FX_PN_LEN];  // the 64 p/n preamble symbols (equaliser training)
    float    proto[FX_PROTO_LEN + 3];
    float    s2sum;          // sum |s|^2
    float    td2sum;         // sum |td|^2
    float    pad_[2];
    float    eq0[16];        // initial equaliser taps (13, real)
    uint16_t perm54[FX_HDR_ENC * 8];   // bit gather tables of the header de-interleavers
    uint16_t perm27[FX_HDR_E0 * 8];
    uint8_t  h84dec[256];
    uint8_t  sdcol[64];
    uint8_t  sd22col[16], sd39col[32];
    uint8_t  h74dec[128];
    uint8_t  h128dec[4096];
    uint32_t golenc[4096], golerr[4096];   // Golay(24,12): codeword of a 12-bit word, error pattern of a syndrome
    uint8_t  rsexp[512], rslog[256];       // GF(2^8)/0x11d for Reed-Solomon RS(255,223)
};

#define FX_DEV __device__ __forceinline__

FX_DEV float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
FX_DEV float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * w
FX_DEV float2 cmul(float2 a, float2 w)
{
    float t = a.y * w.y, u = a.y * w.x;
    return make_float2(fmaf(a.x, w.x, -t), fmaf(a.x, w.y, u));
}
// a * conj(b)
FX_DEV float2 cmulc(float2 a, float2 b)
{
    float t = a.y * b.y, u = a.x * b.y;
    return make_float2(fmaf(a.x, b.x, t), fmaf(a.y, b.x, -u));
}
FX_DEV float cm2(float2 a) { return fmaf(a.x, a.x, a.y * a.y); }

FX_DEV uint32_t rad2u32(float rad)
{
    float t = rintf(rad * 683565248.0f);
    return (uint32_t)(long long)t;
}

// PLL increments, already in phase units: round, clamp below 2^31, plain 32-bit convert
FX_DEV uint32_t phase_inc(float units)
{
    float t = rintf(units);
    t = fminf(fmaxf(t, -2147483520.0f), 2147483520.0f);
    return (uint32_t)(int)t;
}

FX_DEV void sincos_u32(uint32_t th, const float2 *sc, float &c, float &s)
{
    const float2 t = sc[th >> 22];
    float d  = (float)(th & 0x3FFFFFu) * 1.4629180792671596e-9f;
    float d2 = d * d;
    float cd = fmaf(d2, -0.5f, 1.0f);
    float sd = fmaf(d2 * d, -0.16666667f, d);
    c = fmaf(t.x, cd, -(t.y * sd));
    s = fmaf(t.y, cd, t.x * sd);
}

// payload PLL: the phase advance of one symbol, in whole phase units (rounded, clamped below 2^31 so that the conversion
// to an integer is the same everywhere), and the cos / sin of that advance by a 5th-order series -- the loop turns its
// carrier phasor by this between table look-ups
FX_DEV float phase_step(float units)
{
    return fminf(fmaxf(rintf(units), -2147483520.0f), 2147483520.0f);
}
FX_DEV void sincos_small(float step_units, float &c, float &s)
{
    float x  = step_units * 1.4629180792671596e-9f;
    float x2 = x * x;
    c = fmaf(x2, fmaf(x2, 4.16666679e-2f, -0.5f), 1.0f);
    s = fmaf(x * x2, fmaf(x2, 8.33333377e-3f, -0.16666667f), x);
}

// x * exp(-j theta)
FX_DEV float2 derot(float2 x, uint32_t th, const float2 *sc)
{
    float c, s; sincos_u32(th, sc, c, s);
    return make_float2(fmaf(x.x, c, x.y * s), fmaf(x.y, c, -(x.x * s)));
}

FX_DEV float atan2c(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    const bool big = mn > 0.41421356f * mx;
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float base = big ? 0.78539816f : 0.0f;
    float a = num / den;                        // the only division; 0/0 is masked by the last select
    float z = a * a;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    float r = fmaf(p * z, a, a) + base;
    if (ay > ax) r = 1.57079633f - r;
    if (x < 0.0f) r = 3.14159265f - r;
    r = y < 0.0f ? -r : r;
    return mx == 0.0f ? 0.0f : r;
}

// ---------------------------------------------------------------- 8-point DFT in registers
// Packed fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one instruction per complex add, two per complex multiply), written
// as inline assembly because what makes them pay is the operand modifiers: op_sel / op_sel_hi pick which half of a register pair
// feeds the low / high result, neg_lo / neg_hi negate it -- so a multiplication by -j, a conjugate, a swap of real and imaginary
// part or the (x + y, y - x) of a 45-degree twiddle cost nothing, where the compiler spends a v_mov per swizzle (60 of the 258
// VALU instructions per detector bin before this).  Every component is still ONE IEEE operation of the canonical order (DESIGN.md
// section 4): a - b is a + (-b), -((x + y) c) is (-(x + y)) c, bit for bit.
#define FX_C8 0.70710678118654752f
typedef float fx_v2 __attribute__((ext_vector_type(2)));
FX_DEV fx_v2 to_v2(float2 a) { fx_v2 r; r.x = a.x; r.y = a.y; return r; }
FX_DEV float2 to_f2(fx_v2 a) { return make_float2(a.x, a.y); }
FX_DEV fx_v2 pk_add(fx_v2 a, fx_v2 b) { fx_v2 o; asm("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }
FX_DEV fx_v2 pk_sub(fx_v2 a, fx_v2 b) { fx_v2 o; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b)); return o; }
// a + (-j) b = a + (b.y, -b.x)          and a - (-j) b
FX_DEV fx_v2 pk_add_mj(fx_v2 a, fx_v2 b) { fx_v2 o; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b)); return o; }
FX_DEV fx_v2 pk_sub_mj(fx_v2 a, fx_v2 b) { fx_v2 o; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o) : "v"(a), "v"(b)); return o; }
// (t.y - t.x, t.x + t.y)
FX_DEV fx_v2 pk_rot7(fx_v2 t) { fx_v2 o; asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(o) : "v"(t)); return o; }
FX_DEV fx_v2 pk_scale(fx_v2 a, fx_v2 c) { fx_v2 o; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(c)); return o; }
// (a.x c.x, -(a.y c.y))
FX_DEV fx_v2 pk_scale_conj(fx_v2 a, fx_v2 c) { fx_v2 o; asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(o) : "v"(a), "v"(c)); return o; }
// a * w, component for component what cmul() computes: t = a.y w.y, u = a.y w.x; (fma(a.x, w.x, -t), fma(a.x, w.y, u))
FX_DEV fx_v2 pk_cmul(fx_v2 a, fx_v2 w)
{
    fx_v2 m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(m));
    return r;
}
// a * conj(b) with real and imaginary part swapped (the inverse transform by a forward one), component for component what
// cmulc() computes: t = a.y b.y, u = a.x b.y; y = (fma(a.x, b.x, t), fma(a.y, b.x, -u)); result (y.y, y.x)
FX_DEV fx_v2 pk_cmulc_swap(fx_v2 a, fx_v2 b)
{
    fx_v2 m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
FX_DEV void dft4(fx_v2 c0, fx_v2 c1, fx_v2 c2, fx_v2 c3, fx_v2 &o0, fx_v2 &o1, fx_v2 &o2, fx_v2 &o3)
{
    const fx_v2 d0 = pk_add(c0, c2), d1 = pk_add(c1, c3), d2 = pk_sub(c0, c2), e = pk_sub(c1, c3);
    o0 = pk_add(d0, d1); o2 = pk_sub(d0, d1);
    o1 = pk_add_mj(d2, e); o3 = pk_sub_mj(d2, e);
}
FX_DEV void dft8(fx_v2 a[8])
{
    const fx_v2 c8 = { FX_C8, FX_C8 };
    const fx_v2 b0 = pk_add(a[0], a[4]), b4 = pk_sub(a[0], a[4]);
    const fx_v2 b1 = pk_add(a[1], a[5]), t5 = pk_sub(a[1], a[5]);
    const fx_v2 b2 = pk_add(a[2], a[6]), t6 = pk_sub(a[2], a[6]);
    const fx_v2 b3 = pk_add(a[3], a[7]), t7 = pk_sub(a[3], a[7]);
    const fx_v2 b5 = pk_scale(pk_add_mj(t5, t5), c8);             // ((t5.x + t5.y) c, (t5.y - t5.x) c)
    const fx_v2 b7 = pk_scale_conj(pk_rot7(t7), c8);              // ((t7.y - t7.x) c, -((t7.x + t7.y) c))
    dft4(b0, b1, b2, b3, a[0], a[2], a[4], a[6]);
    {   // dft4(b4, b5, b6, b7) with b6 = -j t6 folded into its two uses
        const fx_v2 d0 = pk_add_mj(b4, t6), d2 = pk_sub_mj(b4, t6), d1 = pk_add(b5, b7), e = pk_sub(b5, b7);
        a[1] = pk_add(d0, d1); a[5] = pk_sub(d0, d1);
        a[3] = pk_add_mj(d2, e); a[7] = pk_sub_mj(d2, e);
    }
}
FX_DEV void dft8(float2 a[8])
{
    fx_v2 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = to_v2(a[i]);
    dft8(v);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = to_f2(v[i]);
}

// ---------------------------------------------------------------- FFT-512 by one wavefront
// In : lane j holds a[q] = x[j + 64 q].          (decimation in frequency, 8 x 8 x 8)
// Out: lane L holds a[t] = X[(L>>3) + 8 (L&7) + 64 t].
// scr: 576 float2 of LDS private to the wave (8 rows of 72: 64 data + 8 pad, conflict-free for
//      both ds_write_b64 and ds_read_b64 in either exchange).
// twA[r-1] = W512^(lane r), twB[s-1] = W512^(8 (lane&7) s), r,s = 1..7.
FX_DEV void fft512_wave(fx_v2 a[8], float2 *scr, int lane, const float2 twA[7], const float2 twB[7])
{
    dft8(a);
#pragma unroll
    for (int r = 1; r < 8; r++) a[r] = pk_cmul(a[r], to_v2(twA[r - 1]));
#pragma unroll
    for (int r = 0; r < 8; r++) scr[r * 72 + lane] = to_f2(a[r]);
    __builtin_amdgcn_wave_barrier();
    {
        const int r = lane >> 3, j0 = lane & 7;
#pragma unroll
        for (int p = 0; p < 8; p++) a[p] = to_v2(scr[r * 72 + j0 + 8 * p]);
        __builtin_amdgcn_wave_barrier();
        dft8(a);
#pragma unroll
        for (int s = 1; s < 8; s++) a[s] = pk_cmul(a[s], to_v2(twB[s - 1]));
#pragma unroll
        for (int s = 0; s < 8; s++) scr[r * 72 + s * 9 + j0] = to_f2(a[s]);
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int r = lane >> 3, s = lane & 7;
#pragma unroll
        for (int j0 = 0; j0 < 8; j0++) a[j0] = to_v2(scr[r * 72 + s * 9 + j0]);
        __builtin_amdgcn_wave_barrier();
        dft8(a);
    }
}
// two independent transforms by the same wave, stage by stage side by side: one transform alone is a chain of LDS round trips and
// dependent butterflies that a lone wave on its SIMD cannot hide (the flex_rx walker: one workgroup per CU); two give the scheduler
// something to put into the gaps.  Same arithmetic per transform.
FX_DEV void fft512_wave2(fx_v2 a[8], fx_v2 b[8], float2 *scrA, float2 *scrB, int lane, const float2 twA[7], const float2 twB[7])
{
    dft8(a); dft8(b);
#pragma unroll
    for (int r = 1; r < 8; r++) { a[r] = pk_cmul(a[r], to_v2(twA[r - 1])); b[r] = pk_cmul(b[r], to_v2(twA[r - 1])); }
#pragma unroll
    for (int r = 0; r < 8; r++) { scrA[r * 72 + lane] = to_f2(a[r]); scrB[r * 72 + lane] = to_f2(b[r]); }
    __builtin_amdgcn_wave_barrier();
    {
        const int r = lane >> 3, j0 = lane & 7;
#pragma unroll
        for (int p = 0; p < 8; p++) { a[p] = to_v2(scrA[r * 72 + j0 + 8 * p]); b[p] = to_v2(scrB[r * 72 + j0 + 8 * p]); }
        __builtin_amdgcn_wave_barrier();
        dft8(a); dft8(b);
#pragma unroll
        for (int s = 1; s < 8; s++) { a[s] = pk_cmul(a[s], to_v2(twB[s - 1])); b[s] = pk_cmul(b[s], to_v2(twB[s - 1])); }
#pragma unroll
        for (int s = 0; s < 8; s++) { scrA[r * 72 + s * 9 + j0] = to_f2(a[s]); scrB[r * 72 + s * 9 + j0] = to_f2(b[s]); }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int r = lane >> 3, s = lane & 7;
#pragma unroll
        for (int j0 = 0; j0 < 8; j0++) { a[j0] = to_v2(scrA[r * 72 + s * 9 + j0]); b[j0] = to_v2(scrB[r * 72 + s * 9 + j0]); }
        __builtin_amdgcn_wave_barrier();
        dft8(a); dft8(b);
    }
}
FX_DEV void fft512_wave(float2 a[8], float2 *scr, int lane, const float2 twA[7], const float2 twB[7])
{
    fx_v2 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = to_v2(a[i]);
    fft512_wave(v, scr, lane, twA, twB);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = to_f2(v[i]);
}

// balanced-tree sum across the 64 lanes (xor butterfly; a+b is commutative bit-for-bit)
FX_DEV float wave_sum(float v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// argmax with "first maximum in key order wins": returns the pair on all lanes
FX_DEV void wave_argmax(float &v, uint32_t &key)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        float ov = __shfl_xor(v, m, 64);
        uint32_t ok = __shfl_xor(key, m, 64);
        bool take = (ov > v) || (ov == v && ok < key);
        v = take ? ov : v; key = take ? ok : key;
    }
}

// ---------------------------------------------------------------- modem (hard decision + phase error)
FX_DEV unsigned gray_enc(unsigned x) { return x ^ (x >> 1); }

FX_DEV unsigned modem_bps(unsigned ms)
{
    switch (ms) {
    case FX_MODEM_PSK2: case FX_MODEM_DPSK2: return 1;
    case FX_MODEM_PSK4: case FX_MODEM_DPSK4: case FX_MODEM_ASK4: case FX_MODEM_QPSK: return 2;
    case FX_MODEM_PSK8: case FX_MODEM_DPSK8: return 3;
    case FX_MODEM_PSK16: case FX_MODEM_QAM16: return 4;
    case FX_MODEM_QAM32: return 5;
    case FX_MODEM_QAM64: return 6;
    default: return 0;
    }
}

FX_DEV unsigned pam_index(float v, float inv2al, unsigned Lv)
{
    float t = floorf(fmaf(v, inv2al, 0.5f * (float)Lv));
    if (t < 0.0f) t = 0.0f;
    if (t > (float)(Lv - 1)) t = (float)(Lv - 1);
    return (unsigned)t;
}

FX_DEV unsigned psk_index(float2 r, unsigned bps)
{
    if (bps == 1) return r.x > 0.0f ? 0u : 1u;
    if (bps == 2) {
        if (fabsf(r.x) >= fabsf(r.y)) return r.x > 0.0f ? 0u : 2u;
        return r.y > 0.0f ? 1u : 3u;
    }
    float th = atan2c(r.y, r.x);
    float t = rintf(th * ((float)(1u << bps) * 0.159154943f));
    return (unsigned)((int)t) & ((1u << bps) - 1u);
}

// unit phasor at phase index idx of 2^bps: exact axis points up to 4-PSK, the sincos table beyond
FX_DEV float2 psk_point(unsigned idx, unsigned bps, const float2 *sc)
{
    if (bps <= 2) {
        const unsigned q = bps == 1 ? 2u * (idx & 1u) : (idx & 3u);
        return make_float2(q == 0 ? 1.0f : (q == 2 ? -1.0f : 0.0f), q == 1 ? 1.0f : (q == 3 ? -1.0f : 0.0f));
    }
    return sc[(idx << (32 - bps)) >> 22];
}

// dpsk_prev: running phase index of differential schemes (in/out)
FX_DEV unsigned modem_demod(unsigned ms, unsigned bps, float2 r, unsigned &dpsk_prev, const float2 *sc,
                            float2 &xh, float &pe)
{
    unsigned sym;
    switch (ms) {
    case FX_MODEM_QPSK:
        sym = (r.x > 0.0f ? 0u : 1u) | (r.y > 0.0f ? 0u : 2u);
        xh = make_float2((sym & 1) ? -0.70710678118654752f : 0.70710678118654752f,
                         (sym & 2) ? -0.70710678118654752f : 0.70710678118654752f);
        break;
    case FX_MODEM_PSK2: case FX_MODEM_PSK4: case FX_MODEM_PSK8: case FX_MODEM_PSK16: {
        unsigned idx = psk_index(r, bps);
        sym = gray_enc(idx); xh = psk_point(idx, bps, sc);
        break; }
    case FX_MODEM_DPSK2: case FX_MODEM_DPSK4: case FX_MODEM_DPSK8: {
        unsigned idx = psk_index(r, bps);
        sym = gray_enc((idx - dpsk_prev) & ((1u << bps) - 1u));
        dpsk_prev = idx; xh = psk_point(idx, bps, sc);
        break; }
    case FX_MODEM_ASK4: {
        unsigned idx = pam_index(r.x, 1.11803399f, 4);
        sym = gray_enc(idx); xh = make_float2((2.0f * (float)idx - 3.0f) * 0.447213595f, 0.0f);
        break; }
    default: {
        unsigned mi, mq; float al;
        if (ms == FX_MODEM_QAM16) { mi = 2; mq = 2; al = 0.316227766f; }
        else if (ms == FX_MODEM_QAM32) { mi = 3; mq = 2; al = 0.196116135f; }
        else { mi = 3; mq = 3; al = 0.154303350f; }
        float inv = 0.5f / al;
        unsigned ii = pam_index(r.x, inv, 1u << mi), iq = pam_index(r.y, inv, 1u << mq);
        sym = (gray_enc(ii) << mq) | gray_enc(iq);
        xh = make_float2((2.0f * (float)ii - (float)((1u << mi) - 1u)) * al,
                         (2.0f * (float)iq - (float)((1u << mq) - 1u)) * al);
        break; }
    }
    pe = fmaf(r.y, xh.x, -(r.x * xh.y));          // imag(r conj(xhat))
    return sym;
}

// equaliser output: sum over the 13 taps of x[i] conj(w[i]), in the balanced-tree order of a 16-lane xor butterfly
// (slots 13..15 are zeros; the additions of zero are kept, they are part of the canonical order)
FX_DEV float2 eq_sum16(const float2 *x, const float2 *w)
{
    float2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = i < FX_EQ_TAPS ? cmulc(x[i], w[i]) : make_float2(0.0f, 0.0f);
#pragma unroll
    for (int st = 1; st < 16; st <<= 1)
#pragma unroll
        for (int i = 0; i < 16; i += 2 * st) p[i] = cadd(p[i], p[i + st]);
    return p[0];
}

// sample index (relative to the aligned start) at which MF output symbol c appears
FX_DEV int64_t sym_sample(int64_t c, int mfc0) { return mfc0 == 0 ? 2 * c : (c == 0 ? 0 : 2 * c - 1); }

// ---------------------------------------------------------------- coded lengths (device copy of the host rule)
FX_DEV unsigned crc_len(unsigned check)
{
    switch (check) {
    case FX_CRC_CHECKSUM: case FX_CRC_8: return 1;
    case FX_CRC_16: return 2;
    case FX_CRC_24: return 3;
    case FX_CRC_32: return 4;
    default: return 0;
    }
}
FX_DEV int conv_p(unsigned fs)   // puncturing period; 0 = not a K=7 convolutional scheme
{
    switch (fs) {
    case FX_FEC_CONV_V27: return 1;
    case FX_FEC_CONV_V27P23: return 2;
    case FX_FEC_CONV_V27P34: return 3;
    case FX_FEC_CONV_V27P45: return 4;
    case FX_FEC_CONV_V27P56: return 5;
    case FX_FEC_CONV_V27P67: return 6;
    case FX_FEC_CONV_V27P78: return 7;
    default: return 0;
    }
}
FX_DEV bool blk_spec(unsigned fs, unsigned &k, unsigned &n)    // bit-packed block codes
{
    switch (fs) {
    case FX_FEC_HAMMING74: k = 4; n = 7; return true;
    case FX_FEC_HAMMING128: k = 8; n = 12; return true;
    case FX_FEC_GOLAY2412: k = 12; n = 24; return true;
    default: k = n = 0; return false;
    }
}
FX_DEV bool fec_supported(unsigned fs)
{
    unsigned k, n;
    return fs == FX_FEC_NONE || fs == FX_FEC_HAMMING84 || fs == FX_FEC_SECDED7264 || fs == FX_FEC_SECDED2216 ||
           fs == FX_FEC_SECDED3932 || fs == FX_FEC_RS_M8 || blk_spec(fs, k, n) || conv_p(fs) != 0;
}
FX_DEV unsigned fec_enc_len(unsigned fs, unsigned n)
{
    int p = conv_p(fs);
    if (p) {
        unsigned T = 8 * n + 6;
        unsigned bits = p == 1 ? 2 * T : T + (T + (unsigned)p - 1) / (unsigned)p;
        return (bits + 7) / 8;
    }
    unsigned bk, bn;
    if (blk_spec(fs, bk, bn)) { unsigned nb = (8 * n + bk - 1) / bk; return (nb * bn + 7) / 8; }
    if (fs == FX_FEC_RS_M8) { unsigned nb = (n + 222) / 223; if (nb == 0) nb = 1; const unsigned dl = (n + nb - 1) / nb; return nb * (dl + 32); }
    if (fs == FX_FEC_SECDED2216) return 3 * (n / 2) + ((n % 2) ? (n % 2) + 1 : 0);
    if (fs == FX_FEC_SECDED3932) return 5 * (n / 4) + ((n % 4) ? (n % 4) + 1 : 0);
    if (fs == FX_FEC_HAMMING84) return 2 * n;
    if (fs == FX_FEC_SECDED7264) return 9 * (n / 8) + ((n % 8) ? (n % 8) + 1 : 0);
    return n;
}
