"""Known-answer / closed-form tests of the CPU oracle's primitives (no GPU).

The reference ships no golden vectors for this path (python/qa_flex_rx.py:34-37 and lib/qa_liquiddsp.cc:30-36
are empty), so the oracle is pinned here against published check values and first-principles identities.
Agreement with a real libliquid stays unverified ("parity unpinned", oracle/fxref.h)."""
import ctypes as C
import numpy as np
import pytest


def test_fft512_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for _ in range(4):
        x = (rng.standard_normal(512) + 1j * rng.standard_normal(512)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.abs(oracle.fft512(x) - ref).max() / np.abs(ref).max() < 5e-7
        refi = np.fft.ifft(x.astype(np.complex128)) * 512
        assert np.abs(oracle.fft512(x, inverse=True) - refi).max() / np.abs(refi).max() < 5e-7
    # impulse / linearity known answers
    e = np.zeros(512, np.complex64); e[3] = 1
    assert np.allclose(oracle.fft512(e), np.exp(-2j * np.pi * 3 * np.arange(512) / 512), atol=1e-6)


def test_sincos_and_atan2_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    c, s = C.c_float(), C.c_float()
    for th in rng.integers(0, 2**32, 2000, dtype=np.uint64):
        L.fxr_sincos_u32(int(th), C.byref(c), C.byref(s))
        a = 2 * np.pi * int(th) / 2**32
        assert abs(c.value - np.cos(a)) < 3e-7 and abs(s.value - np.sin(a)) < 3e-7
    for _ in range(2000):
        y, x = rng.standard_normal(2) * 10 ** rng.uniform(-3, 3)
        assert abs(L.fxr_atan2(C.c_float(y), C.c_float(x)) - np.arctan2(np.float32(y), np.float32(x))) < 1e-6
    assert L.fxr_atan2(C.c_float(0), C.c_float(0)) == 0.0
    assert L.fxr_rad2u32(C.c_float(np.pi)) in (2**31, 2**31 - 128, 2**31 + 128)   # float32(pi) * float32(2^32/2pi)
    assert L.fxr_rad2u32(C.c_float(-np.pi / 2)) == (2**32 - 2**30) or abs(L.fxr_rad2u32(C.c_float(-np.pi / 2)) - (2**32 - 2**30)) <= 128


def test_msequence_m7_is_maximal_length(oracle):
    class MS(C.Structure):
        _fields_ = [(k, C.c_uint) for k in ("m", "g", "a", "n", "v")]
    L = oracle.lib()
    ms = MS()
    L.fxr_mseq_init(C.byref(ms), 7, 0x0089, 1)
    L.fxr_mseq_advance.restype = C.c_uint
    bits = [L.fxr_mseq_advance(C.byref(ms)) for _ in range(254)]
    assert bits[:127] == bits[127:]                                    # period 127
    assert all(bits[:127][k:] + bits[:127][:k] != bits[:127] for k in range(1, 127))
    assert sum(bits[:127]) == 64                                       # balance property
    pn = oracle.table("fxr_preamble_pn", 64)
    assert np.allclose(np.abs(pn), 1.0, atol=1e-6)


def test_rrc_pair_is_nyquist(oracle):
    h = oracle.table("fxr_tx_taps", 29, False)
    H = oracle.table("fxr_mf_proto", 897, False)
    assert abs((h ** 2).sum() - 2.0) < 1e-5 and abs((H ** 2).sum() - 64.0) < 1e-3
    assert np.allclose(h, h[::-1], atol=1e-6)                          # linear phase
    L = oracle.lib()
    for br in (7, 16, 31):                                             # branch br undoes a TX delay of br/32 sample
        best = 1.0
        for sign in (-1.0, 1.0):
            hd = np.zeros(29, np.float32)
            L.fxr_firdes_arkaiser(2, 7, C.c_float(0.3), C.c_float(sign * br / 32.0), hd.ctypes.data)
            c = np.convolve(hd, H[br::32][:28])
            k = int(np.argmax(np.abs(c)))
            best = min(best, float(np.abs(np.delete(c[k % 2::2], k // 2)).max() / c[k]))
        assert best < 5e-3
    c = np.convolve(h, H[0::32][:28])
    assert np.argmax(c) == 28 and np.abs(np.delete(c[0::2], 14)).max() < 5e-3    # aligned branch: ISI-free


def test_crc_check_values(oracle):
    L = oracle.lib()
    msg = np.frombuffer(b"123456789", dtype=np.uint8).copy()
    assert L.fxr_crc_key(oracle.CRC_32, msg.ctypes.data, 9) == 0xCBF43926    # CRC-32/ISO-HDLC check value
    assert L.fxr_crc_key(oracle.CRC_16, msg.ctypes.data, 9) == 0xB4C8        # CRC-16/MODBUS family w/ final xor: poly 0x8005 reflected, init/xorout 0xFFFF
    assert L.fxr_crc_key(oracle.CRC_CHECKSUM, msg.ctypes.data, 9) == (-sum(msg.tolist())) & 0xFF


def test_interleaver_is_a_bit_permutation_and_inverts(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    for n in (1, 2, 3, 27, 54, 100, 1027, 2056):
        x = rng.integers(0, 256, n, dtype=np.uint8)
        y = x.copy(); L.fxr_interleave(y.ctypes.data, n, 0)
        assert np.unpackbits(y).sum() == np.unpackbits(x).sum()
        z = y.copy(); L.fxr_interleave(z.ctypes.data, n, 1)
        assert np.array_equal(z, x)
        if n >= 27:
            assert not np.array_equal(y, x)


@pytest.mark.parametrize("fec", [1, 4, 5, 6, 7, 8, 9, 10, 11, 15, 16, 17, 18, 19, 20, 27])
def test_fec_roundtrip_and_error_correction(oracle, fec):
    L = oracle.lib()
    rng = np.random.default_rng(fec)
    for n in (1, 8, 13, 64, 257):
        msg = rng.integers(0, 256, n, dtype=np.uint8)
        el = L.fxr_fec_enc_len(fec, n)
        enc = np.zeros(el + 8, np.uint8); dec = np.zeros(n + 8, np.uint8)
        L.fxr_fec_encode(fec, n, msg.ctypes.data, enc.ctypes.data)
        L.fxr_fec_decode(fec, n, enc.ctypes.data, dec.ctypes.data)
        assert np.array_equal(dec[:n], msg)
        if fec == 1:
            continue
        # isolated single-bit errors, far apart, must be corrected by every code in the menu
        bad = enc.copy()
        step = 64 if 11 <= fec <= 20 else {4: 7, 5: 8, 6: 12, 7: 24, 8: 24, 9: 40, 10: 72, 27: 200}[fec]
        for b in range(3, 8 * el, step * 4 if 11 <= fec <= 20 else step):
            bad[b >> 3] ^= 0x80 >> (b & 7)
        L.fxr_fec_decode(fec, n, bad.ctypes.data, dec.ctypes.data)
        assert np.array_equal(dec[:n], msg)


def test_conv_code_lengths_match_reference_sizes(oracle):
    L = oracle.lib()
    # 1024 B + CRC24 -> 2056 B (r=1/2) -> 8224 PSK4 symbols; header 20 B -> 54 B -> 216 QPSK
    assert L.fxr_packet_enc_len(1024, oracle.CRC_24, oracle.FEC_CONV_V27, oracle.FEC_NONE) == 2056
    assert L.fxr_qpm_sym_len(1024, oracle.CRC_24, oracle.FEC_CONV_V27, oracle.FEC_NONE, 2) == 8224
    assert L.fxr_packet_enc_len(20, oracle.CRC_32, oracle.FEC_SECDED7264, oracle.FEC_HAMMING84) == 54
    assert len(oracle.gen_frame(np.zeros(1024, np.uint8))) == 17066


def test_packet_roundtrip_and_crc_detects_damage(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for check in (oracle.CRC_NONE, oracle.CRC_CHECKSUM, oracle.CRC_8, oracle.CRC_16, oracle.CRC_24, oracle.CRC_32):
        for fec0, fec1 in ((1, 1), (11, 1), (15, 5), (20, 10)):
            n = int(rng.integers(0, 200))
            msg = rng.integers(0, 256, max(n, 1), dtype=np.uint8)[:n]
            el = L.fxr_packet_enc_len(n, check, fec0, fec1)
            pkt = np.zeros(el + 8, np.uint8); out = np.zeros(n + 8, np.uint8)
            L.fxr_packet_encode(n, check, fec0, fec1, msg.ctypes.data, pkt.ctypes.data)
            assert L.fxr_packet_decode(n, check, fec0, fec1, pkt.ctypes.data, out.ctypes.data) == 1
            assert np.array_equal(out[:n], msg)
            if check >= oracle.CRC_16 and fec0 == 1 and fec1 == 1 and n > 4:
                pkt[1] ^= 0x5A
                assert L.fxr_packet_decode(n, check, fec0, fec1, pkt.ctypes.data, out.ctypes.data) == 0


def test_modem_roundtrip_all_schemes(oracle):
    L = oracle.lib()

    class Modem(C.Structure):
        _fields_ = [("ms", C.c_int), ("bps", C.c_uint), ("dpsk_phi", C.c_float)]

    class C32(C.Structure):
        _fields_ = [("re", C.c_float), ("im", C.c_float)]
    L.fxr_modem_mod.restype = C32
    L.fxr_modem_mod.argtypes = [C.POINTER(Modem), C.c_uint]
    L.fxr_modem_demod.restype = C.c_uint
    L.fxr_modem_demod.argtypes = [C.POINTER(Modem), C32, C.POINTER(C32), C.POINTER(C.c_float)]
    rng = np.random.default_rng(4)
    for name, ms in oracle.MODEM.items():
        bps = L.fxr_modem_bps(ms)
        tx, rx = Modem(), Modem()
        L.fxr_modem_init(C.byref(tx), ms); L.fxr_modem_init(C.byref(rx), ms)
        pts, e = set(), 0.0
        for s in list(range(1 << bps)) + rng.integers(0, 1 << bps, 200).tolist():
            p = L.fxr_modem_mod(C.byref(tx), s)
            pts.add((round(p.re, 4), round(p.im, 4))); e += p.re ** 2 + p.im ** 2
            xh, pe = C32(), C.c_float()
            noisy = C32(p.re + 0.01, p.im - 0.01)
            assert L.fxr_modem_demod(C.byref(rx), noisy, C.byref(xh), C.byref(pe)) == s, name
            assert abs(xh.re - p.re) < 1e-6 and abs(xh.im - p.im) < 1e-6 and abs(pe.value) < 0.2
        assert len(pts) == 1 << bps, name                               # all constellation points distinct
        if not name.startswith("D"):
            assert abs(e / (200 + (1 << bps)) - 1.0) < 0.12, name       # unit average energy
