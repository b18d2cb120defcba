"""BASELINE config 1 -- what python/qa_flex_rx.py never tested: flex_tx -> flex_rx loopback, QPSK (PSK4)
r=1/2, 1024-B payloads, CRC-24, driven through the oracle in 256-sample execute calls exactly like
lib/flex_rx_impl.cc:212-215.  CPU only."""
import numpy as np
import pytest


def _channel(x, cfo, ph, snr_db, rng, gain=1.0):
    n = np.arange(len(x))
    y = gain * x * np.exp(1j * (cfo * n + ph))
    s = np.sqrt(0.5 * 10 ** (-snr_db / 10))
    return (y + s * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))).astype(np.complex64)


def test_config1_loopback_100_frames(oracle):
    rng = np.random.default_rng(0x5EED)
    payloads, parts = [], []
    for _ in range(100):
        pl = rng.integers(0, 256, 1024, dtype=np.uint8)
        payloads.append(pl.tobytes())
        parts += [oracle.gen_frame(pl, mod=2, fec0=oracle.FEC_CONV_V27, fec1=oracle.FEC_NONE, check=oracle.CRC_24,
                                   dt=float(rng.uniform(-0.5, 0.5))), np.zeros(256, np.complex64)]
    x = _channel(np.concatenate(parts), 0.013, -2.0, 20.0, rng)
    x = x[: len(x) // 256 * 256]
    s = oracle.Sync()
    fr = s.execute(x, chunk=256)
    assert len(fr) == 100
    for f, pl in zip(fr, payloads):
        assert f.header_valid == 1 and f.payload_valid == 1 and f.payload == pl
        assert f.header == bytes(14)                                   # lib/flex_tx_impl.cc:58-59: 14 zero bytes
        # packet_info as the reference would publish it: (modulation, inner, outer) = (1, 1, 0)
        assert oracle.MOD_BY_INDEX.index(f.mod_scheme) == 1 and oracle.INNER_BY_INDEX.index(f.fec0) == 1 and f.fec1 == oracle.FEC_NONE
        assert len(f.framesyms) == 8224 and f.evm < -15.0
    # chunking must not matter
    s2 = oracle.Sync(); fr2 = s2.execute(x, chunk=None)
    assert [f.info["start"] for f in fr2] == [f.info["start"] for f in fr]


@pytest.mark.parametrize("mod_idx", range(11))
def test_every_modulation_decodes(oracle, mod_idx):
    rng = np.random.default_rng(100 + mod_idx)
    mod = oracle.MOD_BY_INDEX[mod_idx]
    fec0 = oracle.INNER_BY_INDEX[mod_idx % 7]
    pl = rng.integers(0, 256, 200, dtype=np.uint8)
    x = np.concatenate([np.zeros(300, np.complex64), oracle.gen_frame(pl, mod=mod, fec0=fec0, dt=0.2), np.zeros(700, np.complex64)])
    x = _channel(x, -0.02, 0.7, 32.0, rng, gain=0.7)
    fr = oracle.Sync().execute(x)
    assert len(fr) == 1 and fr[0].header_valid and fr[0].payload_valid and fr[0].payload == pl.tobytes()
    assert fr[0].mod_scheme == mod and fr[0].fec0 == fec0


def test_edge_cases(oracle):
    rng = np.random.default_rng(7)
    # empty / noise only / silence: no callbacks, no crash
    assert oracle.Sync().execute(np.zeros(0, np.complex64)) == []
    assert oracle.Sync().execute(np.zeros(4096, np.complex64)) == []
    noise = (0.1 * (rng.standard_normal(20000) + 1j * rng.standard_normal(20000))).astype(np.complex64)
    assert all(not f.header_valid for f in oracle.Sync().execute(noise))
    # zero-length payload and maximum user header content
    hdr = np.arange(14, dtype=np.uint8) + 200
    x = np.concatenate([np.zeros(100, np.complex64), oracle.gen_frame(np.zeros(0, np.uint8), header=hdr), np.zeros(600, np.complex64)])
    fr = oracle.Sync().execute(_channel(x, 0.0, 0.0, 30.0, rng))
    assert len(fr) == 1 and fr[0].payload_valid and fr[0].payload == b"" and fr[0].header == hdr.tobytes()
    # a frame whose header symbols are destroyed -> header_valid = 0, synchroniser recovers for the next frame
    pl = rng.integers(0, 256, 64, dtype=np.uint8)
    f1 = oracle.gen_frame(pl).copy(); f1[200:700] = 0
    x = np.concatenate([np.zeros(64, np.complex64), f1, np.zeros(300, np.complex64), oracle.gen_frame(pl), np.zeros(600, np.complex64)])
    fr = oracle.Sync().execute(_channel(x, 0.01, 0.3, 30.0, rng))
    assert [f.header_valid for f in fr][-1] == 1 and fr[-1].payload == pl.tobytes()
    assert any(not f.header_valid for f in fr)


def test_detector_counts_frames_like_frame_detector_cc(oracle):
    rng = np.random.default_rng(9)
    parts = []
    for _ in range(6):
        parts += [oracle.gen_frame(rng.integers(0, 256, 100, dtype=np.uint8)), np.zeros(int(rng.integers(200, 900)), np.complex64)]
    x = _channel(np.concatenate(parts), 0.03, 1.0, 25.0, rng)
    d = oracle.Detector(0.45)
    dets = d.run(x)
    starts = np.cumsum([0] + [len(p) for p in parts])[0::2][:6]
    found = [x_["pos"] for x_ in dets]
    for s in starts:
        assert any(abs(f - s) <= 1 for f in found), (s, found)
    assert all(abs(x_["dphi"] - 0.03) < 5e-3 for x_ in dets if any(abs(x_["pos"] - s) <= 1 for s in starts))
