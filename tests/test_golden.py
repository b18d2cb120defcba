"""Committed regression vectors (tests/golden/rx_vectors.npz, made by tests/golden/make_golden.py).
Self-generated: the reference holds no fixtures for this path -- see the generator's docstring."""
import os
import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rx_vectors.npz")
FLT = ("rxy", "tau", "gamma", "dphi", "phi", "pilot_dphi", "pilot_phi", "pilot_gain", "evm_sum")


def _check(g, i, start, offset, pfb, hv, pv, ms, bps, chk, f0, f1, flt, header, payload, syms):
    exp = g["f%d_int" % i]
    assert [start, offset, pfb, hv, pv, ms, bps, chk, f0, f1] == exp.tolist()
    assert header == g["f%d_header" % i].tobytes() and payload == g["f%d_payload" % i].tobytes()
    assert np.allclose(flt, g["f%d_flt" % i], rtol=0, atol=1e-5 * np.maximum(1.0, np.abs(g["f%d_flt" % i])))
    if syms is not None:
        assert np.abs(syms - g["f%d_syms" % i]).max() <= 1e-4


def test_oracle_reproduces_golden_vectors(oracle):
    g = np.load(G)
    fr = oracle.Sync().execute(g["iq"], chunk=256)
    assert len(fr) == int(g["n_frames"])
    for i, f in enumerate(fr):
        _check(g, i, f.info["start"], f.info["offset"], f.info["pfb_index"], f.header_valid, f.payload_valid, f.mod_scheme,
               f.mod_bps, f.check, f.fec0, f.fec1, [f.info[k] for k in FLT], f.header, f.payload, f.framesyms)
    dets = oracle.Detector(0.45).run(g["iq"])
    assert [d["pos"] for d in dets] == g["det_pos"].tolist() and [d["offset"] for d in dets] == g["det_off"].tolist()


@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors(fx):
    g = np.load(G)
    for seg in (0, 4096):
        ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
        fr = ctx.process([g["iq"]])
        assert len(fr) == int(g["n_frames"])
        for i, f in enumerate(fr):
            _check(g, i, f["start"], f["cfo_bin"], f["pfb_index"], f["header_valid"], f["payload_valid"], f["mod_scheme"],
                   f["mod_bps"], f["check"], f["fec0"], f["fec1"], [f[k] for k in FLT], f["header"][:14], f["payload"], f["framesyms"])
        ctx.close()
    ctx = fx.RxContext(1, mode=fx.MODE_DETECTOR, threshold=0.45)
    d = ctx.process([g["iq"]])
    assert [t["start"] for t in d] == g["det_pos"].tolist() and [t["cfo_bin"] for t in d] == g["det_off"].tolist()
    flt = np.array([[t[k] for k in ("tau", "gamma", "dphi", "phi", "rxy")] for t in d], np.float32)
    assert np.allclose(flt, g["det_flt"], rtol=0, atol=1e-5)
