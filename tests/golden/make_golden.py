"""Generates tests/golden/rx_vectors.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

These are SELF-GENERATED regression vectors: the reference (gvanhoy/gr-liquiddsp) ships no golden
vectors or fixtures for this path and liquid-dsp itself is not available here, so nothing in this file
pins agreement with libliquid ("parity unpinned").  What it pins: the oracle's and the HIP path's outputs
on a fixed IQ capture stay what they were when this file was committed."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_ffi as o  # noqa: E402


def build():
    rng = np.random.default_rng(20261004)
    cfgs = [(2, o.FEC_CONV_V27, o.FEC_NONE, o.CRC_24, 96),          # headline scheme, short payload
            (27, o.FEC_CONV_V27P23, o.FEC_NONE, o.CRC_24, 120),      # config-4 scheme (QAM16 r2/3)
            (3, o.FEC_NONE, o.FEC_HAMMING84, o.CRC_32, 40),          # PSK8, block outer code
            (10, o.FEC_CONV_V27P78, o.FEC_SECDED7264, o.CRC_16, 64)]  # DPSK4, r7/8 + SECDED
    parts, payloads = [np.zeros(100, np.complex64)], []
    for mod, f0, f1, chk, n in cfgs:
        pl = rng.integers(0, 256, n, dtype=np.uint8)
        hdr = rng.integers(0, 256, 14, dtype=np.uint8)
        payloads.append(pl)
        parts += [o.gen_frame(pl, mod=mod, fec0=f0, fec1=f1, check=chk, header=hdr, dt=float(rng.uniform(-0.5, 0.5))),
                  np.zeros(int(rng.integers(256, 700)), np.complex64)]
    x = np.concatenate(parts)
    n = np.arange(len(x))
    x = 0.8 * x * np.exp(1j * (0.031 * n - 1.1))
    x = (x + 0.02 * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))).astype(np.complex64)
    x = x[: len(x) // 256 * 256]
    s = o.Sync()
    fr = s.execute(x, chunk=256)
    assert len(fr) == len(cfgs) and all(f.payload_valid for f in fr)
    out = dict(iq=x, n_frames=np.int32(len(fr)))
    for i, f in enumerate(fr):
        out["f%d_int" % i] = np.array([f.info["start"], f.info["offset"], f.info["pfb_index"], f.header_valid, f.payload_valid,
                                       f.mod_scheme, f.mod_bps, f.check, f.fec0, f.fec1], np.int64)
        out["f%d_flt" % i] = np.array([f.info[k] for k in ("rxy", "tau", "gamma", "dphi", "phi", "pilot_dphi", "pilot_phi",
                                                          "pilot_gain", "evm_sum")], np.float32)
        out["f%d_header" % i] = np.frombuffer(f.header, np.uint8)
        out["f%d_payload" % i] = np.frombuffer(f.payload, np.uint8)
        out["f%d_syms" % i] = f.framesyms
    d = o.Detector(0.45)
    dets = d.run(x)
    out["det_pos"] = np.array([t["pos"] for t in dets], np.int64)
    out["det_off"] = np.array([t["offset"] for t in dets], np.int32)
    out["det_flt"] = np.array([[t[k] for k in ("tau", "gamma", "dphi", "phi", "rxy")] for t in dets], np.float32)
    return out


if __name__ == "__main__":
    out = build()
    np.savez_compressed(os.path.join(HERE, "rx_vectors.npz"), **out)
    print("wrote rx_vectors.npz:", len(out["iq"]), "samples,", int(out["n_frames"]), "frames,", len(out["det_pos"]), "detections")
