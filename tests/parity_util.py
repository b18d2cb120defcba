"""Helpers shared by the GPU parity tests: run the same IQ through the CPU oracle and through the
C-ABI of libfxrx.so and compare frame by frame."""
import numpy as np

# stated tolerances: integer/byte/index results must be identical; floating-point
# outputs are compared against these bounds (in practice they come out bit-identical).
TOL_EST = 1e-5      # tau, gamma, dphi, phi, pilot estimates (absolute)
TOL_SYM = 1e-4      # payload symbols after carrier recovery, unit-energy constellation (absolute)


def oracle_frames(oracle, x, chunk=256, threshold=None, equalizer=False, soft=False):
    s = oracle.Sync(threshold=threshold, equalizer=equalizer, soft=soft)
    fr = list(s.execute(x, chunk=chunk))
    s.close()
    return fr


def compare_frames(of, gf, check_syms=True):
    """of: oracle Frame objects, gf: product dicts.  Returns max float deviations."""
    assert len(of) == len(gf), "frame count: oracle %d vs gpu %d" % (len(of), len(gf))
    dev = dict(est=0.0, sym=0.0, evm=0.0, bitexact_syms=True)
    for a, b in zip(of, gf):
        i = a.info
        assert i["start"] == b["start"], (i["start"], b["start"])
        assert i["offset"] == b["cfo_bin"]
        assert a.header_valid == b["header_valid"]
        assert a.header == b["header"][:14]
        assert i["pfb_index"] == b["pfb_index"]
        for k, kb in (("rxy", "rxy"), ("tau", "tau"), ("gamma", "gamma"), ("dphi", "dphi"), ("phi", "phi")):
            dev["est"] = max(dev["est"], abs(i[k] - b[kb]))
        if not a.header_valid:
            continue
        for k in ("pilot_dphi", "pilot_phi", "pilot_gain"):
            dev["est"] = max(dev["est"], abs(i[k] - b[k]))
        assert a.payload_valid == b["payload_valid"]
        assert a.payload == b["payload"], "payload bytes differ at frame start %d" % i["start"]
        assert (a.mod_scheme, a.mod_bps, a.check, a.fec0, a.fec1) == (
            b["mod_scheme"], b["mod_bps"], b["check"], b["fec0"], b["fec1"])
        assert len(a.framesyms) == b["num_framesyms"]
        dev["evm"] = max(dev["evm"], abs(i["evm_sum"] - b["evm_sum"]) / max(i["evm_sum"], 1e-9))
        if check_syms and b["framesyms"] is not None:
            d = np.abs(a.framesyms - b["framesyms"]).max() if len(a.framesyms) else 0.0
            dev["sym"] = max(dev["sym"], float(d))
            if not np.array_equal(a.framesyms.view(np.uint32), b["framesyms"].view(np.uint32)):
                dev["bitexact_syms"] = False
    assert dev["est"] <= TOL_EST, dev
    assert dev["sym"] <= TOL_SYM, dev
    return dev
