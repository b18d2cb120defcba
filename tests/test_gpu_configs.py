"""BASELINE configs 3, 4 and 5 at full size on one GPU (`-m gpu`): size-independent properties over every stream -- every
injected frame is found (detector) / decoded byte-exact (flex_rx), starts strictly increase per stream -- plus the complete
oracle comparison on one stream of each.  The streams are 16 (28 for the sweep) distinct synthetic captures tiled to the
configured stream count (the device buffers are shared between the tiles; every tile is still walked and decoded)."""
import numpy as np
import pytest
from parity_util import oracle_frames, compare_frames

pytestmark = pytest.mark.gpu


def _run(fx, n_streams, n_samples, gen, mode, distinct, **ctx_kw):
    import torch
    xs, inj, dev = [], [], []
    for i in range(distinct):
        x, f = gen(i, n_samples)
        xs.append(x); inj.append(f); dev.append(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    ctx = fx.RxContext(n_streams, mode=mode, **ctx_kw)
    n = ctx.process_raw([dev[s % distinct].data_ptr() for s in range(n_streams)], [n_samples] * n_streams, True)
    res = ctx.results(n)
    per = [[] for _ in range(n_streams)]
    for g in res: per[g["stream"]].append(g)
    tm = ctx.timing()
    ctx.close()
    return xs, inj, per, tm


def test_config3_detector_256_streams_full_size(fx, oracle):
    """config 3: frame_detector_cc only, 256 streams x 2^20 samples (2 GiB of IQ)."""
    xs, inj, per, tm = _run(fx, 256, 1 << 20, lambda i, n: fx.synth_stream(n, stream_id=3000 + i), fx.MODE_DETECTOR, 16, threshold=0.45)
    for s in range(256):
        pos = [g["start"] for g in per[s]]
        assert all(b > a for a, b in zip(pos, pos[1:]))
        ps = set(pos)
        assert all((p in ps or p - 1 in ps or p + 1 in ps) for p, _ in inj[s % 16]), "stream %d misses an injected frame" % s
        assert pos == [g["start"] for g in per[s % 16]]                       # tiles of the same capture agree
    od = oracle.Detector(0.45).run(xs[5])
    assert [d["pos"] for d in od] == [g["start"] for g in per[5]] and [d["offset"] for d in od] == [g["cfo_bin"] for g in per[5]]
    for d, g in zip(od, per[5]):
        for k in ("tau", "gamma", "dphi", "phi", "rxy"):
            assert abs(d[k] - g[k]) <= 1e-5
    print("config 3: %d detections, walk %.2f ms" % (sum(len(p) for p in per), tm["walk_ms"]))


def test_config4_qam16_r23_128_streams_full_size(fx, oracle):
    """config 4: flex_rx batched, 128 streams x 2^21 samples, QAM16 r=2/3 (V27P23)."""
    xs, inj, per, tm = _run(fx, 128, 1 << 21, lambda i, n: fx.synth_stream(n, stream_id=4000 + i, mod=27, fec0=15, snr_db=25.0), fx.MODE_FLEX_RX, 16)
    missed = []
    for s in range(128):
        mine = per[s]
        st = [g["start"] for g in mine]
        assert all(b > a for a, b in zip(st, st[1:]))
        by = {g["start"]: g for g in mine}
        for p, pl in inj[s % 16]:
            g = by.get(p) or by.get(p - 1) or by.get(p + 1)
            if g is None: missed.append((s % 16, p)); continue
            assert g["payload_valid"] and g["payload"] == pl and g["mod_scheme"] == 27 and g["fec0"] == 15
        assert [(g["start"], g["payload"]) for g in mine] == [(g["start"], g["payload"]) for g in per[s % 16]]   # tiles of a capture agree
    # A sequential synchroniser does not find every injected frame: a false alarm on the tail of a frame costs it 618 samples,
    # and a preamble inside them is gone (one such place in these 16 captures).  What is missed must be what the oracle misses.
    assert len(set(missed)) <= 2, missed
    for s in sorted(set([3] + [m[0] for m in missed])):
        compare_frames(oracle_frames(oracle, xs[s], chunk=1 << 16), per[s], check_syms=False)
    print("config 4: %d frames, kernels %s" % (sum(len(p) for p in per), {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}))


def test_config5_mod_fec_sweep_one_gpu_share_full_size(fx, oracle):
    """config 5, one GPU's share (128 of the 1024 streams x 2^20 samples): stream s uses modulation {PSK4, QAM16, QAM32, QAM64}
    [s mod 4] and inner code 0..6 [s mod 7] -- the part of the cognitive engine's 616-arm grid that BASELINE names."""
    mods, inner = [2, 27, 28, 29], fx.INNER_BY_INDEX
    xs, inj, per, tm = _run(fx, 128, 1 << 20, lambda i, n: fx.synth_stream(n, stream_id=5000 + i, mod=mods[i % 4], fec0=inner[i % 7], snr_db=32.0),
                            fx.MODE_FLEX_RX, 28)
    for s in range(128):
        by = {g["start"]: g for g in per[s]}
        for p, pl in inj[s % 28]:
            g = by.get(p) or by.get(p - 1) or by.get(p + 1)
            assert g is not None and g["payload_valid"] and g["payload"] == pl
            assert g["mod_scheme"] == mods[(s % 28) % 4] and g["fec0"] == inner[(s % 28) % 7]
    for s in (1, 10, 23):
        compare_frames(oracle_frames(oracle, xs[s], chunk=1 << 16), per[s], check_syms=False)
    print("config 5 share: %d frames, kernels %s" % (sum(len(p) for p in per), {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}))
