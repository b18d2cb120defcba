"""Randomised differential test of the GPU path against the CPU oracle (used by tests/test_gpu_fuzz.py and tools/dev/dev_fuzz.py):
random modulations, inner / outer codes, payload lengths, gaps, SNRs (down to 3 dB), stream counts, segment sizes, pipeline depths,
block cuts, trellis block lengths, plan grids, batch-Viterbi debug paths and hop skipping on / off."""
import os, sys, time
import numpy as np
from parity_util import oracle_frames, compare_frames

KNOBS = ("FXRX_VB_BLK", "FXRX_PLAN_GRID", "FXRX_VB_DEBUG", "FXRX_SKIP_SEEK")


def run_fuzz(fx, oracle, iters, seed, verbose=True):
    """Returns (frames compared, frames whose payload failed its CRC in both); raises AssertionError on the first difference."""
    saved = {k: os.environ.get(k) for k in KNOBS}
    try:
        return _run(fx, oracle, iters, seed, verbose)
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _run(fx, oracle, iters, seed, verbose):
    rng = np.random.default_rng(seed)
    mods, inner, outer = list(fx.MOD_BY_INDEX), list(fx.INNER_BY_INDEX), list(fx.OUTER_BY_INDEX)
    nframes = 0; nbad_payload = 0
    for it in range(iters):
        ns = int(rng.integers(1, 7))
        xs, desc = [], []
        for s in range(ns):
            m, f0, f1 = mods[rng.integers(len(mods))], inner[rng.integers(len(inner))], outer[rng.integers(len(outer))]
            if rng.random() < 0.5: f1 = 1                                   # most traffic has no outer code
            pl = int(rng.choice([0, 1, 7, 16, 100, 333, 1024, 2000])); gap = int(rng.choice([0, 64, 256, 300, 1500]))
            snr = float(rng.choice([3.0, 5.0, 8.0, 12.0, 20.0, 30.0])); n = int(rng.integers(60_000, 400_000))
            sid = int(rng.integers(1 << 20))
            xs.append(fx.synth_stream(n, stream_id=sid, mod=m, fec0=f0, fec1=f1, payload_len=pl, gap=gap, snr_db=snr)[0])
            desc.append((m, f0, f1, pl, gap, snr, n, sid))
        env = dict(FXRX_VB_BLK=str(rng.choice([0, 128, 192, 448])), FXRX_PLAN_GRID=str(rng.choice([0, 3])), FXRX_VB_DEBUG=str(rng.choice([0, 0, 1, 2])),
                   FXRX_SKIP_SEEK=str(rng.choice([1, 1, 0])))
        os.environ.update(env)
        seg = int(rng.choice([0, 8192, 50_000])); depth = int(rng.choice([1, 3])); ncut = int(rng.choice([1, 1, 3]))
        eq = bool(rng.random() < 0.15); soft = bool(rng.random() < 0.15)            # the optional stages, now and then
        if rng.random() < 0.12:                                                      # ... and the detector-only mode (frame_detector_cc)
            ctx = fx.RxContext(ns, mode=fx.MODE_DETECTOR, threshold=0.45, segment_len=seg); ctx.set_depth(depth)
            got, inflight, keep = [], 0, []
            cuts = [[len(x) * k // ncut for k in range(ncut + 1)] for x in xs]
            for k in range(ncut):
                parts = [np.ascontiguousarray(x[c[k]:c[k + 1]]) for x, c in zip(xs, cuts)]; keep.append(parts)
                if inflight == depth: got += ctx.results(ctx.collect_raw()); inflight -= 1
                ctx.submit_raw([p.ctypes.data for p in parts], [len(p) for p in parts], False); inflight += 1
            while inflight: got += ctx.results(ctx.collect_raw()); inflight -= 1
            ctx.close()
            for s_, x in enumerate(xs):
                want = [d["pos"] for d in oracle.Detector(0.45).run(x) if d["pos"] + 512 <= len(x)]
                mine = [g["start"] for g in got if g["stream"] == s_]
                if mine[:len(want)] != want:
                    k = next((i for i, (a, b) in enumerate(zip(mine, want)) if a != b), min(len(mine), len(want)))
                    raise AssertionError("fuzz iteration %d (seed %d), detector mode: stream %d %s seg %d depth %d cuts %d: %d vs %d detections, first difference at #%d: gpu %s oracle %s (len %d)"
                                         % (it, seed, s_, desc[s_], seg, depth, ncut, len(mine), len(want), k, mine[max(0, k - 2):k + 3], want[max(0, k - 2):k + 3], len(x)))
                nframes += len(want)
            if verbose: print("it %d ok: detector mode, %d streams seg %d depth %d cuts %d" % (it, ns, seg, depth, ncut), flush=True)
            continue
        ofs = [oracle_frames(oracle, x, equalizer=eq, soft=soft) for x in xs]
        ctx = fx.RxContext(ns, want_framesyms=True, segment_len=seg, equalizer=eq, soft_decision=soft); ctx.set_depth(depth)
        got, inflight = [], 0
        cuts = [[len(x) * k // ncut for k in range(ncut + 1)] for x in xs]
        keep = []
        on_dev = bool(rng.random() < 0.25)                                       # inputs already in device memory (used in place)
        for k in range(ncut):
            parts = [np.ascontiguousarray(x[c[k]:c[k + 1]]) for x, c in zip(xs, cuts)]
            if on_dev:
                import torch
                parts = [torch.from_numpy(p).cuda() for p in parts]
            keep.append(parts)
            if inflight == depth: got += ctx.results(ctx.collect_raw()); inflight -= 1
            ctx.submit_raw([p.data_ptr() if on_dev else p.ctypes.data for p in parts], [len(p) for p in parts], on_dev); inflight += 1
        while inflight: got += ctx.results(ctx.collect_raw()); inflight -= 1
        tm = ctx.timing(); ctx.close()
        try:
            for s in range(ns):
                mine = sorted([g for g in got if g["stream"] == s], key=lambda g: g["start"])
                compare_frames(ofs[s], mine)
                nframes += len(ofs[s]); nbad_payload += sum(1 for f in ofs[s] if f.header_valid and not f.payload_valid)
        except AssertionError as e:
            raise AssertionError("fuzz iteration %d (seed %d): streams %s env %s seg %d depth %d cuts %d eq %d soft %d -> %s" % (it, seed, desc, env, seg, depth, ncut, eq, soft, e))
        if verbose:
            print("it %d ok: %d streams, env %s seg %d depth %d cuts %d eq %d soft %d, repairs %d vb_repairs %d fallbacks %d" % (it, ns, env, seg, depth, ncut, eq, soft, tm["repairs"], tm["vb_repairs"], tm["vb_fallbacks"]), flush=True)
    return nframes, nbad_payload
