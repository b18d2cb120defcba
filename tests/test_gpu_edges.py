"""GPU parity tests for the corners of the exact-speculation machinery: frames with no gap,
overlapping frames, a valid preamble right behind a broken header, the densest possible traffic -- each over several segment sizes (so that segment boundaries fall everywhere) and against the sequential oracle."""
import numpy as np
import pytest
from parity_util import oracle_frames, compare_frames

pytestmark = pytest.mark.gpu


def _chan(x, cfo, ph, snr_db, rng, gain=1.0):
    n = np.arange(len(x))
    y = gain * x * np.exp(1j * (cfo * n + ph))
    s = np.sqrt(0.5 * 10 ** (-snr_db / 10))
    return (y + s * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))).astype(np.complex64)


def _place(parts, total=None):
    """parts: list of (offset, samples); overlapping parts add."""
    n = max(o + len(f) for o, f in parts) + 1200 if total is None else total
    x = np.zeros(n, np.complex64)
    for o, f in parts:
        x[o:o + len(f)] += f
    return x


SEGS = (0, 4096, 6000, 8192, 20000)


def _check(fx, oracle, x, segs=SEGS, min_frames=1):
    of = oracle_frames(oracle, x)
    assert len(of) >= min_frames, "test input produced only %d oracle frames" % len(of)
    reps = 0
    for seg in segs:
        ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
        compare_frames(of, ctx.process([x]))
        reps += ctx.timing()["repairs"]
        ctx.close()
    return of, reps


def test_zero_gap_and_overlapping_frames(fx, oracle):
    """Frames back to back with no gap at all, and frames whose head overlaps the flush tail of the frame before: the
    detector restarts right on top of the next preamble (window half zeros, detection below the floor)."""
    rng = np.random.default_rng(31)
    g = fx.FrameGen()
    for trial, (gaps, plen) in enumerate([([0] * 30, 100), ([0, -6, -12, -20, -28, 3, 1, -2] * 4, 64), ([-27, -26, -25, -24] * 6, 33)]):
        parts, p = [], 300
        for k, gp in enumerate(gaps):
            f = g.frame(rng.integers(0, 256, plen + k, dtype=np.uint8))
            parts.append((p, f)); p += len(f) + gp
        x = _chan(_place(parts), 0.011 * (trial + 1), 0.7 * trial, 26.0, rng)
        of, _ = _check(fx, oracle, x, min_frames=len(gaps) // 2)
        assert sum(f.payload_valid for f in of) >= len(gaps) // 2
    g.close()


def test_valid_frame_close_behind_a_broken_header(fx, oracle):
    """A preamble whose header is destroyed costs the synchroniser 618 samples (reset with floor = start + 618); a valid
    frame starting inside or just behind that span is detected with its start below the floor.  Spread such pairs so
    that segment boundaries of every size fall between them."""
    rng = np.random.default_rng(32)
    g = fx.FrameGen()
    parts, p = [], 200
    for k in range(40):
        bad = g.frame(rng.integers(0, 256, 20, dtype=np.uint8)).copy()
        bad[180:] = 0                                                # preamble intact, header and everything after gone
        bad = bad[:620]
        parts.append((p, bad))
        d = [400, 500, 590, 610, 617, 618, 619, 640, 700, 900][k % 10] + int(rng.integers(0, 3))
        good = g.frame(rng.integers(0, 256, 48 + k, dtype=np.uint8))
        parts.append((p + d, good))
        p += d + len(good) + int(rng.integers(100, 1500))
    x = _chan(_place(parts), -0.017, 1.1, 27.0, rng)
    of, _ = _check(fx, oracle, x, min_frames=50)
    assert sum(1 for f in of if not f.header_valid) >= 20 and sum(f.payload_valid for f in of) >= 25
    g.close()


@pytest.mark.parametrize("plen", [0, 7, 16])
def test_dense_tiny_frames_overflow_the_frame_table(fx, oracle, plen):
    """Header-only and tiny-payload frames are ~630-680 samples long: the densest traffic there is.  (Walk jobs' frame
    tables are sized for exactly this, one slot per 600 samples of segment.)"""
    rng = np.random.default_rng(33 + plen)
    g = fx.FrameGen(mod=29, fec0=1)                                 # QAM64, no FEC: shortest frames
    parts, p = [], 100
    for k in range(420):
        f = g.frame(rng.integers(0, 256, plen, dtype=np.uint8))
        parts.append((p, f)); p += len(f) + int(rng.integers(0, 24))
    g.close()
    x = _chan(_place(parts), 0.004, -0.4, 30.0, rng)
    of, reps = _check(fx, oracle, x, segs=(0, 8192, 40000, 1 << 18), min_frames=400)
    assert sum(f.payload_valid for f in of) >= 400


def test_dense_traffic_across_speculative_blocks(fx, oracle):
    """The same dense traffic fed as big blocks of one continuing stream, several in flight (cross-block speculation)."""
    rng = np.random.default_rng(36)
    g = fx.FrameGen(mod=2, fec0=1)
    parts, p = [], 50
    while p < 1_150_000:
        f = g.frame(rng.integers(0, 256, int(rng.integers(0, 17)), dtype=np.uint8))
        parts.append((p, f)); p += len(f) + int(rng.integers(0, 40))
    g.close()
    x = _chan(_place(parts), 0.02, 0.2, 28.0, rng)
    of = oracle_frames(oracle, x)
    assert len(of) > 1200
    cuts = [0, 300_000, 600_123, 900_000, len(x)]
    ctx = fx.RxContext(1, want_framesyms=True)
    ctx.set_depth(3)
    keep = [np.ascontiguousarray(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    got, inflight = [], 0
    for pc in keep:
        if inflight == 3:
            got += ctx.results(ctx.collect_raw()); inflight -= 1
        ctx.submit_raw([pc.ctypes.data], [len(pc)], False); inflight += 1
    while inflight:
        got += ctx.results(ctx.collect_raw()); inflight -= 1
    compare_frames(of, got)


def test_repairs_with_blocks_in_flight_behind_them(fx, oracle):
    """Weak preambles around the detector's threshold make skipped hops fire under verification (and hand-off targets go
    missing): the lean chain kernel leaves such a block unstitched, fxrx_collect has the full-size chain kernel walk the
    stretch again -- and the blocks already in flight behind it, which found no state to start from, are enqueued again.
    Continuing blocks, three in flight, against the sequential oracle."""
    rng = np.random.default_rng(2026)
    g = fx.FrameGen()
    parts = [np.zeros(300, np.complex64)]
    for k in range(260):
        amp = 1.0 if k % 7 == 0 else rng.uniform(0.16, 0.5)
        parts += [amp * g.frame(rng.integers(0, 256, 64, dtype=np.uint8)), np.zeros(int(rng.integers(300, 3000)), np.complex64)]
    g.close()
    x = _chan(np.concatenate(parts), 0.013, 0.9, 12.0, rng)
    of = oracle_frames(oracle, x)
    assert len(of) > 150
    nb = 8
    cuts = [len(x) * k // nb for k in range(nb + 1)]
    ctx = fx.RxContext(1, want_framesyms=True, segment_len=16384)
    ctx.set_depth(3)
    keep = [np.ascontiguousarray(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    got, inflight, fails = [], 0, 0
    for pc in keep:
        if inflight == 3:
            got += ctx.results(ctx.collect_raw()); fails += ctx.timing()["verify_failures"]; inflight -= 1
        ctx.submit_raw([pc.ctypes.data], [len(pc)], False); inflight += 1
    while inflight:
        got += ctx.results(ctx.collect_raw()); fails += ctx.timing()["verify_failures"]; inflight -= 1
    compare_frames(of, got)
    assert fails > 0 and ctx.timing()["replays"] > 0, "no block needed a repair: the path was not exercised"


@pytest.mark.parametrize("snr", [3.0, 4.5])
def test_low_snr_hand_off_misses_are_mended_in_parallel_rounds(fx, oracle, snr):
    """Near the detector's threshold, what a walker finds depends on where its hops happen to fall: speculative lists and
    the true chain disagree many times per block (a third of the headers fail at 3 dB).  fxrx_collect mends those in repair
    rounds -- every segment whose predecessor's hand-off target is missing is walked again from the true state, all at
    once, each walk carrying on until it meets a list that holds its hand-off -- instead of one workgroup walking them one
    after the other.  Continuing blocks, small segments (hundreds of them), against the sequential oracle."""
    x = fx.synth_stream(3_000_000, stream_id=77, snr_db=snr, payload_len=300)[0]
    of = oracle_frames(oracle, x)
    assert sum(1 for f in of if not f.header_valid) > 20 or snr > 4.0
    ctx = fx.RxContext(1, want_framesyms=True, segment_len=16384)
    ctx.set_depth(2)
    cuts = [0, 1_000_000, 2_100_000, len(x)]
    keep = [np.ascontiguousarray(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    got, inflight, reps = [], 0, 0
    for pc in keep:
        if inflight == 2:
            got += ctx.results(ctx.collect_raw()); reps += ctx.timing()["repairs"]; inflight -= 1
        ctx.submit_raw([pc.ctypes.data], [len(pc)], False); inflight += 1
    while inflight:
        got += ctx.results(ctx.collect_raw()); reps += ctx.timing()["repairs"]; inflight -= 1
    compare_frames(of, got)
    assert reps > 0, "no hand-off miss: the repair rounds were not exercised"
    ctx.close()
