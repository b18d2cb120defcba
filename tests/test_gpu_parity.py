"""GPU parity tests: the HIP path, called through the C ABI of libfxrx.so, against the CPU oracle on
identical IQ.  Bar: frame positions, CFO bins, polyphase branch, header bytes,
payload bytes and validity flags identical; float estimates within 1e-5, payload symbols within 1e-4
(they are normally bit-identical -- the tests report it)."""
import ctypes as C
import numpy as np
import pytest
from parity_util import oracle_frames, compare_frames

pytestmark = pytest.mark.gpu


def _chan(x, cfo, ph, snr_db, rng, gain=1.0):
    n = np.arange(len(x))
    y = gain * x * np.exp(1j * (cfo * n + ph))
    s = np.sqrt(0.5 * 10 ** (-snr_db / 10))
    return (y + s * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))).astype(np.complex64)


def test_native_library_is_loaded_and_device_present(fx):
    assert fx.lib().fxrx_device_count() >= 1
    assert "libfxrx.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("seg", [0, 4096, 8192, 50000, 1 << 20])
def test_single_stream_qpsk_r12_parity_any_segmentation(fx, oracle, seg):
    """BASELINE config 2 scheme at a size the oracle does in a second; speculation granularity must not matter."""
    x, inj = fx.synth_stream(400_000, stream_id=11)
    of = oracle_frames(oracle, x)
    ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
    gf = ctx.process([x])
    dev = compare_frames(of, gf)
    assert len(gf) == len(inj) and all(g["payload"] == pl for g, (_, pl) in zip(gf, inj))
    print("seg=%d jobs=%d repairs=%d deviations=%s" % (seg, ctx.timing()["walk_jobs"], ctx.timing()["repairs"], dev))


def test_all_modulations_and_codes_batched(fx, oracle):
    """One batch, 11 streams, each its own modulation / inner code (the mod/FEC sweep of config 5 in small)."""
    xs, exp = [], []
    for i, mod in enumerate(fx.MOD_BY_INDEX):
        fec0 = fx.INNER_BY_INDEX[i % 7]
        x, inj = fx.synth_stream(60_000 + 1000 * i, stream_id=100 + i, mod=mod, fec0=fec0, payload_len=300 + 17 * i,
                                 snr_db=32.0, gap=256 + 50 * i, lead=10 * i)
        xs.append(x); exp.append(inj)
    ctx = fx.RxContext(len(xs), want_framesyms=True)
    gf = ctx.process(xs)
    for s, x in enumerate(xs):
        of = oracle_frames(oracle, x)
        mine = [g for g in gf if g["stream"] == s]
        compare_frames(of, mine)
        assert len(mine) == len(exp[s]) and all(g["payload_valid"] and g["payload"] == pl for g, (_, pl) in zip(mine, exp[s]))
        assert all(g["mod_scheme"] == fx.MOD_BY_INDEX[s] and g["fec0"] == fx.INNER_BY_INDEX[s % 7] for g in mine)


def test_outer_block_codes_and_checks(fx, oracle):
    xs = []
    for i, (f0, f1, chk) in enumerate([(1, 5, 6), (11, 10, 4), (20, 5, 3), (1, 10, 2), (15, 1, 1),
                                       (11, 7, 5), (1, 4, 5), (15, 6, 5), (17, 8, 6), (1, 9, 5), (19, 7, 4), (11, 27, 5), (1, 27, 6)]):
        x, _ = fx.synth_stream(40_000, stream_id=300 + i, mod=27, fec0=f0, fec1=f1, check=chk, payload_len=257, snr_db=30.0)
        xs.append(x)
    ctx = fx.RxContext(len(xs), want_framesyms=True)
    gf = ctx.process(xs)
    for s, x in enumerate(xs):
        of = oracle_frames(oracle, x)
        assert len(of) >= 2
        compare_frames(of, [g for g in gf if g["stream"] == s])


def test_chunked_feeding_carries_state_exactly(fx, oracle):
    """Feeding a stream in irregular pieces (state carried across calls) == one shot == oracle."""
    x, inj = fx.synth_stream(300_000, stream_id=21, payload_len=700)
    of = oracle_frames(oracle, x)
    ctx = fx.RxContext(1, want_framesyms=True, segment_len=16384)
    got, p = [], 0
    rng = np.random.default_rng(5)
    while p < len(x):
        n = int(rng.choice([1, 255, 256, 1000, 4096, 30000, 70001]))
        got += ctx.process([x[p:p + n]])
        p += n
    compare_frames(of, got)
    assert [g["start"] for g in got] == [f.info["start"] for f in of]


def test_ragged_and_empty_streams(fx, oracle):
    xs = [fx.synth_stream(90_000, stream_id=31)[0], np.zeros(0, np.complex64), fx.synth_stream(37_123, stream_id=32)[0],
          np.zeros(5000, np.complex64), fx.synth_stream(200, stream_id=33)[0]]
    ctx = fx.RxContext(len(xs), want_framesyms=True)
    gf = ctx.process(xs)
    for s, x in enumerate(xs):
        compare_frames(oracle_frames(oracle, x), [g for g in gf if g["stream"] == s])
    assert ctx.process([np.zeros(0, np.complex64)] * len(xs)) == []


def test_noise_false_alarms_and_broken_headers_match(fx, oracle):
    rng = np.random.default_rng(77)
    noise = (0.3 * (rng.standard_normal(300_000) + 1j * rng.standard_normal(300_000))).astype(np.complex64)
    pl = rng.integers(0, 256, 64, dtype=np.uint8)
    g = fx.FrameGen()
    f1 = g.frame(pl).copy(); f1[200:700] = 0                      # header symbols destroyed
    x2 = _chan(np.concatenate([np.zeros(64, np.complex64), f1, np.zeros(300, np.complex64), g.frame(pl), np.zeros(600, np.complex64)]),
               0.01, 0.3, 30.0, rng)
    hdr = np.arange(14, dtype=np.uint8) + 200
    x3 = _chan(np.concatenate([np.zeros(100, np.complex64), g.frame(np.zeros(0, np.uint8), header=hdr), np.zeros(600, np.complex64)]),
               0.0, 0.0, 30.0, rng)
    low = fx.synth_stream(120_000, stream_id=41, snr_db=3.0)[0]   # marginal SNR: CRC failures must match too
    for x in (noise, x2, x3, low):
        of = oracle_frames(oracle, x)
        ctx = fx.RxContext(1, want_framesyms=True, segment_len=8192)
        compare_frames(of, ctx.process([x]))
    of = oracle_frames(oracle, x2)
    assert any(not f.header_valid for f in of) and of[-1].payload == pl.tobytes()


def test_detector_mode_multi_stream_parity(fx, oracle):
    """frame_detector_cc path (config 3 in small): positions and bins identical, estimates within 1e-5."""
    xs = [fx.synth_stream(150_000, stream_id=50 + i, payload_len=64 + 100 * i, snr_db=15.0 + 2 * i)[0] for i in range(6)]
    ctx = fx.RxContext(len(xs), mode=fx.MODE_DETECTOR, threshold=0.45, segment_len=20000)
    gd = ctx.process(xs)
    tot = 0
    for s, x in enumerate(xs):
        od = oracle.Detector(0.45).run(x)
        mine = [g for g in gd if g["stream"] == s]
        assert [d["pos"] for d in od] == [g["start"] for g in mine]
        assert [d["offset"] for d in od] == [g["cfo_bin"] for g in mine]
        for d, g in zip(od, mine):
            for k in ("tau", "gamma", "dphi", "phi", "rxy"):
                assert abs(d[k] - g[k]) <= 1e-5, (k, d[k], g[k])
        tot += len(mine)
    assert tot >= 6 * 8


def test_dropin_flexframesync_callback_contract(fx, oracle):
    """The liquid entry points flex_rx calls (lib/flex_rx_impl.cc:49,213,71): frames arrive through the callback,
    at most one per execute call, in order, with buffers valid inside the callback."""
    L = fx.lib()
    x, inj = fx.synth_stream(120_000, stream_id=61, payload_len=500)
    of = oracle_frames(oracle, x)
    got, per_call = [], []

    def cb(header, hv, payload, plen, pv, st, ud):
        syms = np.frombuffer(C.cast(st.framesyms, C.POINTER(C.c_float * (2 * st.num_framesyms))).contents, np.complex64).copy() \
            if st.num_framesyms else np.zeros(0, np.complex64)
        got.append(dict(header=bytes(header[i] for i in range(14)), hv=hv, pv=pv, payload=C.string_at(payload, plen) if plen else b"",
                        syms=syms, ms=st.mod_scheme, fec0=st.fec0, fec1=st.fec1, evm=st.evm, rssi=st.rssi, cfo=st.cfo))
        return 0
    cbf = fx._ffi.FRAMESYNC_CALLBACK(cb)
    q = L.flexframesync_create(cbf, None)
    assert q
    L.fxrx_sync_set_block(q, 30000)
    for i in range(0, len(x), 256):                                # lib/flex_rx_impl.cc:212-215
        n0 = len(got)
        blk = x[i:i + 256]
        L.flexframesync_execute(q, blk.ctypes.data, len(blk))
        per_call.append(len(got) - n0)
    L.fxrx_sync_flush(q)
    while L.fxrx_sync_pending(q):
        L.flexframesync_execute(q, None, 0)
    L.flexframesync_destroy(q)
    assert max(per_call) <= 1
    assert len(got) == len(of)
    for a, b in zip(of, got):
        assert (a.header_valid, a.payload_valid, a.payload, a.header) == (b["hv"], b["pv"], b["payload"], b["header"])
        assert (a.mod_scheme, a.fec0, a.fec1) == (b["ms"], b["fec0"], b["fec1"])
        assert np.abs(a.framesyms - b["syms"]).max() <= 1e-4
        assert abs(a.evm - b["evm"]) < 1e-3 and abs(a.rssi - b["rssi"]) < 1e-3 and abs(a.cfo - b["cfo"]) < 1e-6


def test_block_api_messages_like_the_reference(fx, oracle):
    """flex_tx -> flex_rx loopback through the block mirrors (BASELINE config 1 plumbing, on the GPU path)."""
    tx = fx.flex_tx.make(1, 1, 0)                                   # PSK4, V27, none
    rng = np.random.default_rng(8)
    payloads = [rng.integers(0, 256, 1024, dtype=np.uint8).tobytes() for _ in range(5)]
    for pl in payloads:
        tx.send_pkt((None, pl))
    parts = []
    for _, vec in tx.messages["pdus"]:
        assert len(vec) == 17066
        parts += [vec, np.zeros(256, np.complex64)]
    x = _chan(np.concatenate(parts), 0.02, 0.5, 25.0, rng)
    x = np.concatenate([x, np.zeros((-len(x)) % 256, np.complex64)])
    rx = fx.flex_rx.make()
    assert rx.output_multiple == 256
    for i in range(0, len(x), 8192):
        blk = x[i:i + 8192]
        assert rx.work(len(blk), [blk], []) == len(blk)
    assert [m[1] for m in rx.messages["payload_data"]] == payloads
    assert rx.messages["packet_info"] == [dict(header_valid=1, payload_valid=1, modulation=1, inner_code=1, outer_code=0)] * 5
    assert [len(m[1]) for m in rx.messages["constellation"]] == [8224] * 5
    with pytest.raises(RuntimeError):
        tx.work(1, [], [])
    det = fx.frame_detector_cc.make()
    out = np.zeros_like(x)
    assert det.work(len(x), [x], [out]) == len(x)
    assert np.array_equal(out, x) and det.d_num_frames == len(oracle.Detector(0.45).run(x))


def test_full_size_config2_roundtrip_and_oracle(fx, oracle):
    """BASELINE config 2 at full size: 20 Msamples, one stream, device-resident.  Size-independent properties
    (every injected payload comes back byte-exact; starts strictly increasing; two segmentations agree) and the
    complete oracle comparison (the oracle needs ~10 s for this)."""
    import torch
    x, inj = fx.synth_stream(20_000_000, stream_id=0)
    xd = torch.from_numpy(x).cuda()
    res = []
    for seg in (0, 100_000):
        ctx = fx.RxContext(1, segment_len=seg)
        res.append(ctx.process([xd]))
        ctx.close()
    a, b = res
    assert len(a) == len(inj) == 1154
    assert all(g["payload_valid"] and g["payload"] == pl for g, (_, pl) in zip(a, inj))
    st = [g["start"] for g in a]
    assert all(y > x_ for x_, y in zip(st, st[1:]))
    key = lambda g: (g["start"], g["cfo_bin"], g["rxy"], g["tau"], g["dphi"], g["phi"], g["payload"], g["evm_sum"])
    assert [key(g) for g in a] == [key(g) for g in b]
    of = oracle_frames(oracle, x, chunk=1 << 16)
    compare_frames(of, a, check_syms=False)


def test_config3_and_config4_shapes_scaled(fx, oracle):
    """configs 3 and 4 at reduced stream count (oracle-checked on a subset, round-trip on all)."""
    # config 4 scheme: QAM16, V27P23
    xs, inj = zip(*[fx.synth_stream(1 << 19, stream_id=400 + i, mod=27, fec0=15, snr_db=25.0) for i in range(8)])
    ctx = fx.RxContext(8, want_framesyms=True)
    gf = ctx.process(list(xs))
    for s in range(8):
        mine = [g for g in gf if g["stream"] == s]
        assert len(mine) == len(inj[s]) and all(g["payload_valid"] and g["payload"] == pl for g, (_, pl) in zip(mine, inj[s]))
    compare_frames(oracle_frames(oracle, xs[3], chunk=1 << 16), [g for g in gf if g["stream"] == 3])
    # config 3: detector over many streams; detections == frames injected (+-1 sample), oracle on one stream
    ctx = fx.RxContext(8, mode=fx.MODE_DETECTOR, threshold=0.45)
    gd = ctx.process(list(xs))
    for s in range(8):
        pos = [g["start"] for g in gd if g["stream"] == s]
        for p, _ in inj[s]:
            assert any(abs(p - q) <= 1 for q in pos)
    od = oracle.Detector(0.45).run(xs[5])
    assert [d["pos"] for d in od] == [g["start"] for g in gd if g["stream"] == 5]


def test_pipelined_submit_collect_equals_blocking(fx, oracle):
    """Several blocks in flight (fxrx_submit / fxrx_collect) must give exactly what one-block-at-a-time gives,
    including state carried from block to block of the same stream."""
    x, inj = fx.synth_stream(600_000, stream_id=71, payload_len=400)
    of = oracle_frames(oracle, x)
    blocks = [x[i:i + 75_000] for i in range(0, len(x), 75_000)]
    ref_ctx = fx.RxContext(1, want_framesyms=True)
    ref = sum((ref_ctx.process([b]) for b in blocks), [])
    ctx = fx.RxContext(1, want_framesyms=True)
    ctx.set_depth(3)
    got, inflight = [], 0
    for b in blocks:
        if inflight == 3:
            got += ctx.results(ctx.collect_raw()); inflight -= 1
        ctx.submit_raw([b.ctypes.data], [len(b)], False); inflight += 1
    while inflight:
        got += ctx.results(ctx.collect_raw()); inflight -= 1
    compare_frames(of, got)
    assert [(g["start"], g["payload"], g["evm_sum"]) for g in got] == [(g["start"], g["payload"], g["evm_sum"]) for g in ref]
    with pytest.raises(fx.rx.RxError):
        ctx.collect_raw()                                      # nothing in flight


def test_cpp_block_shells_loopback(fx, tmp_path):
    """The C++ shells with the reference's class names / make() / work() signature (csrc/blocks/fx_blocks.hpp),
    linked against libfxrx.so only: flex_tx -> flex_rx messages, frame_detector_cc pass-through + count."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_blocks")
    lib = os.path.join(root, "gr-liquiddsp_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "cpp", "test_blocks.cpp"),
                           "-L" + lib, "-lfxrx", "-Wl,-rpath," + lib])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


def test_reference_call_sites_run_through_the_liquid_shim(fx, tmp_path):
    """tests/cpp/test_reference_callsites.cpp: the liquid calls of lib/flex_rx_impl.cc:49,71,181-201,213,
    lib/frame_detector_cc_impl.cc:46-55,63,77 and lib/flex_tx_impl.cc:51-56,188,198-201, with the reference's argument types,
    compiled against include/liquid/liquid.h and libfxrx.so only: TX -> channel -> 256-sample execute calls -> payloads."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_reference_callsites")
    lib = os.path.join(root, "gr-liquiddsp_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-O1", "-I" + os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "cpp", "test_reference_callsites.cpp"), "-L" + lib, "-lfxrx", "-Wl,-rpath," + lib])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


def test_dropin_qdetector_and_msequence(fx, oracle):
    """The per-sample liquid names frame_detector_cc calls (lib/frame_detector_cc_impl.cc:47-55,63,77): every detection
    is reported exactly once with the oracle's estimates; the returned pointer holds the 512 aligned samples."""
    L = fx.lib()
    ms = L.msequence_create(7, 0x0089, 1)
    pn = np.zeros(64, np.complex64)
    for i in range(64):
        re = np.sqrt(0.5) if L.msequence_advance(ms) else -np.sqrt(0.5)
        im = np.sqrt(0.5) if L.msequence_advance(ms) else -np.sqrt(0.5)
        pn[i] = re + 1j * im
    L.msequence_destroy(ms)
    assert L.qdetector_cccf_create_linear(pn.ctypes.data, 63, 7, 2, 7, C.c_float(0.3)) is None      # not the flexframe preamble
    q = L.qdetector_cccf_create_linear(pn.ctypes.data, 64, 7, 2, 7, C.c_float(0.3))
    assert q
    L.qdetector_cccf_set_threshold(q, 0.45)
    assert L.qdetector_cccf_get_buf_len(q) == 512
    x, inj = fx.synth_stream(140_000, stream_id=81, payload_len=200)
    x = np.concatenate([x, np.zeros(70_000, np.complex64)])           # flush the block queue (64 Ki samples)
    od = oracle.Detector(0.45).run(x)
    got = []
    xs = x.view(np.float32).reshape(-1, 2)
    for i in range(len(x)):
        p = L.qdetector_cccf_execute(q, fx._ffi.FxComplex(float(xs[i, 0]), float(xs[i, 1])))
        if p:
            win = np.frombuffer(C.cast(p, C.POINTER(C.c_float * 1024)).contents, np.complex64).copy()
            got.append((L.qdetector_cccf_get_tau(q), L.qdetector_cccf_get_gamma(q), L.qdetector_cccf_get_dphi(q),
                        L.qdetector_cccf_get_phi(q), win))
    L.qdetector_cccf_destroy(q)
    od = [d for d in od if d["pos"] + 512 <= 131072]                  # detections inside the blocks that were run
    assert len(got) >= len(od) >= 8
    for d, g in zip(od, got):
        assert abs(d["tau"] - g[0]) < 1e-5 and abs(d["gamma"] - g[1]) < 1e-5 and abs(d["dphi"] - g[2]) < 1e-5 and abs(d["phi"] - g[3]) < 1e-5
        lo = max(d["pos"], 0)
        assert np.array_equal(g[4][lo - d["pos"]:], x[lo:d["pos"] + 512])


def test_pipelined_multi_stream_detector_mode(fx, oracle):
    xs = [fx.synth_stream(200_000, stream_id=90 + i, payload_len=300)[0] for i in range(3)]
    ctx = fx.RxContext(3, mode=fx.MODE_DETECTOR, threshold=0.45)
    ctx.set_depth(2)
    got = []
    for lo in range(0, 200_000, 50_000):
        blk = [x[lo:lo + 50_000] for x in xs]
        ctx.submit_raw([b.ctypes.data for b in blk], [len(b) for b in blk], False)
        if lo:
            got += ctx.results(ctx.collect_raw())
    got += ctx.results(ctx.collect_raw())
    for s, x in enumerate(xs):
        od = oracle.Detector(0.45).run(x)
        mine = [g for g in got if g["stream"] == s]
        assert [d["pos"] for d in od if d["pos"] + 512 <= 200_000] == [g["start"] for g in mine][:len([d for d in od if d["pos"] + 512 <= 200_000])]
        assert len(mine) >= 10


@pytest.mark.gpu
def test_skipped_hops_are_verified_and_weak_preambles_fall_back(fx, oracle):
    """Locked walkers skip hops their coarse scan finds empty; fx_seekverify_kernel re-checks each with the full
    detector.  Weak preambles (full correlator fires, differential coarse scan does not) must come out exactly as the
    sequential oracle has them -- through the per-stream fallback walk -- and the fallback must actually be exercised."""
    rng = np.random.default_rng(2025)
    g = fx.FrameGen()
    failures = hops = 0
    for trial in range(6):
        parts = [np.zeros(300, np.complex64), g.frame(rng.integers(0, 256, 64, dtype=np.uint8))]
        for k in range(10):
            amp = rng.uniform(0.18, 0.5)                                     # around the detector's threshold in this noise
            parts += [np.zeros(int(rng.integers(300, 3000)), np.complex64), amp * g.frame(rng.integers(0, 256, 64, dtype=np.uint8))]
        parts += [np.zeros(700, np.complex64), g.frame(rng.integers(0, 256, 64, dtype=np.uint8)), np.zeros(2000, np.complex64)]
        x = _chan(np.concatenate(parts), 0.02 * (trial - 2), 0.4 * trial, 12.0, rng)
        of = oracle_frames(oracle, x)
        for seg in (0, 16384):
            ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
            compare_frames(of, ctx.process([x]))
            tm = ctx.timing(); failures += tm["verify_failures"]; hops += tm["verify_hops"]
            ctx.close()
    assert hops > 0, "no hop was ever re-checked: skipping is not active"
    assert failures > 0, "the fallback walk was never exercised by these inputs"


@pytest.mark.gpu
def test_independent_captures_overlap_their_walks_exactly(fx, oracle):
    """After fxrx_reset the next block does not depend on the one still being walked, so its walk is launched before
    that one is stitched (two walk streams).  Different captures, device-resident, several in flight: every capture must
    come out exactly as when processed alone, and a continuing stream (no reset) must still carry its state."""
    import torch
    caps = [fx.synth_stream(150_000 + 7_000 * i, stream_id=90 + i, payload_len=200 + 50 * i, mod=[2, 27, 3, 28][i % 4],
                            fec0=[11, 15, 1, 17][i % 4])[0] for i in range(7)]
    dev = [torch.from_numpy(c).cuda() for c in caps]
    key = lambda g: (g["start"], g["payload"], g["payload_valid"], g["evm_sum"], g["rxy"])
    alone = []
    for c in caps:
        ctx1 = fx.RxContext(1); alone.append([key(g) for g in ctx1.process([c])]); ctx1.close()
    for i in (0, 3):
        compare_frames(oracle_frames(oracle, caps[i]), fx.RxContext(1, want_framesyms=True).process([caps[i]]))
    ctx = fx.RxContext(1)
    ctx.set_depth(4)
    got, inflight = [], 0
    for d in dev:
        if inflight == 4:
            got.append([key(g) for g in ctx.results(ctx.collect_raw())]); inflight -= 1
        ctx.reset()
        ctx.submit_raw([d.data_ptr()], [d.numel()], True); inflight += 1
    while inflight:
        got.append([key(g) for g in ctx.results(ctx.collect_raw())]); inflight -= 1
    assert got == alone
    # the same context, now as one continuing stream cut in three (state must carry: no reset in between)
    x = caps[2]; cuts = [0, 50_000, 101_234, len(x)]
    ctx.reset()
    parts = [torch.from_numpy(x[a:b].copy()).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
    cont = []
    for p in parts:
        ctx.submit_raw([p.data_ptr()], [p.numel()], True)
    for _ in parts:
        cont += [key(g) for g in ctx.results(ctx.collect_raw())]
    assert cont == alone[2]


@pytest.mark.gpu
@pytest.mark.parametrize("on_device", [True, False])
def test_continuing_stream_is_walked_speculatively_across_blocks(fx, oracle, on_device):
    """Big blocks of ONE continuing stream, several in flight: the speculative walkers of block k+1 are launched before
    block k is stitched, its true walker joins them once block k's tail and resume state are known.  Frames (absolute
    positions included) must be exactly those of a single pass / of the sequential oracle, whatever the cut."""
    import torch
    x, inj = fx.synth_stream(1_900_000, stream_id=77, payload_len=300, gap=200)
    key = lambda g: (g["start"], g["payload"], g["payload_valid"], g["evm_sum"], g["rxy"], g["header_valid"])
    one = fx.RxContext(1, want_framesyms=True)
    ref = one.process([x]); one.close()
    compare_frames(oracle_frames(oracle, x), ref)
    for cuts in ([0, 300_000, 650_123, 950_000, 1_300_777, 1_600_000, len(x)], [0, 400_000, 800_000, 1_000_000, 1_100_000, 1_500_000, len(x)]):
        parts = [np.ascontiguousarray(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        keep = [torch.from_numpy(p).cuda() for p in parts] if on_device else parts
        ctx = fx.RxContext(1)
        ctx.set_depth(3)
        got, inflight = [], 0
        for p in keep:
            if inflight == 3:
                got += ctx.results(ctx.collect_raw()); inflight -= 1
            if on_device: ctx.submit_raw([p.data_ptr()], [p.numel()], True)
            else: ctx.submit_raw([p.ctypes.data], [len(p)], False)
            inflight += 1
        while inflight:
            got += ctx.results(ctx.collect_raw()); inflight -= 1
        assert [key(g) for g in got] == [key(g) for g in ref]
        assert ctx.timing()["walk_mode"] == 1                     # the last block was walked speculatively
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("carry", [0, 65536, 4096])
def test_continuing_multi_stream_blocks_and_tail_longer_than_the_carry_buffer(fx, oracle, monkeypatch, carry):
    """Several streams of different block sizes fed as continuing blocks, several in flight, with frames long enough (about
    260 k samples) that the tail carried over a cut exceeds the carry buffer (FXRX_CARRY_SAMPLES shrinks it): the state the
    overflowing block leaves is invalid, the blocks behind it do nothing, and fxrx_collect grows the buffers and enqueues them
    again.  Everything must still equal the single pass."""
    key = lambda g: (g["stream"], g["start"], g["payload"], g["payload_valid"], g["evm_sum"], g["rxy"], g["header_valid"])
    # stream 0: short frames first, then PSK2 r1/2 frames of about 260 k samples: the cut at 700 k falls 150 k samples into
    # the first of them
    xs = [np.concatenate([fx.synth_stream(550_000, stream_id=84, payload_len=64)[0],
                          fx.synth_stream(850_000, stream_id=81, payload_len=8000, mod=1, fec0=11, gap=3000)[0]]),
          fx.synth_stream(1_150_000, stream_id=82, payload_len=500, mod=27, fec0=15)[0],
          fx.synth_stream(1_300_000, stream_id=83, payload_len=64, mod=2, fec0=1)[0]]
    one = fx.RxContext(3)
    ref = [key(g) for g in one.process(xs)]; one.close()
    compare_frames(oracle_frames(oracle, xs[0]), [g for g in fx.RxContext(3, want_framesyms=True).process(xs) if g["stream"] == 0])
    assert len([r for r in ref if r[0] == 0 and len(r[2]) == 8000]) >= 2
    nblk = 4
    cuts = [[int(len(x) * k / nblk) + (137 * s if 0 < k < nblk else 0) for k in range(nblk + 1)] for s, x in enumerate(xs)]
    if carry: monkeypatch.setenv("FXRX_CARRY_SAMPLES", str(carry))
    ctx = fx.RxContext(3)
    ctx.set_depth(3)
    got, inflight, parts_alive, modes = [], 0, [], []
    for k in range(nblk):
        parts = [np.ascontiguousarray(x[cuts[s][k]:cuts[s][k + 1]]) for s, x in enumerate(xs)]
        parts_alive.append(parts)
        if inflight == 3:
            got += ctx.results(ctx.collect_raw()); modes.append(ctx.timing()["walk_mode"]); inflight -= 1
        ctx.submit_raw([p.ctypes.data for p in parts], [len(p) for p in parts], False); inflight += 1
    while inflight:
        got += ctx.results(ctx.collect_raw()); modes.append(ctx.timing()["walk_mode"]); inflight -= 1
    assert sorted(key(g) for g in got) == sorted(ref)
    assert modes == [0, 1, 1, 1]
    if carry: assert ctx.timing()["replays"] >= 1, "the carry buffer never overflowed: the replay path was not exercised"
    ctx.close()


@pytest.mark.gpu
def test_chain_kernel_general_path_equals_fast_path(fx, oracle, monkeypatch):
    """fx_chain_kernel stitches through a parallel fast path when nothing needs repairing and through a sequential general
    path otherwise; FXRX_CHAIN_SLOW=1 forces the general one.  Same frames either way, and both equal the oracle."""
    x, inj = fx.synth_stream(700_000, stream_id=12, payload_len=200)
    of = oracle_frames(oracle, x)
    monkeypatch.setenv("FXRX_CHAIN_SLOW", "1")
    for seg in (0, 4096, 30000):
        ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
        compare_frames(of, ctx.process([x]))
        ctx.close()


@pytest.mark.gpu
def test_reset_between_speculative_blocks(fx, oracle):
    """fxrx_reset while speculative blocks of the old stream are still pending: they must finish with the state their
    predecessors leave behind, and the new stream must start clean.  Reference: the same calls, one block at a time."""
    import torch
    xa = fx.synth_stream(900_000, stream_id=85, payload_len=350)[0]
    xb = fx.synth_stream(800_000, stream_id=86, payload_len=120, mod=3, fec0=16)[0]
    key = lambda g: (g["start"], g["payload"], g["payload_valid"], g["evm_sum"], g["rxy"])
    seq = [("a", xa[:300_000]), ("a", xa[300_000:600_000]), ("a", xa[600_000:]), ("reset", None),
           ("b", xb[:400_123]), ("b", xb[400_123:]), ("reset", None), ("a", xa[:450_000]), ("a", xa[450_000:])]
    blocking = fx.RxContext(1)
    ref = []
    for tag, blk in seq:
        if tag == "reset": blocking.reset()
        else: ref.append([key(g) for g in blocking.process([np.ascontiguousarray(blk)])])
    ctx = fx.RxContext(1); ctx.set_depth(4)
    dev = [None if b is None else torch.from_numpy(np.ascontiguousarray(b)).cuda() for _, b in seq]
    got, inflight = [], 0
    for (tag, _), d in zip(seq, dev):
        if tag == "reset": ctx.reset(); continue
        if inflight == 4:
            got.append([key(g) for g in ctx.results(ctx.collect_raw())]); inflight -= 1
        ctx.submit_raw([d.data_ptr()], [d.numel()], True); inflight += 1
    while inflight:
        got.append([key(g) for g in ctx.results(ctx.collect_raw())]); inflight -= 1
    assert got == ref
    assert sum(len(r) for r in ref) > 300


@pytest.mark.gpu
@pytest.mark.parametrize("mod,fec0,fec1,check", [(2, 11, 1, 6), (29, 27, 6, 5), (1, 17, 10, 3)])
def test_maximum_payload_length(fx, oracle, mod, fec0, fec1, check):
    """The header's 16-bit length field at its maximum: one 65535-byte frame (0.15 to 1.2 M samples depending on the
    scheme; 0.5 M trellis steps for the convolutional codes) between two short ones, against the oracle."""
    rng = np.random.default_rng(65535 + mod)
    g = fx.FrameGen(mod, fec0, fec1, check)
    big = rng.integers(0, 256, 65535, dtype=np.uint8)
    small = rng.integers(0, 256, 40, dtype=np.uint8)
    x = _chan(np.concatenate([np.zeros(500, np.complex64), g.frame(small), np.zeros(300, np.complex64), g.frame(big),
                              np.zeros(280, np.complex64), g.frame(small), np.zeros(1000, np.complex64)]), 0.013, -0.8, 28.0, rng)
    g.close()
    of = oracle_frames(oracle, x)
    assert [len(f.payload) for f in of] == [40, 65535, 40] and all(f.payload_valid for f in of)
    for seg in (0, 1 << 16):
        ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
        gf = ctx.process([x])
        compare_frames(of, gf)
        assert gf[1]["payload"] == big.tobytes()
        ctx.close()


@pytest.mark.gpu
def test_gpu_frame_generator_bit_exact_and_loopback_on_device(fx, oracle):
    """fxtx_generate (the batched GPU counterpart of flex_tx / flexframegen): every frame bit-identical to the oracle's
    generator for all 11 modulations, the FEC menu, fractional delays and odd/even output offsets; then a loopback that
    never leaves the device: GPU generator -> GPU receiver, payloads byte-exact."""
    import torch
    rng = np.random.default_rng(4242)
    mods = list(fx.MOD_BY_INDEX); inner = list(fx.INNER_BY_INDEX); outer = list(fx.OUTER_BY_INDEX)
    frames, off = [], 7
    for i in range(40):
        fr = dict(mod=mods[i % len(mods)], fec0=inner[i % len(inner)], fec1=outer[(i // 3) % len(outer)], check=[3, 4, 5, 6][i % 4],
                  payload=rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8),
                  header=rng.integers(0, 256, 14, dtype=np.uint8) if i % 2 else None,
                  dt=float(rng.uniform(-0.5, 0.5)) if i % 3 else 0.0, offset=off)
        frames.append(fr)
    tx = fx.TxContext()
    for fr in frames:                                              # lay the frames out back to back with small gaps
        fr["offset"] = off
        off += tx.frame_len(fr) + int(rng.integers(0, 5))
    out = torch.zeros(off + 16, dtype=torch.complex64, device="cuda")
    tx.generate(frames, out.data_ptr(), out.numel())
    y = out.cpu().numpy()
    covered = np.zeros(len(y), bool)
    for fr in frames:
        ref = oracle.gen_frame(fr["payload"], mod=fr["mod"], fec0=fr["fec0"], fec1=fr["fec1"], check=fr["check"], header=fr["header"], dt=fr["dt"])
        assert tx.frame_len(fr) == len(ref)
        got = y[fr["offset"]:fr["offset"] + len(ref)]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "frame at %d (mod %d fec %d/%d) differs from the oracle" % (fr["offset"], fr["mod"], fr["fec0"], fr["fec1"])
        prod = fx.FrameGen(fr["mod"], fr["fec0"], fr["fec1"], fr["check"]).frame(fr["payload"], header=fr["header"], dt=fr["dt"])
        assert np.array_equal(prod.view(np.uint32), ref.view(np.uint32))
        covered[fr["offset"]:fr["offset"] + len(ref)] = True
    assert not np.any(y[~covered])                                 # nothing written outside the frames
    # the packet encoder on the GPU (chains without block codes): every modulation x {no code, r1/2, punctured 7/8} x CRCs,
    # empty and odd payload lengths; and the same frames with the encoding forced onto the host
    enc_frames, off2 = [], 3
    for i, m in enumerate(mods):
        for f0 in (1, 11, 20):
            fr = dict(mod=m, fec0=f0, fec1=1, check=[2, 3, 4, 5, 6][(i + f0) % 5], dt=0.25 if i % 2 else 0.0,
                      payload=rng.integers(0, 256, [0, 1, 37, 255, 600][(i + f0) % 5], dtype=np.uint8), offset=off2)
            off2 += tx.frame_len(fr) + 1
            enc_frames.append(fr)
    out2 = torch.zeros(off2 + 8, dtype=torch.complex64, device="cuda")
    tx.generate(enc_frames, out2.data_ptr(), out2.numel())
    y2 = out2.cpu().numpy()
    for fr in enc_frames:
        ref = oracle.gen_frame(fr["payload"], mod=fr["mod"], fec0=fr["fec0"], fec1=fr["fec1"], check=fr["check"], dt=fr["dt"])
        assert np.array_equal(y2[fr["offset"]:fr["offset"] + len(ref)].view(np.uint32), ref.view(np.uint32)), "GPU-encoded frame (mod %d fec0 %d check %d len %d) differs" % (fr["mod"], fr["fec0"], fr["check"], len(fr["payload"]))
    # maximum payload length through the GPU encoder (Reed-Solomon + punctured convolutional code, 0.5 M symbols)
    bigfr = dict(mod=29, fec0=17, fec1=27, check=6, payload=rng.integers(0, 256, 65535, dtype=np.uint8), dt=-0.3, offset=5)
    outb = torch.zeros(tx.frame_len(bigfr) + 16, dtype=torch.complex64, device="cuda")
    tx.generate([bigfr], outb.data_ptr(), outb.numel())
    refb = oracle.gen_frame(bigfr["payload"], mod=29, fec0=17, fec1=27, check=6, dt=-0.3)
    assert np.array_equal(outb.cpu().numpy()[5:5 + len(refb)].view(np.uint32), refb.view(np.uint32))
    # loopback on the device: 200 frames, QAM16 r2/3 + PSK4 r1/2 alternating, straight into the receiver
    lb, off = [], 300
    for i in range(200):
        fr = dict(mod=[27, 2][i % 2], fec0=[15, 11][i % 2], fec1=1, check=5, payload=rng.integers(0, 256, 300 + i, dtype=np.uint8), offset=off)
        off += tx.frame_len(fr) + 240
        lb.append(fr)
    sig = torch.zeros(off + 1000, dtype=torch.complex64, device="cuda")
    tx.generate(lb, sig.data_ptr(), sig.numel())
    sig += 0.02 * torch.randn(sig.numel(), dtype=torch.complex64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    rx = fx.RxContext(1)
    n = rx.process_raw([sig.data_ptr()], [sig.numel()], True)
    res = rx.results(n)
    assert [(g["start"], g["payload"], g["payload_valid"]) for g in res] == [(fr["offset"], fr["payload"].tobytes(), 1) for fr in lb]
    with pytest.raises(RuntimeError):
        tx.generate([dict(payload=np.zeros(10, np.uint8), offset=sig.numel() - 5)], sig.data_ptr(), sig.numel())   # does not fit
    tx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,depth", [("independent", 5), ("continuous", 4), ("reset-every-3", 7)])
def test_pipeline_soak_every_block_checked(fx, mode, depth):
    """120 blocks through submit/collect, every block's results compared: the first 24 with the same calls made one block
    at a time (the serial host path), the rest with the block one period earlier (the capture repeats).  Catches the
    races a handful of blocks would not: stream / event / buffer reuse across slots and walk streams."""
    import torch
    x = fx.synth_stream(1_200_000, stream_id=91, payload_len=256)[0]
    xd = torch.from_numpy(x).cuda()
    ptrs, counts = [xd.data_ptr()], [xd.numel()]
    key = lambda g: (g["start"], g["payload_valid"], hash(g["payload"]), g["evm_sum"], g["rxy"])
    resets = lambda b: mode == "independent" or (mode == "reset-every-3" and b % 3 == 0)
    per = 3 if mode == "reset-every-3" else 1
    ref_ctx, refs = fx.RxContext(1), []
    for b in range(24):
        if resets(b): ref_ctx.reset()
        refs.append([key(g) for g in ref_ctx.results(ref_ctx.process_raw(ptrs, counts, True))])
    assert len(refs[0]) > 150
    ctx = fx.RxContext(1); ctx.set_depth(depth)
    got, inflight = [], 0
    def collect():
        g = [key(r) for r in ctx.results(ctx.collect_raw())]
        b = len(got)
        if b < 24: want = refs[b]
        else: want = [(k[0] + (len(x) * per if mode == "continuous" else 0),) + k[1:] for k in got[b - per]]
        assert g == want, "%s: block %d differs" % (mode, b)
        got.append(g)
    for b in range(120):
        if inflight == depth: collect(); inflight -= 1
        if resets(b): ctx.reset()
        ctx.submit_raw(ptrs, counts, True); inflight += 1
    while inflight: collect(); inflight -= 1
    assert len(got) == 120


@pytest.mark.gpu
def test_equalizer_stage_on_a_multipath_channel(fx, oracle):
    """The optional equaliser stage (liquid's FLEXFRAMESYNC_ENABLE_EQ, which a stock libliquid compiles out --
    hence off by default): 13-tap eqlms at 2 samples/symbol behind the matched filter, trained on the 64 p/n symbols,
    frozen afterwards, symbol instants 3 symbols later.  Three-ray channel, HIP path against the oracle with the stage on
    (usual tolerances), several segmentations and continuing blocks; and the stage must do its job (EVM improves)."""
    h = np.zeros(8, np.complex64); h[0] = 1.0; h[3] = 0.35 * np.exp(1j * 1.1); h[5] = 0.2 * np.exp(-1j * 0.4)
    xs = []
    for i, (mod, f0, pl) in enumerate([(2, 11, 300), (27, 15, 200), (3, 1, 64)]):
        x, _ = fx.synth_stream(260_000, stream_id=500 + i, mod=mod, fec0=f0, payload_len=pl, snr_db=27.0, gap=100 + 77 * i)
        xs.append(np.convolve(x, h)[:len(x)].astype(np.complex64))
    for s, x in enumerate(xs):
        of_on = oracle_frames(oracle, x, equalizer=True)
        of_off = oracle_frames(oracle, x)
        assert len(of_on) >= 10 and sum(f.payload_valid for f in of_on) >= sum(f.payload_valid for f in of_off)
        assert np.mean([f.evm for f in of_on if f.header_valid]) < np.mean([f.evm for f in of_off if f.header_valid]) - 1.0    # dB
        for seg in (0, 8192):
            ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg, equalizer=True)
            dev = compare_frames(of_on, ctx.process([x]))
            ctx.close()
        # default: stage off, the same samples give what the synchroniser without equaliser gives
        ctx = fx.RxContext(1, want_framesyms=True)
        compare_frames(of_off, ctx.process([x])); ctx.close()
    # continuing blocks, several in flight, all three streams in one context
    ctx = fx.RxContext(3, want_framesyms=True, equalizer=True); ctx.set_depth(3)
    cuts = [0, 70_000, 140_123, 200_000, 260_000]
    got, inflight, keep = [], 0, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        parts = [np.ascontiguousarray(x[a:b]) for x in xs]; keep.append(parts)
        if inflight == 3: got += ctx.results(ctx.collect_raw()); inflight -= 1
        ctx.submit_raw([p.ctypes.data for p in parts], [len(p) for p in parts], False); inflight += 1
    while inflight: got += ctx.results(ctx.collect_raw()); inflight -= 1
    for s, x in enumerate(xs):
        compare_frames(oracle_frames(oracle, x, equalizer=True), [g for g in got if g["stream"] == s])
    ctx.close()
    # the drop-in handle: fxrx_sync_set_equalizer
    L = fx.lib(); frames = []
    cbf = fx._ffi.FRAMESYNC_CALLBACK(lambda hd, hv, pl, n, pv, st, ud: frames.append((hv, pv, C.string_at(pl, n) if n else b"")) or 0)
    q = L.flexframesync_create(cbf, None)
    L.fxrx_sync_set_equalizer(q, 1); L.fxrx_sync_set_block(q, 1 << 18)
    x = np.concatenate([xs[0], np.zeros((-len(xs[0])) % 256, np.complex64)])
    L.flexframesync_execute(q, x.ctypes.data, len(x)); L.fxrx_sync_flush(q)
    while L.fxrx_sync_pending(q): L.flexframesync_execute(q, None, 0)
    L.flexframesync_destroy(q)
    of_on = oracle_frames(oracle, xs[0], equalizer=True)
    assert [(f.header_valid, f.payload_valid, f.payload) for f in of_on] == frames[:len(of_on)] and len(frames) >= len(of_on)


@pytest.mark.gpu
def test_soft_decision_demod_and_decode(fx, oracle):
    """Soft-decision option (north_star "soft demod outputs"; liquid's flexframesync_decode_payload_soft, which the
    reference never calls -- hence off by default): per-bit soft values from the carrier-recovered symbols for every scheme,
    soft-input Viterbi for the convolutional stage(s) nearest the channel, hard decisions elsewhere.  Low SNR, so that soft
    and hard decoding differ.  Stated tolerance: soft values within +-1 of the oracle's (they come out identical -- the
    test reports it); decoded bytes, validity flags and everything else as in the hard-decision tests."""
    cases = [(2, 11, 1, 2.5), (27, 15, 1, 9.0), (29, 11, 1, 14.0), (3, 17, 1, 6.5), (1, 11, 7, 1.5), (10, 11, 1, 5.0), (18, 20, 1, 10.0),
             (28, 1, 11, 12.0), (4, 19, 1, 11.0), (9, 1, 1, 4.0), (11, 11, 1, 9.0), (27, 27, 16, 12.0), (2, 6, 11, 3.0), (2, 11, 11, 1.0)]
    xs = [fx.synth_stream(150_000, stream_id=600 + i, mod=m, fec0=f0, fec1=f1, payload_len=150 + 7 * i, snr_db=snr)[0] for i, (m, f0, f1, snr) in enumerate(cases)]
    ctx = fx.RxContext(len(xs), want_framesyms=True, soft_decision=True)
    gf = ctx.process(xs)
    n_soft_differs, exact = 0, True
    for s, x in enumerate(xs):
        of = oracle_frames(oracle, x, soft=True)
        mine = [g for g in gf if g["stream"] == s]
        compare_frames(of, mine)
        hard = oracle_frames(oracle, x)
        n_soft_differs += sum(1 for a, b in zip(of, hard) if a.payload != b.payload)
        for a, b in zip(of, mine):
            if not a.header_valid: continue
            assert b["soft_bits"] is not None and len(b["soft_bits"]) == len(a.soft)
            d = np.abs(a.soft.astype(np.int32) - b["soft_bits"].astype(np.int32)).max() if len(a.soft) else 0
            assert d <= 1, (cases[s], d)
            exact = exact and d == 0
    assert n_soft_differs > 20, "soft and hard decoding never differed: the SNRs are too kind"
    print("soft values bit-identical:", exact)
    ctx.close()
    # default (hard) contexts report no soft values
    ctx = fx.RxContext(1, want_framesyms=True); g0 = ctx.process([xs[0]]); ctx.close()
    assert all(g["soft_bits"] is None for g in g0)
    # continuing blocks in flight + the equaliser at the same time
    h = np.zeros(6, np.complex64); h[0] = 1.0; h[2] = 0.3j
    y = np.convolve(xs[1], h)[:len(xs[1])].astype(np.complex64)
    of = oracle_frames(oracle, y, soft=True, equalizer=True)
    ctx = fx.RxContext(1, want_framesyms=True, soft_decision=True, equalizer=True); ctx.set_depth(2)
    got, parts = [], [np.ascontiguousarray(y[a:a + 50_000]) for a in range(0, len(y), 50_000)]
    for i, pc in enumerate(parts):
        if i >= 2: got += ctx.results(ctx.collect_raw())
        ctx.submit_raw([pc.ctypes.data], [len(pc)], False)
    got += ctx.results(ctx.collect_raw()); got += ctx.results(ctx.collect_raw())
    compare_frames(of, got)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dbg,blk", [(0, 0), (0, 128), (1, 192), (2, 128), (3, 256)])
def test_batch_viterbi_repairs_retraces_and_fallbacks(fx, oracle, monkeypatch, dbg, blk):
    """The batch Viterbi path (lane per trellis block) speculates twice -- a block's warm-up is taken to arrive at the true
    metric differences, a block's traceback is taken to start from the state a short look-ahead finds -- and verifies both.
    Low SNR makes the first check fail now and then (fx_vbfix_kernel runs the block again; what two passes do not settle is
    handed to the wave-per-frame decoder); FXRX_VB_DEBUG bit 0 makes nearly every end-state guess wrong (fx_vbfinish_kernel
    traces those blocks again), bit 1 leaves every other failed hand-over unrepaired (fallback path).  Every variant,
    several block lengths and all puncturing classes at once, against the oracle."""
    cases = [(2, 11, 3.0, 3_000_000, 1024), (27, 15, 9.5, 600_000, 700), (2, 20, 6.5, 600_000, 300), (28, 17, 12.0, 500_000, 2000), (3, 19, 7.0, 400_000, 50)]
    xs = [fx.synth_stream(n, stream_id=900 + i, mod=m, fec0=f0, payload_len=pl, snr_db=snr)[0] for i, (m, f0, snr, n, pl) in enumerate(cases)]
    monkeypatch.setenv("FXRX_VB_DEBUG", str(dbg))
    if blk: monkeypatch.setenv("FXRX_VB_BLK", str(blk))
    ctx = fx.RxContext(len(xs), want_framesyms=True)
    gf = ctx.process(xs)
    tm = ctx.timing()
    gf2 = ctx.process(xs)                                   # (second block: arenas and grids sized from the first one's traffic)
    tm2 = ctx.timing()
    ctx.close()
    n_bad_payloads = 0
    for s, x in enumerate(xs):
        of = oracle_frames(oracle, x)
        compare_frames(of, [g for g in gf if g["stream"] == s])
        n_bad_payloads += sum(1 for f in of if f.header_valid and not f.payload_valid)
    # the second pass continues the streams (their tails differ), so only count it
    assert tm["vb_blocks"] > 0 and tm2["vb_blocks"] > 0
    assert n_bad_payloads > 10, "every payload decoded: the SNRs are too kind"
    print("repairs", tm["vb_repairs"], "fallbacks", tm["vb_fallbacks"], "blocks", tm["vb_blocks"])
    if dbg & 2: assert tm["vb_fallbacks"] > 0, "no frame was handed back: the fallback path was not exercised"
    else: assert tm["vb_repairs"] + tm["vb_fallbacks"] > 0, "no hand-over check failed: the repair path was not exercised"
    assert len(gf2) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [1, 5, 64])
def test_plan_kernels_any_number_of_workgroups(fx, oracle, monkeypatch, grid):
    """fx_plan_kernel / fx_planlists_kernel lay the payload stage out with several workgroups over contiguous ranges of the
    chain's frames: arena offsets by a decoupled look-back across workgroups, list slots by wave-aggregated counters.  Many
    short frames of every modulation class and code class, several streams (some empty, some without a valid frame), any
    grid -- including more workgroups than chunks of frames -- against the oracle."""
    cases = [(2, 11, 40), (27, 15, 90), (1, 1, 16), (29, 20, 33), (3, 6, 64), (10, 17, 7), (18, 11, 0), (28, 27, 120), (9, 19, 55), (11, 11, 25)]
    xs = [fx.synth_stream(180_000 + 7_000 * i, stream_id=950 + i, mod=m, fec0=f0, payload_len=pl, snr_db=30.0, gap=260 + 11 * i)[0] for i, (m, f0, pl) in enumerate(cases)]
    xs.insert(3, np.zeros(50_000, np.complex64))                                  # a stream without any frame
    rng = np.random.default_rng(5)
    xs.append((0.3 * (rng.standard_normal(60_000) + 1j * rng.standard_normal(60_000))).astype(np.complex64))   # noise only
    monkeypatch.setenv("FXRX_PLAN_GRID", str(grid))
    ctx = fx.RxContext(len(xs), want_framesyms=True)
    gf = ctx.process(xs)
    n = 0
    for s, x in enumerate(xs):
        of = oracle_frames(oracle, x)
        compare_frames(of, [g for g in gf if g["stream"] == s])
        n += len(of)
    assert n > 1500, n
    # the same block again, pipelined with itself (hints, grids and arenas now come from the first pass)
    ctx.reset(); ctx.set_depth(2)
    ptrs = [x.ctypes.data for x in xs]; cnt = [len(x) for x in xs]
    ctx.submit_raw(ptrs, cnt, False); ctx.reset(); ctx.submit_raw(ptrs, cnt, False)
    a = ctx.results(ctx.collect_raw()); b = ctx.results(ctx.collect_raw())
    key = lambda g: (g["stream"], g["start"], g["payload"], g["payload_valid"], g["evm_sum"])
    assert [key(g) for g in a] == [key(g) for g in gf] == [key(g) for g in b]
    ctx.close()


@pytest.mark.gpu
def test_detector_mode_dense_detections_fit_the_tables(fx, oracle):
    """frame_detector_cc on dense traffic of tiny frames: the bare detector fires more often than once per 600 samples (the
    densest flex_rx can see) -- tables are sized for one detection per 256-sample hop in this mode.  (Found by the randomised
    test: results used to be cut off silently at samples / 600 + 8 detections per stream.)"""
    x = fx.synth_stream(126492, stream_id=313177, mod=28, fec0=18, fec1=7, payload_len=7, gap=300, snr_db=30.0)[0]
    thr = 0.35                                                    # (fires on the frames' tails as well: two detections per 1000-sample frame)
    want = [d["pos"] for d in oracle.Detector(thr).run(x) if d["pos"] + 512 <= len(x)]
    assert len(want) > len(x) // 600 + 8
    for seg in (0, 8192):
        ctx = fx.RxContext(1, mode=fx.MODE_DETECTOR, threshold=thr, segment_len=seg)
        mine = [g["start"] for g in ctx.process([x])]
        ctx.close()
        assert mine[:len(want)] == want
