"""The drop-in boundary under load and under failure (`-m gpu`): the asynchronous flexframesync_execute path fed in 256-sample
calls like /root/reference/lib/flex_rx_impl.cc:212-215, two handles on two threads (GNU Radio runs one thread per block
instance, :49), and what a failing fxrx_submit / fxrx_collect leaves behind (include/fxrx.h, "Failure semantics")."""
import ctypes as C
import numpy as np
import pytest
from parity_util import oracle_frames, compare_frames

pytestmark = pytest.mark.gpu


def _key(g):
    return (g["start"], g["payload"], g["payload_valid"], g["header_valid"], g["evm_sum"], g["rxy"])


def _okey(f):
    return (f.info["start"], f.payload, f.payload_valid, f.header_valid, f.info["evm_sum"], f.info["rxy"])


def test_failing_submit_leaves_the_context_as_it_was(fx, oracle):
    """A submit that fails -- a batch too large for the 32-bit arena offsets (caller-recoverable, nothing launched), or an
    injected failure after all of the call's bookkeeping -- must not move the stream: the next valid submit continues where
    the last successful one stopped, positions included.  Also as the very first call on a context (fresh state)."""
    import torch
    L = fx.lib()
    x, _ = fx.synth_stream(400_000, stream_id=311, payload_len=300)
    of = [_okey(f) for f in oracle_frames(oracle, x)]
    cut = 180_000
    a, b = torch.from_numpy(x[:cut].copy()).cuda(), torch.from_numpy(x[cut:].copy()).cuda()
    torch.cuda.synchronize()
    ctx = fx.RxContext(1)
    ctx.set_depth(2)

    def too_large():
        with pytest.raises(fx.rx.RxError, match="too large"):
            ctx.submit_raw([a.data_ptr()], [1 << 34], True)       # (the pointer is never read: the call fails while sizing arenas)

    def injected():
        assert L.fxrx_debug_fail(ctx.h, 1, 0) == 0
        with pytest.raises(fx.rx.RxError, match="injected"):
            ctx.submit_raw([a.data_ptr()], [a.numel()], True)
    too_large(); injected()                                        # first calls after create
    assert L.fxrx_inflight(ctx.h) == 0
    ctx.submit_raw([a.data_ptr()], [a.numel()], True)
    too_large(); injected()                                        # with a block in flight
    ctx.submit_raw([b.data_ptr()], [b.numel()], True)
    got = [_key(g) for g in ctx.results(ctx.collect_raw())] + [_key(g) for g in ctx.results(ctx.collect_raw())]
    assert got == of
    ctx.close()


def test_failing_collect_drops_what_is_in_flight_and_the_context_recovers(fx, oracle):
    """fxrx_collect reporting an error (injected) drops every block in flight; the context must not be wedged: the next collect
    says "nothing in flight", and the next block -- without a reset -- is searched from a freshly reset synchroniser, at
    positions that keep counting."""
    import torch
    L = fx.lib()
    x, _ = fx.synth_stream(600_000, stream_id=312, payload_len=200)
    parts = [x[:200_000], x[200_000:400_000], x[400_000:]]
    dev = [torch.from_numpy(p.copy()).cuda() for p in parts]
    torch.cuda.synchronize()
    ctx = fx.RxContext(1)
    ctx.set_depth(3)
    ctx.submit_raw([dev[0].data_ptr()], [dev[0].numel()], True)
    ctx.submit_raw([dev[1].data_ptr()], [dev[1].numel()], True)
    assert L.fxrx_debug_fail(ctx.h, 0, 1) == 0
    with pytest.raises(fx.rx.RxError, match="injected"):
        ctx.collect_raw()
    assert L.fxrx_inflight(ctx.h) == 0
    with pytest.raises(fx.rx.RxError, match="nothing in flight"):
        ctx.collect_raw()
    # the third block: as if it were the start of a capture, at absolute positions 400 000 + ...
    got = ctx.results(ctx.process_raw([dev[2].data_ptr()], [dev[2].numel()], True))
    alone = oracle_frames(oracle, parts[2])
    assert [(g["start"] - 400_000, g["payload"], g["payload_valid"]) for g in got] == [(f.info["start"], f.payload, f.payload_valid) for f in alone]
    assert len(got) > 5
    # and after a reset the context is as good as new
    ctx.reset()
    got = ctx.results(ctx.process_raw([dev[0].data_ptr()], [dev[0].numel()], True))
    assert [_key(g) for g in got] == [_okey(f) for f in oracle_frames(oracle, parts[0])]
    ctx.close()


def _feed_256(L, q, x):
    for i in range(0, len(x) - len(x) % 256, 256):                 # lib/flex_rx_impl.cc:212-215
        blk = x[i:i + 256]
        L.flexframesync_execute(q, blk.ctypes.data, 256)


def _drain(L, q):
    L.fxrx_sync_flush(q)
    while L.fxrx_sync_pending(q):
        L.flexframesync_execute(q, None, 0)


def test_dropin_never_feeds_samples_twice_after_a_failed_block(fx, oracle):
    """flexframesync_execute returns void: a GPU-side failure is counted and reported, the samples of the blocks concerned are
    dropped, and the synchroniser restarts behind the gap -- every frame the callback sees is a distinct injected frame, in
    order, and the frames behind the gap do arrive."""
    L = fx.lib()
    x, inj = fx.synth_stream(500_000, stream_id=313, payload_len=150)
    want = [pl for _, pl in inj]
    got = []
    cbf = fx._ffi.FRAMESYNC_CALLBACK(lambda hd, hv, pl, n, pv, st, ud: got.append((hv, pv, C.string_at(pl, n) if n else b"")) or 0)
    q = L.flexframesync_create(cbf, None)
    assert q
    L.fxrx_sync_set_block(q, 1 << 16)
    half = 250_112
    _feed_256(L, q, x[:half])
    assert L.fxrx_debug_fail(L.fxrx_sync_context(q), 0, 1) == 0    # the next collect fails: the blocks in flight are dropped
    _feed_256(L, q, x[half:])
    _drain(L, q)
    assert L.fxrx_sync_errors(q) == 1
    L.flexframesync_destroy(q)
    payloads = [p for hv, pv, p in got if hv and pv]
    idx = [want.index(p) for p in payloads]                        # every delivered payload is an injected one ...
    assert idx == sorted(set(idx))                                 # ... at most once, in order
    assert len(idx) < len(want)                                    # something was dropped
    assert idx[-1] == len(want) - 1 and idx[0] == 0                # frames before and behind the gap arrive
    period = inj[1][0] - inj[0][0]
    assert len(want) - len(idx) <= 4 * 65536 // period + 4        # at most the blocks in flight (three) and the one being collected went missing


@pytest.mark.parametrize("block", [1 << 16, 0])
def test_dropin_block_shell_in_256_sample_calls_matches_the_oracle(fx, oracle, block, monkeypatch):
    """The C++ flex_rx shell (csrc/blocks/fx_blocks.hpp) driven like GNU Radio drives the reference block -- work() calls of
    8192 items from pageable memory, 256-sample flexframesync_execute calls inside -- publishes exactly the oracle's frames:
    payload bytes (hashed in order), header / payload verdicts, one constellation per frame.  block = 0: the default block
    length (2^20: the whole stream is one partial block, pushed through by flush())."""
    F = fx._ffi.feed_lib()
    if block: monkeypatch.setenv("FXRX_SYNC_BLOCK", str(block))
    x, inj = fx.synth_stream(700_000, stream_id=314)
    of = oracle_frames(oracle, x)
    st = fx._ffi.DropinStats()
    assert F.dropin_feed(x.ctypes.data, len(x), 8192, 1, C.byref(st)) == 0
    assert st.errors == 0
    assert st.frames == len(of) and st.header_valid == sum(f.header_valid for f in of) and st.payload_valid == sum(f.payload_valid for f in of)
    assert st.constellation_syms == sum(len(f.framesyms) for f in of)
    assert st.payload_hash == fx._ffi.fnv1a([f.payload for f in of if f.header_valid])
    assert st.frames == len(inj)


def test_two_flexframesync_handles_on_two_threads(fx, oracle):
    """SURVEY 8(b) "Threading": one flexframesync per block instance (lib/flex_rx_impl.cc:49), GNU Radio runs one thread per
    block.  Two handles, two threads, two different streams, both fed in 256-sample calls at the same time; each must publish
    exactly its own stream's oracle frames."""
    F = fx._ffi.feed_lib()
    xs = [fx.synth_stream(900_000, stream_id=315)[0], fx.synth_stream(700_000, stream_id=316, mod=27, fec0=15, payload_len=600, snr_db=25.0)[0]]
    ofs = [oracle_frames(oracle, x) for x in xs]
    ptrs = (C.c_void_p * 2)(*[x.ctypes.data for x in xs])
    ns = (C.c_ulonglong * 2)(*[len(x) for x in xs])
    st = (fx._ffi.DropinStats * 2)()
    assert F.dropin_feed_threads(ptrs, ns, 2, 4096, st) == 0
    for t in range(2):
        of = ofs[t]
        assert st[t].errors == 0 and st[t].frames == len(of) and len(of) > 20
        assert st[t].payload_valid == sum(f.payload_valid for f in of)
        assert st[t].payload_hash == fx._ffi.fnv1a([f.payload for f in of if f.header_valid])
        assert st[t].constellation_syms == sum(len(f.framesyms) for f in of)
    # and two batched contexts driven from two Python threads (ctypes releases the GIL inside the calls)
    import threading
    res = [None, None]

    def run(t):
        ctx = fx.RxContext(1, want_framesyms=True)
        ctx.set_depth(2)
        cut = len(xs[t]) // 2
        a, b = np.ascontiguousarray(xs[t][:cut]), np.ascontiguousarray(xs[t][cut:])
        ctx.submit_raw([a.ctypes.data], [len(a)], False); ctx.submit_raw([b.ctypes.data], [len(b)], False)
        res[t] = ctx.results(ctx.collect_raw()) + ctx.results(ctx.collect_raw())
        ctx.close()
    th = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for t in range(2):
        compare_frames(ofs[t], res[t])


def test_pinned_host_blocks_take_the_upload_kernel_and_give_the_same_frames(fx, oracle):
    """Host blocks in page-locked memory are fetched by fx_upload_kernel (fx_host.cpp: pinned_device_ptr), pageable ones by the
    runtime's copy: both must give what the same samples give when they are resident on the device -- odd lengths, a block that
    starts at an odd sample of the pinned allocation, consecutive blocks of one stream."""
    import torch
    x, _ = fx.synth_stream(333_333, stream_id=317, payload_len=120)
    of = [_okey(f) for f in oracle_frames(oracle, x)]
    xp = torch.from_numpy(np.concatenate([np.zeros(1, np.complex64), x])).pin_memory()      # the stream starts 8 bytes into the allocation
    base = xp.data_ptr() + 8
    cuts = [0, 100_001, 250_000, len(x)]
    for on_pinned in (True, False):
        ctx = fx.RxContext(1)
        ctx.set_depth(2)
        got = []
        for a, b in zip(cuts, cuts[1:]):
            if on_pinned: ctx.submit_raw([base + 8 * a], [b - a], False)
            else:
                blk = np.ascontiguousarray(x[a:b]); ctx.submit_raw([blk.ctypes.data], [b - a], False)
            got += [_key(g) for g in ctx.results(ctx.collect_raw())]
        assert got == of and len(of) > 10
        ctx.close()
