"""Bring-up of the device-side chain: a few parity checks against the oracle with timing / counters printed."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi as oracle
from parity_util import oracle_frames, compare_frames
fx = importlib.import_module("gr-liquiddsp_amd")

x, inj = fx.synth_stream(400_000, stream_id=11)
of = oracle_frames(oracle, x)
for slow in ("0", "1"):
    os.environ["FXRX_CHAIN_SLOW"] = slow
    for seg in (0, 8192, 50000):
        ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
        t = time.time(); gf = ctx.process([x]); dt = time.time() - t
        print("slow", slow, "seg", seg, "frames", len(gf), "of", len(of), "t %.3f" % dt, {k: v for k, v in ctx.timing().items() if v}, flush=True)
        dev = compare_frames(of, gf)
        print("   dev", dev, flush=True)
        ctx.close()
os.environ["FXRX_CHAIN_SLOW"] = "0"
# chunked feeding (continuing blocks, carried tails)
ctx = fx.RxContext(1, want_framesyms=True, segment_len=16384)
got, p = [], 0
rng = np.random.default_rng(5)
while p < len(x):
    n = int(rng.choice([1, 255, 256, 1000, 4096, 30000, 70001]))
    got += ctx.process([x[p:p + n]]); p += n
print("chunked", len(got), compare_frames(of, got), flush=True)
# multi-stream, all mods
xs = []
for i, mod in enumerate(fx.MOD_BY_INDEX):
    xx, _ = fx.synth_stream(60_000 + 1000 * i, stream_id=100 + i, mod=mod, fec0=fx.INNER_BY_INDEX[i % 7], payload_len=300 + 17 * i, snr_db=32.0, gap=256 + 50 * i, lead=10 * i)
    xs.append(xx)
ctx = fx.RxContext(len(xs), want_framesyms=True)
gf = ctx.process(xs)
for s, xx in enumerate(xs):
    compare_frames(oracle_frames(oracle, xx), [g for g in gf if g["stream"] == s])
print("all mods ok", ctx.timing()["frames"], flush=True)
