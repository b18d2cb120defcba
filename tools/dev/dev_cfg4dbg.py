import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_ffi as oracle
from parity_util import oracle_frames, compare_frames
fx = importlib.import_module("gr-liquiddsp_amd")
n = 1 << 21
xs, inj, dev = [], [], []
for i in range(16):
    x, f = fx.synth_stream(n, stream_id=4000 + i, mod=27, fec0=15, snr_db=25.0)
    xs.append(x); inj.append(f); dev.append(torch.from_numpy(x).cuda())
for NS in (16, 128):
    ctx = fx.RxContext(NS)
    nres = ctx.process_raw([dev[s % 16].data_ptr() for s in range(NS)], [n] * NS, True)
    res = ctx.results(nres)
    print("NS", NS, "frames", len(res), {k: v for k, v in ctx.timing().items() if k in ("walk_jobs", "repairs", "verify_failures", "replays", "hops")})
    for s in range(NS):
        mine = [g for g in res if g["stream"] == s]
        st = [g["start"] for g in mine]
        want = [p for p, _ in inj[s % 16]]
        miss = [p for p in want if not any(abs(p - q) <= 1 for q in st)]
        extra = [q for q in st if not any(abs(p - q) <= 1 for p in want)]
        if miss or extra or len(st) != len(want):
            print(" stream", s, "frames", len(st), "want", len(want), "missing", miss[:5], "extra", extra[:5])
            i0 = want.index(miss[0]) if miss else 0
            print("   around:", st[max(0, i0 - 2):i0 + 3], want[max(0, i0 - 2):i0 + 3])
            of = oracle_frames(oracle, xs[s % 16], chunk=1 << 16)
            print("   oracle:", [f.info["start"] for f in of][max(0, i0 - 2):i0 + 3], len(of))
            break
    ctx.close()
