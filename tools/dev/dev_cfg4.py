import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle_ffi as o
from parity_util import oracle_frames, compare_frames
fx = importlib.import_module("gr-liquiddsp_amd")
xs, injs = zip(*[fx.synth_stream(1 << 21, stream_id=4000 + i, mod=27, fec0=15, snr_db=25.0) for i in range(16)])
ctx = fx.RxContext(16, want_framesyms=False)
gf = ctx.process(list(xs))
for s in range(16):
    mine = [g for g in gf if g["stream"] == s]
    by = {g["start"]: g for g in mine}
    miss = [p for p, pl in injs[s] if not any(q in by for q in (p - 1, p, p + 1))]
    bad = [p for p, pl in injs[s] if any(q in by for q in (p - 1, p, p + 1)) and not any((q in by and by[q]["payload"] == pl and by[q]["payload_valid"]) for q in (p - 1, p, p + 1))]
    if miss or bad or len(mine) != len(injs[s]):
        print("stream", s, "gpu", len(mine), "inj", len(injs[s]), "missing", miss, "bad", bad)
        of = oracle_frames(o, xs[s], chunk=1 << 16)
        print("   oracle frames", len(of), "oracle starts near:", [(f.info["start"], f.header_valid, f.payload_valid) for f in of if miss and abs(f.info["start"] - miss[0]) < 20000])
        print("   gpu near:", [(g["start"], g["header_valid"], g["payload_valid"]) for g in mine if miss and abs(g["start"] - miss[0]) < 20000])
        try:
            print("   compare:", compare_frames(of, mine, check_syms=False))
        except AssertionError as e:
            print("   COMPARE FAILED", str(e)[:300])
print("done")
