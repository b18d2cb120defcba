import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle_ffi as o
from parity_util import oracle_frames, compare_frames
fx = importlib.import_module("gr-liquiddsp_amd")
x, inj = fx.synth_stream(1 << 21, stream_id=4000, mod=27, fec0=15, snr_db=25.0)
of = oracle_frames(o, x, chunk=1 << 16)
print("injected", len(inj), "oracle", len(of))
for seg in (1 << 20, 400000, 0):
    ctx = fx.RxContext(1, want_framesyms=False, segment_len=seg)
    gf = ctx.process([x])
    tm = ctx.timing()
    st_o = [f.info["start"] for f in of]; st_g = [g["start"] for g in gf]
    print("seg", seg, "gpu", len(gf), "jobs", tm["walk_jobs"], "repairs", tm["repairs"], "missing", sorted(set(st_o) - set(st_g))[:5], "extra", sorted(set(st_g) - set(st_o))[:5], "dups", len(st_g) - len(set(st_g)))
    bad = [(a.info["start"], a.payload_valid, g["payload_valid"]) for a, g in zip(of, gf) if a.info["start"] == g["start"] and (a.payload != g["payload"] or a.payload_valid != g["payload_valid"])]
    print("   payload mismatches", len(bad), bad[:3])
