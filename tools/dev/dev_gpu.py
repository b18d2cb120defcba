"""Ad-hoc GPU bring-up script (not a test): python tests/dev_gpu.py"""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle_ffi as o
from parity_util import oracle_frames, compare_frames
fx = importlib.import_module("gr-liquiddsp_amd")

n = 200000
x, frames = fx.synth_stream(n, stream_id=1)
print("frames injected", len(frames))
t = time.time(); of = oracle_frames(o, x, chunk=256); print("oracle: %d frames in %.2fs" % (len(of), time.time() - t))
for seg in (0, 8192, 1 << 20):
    ctx = fx.RxContext(1, want_framesyms=True, segment_len=seg)
    t = time.time(); gf = ctx.process([x]); dt = time.time() - t
    print("seg", seg, "gpu frames", len(gf), "time %.3f" % dt, ctx.timing())
    for a, b in zip(of[:3], gf[:3]):
        print(" oracle", a.info["start"], a.info["offset"], a.info["tau"], a.info["dphi"], a.header_valid, a.payload_valid)
        print(" gpu   ", b["start"], b["cfo_bin"], b["tau"], b["dphi"], b["header_valid"], b["payload_valid"])
    try:
        print(" compare:", compare_frames(of, gf))
    except AssertionError as e:
        print(" COMPARE FAILED:", repr(e)[:400])
    ctx.close()
# detector mode
d = o.Detector(0.45); od = d.run(x)
ctx = fx.RxContext(1, mode=fx.MODE_DETECTOR, segment_len=16384)
gd = ctx.process([x])
print("detector: oracle", len(od), "gpu", len(gd), ctx.timing())
for a, b in list(zip(od, gd))[:4]:
    print(" ", a["pos"], a["offset"], a["tau"], a["dphi"], a["phi"], "|", b["start"], b["cfo_bin"], b["tau"], b["dphi"], b["phi"])
print(" same positions:", [a["pos"] for a in od] == [b["start"] for b in gd])

# ---- full-size timing (config 2): 20 Msamples single stream, device-resident ----
import torch
t = time.time(); xb, fb = fx.synth_stream(20_000_000, stream_id=0); print("synth 20M: %.1fs, %d frames" % (time.time() - t, len(fb)))
xd = torch.from_numpy(xb).cuda()
for seg in (0, 16384, 65536):
    ctx = fx.RxContext(1, segment_len=seg)
    for it in range(3):
        t = time.time(); gf = ctx_res = ctx.process([xd]); dt = time.time() - t
        ctx.reset()
    tm = ctx.timing()
    ok = sum(1 for g, (p, pl) in zip(gf, fb) if g["payload_valid"] and g["payload"] == pl)
    print("seg", seg, "frames", len(gf), "ok", ok, "wall %.2f ms" % (dt * 1e3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
    ctx.close()
