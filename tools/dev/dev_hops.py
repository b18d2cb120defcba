"""Exact-detector sweeps per frame of a locked walker: one segment (no speculation), skipping on / off."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
for gap in (256, 300, 384, 512):
    xb, fb = fx.synth_stream(4_000_000, stream_id=0, gap=gap)
    xd = torch.from_numpy(xb).cuda()
    for skip in ("1", "0"):
        os.environ["FXRX_SKIP_SEEK"] = skip
        ctx = fx.RxContext(1, segment_len=4_000_000)
        gf = ctx.process([xd]); tm = ctx.timing()
        print("gap", gap, "skip", skip, "frames", len(gf), "exact hops", tm["hops"], "cheap", tm["hops_cheap"], "verify hops", tm["verify_hops"], "per frame %.2f" % (tm["hops"] / max(1, len(gf))), "walk_ms %.2f" % tm["walk_ms"], flush=True)
        ctx.close()
