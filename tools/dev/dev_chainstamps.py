"""Chain-kernel phase clocks on config 2 (one 20-Msample stream): python tools/dev/dev_chainstamps.py"""
import importlib, sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(xb).cuda()
ctx = fx.RxContext(1)
for it in range(3):
    gf = ctx.process([xd]); ctx.reset()
out = (C.c_uint32 * 8)()
fx.lib().fxrx_debug_chain_stamps(ctx.h, C.byref(out))
tm = ctx.timing()
print("chain+plan %.3f ms; chainfast cycles: look-ups %d, + list ranking %d, + compaction = fast path %d, kernel total %d (jobs %d, frames %d)" % (tm["chain_ms"], out[4], out[5], out[0], out[3], tm["walk_jobs"], tm["frames"]))
print({k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")})
