import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(xb).cuda()
for seg in (0, 65536, 131072):
    ctx = fx.RxContext(1, segment_len=seg)
    for it in range(3):
        gf = ctx.process([xd]); ctx.reset()
    tm = ctx.timing()
    print("seg", seg, "frames", len(gf), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
    ctx.close()
