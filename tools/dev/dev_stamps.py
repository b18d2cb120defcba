import importlib, sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(xb).cuda()
ctx = fx.RxContext(1)
for it in range(2):
    gf = ctx.process([xd]); ctx.reset()
L = fx.lib()
acc = np.zeros(8)
n = 0
for i in range(0, len(gf), 7):
    out = (C.c_uint32 * 8)()
    if L.fxrx_debug_stamps(ctx.h, i, C.byref(out)) == 0:
        acc += np.array(list(out), dtype=np.float64); n += 1
print("avg cycles: pack %.0f perm1 %.0f fec1 %.0f perm0 %.0f fec0(viterbi) %.0f tail %.0f | viterbi fwd %.0f" % tuple((acc / n)[:7]), ctx.timing()["paydec_ms"])
