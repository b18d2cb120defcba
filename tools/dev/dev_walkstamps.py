"""Walker phase clocks (needs a -DFX_STAMPS build: make -C gr-liquiddsp_amd/csrc EXTRA=-DFX_STAMPS OUT=libfxrx_stamps.so,
then FXRX_LIB=gr-liquiddsp_amd/csrc/libfxrx_stamps.so python tools/dev/dev_walkstamps.py)."""
import importlib, sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(xb).cuda()
ctx = fx.RxContext(1)
for it in range(3):
    gf = ctx.process([xd]); ctx.reset()
L = fx.lib()
tm = ctx.timing()
tot = (C.c_uint64 * 4)(); mx = (C.c_uint64 * 8)()
L.fxrx_debug_walk_stamps(ctx.h, C.byref(tot)); L.fxrx_debug_walk_maxjob(ctx.h, C.byref(mx))
nj = tm["walk_jobs"]
names = ["coarse-scan", "seek(FFT sweep)", "align", "header"]
print("walk_ms %.3f jobs %d frames %d hops %d cheap %d" % (tm["walk_ms"], nj, len(gf), tm["hops"], tm["hops_cheap"]))
print("sum over jobs (Mcycles):", {n: round(tot[i] / 1e6, 2) for i, n in enumerate(names)}, " avg/job kcycles:", round(sum(tot) / nj / 1e3, 1))
print("slowest job (kcycles):", {n: round(mx[i] / 1e3, 1) for i, n in enumerate(names)}, "hops", mx[4], "cheap", mx[5], "frames", mx[6], "total", round(mx[7] / 1e3, 1))
