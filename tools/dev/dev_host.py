import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(xb).cuda()
ctx = fx.RxContext(1, segment_len=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for it in range(4):
    ctx.reset(); t = time.perf_counter(); n = ctx.process_raw([xd.data_ptr()], [xd.numel()], True); dt = time.perf_counter() - t
tm = ctx.timing()
print("wall %.3f ms" % (dt * 1e3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
import ctypes as C
L = fx.lib()
try:
    out = (C.c_uint64 * 4)()
    L.fxrx_debug_walk_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 4)]
    L.fxrx_debug_walk_stamps(ctx.h, C.byref(out))
    nj = tm["walk_jobs"]
    print("walker avg cycles per job: coarse %.0f seek %.0f align %.0f header %.0f" % tuple(v / nj for v in out))
except Exception as e:
    print("no walk stamps", e)
out8 = (C.c_uint64 * 8)()
L.fxrx_debug_walk_maxjob.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 8)]
L.fxrx_debug_walk_maxjob(ctx.h, C.byref(out8))
print("slowest job: coarse %d seek %d align %d header %d | hops %d cheap %d frames %d total %d" % tuple(out8))
