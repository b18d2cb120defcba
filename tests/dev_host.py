import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(xb).cuda()
ctx = fx.RxContext(1, segment_len=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for it in range(4):
    ctx.reset(); t = time.perf_counter(); n = ctx.process_raw([xd.data_ptr()], [xd.numel()], True); dt = time.perf_counter() - t
tm = ctx.timing()
print("wall %.3f ms" % (dt * 1e3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
