"""Is one context's pipeline limited by the chip or by something serial inside a context (host thread, ordering of the chain
kernels)?  N contexts on N host threads, each with its own pipelined loop over the bench capture: if the summed rate goes up
with N, it is not the chip.  python tools/dev/dev_two_ctx_rate.py"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
fx = importlib.import_module("gr-liquiddsp_amd")
x, inj = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(x).cuda()
ptrs, counts = [xd.data_ptr()], [xd.numel()]
def loop(ctx, depth, k):
    infl = 0
    for _ in range(k):
        if infl == depth: ctx.collect_raw(); infl -= 1
        ctx.reset(); ctx.submit_raw(ptrs, counts, True); infl += 1
    while infl: ctx.collect_raw(); infl -= 1
for nctx, depth in ((1, 12), (2, 6), (2, 10), (3, 6), (4, 5)):
    cs = [fx.RxContext(1) for _ in range(nctx)]
    for c in cs:
        c.reset(); c.process_raw(ptrs, counts, True); c.reset(); c.process_raw(ptrs, counts, True); c.set_depth(depth)
    for c in cs: loop(c, depth, 40)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(c, depth, 800)) for c in cs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d context(s) x depth %2d: %.4f ms per block overall, %.1f Gsamples/s" % (nctx, depth, dt / (800 * nctx) * 1e3, 800 * nctx * 20e6 / dt / 1e9), flush=True)
    for c in cs: c.close()
