"""How much of the chip does one context's host-serial walk chain leave idle?  N threads, one RxContext each (own HIP
streams), all running the bench workload pipelined; prints the aggregate rate.  python tools/dev/dev_two_ctx.py [threads] [depth]"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("FXRX_PAYLOAD_STREAMS", "3")
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
nthr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = 150
x, inj = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(x).cuda()
ptrs, counts = [xd.data_ptr()], [xd.numel()]
ctxs = [fx.RxContext(1) for _ in range(nthr)]
for c in ctxs:
    c.process_raw(ptrs, counts, True); c.reset(); c.set_depth(depth)


def run(c, k):
    infl = 0
    for _ in range(k):
        if infl == depth:
            c.collect_raw(); infl -= 1
        c.reset(); c.submit_raw(ptrs, counts, True); infl += 1
    while infl:
        c.collect_raw(); infl -= 1


for c in ctxs: run(c, 2 * depth)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(c, steps)) for c in ctxs]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("threads %d depth %d: %.0f Msamples/s aggregate (%.3f ms per block)" % (nthr, depth, nthr * steps * 20.0 / dt, dt / (nthr * steps) * 1e3))
