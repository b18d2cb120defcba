"""Per block, on a common clock: when the host submitted it, when its first stage event fired on the GPU, when its last one did,
when the host had it back -- the bench workload with 12 blocks in flight.  python tools/dev/dev_block_times.py"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
fx = importlib.import_module("gr-liquiddsp_amd")
x, inj = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(x).cuda()
ctx = fx.RxContext(1); ctx.set_timing(2)
ptrs, counts = [xd.data_ptr()], [xd.numel()]
ctx.reset(); ctx.process_raw(ptrs, counts, True); ctx.reset(); ctx.process_raw(ptrs, counts, True)
depth = 12; ctx.set_depth(depth)
rows = []
def collect():
    ctx.collect_raw(); o = (C.c_double * 4)(); ctx.L.fxrx_debug_block_times(ctx.h, C.byref(o)); rows.append(tuple(o))
infl = 0
for _ in range(400):
    if infl == depth: collect(); infl -= 1
    ctx.reset(); ctx.submit_raw(ptrs, counts, True); infl += 1
while infl: collect(); infl -= 1
r = rows[200:260]
print("block: submit(host) -> gpu first event -> gpu last event -> collected(host)   [ms, relative to the first of these blocks]")
t0 = r[0][0]
for a, b, c_, d in r[:26]: print("  %8.3f  %8.3f  %8.3f  %8.3f   queue wait %.3f  gpu span %.3f  done->collected %.3f" % (a - t0, b - t0, c_ - t0, d - t0, b - a, c_ - b, d - c_))
n = len(r)
print("means over %d blocks: submit->first event %.3f ms, gpu span %.3f ms, last event->collected %.3f ms, cadence %.3f ms" % (n, sum(b - a for a, b, _, _ in r) / n, sum(c_ - b for _, b, c_, _ in r) / n, sum(d - c_ for _, _, c_, d in r) / n, (r[-1][0] - r[0][0]) / (n - 1)))
