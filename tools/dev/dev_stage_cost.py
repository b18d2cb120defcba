"""What each stage of a block costs the CHIP with blocks in flight: the bench workload (20-Msample capture, 12 in flight) with the
kernel chain cut after the plan stage / the matched filter / the PLL / not at all (FXRX_DEBUG_STOP_AFTER); differences of the
steady-state ms per block are the stages' shares.  One context per setting: run each as its own process.
python tools/dev/dev_stage_cost.py [stop_after]"""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
cont = "--continuous" in sys.argv                       # the passes as consecutive blocks of one stream (no reset between them)
if cont: sys.argv.remove("--continuous")
if len(sys.argv) < 2:
    for stop in ("4", "5", "6", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), stop] + (["--continuous"] if cont else []), env=dict(os.environ, FXRX_DEBUG_STOP_AFTER=stop))
    sys.exit(0)
import torch
fx = importlib.import_module("gr-liquiddsp_amd")
x, inj = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(x).cuda()
ctx = fx.RxContext(1)
ptrs, counts = [xd.data_ptr()], [xd.numel()]
os.environ["FXRX_DEBUG_STOP_AFTER"] = "0"
depth = 12
ctx.reset(); ctx.process_raw(ptrs, counts, True); ctx.reset(); ctx.process_raw(ptrs, counts, True)
ctx.set_depth(depth)
acc = [0.0, 0.0, 0]
def run(k):
    infl = 0
    for _ in range(k):
        if infl == depth:
            t = time.perf_counter(); ctx.collect_raw(); acc[0] += time.perf_counter() - t; infl -= 1
        if not cont: ctx.reset()
        t = time.perf_counter(); ctx.submit_raw(ptrs, counts, True); acc[1] += time.perf_counter() - t; acc[2] += 1; infl += 1
    while infl: ctx.collect_raw(); infl -= 1
run(100); torch.cuda.synchronize(); acc[:] = [0.0, 0.0, 0]; t0 = time.perf_counter(); run(1500); torch.cuda.synchronize()
print(("continuing stream, " if cont else "") + "stop after %s: %.4f ms per block (host: %.4f ms in collect, %.4f ms in submit per block)" % (sys.argv[1], (time.perf_counter() - t0) / 1500 * 1e3, acc[0] / acc[2] * 1e3, acc[1] / acc[2] * 1e3), flush=True)
