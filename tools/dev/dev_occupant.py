"""Bench workload (12 blocks in flight) with an idle occupant holding part of every CU's registers / LDS: tools/dev/occupant.hip.
python tools/dev/dev_occupant.py"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
fx = importlib.import_module("gr-liquiddsp_amd")
O = C.CDLL(os.path.join(ROOT, "tools", "dev", "liboccupant.so"))
x, inj = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(x).cuda()
ctx = fx.RxContext(1)
ptrs, counts = [xd.data_ptr()], [xd.numel()]
ctx.reset(); ctx.process_raw(ptrs, counts, True); ctx.reset(); ctx.process_raw(ptrs, counts, True)
depth = 12; ctx.set_depth(depth)
def run(k):
    infl = 0
    for _ in range(k):
        if infl == depth: ctx.collect_raw(); infl -= 1
        ctx.reset(); ctx.submit_raw(ptrs, counts, True); infl += 1
    while infl: ctx.collect_raw(); infl -= 1
def measure(tag):
    # (no device-wide synchronise in here: it would wait for the occupant, which only leaves when told to)
    run(60); t0 = time.perf_counter(); run(800)
    print("%-64s %.4f ms per block" % (tag, (time.perf_counter() - t0) / 800 * 1e3), flush=True)
measure("no occupant")
for regs, lds, wgs, what in ((128, 0, 256, "one 4-wave workgroup per CU holding 104 VGPRs per lane: 20 % of the register file"),
                             (256, 0, 256, "one per CU holding ~224 VGPRs: 44 %"), (128, 0, 512, "two per CU x 104 VGPRs: 41 %"),
                             (64, 1, 256, "one per CU holding 32 KB of LDS and 44 VGPRs (9 %): 20 % of the LDS"), (64, 1, 512, "two per CU: 40 % of the LDS, 17 % of the registers")):
    O.occupant_start(regs, lds, wgs); time.sleep(0.05)
    try: measure(what)
    finally: O.occupant_stop()
measure("no occupant again")
