import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
x, inj = fx.synth_stream(1_900_000, stream_id=77, payload_len=300, gap=200)
one = fx.RxContext(1); ref = one.process([x]); one.close()
cuts = [0, 300_000, 650_123, 950_000, 1_300_777, 1_600_000, len(x)]
parts = [torch.from_numpy(np.ascontiguousarray(x[a:b])).cuda() for a, b in zip(cuts[:-1], cuts[1:])]
for rep in range(12):
    ctx = fx.RxContext(1); ctx.set_depth(3)
    got, infl, tms = [], 0, []
    for p in parts:
        if infl == 3:
            got += ctx.results(ctx.collect_raw()); tms.append(ctx.timing()); infl -= 1
        ctx.submit_raw([p.data_ptr()], [p.numel()], True); infl += 1
    while infl:
        got += ctx.results(ctx.collect_raw()); tms.append(ctx.timing()); infl -= 1
    for i, (a, b) in enumerate(zip(ref, got)):
        d = {k: (a[k], b[k]) for k in ("start", "rxy", "tau", "gamma", "dphi", "phi", "header_valid", "payload_valid", "evm_sum", "cfo_bin") if a[k] != b[k]}
        if d or a["payload"] != b["payload"]:
            print("first diff at", i, d, "payload equal", a["payload"] == b["payload"]); break
    print("rep", rep, "frames", len(got), "repairs", [t["repairs"] for t in tms])
    ctx.close()
