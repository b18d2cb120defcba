"""Soak: many blocks through the submit/collect pipeline, EVERY block's results compared with a reference pass
(independent captures, then one continuing stream, then alternating resets).  python tools/dev/dev_soak.py [blocks] [depth]"""
import importlib, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 600
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x, inj = fx.synth_stream(6_000_000, stream_id=3, snr_db=float(os.environ.get("SOAK_SNR", "20")))
xd = torch.from_numpy(x).cuda()
ptrs, counts = [xd.data_ptr()], [xd.numel()]
key = lambda g: (g["start"], g["payload_valid"], hashlib.md5(g["payload"]).hexdigest(), g["evm_sum"], g["rxy"])
ref_ctx = fx.RxContext(1)                      # reference: the same calls, one block at a time (serial host path)
ctx = fx.RxContext(1); ctx.set_depth(depth)
for mode in ("independent", "continuous", "reset-every-3"):
    nref = min(nblk, 60)                       # the blocking reference is slow: check the first 60 blocks exactly, the rest for
    ref_ctx.reset(); refs = []                 # periodicity (the stream repeats, so block b and block b-3 agree up to the offset)
    for b in range(nref):
        if mode == "independent" or (mode == "reset-every-3" and b % 3 == 0): ref_ctx.reset()
        refs.append([key(g) for g in ref_ctx.results(ref_ctx.process_raw(ptrs, counts, True))])
    ctx.reset()
    bad, infl, done, got_all = 0, 0, 0, []
    def collect():
        global bad, done
        got = [key(g) for g in ctx.results(ctx.collect_raw())]
        if done < nref: want = refs[done]
        else:
            per = 3 if mode == "reset-every-3" else 1
            shift = per * len(x) if mode != "independent" and not (mode == "reset-every-3") else 0
            prev = got_all[done - per]
            want = [(k[0] + (len(x) * per if mode == "continuous" else 0),) + k[1:] for k in prev]
        if got != want:
            bad += 1
            if bad <= 3: print(mode, "block", done, "differs: frames", len(got), "vs", len(want), [i for i, (a, b) in enumerate(zip(got, want)) if a != b][:5])
        got_all.append(got); done += 1
    for b in range(nblk):
        if infl == depth: collect(); infl -= 1
        if mode == "independent" or (mode == "reset-every-3" and b % 3 == 0): ctx.reset()
        ctx.submit_raw(ptrs, counts, True); infl += 1
    while infl: collect(); infl -= 1
    print(mode, "blocks", done, "mismatching", bad, flush=True)
