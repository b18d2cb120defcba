"""Every rank of bench.py --gpus N receives stream_id = rank: check that streams 0..7 decode completely (bench refuses to
report a rate otherwise)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
ctx = fx.RxContext(1)
for sid in range(8):
    x, inj = fx.synth_stream(20_000_000, stream_id=sid)
    xd = torch.from_numpy(x).cuda()
    ctx.reset()
    res = ctx.results(ctx.process_raw([xd.data_ptr()], [xd.numel()], True))
    ok = sum(1 for g, (_, pl) in zip(res, inj) if g["payload_valid"] and g["payload"] == pl)
    print("stream", sid, "injected", len(inj), "found", len(res), "ok", ok, flush=True)
