import importlib, sys, os
sys.path.insert(0, "/root/repo")
os.environ["FXRX_DEBUG_ROUNDS"] = "1"
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
xb, fb = fx.synth_stream(20_000_000, stream_id=0, snr_db=float(os.environ.get("SNR","4")))
xd = torch.from_numpy(xb).cuda()
ctx = fx.RxContext(1)
gf = ctx.process([xd])
print(ctx.timing()["chain_ms"], ctx.timing()["repairs"])
