"""Low-SNR behaviour of the speculative walk: counters and times over an SNR sweep of the headline stream."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
KEYS = ("walk_ms", "seekverify_ms", "chain_ms", "total_ms", "repairs", "replays", "verify_failures", "verify_hops", "hops", "hops_cheap", "walk_jobs", "frames")
for snr in (20.0, 10.0, 6.0, 4.0, 3.0):
    xb, fb = fx.synth_stream(20_000_000, stream_id=0, snr_db=snr)
    xd = torch.from_numpy(xb).cuda()
    for skip in ("1", "0"):
        os.environ["FXRX_SKIP_SEEK"] = skip
        ctx = fx.RxContext(1)
        for it in range(3):
            ctx.reset(); gf = ctx.process([xd])
        tm = ctx.timing()
        print("snr", snr, "skip", skip, "valid", sum(g["payload_valid"] for g in gf), "hdr_valid", sum(g["header_valid"] for g in gf),
              {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items() if k in KEYS}, flush=True)
        ctx.close()
