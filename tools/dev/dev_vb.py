"""Batch Viterbi path on config 2: timing, block counts, repairs; and a low-SNR stream where hand-over checks fail."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
for snr in (20.0, 3.0):
    xb, fb = fx.synth_stream(20_000_000, stream_id=0, snr_db=snr)
    xd = torch.from_numpy(xb).cuda()
    for bv in ("1", "0"):
        os.environ["FXRX_BATCH_VITERBI"] = bv
        ctx = fx.RxContext(1)
        for it in range(3):
            ctx.reset(); gf = ctx.process([xd])
        tm = ctx.timing()
        print("snr", snr, "batch", bv, "frames", len(gf), "valid", sum(g["payload_valid"] for g in gf),
              {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items() if k in ("paydec_ms", "paypll_ms", "total_ms", "vb_blocks", "vb_repairs", "late_decodes", "chain_ms", "walk_ms")})
        ctx.close()
