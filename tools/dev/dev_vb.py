"""Batch Viterbi path on config 2: timing, block counts, repairs; and a low-SNR stream where hand-over checks fail."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
KEYS = ("paydec_ms", "paypll_ms", "total_ms", "vb_blocks", "vb_repairs", "vb_fallbacks", "late_decodes", "chain_ms", "walk_ms")
for snr in (20.0, 3.0):
    xb, fb = fx.synth_stream(20_000_000, stream_id=0, snr_db=snr)
    xd = torch.from_numpy(xb).cuda()
    ref = None
    for bv, dbg, blk in (("0", "0", "0"), ("1", "0", "0"), ("1", "1", "0"), ("1", "2", "0"), ("1", "3", "256"), ("1", "0", "4096")):
        os.environ["FXRX_BATCH_VITERBI"] = bv; os.environ["FXRX_VB_DEBUG"] = dbg; os.environ["FXRX_VB_BLK"] = blk
        ctx = fx.RxContext(1)
        for it in range(3):
            ctx.reset(); gf = ctx.process([xd])
        tm = ctx.timing()
        sig = [(g["start"], g["payload_valid"], bytes(g["payload"])) for g in gf]
        if ref is None: ref = sig
        print("snr", snr, "batch", bv, "dbg", dbg, "blk", blk, "frames", len(gf), "valid", sum(g["payload_valid"] for g in gf), "same_as_wave_per_frame", sig == ref,
              {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items() if k in KEYS}, flush=True)
        ctx.close()
