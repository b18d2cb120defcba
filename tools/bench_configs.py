#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one GPU (not the headline bench line; see bench.py for that).

  config 3: frame_detector_cc only, 256 streams x 2^20 samples (detector mode)
  config 4: flex_rx batched, 128 streams x 2^21 samples, QAM16 r=2/3 (V27P23)
  config 5 (one GPU's share): 128 streams x 2^20 samples, modulation in {PSK4,QAM16,QAM32,QAM64} x inner 0..6 cycling

Streams are synthesised on the host (16 distinct ones per config, tiled to the stream count), uploaded once, then
every pass = reset + process of all streams, first one pass at a time, then with four passes in flight.  Every injected frame is checked before a rate is printed."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def run(fx, torch, name, n_streams, n_samples, gen, mode, passes, distinct=16, pipeline=True):
    xs, inj = [], []
    for i in range(distinct):
        x, f = gen(i, n_samples)
        xs.append(torch.from_numpy(x).cuda()); inj.append(f)
    ptrs = [xs[s % distinct].data_ptr() for s in range(n_streams)]
    counts = [n_samples] * n_streams
    ctx = fx.RxContext(n_streams, mode=mode, threshold=0.45 if mode == fx.MODE_DETECTOR else 0.0)
    ctx.reset(); ctx.process_raw(ptrs, counts, True)            # (twice: the second pass sizes grids and arenas from the first one's traffic)
    ctx.reset(); n = ctx.process_raw(ptrs, counts, True)
    res = ctx.results(n)
    ok = True; n_inj = n_found = n_bytes_ok = 0
    for s in range(min(n_streams, distinct)):
        mine = [g for g in res if g["stream"] == s]
        n_inj += len(inj[s])
        if mode == fx.MODE_DETECTOR:
            pos = set(g["start"] for g in mine)
            hit = sum(1 for p, _ in inj[s] if (p in pos or p - 1 in pos or p + 1 in pos))
            n_found += hit; n_bytes_ok += hit
        else:
            by_start = {g["start"]: g for g in mine}
            for p, pl in inj[s]:
                g = by_start.get(p) or by_start.get(p - 1) or by_start.get(p + 1)
                if g is not None:
                    n_found += 1
                    n_bytes_ok += int(g["payload_valid"] and g["payload"] == pl)
    ok = n_bytes_ok == n_inj
    torch.cuda.synchronize(); t0 = time.perf_counter()
    per_pass = []
    for _ in range(passes):
        t1 = time.perf_counter()
        ctx.reset(); ctx.process_raw(ptrs, counts, True)
        if os.environ.get("BENCH_CONFIGS_DEBUG"):
            t_ = ctx.timing(); per_pass.append((round((time.perf_counter() - t1) * 1e3, 2), t_["replays"], t_["late_decodes"], t_["repairs"], round(t_["host_submit_ms"], 2), round(t_["total_ms"], 2)))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / passes
    if per_pass: print("per pass (ms, replays, late_decodes, repairs, submit ms, kernels ms):", per_pass, file=sys.stderr, flush=True)
    tm = ctx.timing()
    # the same passes with several in flight (each pass is an independent capture of all streams: reset in between)
    depth = 4
    if pipeline: ctx.set_depth(depth)
    acc = {}
    def collect():
        ctx.collect_raw()
        t = ctx.timing()
        for k in ("late_decodes", "replays", "vb_repairs", "vb_blocks", "total_ms", "paydec_ms", "walk_ms", "chain_ms"):
            acc[k] = acc.get(k, 0) + t[k]
    def pipelined(k):
        infl = 0
        for _ in range(k):
            if infl == depth:
                collect(); infl -= 1
            ctx.reset(); ctx.submit_raw(ptrs, counts, True); infl += 1
        while infl:
            collect(); infl -= 1
    dtp = float("nan")
    if pipeline:
        pipelined(2 * depth + 2); acc.clear()          # (every slot of the ring -- depth + 1 -- has its arenas, hints have settled)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipelined(3 * passes)
        torch.cuda.synchronize(); dtp = (time.perf_counter() - t0) / (3 * passes)
    out = dict(config=name, streams=n_streams, samples_per_stream=n_samples, frames=len(res), checked_streams=min(n_streams, distinct),
               injected=n_inj, found=n_found, payload_ok=n_bytes_ok, all_frames_ok=bool(ok),
               ms_per_pass=round(dt * 1e3, 3), msamples_per_s=round(n_streams * n_samples / dt / 1e6, 1),
               ms_per_pass_4_in_flight=(round(dtp * 1e3, 3) if pipeline else None), msamples_per_s_4_in_flight=(round(n_streams * n_samples / dtp / 1e6, 1) if pipeline else None),
               kernels_ms={k: round(tm[k], 3) for k in ("walk_ms", "seekverify_ms", "chain_ms", "paymf_ms", "paypll_ms", "paydec_ms")},
               host_ms={k: round(tm[k], 3) for k in ("host_submit_ms", "host_collectwait_ms")}, late_decodes=tm["late_decodes"], replays=tm["replays"],
               in_flight_sums={k: round(v, 3) for k, v in acc.items()},
               hops=tm["hops"], hops_cheap=tm["hops_cheap"], walk_jobs=tm["walk_jobs"], repairs=tm["repairs"])
    print(json.dumps(out), flush=True)
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--passes", type=int, default=5); ap.add_argument("--only", default=""); ap.add_argument("--no-pipeline", action="store_true")
    a = ap.parse_args()
    import torch
    fx = importlib.import_module("gr-liquiddsp_amd")
    mods, inner = [2, 27, 28, 29], fx.INNER_BY_INDEX
    cfgs = {
        "3": lambda: run(fx, torch, "3: frame_detector_cc, 256 x 2^20", 256, 1 << 20,
                         lambda i, n: fx.synth_stream(n, stream_id=3000 + i), fx.MODE_DETECTOR, a.passes, pipeline=not a.no_pipeline),
        "4": lambda: run(fx, torch, "4: flex_rx 128 x 2^21, QAM16 r2/3", 128, 1 << 21,
                         lambda i, n: fx.synth_stream(n, stream_id=4000 + i, mod=27, fec0=15, snr_db=25.0), fx.MODE_FLEX_RX, a.passes, pipeline=not a.no_pipeline),
        "5": lambda: run(fx, torch, "5 (one GPU's share): 128 x 2^20, mod/FEC sweep", 128, 1 << 20,
                         lambda i, n: fx.synth_stream(n, stream_id=5000 + i, mod=mods[i % 4], fec0=inner[i % 7], snr_db=32.0),
                         fx.MODE_FLEX_RX, a.passes, distinct=28, pipeline=not a.no_pipeline),
    }
    for k, f in cfgs.items():
        if not a.only or k in a.only.split(","):
            f()


if __name__ == "__main__":
    main()
