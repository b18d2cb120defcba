#!/usr/bin/env python3
"""bench.py -- Msamples/s through flex_rx (QPSK/PSK4 r=1/2, 1024-B payload, CRC-24) on synthetic IQ.

Workload at N=1 = BASELINE.json configs[1]: one stream, 10 Msym = 20 Msamples of complex64, frames back to
back with 256-sample gaps, CFO/phase/delay/AWGN(20 dB) channel.  With --gpus N every rank
receives its own independent stream of that size (streams shard with no collective: "weak" scaling); the only
torch.distributed traffic is the barrier and the max-over-ranks of the elapsed time.

A step = reset the synchroniser + one full pass of the stream through the HIP path (walker, payload MF,
payload PLL, packet decode), decoded payloads copied back to the host, payload symbols left in HBM.
The IQ is resident in HBM before the timed region starts.  Steps are issued through the library's submit/collect
pipeline (--depth blocks in flight) and the timed region runs from an idle GPU to an idle GPU, so it contains the
fill and the drain of that pipeline (about ten block periods): the default K = 200 measures the steady rate, a run
with --steps 20 reports roughly half of it (profiles/README.md has the numbers).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Runtime knobs, set before anything initialises HIP.  The pipeline keeps several blocks in flight on separate HIP
# streams; HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them torch's), and
# streams that share a queue serialise.  Sixteen queues let the two walk streams and the payload streams (the library creates one per block in flight) run
# concurrently.  (libfxrx sets the same default itself when it is loaded before HIP initialises; torch gets there first here.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
N_SAMPLES = 20_000_000         # 10 Msym at k = 2 samples/symbol
FRAME_SAMPLES = 17066
# algorithmic HBM bytes per input sample: IQ read + decoded payload + payload symbols written
BYTES_PER_SAMPLE = 8.0 + 1024.0 / 17322.0 + 8.0 * 8224.0 / 17322.0


def shard_streams(n_streams_total, rank, world):
    """Contiguous block of stream ids for this rank (independent streams, no collective)."""
    per = n_streams_total // world
    rem = n_streams_total % world
    lo = rank * per + min(rank, rem)
    return list(range(lo, lo + per + (1 if rank < rem else 0)))


def reduce_max_time(dt, dist, device=None):
    """Max elapsed time over ranks (the only data that crosses ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(x, seconds_budget=30.0):
    """Times the CPU oracle (oracle/, kind='port': a restatement, not libliquid) on this host, single thread,
    driven in 256-sample execute calls like lib/flex_rx_impl.cc:212-215, on a bounded prefix of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as o
    L = o.lib()
    probe = 2_000_000
    s = o.Sync(); t = time.perf_counter(); s.execute(x[:probe], chunk=256); dt = time.perf_counter() - t; s.close()
    rate = probe / dt
    passes = int(max(1, min(20, round(rate * 12.0 / len(x)))))     # ~12 s of single-core work
    n = len(x) - len(x) % 256
    tot, nfr = 0.0, 0
    for _ in range(passes):
        s = o.Sync(); t = time.perf_counter(); fr = s.execute(x[:n], chunk=256); tot += time.perf_counter() - t; s.close()
        nfr = len(fr)
    return dict(value=passes * n / tot / 1e6, unit="Msamples/s", cores=1, kind="port",
                sample="%d pass(es) over the %d-sample bench stream (%d frames each), 256-sample execute calls, %.1f s of CPU" % (passes, n, nfr, tot))


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json,
    made by profiles/collect_pmc.sh + profiles/summarize_pmc.py).  None when no such profile is committed."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        d = json.load(open(p))
        k = d["kernels"].get(kernel)
        return None if k is None else k["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=N_SAMPLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one block in flight (latency mode)")
    ap.add_argument("--depth", type=int, default=10, help="blocks in flight in the timed region")
    ap.add_argument("--segment", type=int, default=0, help="speculation segment length in samples (0 = library default)")
    ap.add_argument("--continuous", action="store_true",
                    help="feed the passes as consecutive blocks of ONE continuing stream (no reset in between): not the headline number")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    # BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist only to rehearse the N>1 code path on a one-GPU box (all ranks on
    # device 0 over gloo); the driver's multi-GPU run uses one GPU per rank over RCCL ("nccl").
    if "BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    fx = importlib.import_module("gr-liquiddsp_amd")

    (sid,) = shard_streams(world, rank, world)            # one stream per rank
    x, injected = fx.synth_stream(a.samples, stream_id=sid)
    xd = torch.from_numpy(x).to(dev)
    ctx = fx.RxContext(1, device=local, segment_len=a.segment)
    torch.cuda.synchronize()
    ptrs, counts = [xd.data_ptr()], [xd.numel()]

    def check(nres):
        frames = ctx.results(nres)
        ok = sum(1 for g, (_, pl) in zip(frames, injected) if g["payload_valid"] and g["payload"] == pl)
        if ok != len(injected):
            raise SystemExit("bench: decoded %d of %d injected frames -- refusing to report a throughput" % (ok, len(injected)))
        return ok

    # per-kernel device times (HIP events on the library's own streams), taken un-pipelined so that they are
    # pure kernel durations: these feed `roofline` and `kernels_ms`
    kt = dict(walk_ms=0.0, seekverify_ms=0.0, paymf_ms=0.0, paypll_ms=0.0, paydec_ms=0.0, total_ms=0.0)
    for i in range(max(a.warmup, 1) + 3):
        ctx.reset()
        nres = ctx.process_raw(ptrs, counts, True)
        if i >= max(a.warmup, 1):
            tm = ctx.timing()
            for k in kt: kt[k] += tm[k] / 3.0
    ok = check(nres)

    # timed region: K steps, each a full pass (reset + walk + MF + PLL + decode + results to the host), issued
    # through the submit/collect pipeline the way a streaming receiver feeds consecutive blocks
    depth = 1 if a.no_pipeline else a.depth
    ctx.set_depth(depth)

    kt_live = dict(walk_ms=0.0, seekverify_ms=0.0, paymf_ms=0.0, paypll_ms=0.0, paydec_ms=0.0, host_submit_ms=0.0, host_walkwait_ms=0.0, host_collectwait_ms=0.0)

    def collect():
        n = ctx.collect_raw()
        tm_ = ctx.timing()                                   # HIP events around each kernel, on the stream it ran on
        for k in kt_live: kt_live[k] += tm_[k]
        return n

    def run_steps(k):
        inflight, last = 0, 0
        for k_ in kt_live: kt_live[k_] = 0.0
        for _ in range(k):
            if inflight == depth:
                last = collect(); inflight -= 1
            if not a.continuous: ctx.reset()
            ctx.submit_raw(ptrs, counts, True); inflight += 1
        while inflight:
            last = collect(); inflight -= 1
        return last

    run_steps(max(a.warmup, 1) * 4)                      # untimed warm-up of the pipelined path itself
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nres = run_steps(a.steps)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    dt = time.perf_counter() - t0
    dt = reduce_max_time(dt, dist if world > 1 else None, dev if backend == "nccl" else None)
    ok = check(nres)

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = world * a.samples / (dt / a.steps) / 1e6
        names = dict(walk_ms="fx_walk_kernel", seekverify_ms="fx_seekverify_kernel", paymf_ms="fx_paymf_kernel",
                     paypll_ms="fx_paypll_kernel", paydec_ms="fx_paydec_kernel")
        live = {k: kt_live[k] / a.steps for k in kt_live}    # average launch duration inside the timed region
        dom = max(names, key=lambda k: live[k])
        alg_bytes = BYTES_PER_SAMPLE * a.samples
        achieved = alg_bytes / (live[dom] * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s through flex_rx (QPSK r1/2 1024B)", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "flex_rx single stream per GPU, %d samples (10 Msym), PSK4 r=1/2 (CONV_V27), 1024-B payload, CRC-24, "
                                   "256-sample gaps, CFO/phase/delay + AWGN Es/N0=20 dB" % a.samples,
                       "frames_per_stream": len(injected), "frames_decoded_ok": ok, "streams_per_gpu": 1, "blocks_in_flight": depth,
                       "passes": "consecutive blocks of one continuing stream" if a.continuous else "independent captures (reset between passes)",
                       "segments": int(tm["walk_jobs"]), "repairs": int(tm["repairs"])},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(names[dom]),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(live[dom], 4),
                         "note": "path is latency/VALU-bound, not HBM-bound: see DESIGN.md section 6"},
            "kernels_ms": {names[k]: round(live[k], 4) for k in names},
            "kernels_ms_one_block_in_flight": {names[k]: round(kt[k], 4) for k in names},
            "device_ms_per_step": round(kt["total_ms"], 4),
            "host_ms_per_step": {"in_submit": round(live["host_submit_ms"], 4), "of_which_waiting_for_walker": round(live["host_walkwait_ms"], 4),
                                 "in_collect_waiting_for_results": round(live["host_collectwait_ms"], 4)},
            "whole_path_hbm_gbs": round(alg_bytes / (dt / a.steps) / 1e9, 2),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
