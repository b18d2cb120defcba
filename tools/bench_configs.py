#!/usr/bin/env python3
"""Throughput of BASELINE configs 3, 4 and 5 (not the headline bench line; see bench.py for that, or `bench.py --config N`,
which ends up here).

  config 3: frame_detector_cc only, 256 streams x 2^20 samples (detector mode), one GPU
  config 4: flex_rx batched, 128 streams x 2^21 samples, QAM16 r=2/3 (V27P23), one GPU
  config 5: 1024 streams x 2^20 samples sharded over the ranks (128 per GPU on 8 GPUs); stream s uses modulation
            {PSK4, QAM16, QAM32, QAM64}[s mod 4] and inner code index s mod 7 -- the part of the cognitive engine's grid
            (/root/reference/python/cognitive_engine.py:525-527) that BASELINE.json configs[4] names

Every stream is DISTINCT and generated on the device (fxtx_generate + fxtx_apply_channel: own payloads, own CFO / phase /
delay, own noise), so the IQ a pass reads is as large as the config says (2 GiB for configs 3 and 4, 1 GiB per 128 streams of
config 5) -- nothing is tiled.  Es/N0 = 20 dB as BASELINE.md section 3 specifies (--snr to change it); at 20 dB the dense
uncoded constellations of config 5 lose payloads, as they would in any receiver: the check before a rate is printed is that every injected frame is
found with a valid header, and payload_ok / injected is reported.  A rank handles its streams in groups of --group (one context call each); a pass = all groups of all ranks, first
one group at a time, then with four groups in flight.

--gpus N without a launcher starts the N ranks itself (bench.py's launcher: fresh interpreters, one GPU each, rendezvous on
127.0.0.1); the only thing that crosses ranks is the barrier and the max of the elapsed time (streams are independent: no
collective on the data path).  BENCH_STUB=1 rehearses that logic without a GPU (tests/test_dist.py)."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import bench as B

MODS = [2, 27, 28, 29]                      # PSK4, QAM16, QAM32, QAM64 (liquid enum values)
INNER = [1, 11, 15, 17, 18, 19, 20]         # block-API inner_code 0..6 (lib/flex_tx_impl.cc:118-146)

CONFIGS = {
    "3": dict(name="3: frame_detector_cc, 256 x 2^20", streams=256, n=1 << 20, group=256, detect=True, props=lambda sid: {}),
    "4": dict(name="4: flex_rx 128 x 2^21, QAM16 r2/3", streams=128, n=1 << 21, group=128, detect=False, props=lambda sid: dict(mod=27, fec0=15)),
    "5": dict(name="5: 1024 x 2^20 over the ranks, mod/FEC sweep", streams=1024, n=1 << 20, group=128, detect=False,
              props=lambda sid: dict(mod=MODS[sid % 4], fec0=INNER[sid % 7])),
}


def config5_props(sid):
    return CONFIGS["5"]["props"](sid)


def check_group(fx, res, inj, detect, n_streams):
    """every injected frame of every stream found (+-1 sample) with a valid header; counts"""
    per = [dict() for _ in range(n_streams)]
    for g in res:
        per[g["stream"]][g["start"]] = g
    n_inj = n_found = n_ok = 0
    for s in range(n_streams):
        by = per[s]
        for p, pl in inj[s]:
            n_inj += 1
            g = by.get(p) or by.get(p - 1) or by.get(p + 1)
            if g is None: continue
            if detect: n_found += 1; n_ok += 1
            else:
                n_found += int(g["header_valid"]); n_ok += int(g["payload_valid"] and g["payload"] == pl)
    return n_inj, n_found, n_ok


def run_rank(a, key, rank, world, dist, rdev, stub):
    cfg = CONFIGS[key]
    total = a.streams if a.streams else cfg["streams"]
    mine = B.shard_streams(total, rank, world) if key == "5" else list(range(total))     # configs 3 / 4 are one-GPU configs: every rank runs its own copy
    group = min(a.group if a.group else cfg["group"], len(mine))
    n = a.samples if a.samples else cfg["n"]
    groups = [mine[i:i + group] for i in range(0, len(mine), group)]
    groups = [g for g in groups if len(g) == group] or [mine]                           # (whole groups only: one context size)
    out = dict(config=cfg["name"], n_gpus=world, streams=total, streams_this_rank=sum(len(g) for g in groups), groups_per_pass=len(groups), streams_per_group=len(groups[0]),
               samples_per_stream=n, snr_db=a.snr)
    if stub:
        time.sleep(0.01 * (1 + rank))
        dt = B.reduce_max_time(0.01 * (1 + rank), dist if world > 1 else None, rdev)
        out.update(data="stub", distinct_streams=out["streams_this_rank"], first_stream=groups[0][0], last_stream=groups[-1][-1], ms_per_pass=round(dt * 1e3, 3))
        return out
    import torch
    fx = importlib.import_module("gr-liquiddsp_amd")
    dev = torch.cuda.current_device()
    tx = fx.TxContext(dev)
    t0 = time.perf_counter()
    data = []
    for g in groups:        # stream ids are global: every stream of the job is distinct, whatever rank holds it
        x, inj = fx.synth_streams_device(len(g), n, first_stream_id=1000 * int(key) + g[0], props=lambda sid: cfg["props"](sid - 1000 * int(key)), snr_db=a.snr, device=dev, tx=tx)
        data.append((x, inj))
    tx.close()
    torch.cuda.synchronize()
    out["generate_s"] = round(time.perf_counter() - t0, 2)
    ptrs = [[x[s].data_ptr() for s in range(x.shape[0])] for x, _ in data]
    assert len(set(p for pp in ptrs for p in pp)) == out["streams_this_rank"]
    out["distinct_streams"] = len(set(p for pp in ptrs for p in pp))
    out["iq_bytes_this_rank"] = int(sum(x.numel() for x, _ in data) * 8)
    counts = [n] * len(groups[0])
    mode = fx.MODE_DETECTOR if cfg["detect"] else fx.MODE_FLEX_RX
    ctx = fx.RxContext(len(groups[0]), mode=mode, device=dev, threshold=0.45 if cfg["detect"] else 0.0, segment_len=a.segment_len)
    n_inj = n_found = n_ok = n_frames = 0
    for gi, (x, inj) in enumerate(data):                      # (first touch sizes arenas and grids; also the correctness check)
        ctx.reset(); ctx.process_raw(ptrs[gi], counts, True)
        ctx.reset(); res = ctx.results(ctx.process_raw(ptrs[gi], counts, True))
        a_, b_, c_ = check_group(fx, res, inj, cfg["detect"], len(inj)); n_inj += a_; n_found += b_; n_ok += c_; n_frames += len(res)
    out.update(frames=n_frames, injected=n_inj, found=n_found, payload_ok=n_ok)
    # (a sequential synchroniser does not find every injected frame: a false alarm on a frame's tail costs it 618 samples and a
    # preamble inside them is gone, and at 20 dB a header fails now and then -- a few in a hundred thousand, in the oracle as
    # here: tests/test_gpu_configs.py compares such streams with it; more than that is an error)
    out["not_found"] = n_inj - n_found
    if n_inj - n_found > max(2, n_inj // 5000) and not a.allow_missing:
        raise SystemExit("bench_configs: found %d of %d injected frames -- refusing to report a throughput" % (n_found, n_inj))

    def sync():
        torch.cuda.synchronize()
        if world > 1: dist.barrier()
    # one group at a time
    sync(); t0 = time.perf_counter()
    for _ in range(a.passes):
        for gi in range(len(groups)):
            ctx.reset(); ctx.process_raw(ptrs[gi], counts, True)
    sync(); dt = B.reduce_max_time((time.perf_counter() - t0) / a.passes, dist if world > 1 else None, rdev)
    tm = ctx.timing()
    out.update(ms_per_pass=round(dt * 1e3, 3), kernels_ms_last_group={k: round(tm[k], 3) for k in ("walk_ms", "seekverify_ms", "chain_ms", "paymf_ms", "paypll_ms", "paydec_ms")},
               replays=tm["replays"], repairs=tm["repairs"], late_decodes=tm["late_decodes"], hops=tm["hops"], walk_jobs=tm["walk_jobs"])
    # four groups in flight (each an independent capture: reset in between)
    dtp = None
    if not a.no_pipeline:
        depth = 4
        ctx.set_depth(depth)
        acc = {}

        def collect():
            ctx.collect_raw(); t = ctx.timing()
            for k in ("late_decodes", "replays", "vb_repairs", "total_ms"): acc[k] = acc.get(k, 0) + t[k]

        def pipelined(k):
            infl = 0
            for i in range(k):
                if infl == depth: collect(); infl -= 1
                ctx.reset(); ctx.submit_raw(ptrs[i % len(groups)], counts, True); infl += 1
            while infl: collect(); infl -= 1
        pipelined(2 * depth + 2); acc.clear()
        sync(); t0 = time.perf_counter()
        reps = max(3 * a.passes, 1) * len(groups)
        pipelined(reps)
        sync(); dtp = B.reduce_max_time((time.perf_counter() - t0) / reps * len(groups), dist if world > 1 else None, rdev)
        out.update(ms_per_pass_4_in_flight=round(dtp * 1e3, 3), late_decodes_in_flight=int(acc.get("late_decodes", 0)), vb_repairs_in_flight=int(acc.get("vb_repairs", 0)))
    # whole-job samples per pass: config 5 is one job over all ranks; configs 3 / 4 replicate (weak scaling)
    per_rank = sum(len(g) for g in groups) * n
    if world > 1:
        t = torch.tensor([float(per_rank)], dtype=torch.float64, device=rdev if rdev is not None else "cpu")
        dist.all_reduce(t); job = float(t.item())
    else:
        job = float(per_rank)
    out["samples_per_pass_all_ranks"] = int(job)
    out["msamples_per_s"] = round(job / dt / 1e6, 1)
    if dtp: out["msamples_per_s_4_in_flight"] = round(job / dtp / 1e6, 1)
    out["iq_gbs_4_in_flight"] = round(8.0 * job / (dtp or dt) / 1e9, 1)
    ctx.close()
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--only", default="", help="comma-separated config numbers (3,4,5); default all")
    ap.add_argument("--no-pipeline", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--snr", type=float, default=20.0, help="Es/N0 in dB (BASELINE.md section 3: 20)")
    ap.add_argument("--streams", type=int, default=0, help="override the config's stream count (rehearsals)")
    ap.add_argument("--samples", type=int, default=0, help="override samples per stream (rehearsals)")
    ap.add_argument("--group", type=int, default=0, help="streams per context call")
    ap.add_argument("--segment-len", type=int, default=0, help="walker segment length in samples (0: the library's own choice)")
    ap.add_argument("--allow-missing", action="store_true", help="report a rate even if an injected frame was not found (low SNR experiments)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    stub = os.environ.get("BENCH_STUB", "0") == "1"
    if "RANK" not in os.environ and a.gpus > 1:
        raise SystemExit(B.launch_ranks(a.gpus, argv, script=os.path.abspath(__file__)))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench_configs.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench_configs.py needs a HIP device (there is no CPU path to measure)")
    if "BENCH_FORCE_DEVICE" in os.environ: local = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_DIST_BACKEND", "gloo" if stub else "nccl")
    dev = None
    if not stub:
        torch.cuda.set_device(local); dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl": dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else: dist.init_process_group(backend, rank=rank, world_size=world)
    rdev = dev if backend == "nccl" else None
    for k in CONFIGS:
        if a.only and k not in a.only.split(","): continue
        out = run_rank(a, k, rank, world, dist, rdev, stub)
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, dict(rank=rank, first=out.get("first_stream"), last=out.get("last_stream"), streams=out["streams_this_rank"],
                                                  found=out.get("found"), injected=out.get("injected"), payload_ok=out.get("payload_ok"), frames=out.get("frames")))
            out["ranks"] = gathered
        if rank == 0: print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
