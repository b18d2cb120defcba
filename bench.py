#!/usr/bin/env python3
"""bench.py -- Msamples/s through flex_rx (QPSK/PSK4 r=1/2, 1024-B payload, CRC-24) on synthetic IQ.

Workload at N=1 = BASELINE.json configs[1]: one stream, 10 Msym = 20 Msamples of complex64, frames back to
back with 256-sample gaps, CFO/phase/delay/AWGN(20 dB) channel.  With --gpus N every rank
receives its own independent stream of that size (streams shard with no collective: "weak" scaling); the only
torch.distributed traffic is the barrier and the max-over-ranks of the elapsed time.

Launching.  `python bench.py --gpus N` with no RANK in the environment starts N child processes itself -- one per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before this process has made any GPU call (children are fresh
interpreters, nothing is exec'ed over a process that touched the GPU) -- and relays rank 0's JSON line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks come from the environment and
--gpus must agree with WORLD_SIZE (it fails loudly otherwise).

A step = reset the synchroniser + one full pass of the stream through the HIP path (walker, stitch/plan, seek
verification, payload MF, payload PLL, packet decode), decoded payloads copied back to the host, payload symbols left in HBM.
The IQ is resident in HBM before the timed region starts.  Steps are issued through the library's submit/collect
pipeline (--depth blocks in flight) and the timed region runs from an idle GPU to an idle GPU.  So that the fill and
drain of that pipeline do not decide the number, the K-step schedule is repeated R times back to back inside ONE timed
region, R chosen from the warm-up rate so that the region lasts at least --min-time seconds (default 1 s):
ms_per_step = dt / (K R).  `steps`, `repeats` and `passes_timed` are reported.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Runtime knobs, set before anything initialises HIP.  The pipeline keeps several blocks in flight on separate HIP
# streams; HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them torch's), and
# streams that share a queue serialise.  (libfxrx sets the same default itself when it is loaded before HIP initialises;
# torch gets there first here.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def choose_repeats(est_step_s, steps, min_time_s):
    """R such that R * steps passes last at least min_time_s at the warm-up rate (>= 1)."""
    if est_step_s <= 0.0 or steps <= 0:
        return 1
    return max(1, int(math.ceil(min_time_s / (est_step_s * steps))))


def rank_env(rank, world, port, base=None):
    """Environment of child rank `rank` (what torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, script=None):
    """--gpus N without a launcher: start N fresh child processes (one per GPU) and wait for them.  The parent never
    imports torch and never touches the GPU.  Returns the worst child exit code.  (script: tools/bench_configs.py starts its
    ranks through here as well.)"""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=rank_env(r, n, port)) for r in range(n)]
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in pending:          # a rank died: the others would wait in a barrier for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _oracle_pass(o, x, chunk=256):
    s = o.Sync()
    t = time.perf_counter()
    fr = s.execute(x, chunk=chunk)
    dt = time.perf_counter() - t
    n = len(fr)
    s.close()
    return dt, n


def cpu_baseline(x, seconds_budget=12.0):
    """Times the CPU oracle (oracle/, kind='port': a restatement, not libliquid) on this host, driven in 256-sample
    execute calls like lib/flex_rx_impl.cc:212-215, on a bounded prefix of the workload: (i) one core -- the reference's
    own threading model, one thread per block instance; (ii) all host cores, one independent synchroniser (= one stream)
    per thread, SURVEY.md 8(d).  The oracle runs inside ctypes calls, which release the GIL."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as o
    o.lib()
    probe = min(len(x), 2_000_000)
    dt, _ = _oracle_pass(o, x[:probe])
    rate = probe / dt
    n = len(x) - len(x) % 256
    n1 = int(min(n, max(1 << 18, rate * seconds_budget))) // 256 * 256
    passes = int(max(1, min(20, round(rate * seconds_budget / n1))))          # about seconds_budget of single-core work
    dt1, nfr = 0.0, 0
    for _ in range(passes):
        d_, nfr = _oracle_pass(o, x[:n1]); dt1 += d_
    one = dict(value=round(passes * n1 / dt1 / 1e6, 3), unit="Msamples/s", cores=1, kind="port",
               sample="%d pass(es) over the first %d samples of the bench stream (%d frames each), 256-sample execute calls, %.1f s of CPU" % (passes, n1, nfr, dt1))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    pa = 1                                                 # (one pass over a shorter prefix per thread and round: two rounds take ~30 s on 256 threads)
    n_all = min(n1, 8_000_000) // 256 * 256

    def work(i):
        for _ in range(pa): _oracle_pass(o, x[:n_all])
    wall = None
    for _ in range(2):                                     # best of two: the first round also pays thread start-up and cold caches
        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        w = time.perf_counter() - t0
        wall = w if wall is None else min(wall, w)
    allc = dict(value=round(cores * pa * n_all / wall / 1e6, 3), unit="Msamples/s", cores=cores, kind="port", nproc=os.cpu_count(),
                sample="%d threads, each its own synchroniser, %d pass(es) over the first %d samples (one stream per thread), best of two rounds, %.1f s wall" % (cores, pa, n_all, wall))
    return one, allc


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic.json,
    made by profiles/collect_pmc.sh + profiles/summarize_pmc.py).  None when no such profile is committed."""
    try:
        p = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        d = json.load(open(p))
        k = d["kernels"].get(kernel)
        return None if k is None else k["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=N_SAMPLES)
    ap.add_argument("--min-time", type=float, default=1.0, help="the timed region lasts at least this long (seconds): the K-step schedule is repeated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-constellation", action="store_true", help="skip the second figure (payload symbols copied to the host)")
    ap.add_argument("--no-pipeline", action="store_true", help="one block in flight (latency mode)")
    ap.add_argument("--depth", type=int, default=12, help="blocks in flight in the timed region")
    ap.add_argument("--segment", type=int, default=0, help="speculation segment length in samples (0 = library default)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE config: 2 = the headline line (default); 3 / 4 / 5 = tools/bench_configs.py (distinct streams generated on the device; "
                         "config 5: 1024 streams sharded over --gpus ranks)")
    ap.add_argument("--continuous", action="store_true",
                    help="feed the passes as consecutive blocks of ONE continuing stream (no reset in between): not the headline number")
    return ap.parse_args(argv)


class StubPath:
    """BENCH_STUB=1: stands in for the GPU part so that the launch / rendezvous / reduction logic of main() can be
    rehearsed on a CPU-only box (tests/test_dist.py).  Its output line says data = "stub"; it measures nothing."""

    def __init__(self, rank):
        self.rank = rank

    def run_steps(self, k):
        time.sleep(0.0005 * k * (1 + self.rank))
        return 0


def main(argv=None):
    a = parse_args(argv)
    argv = list(sys.argv[1:] if argv is None else argv)
    stub = os.environ.get("BENCH_STUB", "0") == "1"
    if a.config != 2:                                      # the other BASELINE configs have their own runner (same launcher, same rendezvous)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        return bench_configs.main(["--only", str(a.config), "--gpus", str(a.gpus), "--passes", str(max(1, min(a.steps, 5)))] + (["--no-pipeline"] if a.no_pipeline else []))
    if "RANK" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a.gpus, argv))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with a matching world (or without a launcher: "
                         "bench.py starts the ranks itself)" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    # BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist only to rehearse the N>1 code path on a one-GPU box (all ranks on
    # device 0 over gloo); the driver's multi-GPU run uses one GPU per rank over RCCL ("nccl").
    if "BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_DIST_BACKEND", "gloo" if stub else "nccl")
    dev = None
    if not stub:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    rdev = dev if backend == "nccl" else None

    (sid,) = shard_streams(world, rank, world)            # one stream per rank
    depth = 1 if a.no_pipeline else a.depth

    if stub:
        path = StubPath(rank)
        run_steps = path.run_steps
        kt = kt_live = None
    else:
        fx = importlib.import_module("gr-liquiddsp_amd")
        x, injected = fx.synth_stream(a.samples, stream_id=sid)
        xd = torch.from_numpy(x).to(dev)
        ctx = fx.RxContext(1, device=local, segment_len=a.segment)
        torch.cuda.synchronize()
        ptrs, counts = [xd.data_ptr()], [xd.numel()]

        def check(c, nres):
            frames = c.results(nres)
            ok = sum(1 for g, (_, pl) in zip(frames, injected) if g["payload_valid"] and g["payload"] == pl)
            if ok != len(injected):
                raise SystemExit("bench: decoded %d of %d injected frames -- refusing to report a throughput" % (ok, len(injected)))
            return ok

        # per-kernel device times (HIP events on the library's own streams), taken un-pipelined so that they are
        # pure kernel durations: these feed `kernels_ms_one_block_in_flight`
        kt = dict(walk_ms=0.0, seekverify_ms=0.0, chain_ms=0.0, paymf_ms=0.0, paypll_ms=0.0, paydec_ms=0.0, total_ms=0.0)
        for i in range(max(a.warmup, 1) + 3):
            ctx.reset()
            nres = ctx.process_raw(ptrs, counts, True)
            if i >= max(a.warmup, 1):
                tm = ctx.timing()
                for k in kt: kt[k] += tm[k] / 3.0
        ok = check(ctx, nres)

        # timed region: each step a full pass (reset + walk + MF + PLL + decode + results to the host), issued
        # through the submit/collect pipeline the way a streaming receiver feeds consecutive blocks
        ctx.set_depth(depth)
        # Stage times come from HIP events recorded between a block's kernels; every event is one more packet in the block's queue, and
        # with blocks in flight the packet rate is what limits throughput.  The timed region therefore records
        # only the pair around the kernel the roofline line is quoted for (the PLL: fxrx_set_timing level 1); a second, shorter region
        # with all stage events on fills `kernels_ms` (and is reported as value_with_all_stage_events).
        ctx.set_timing(1)
        kt_live = dict(walk_ms=0.0, seekverify_ms=0.0, chain_ms=0.0, paymf_ms=0.0, paypll_ms=0.0, paydec_ms=0.0, host_submit_ms=0.0, host_walkwait_ms=0.0, host_collectwait_ms=0.0)

        def make_runner(c, acc, src=None, on_dev=True, cont=None):
            src = ptrs if src is None else src
            cont = a.continuous if cont is None else cont
            def collect():
                n = c.collect_raw()
                if acc is not None:
                    tm_ = c.timing()                                 # HIP events around each kernel, on the stream it ran on
                    for k in acc: acc[k] += tm_[k]
                return n

            def run(k):
                inflight, last = 0, 0
                if acc is not None:
                    for k_ in acc: acc[k_] = 0.0
                for _ in range(k):
                    if inflight == depth:
                        last = collect(); inflight -= 1
                    if not cont: c.reset()
                    c.submit_raw(src, counts, on_dev); inflight += 1
                while inflight:
                    last = collect(); inflight -= 1
                return last
            return run
        run_steps = make_runner(ctx, kt_live)

    def timed(run, steps, min_time):
        """warm-up, pick the repeat count from the warm-up rate (same on every rank), then ONE timed region of
        steps * repeats passes, barrier + device sync on both sides, max over ranks."""
        nw = max(a.warmup, 1) * 4
        run(nw)                                              # untimed warm-up of the pipelined path itself
        if not stub: torch.cuda.synchronize()
        t0 = time.perf_counter(); run(max(nw, 2 * depth));
        if not stub: torch.cuda.synchronize()
        est = (time.perf_counter() - t0) / max(nw, 2 * depth)
        est = reduce_max_time(est, dist if world > 1 else None, rdev)
        reps = choose_repeats(est, steps, min_time)
        while True:
            if world > 1: dist.barrier()
            if not stub: torch.cuda.synchronize()
            t0 = time.perf_counter()
            nres = run(steps * reps)
            if not stub: torch.cuda.synchronize()
            if world > 1: dist.barrier()
            dt = time.perf_counter() - t0
            dt = reduce_max_time(dt, dist if world > 1 else None, rdev)
            # the warm-up estimate contains the fill and drain of the pipeline and is pessimistic: if the region came out
            # short, time a longer one (same decision on every rank: dt is the max over ranks)
            if dt >= 0.95 * min_time or reps >= 1 << 20: break
            reps = max(reps + 1, int(math.ceil(reps * min_time / max(dt, 1e-9) * 1.1)))
        return dt, reps, nres

    dt, reps, nres = timed(run_steps, a.steps, a.min_time)
    passes = a.steps * reps
    ms_step = dt / passes * 1e3
    value = world * a.samples / (dt / passes) / 1e6

    if stub:
        if rank == 0:
            print(json.dumps({"metric": "Msamples/s through flex_rx (QPSK r1/2 1024B)", "value": round(value, 2), "unit": "Msamples/s",
                              "n_gpus": world, "steps": a.steps, "repeats": reps, "passes_timed": passes, "warmup": a.warmup,
                              "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f32", "data": "stub", "config": {"workload": "BENCH_STUB=1: launch logic only, nothing measured"}}), flush=True)
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return 0

    ok = check(ctx, nres)
    live = {k: kt_live[k] / passes for k in kt_live}          # average launch duration inside the timed region (PLL; host times)
    ctx.set_timing(2)
    dt_ev, reps_ev, nres_ev = timed(run_steps, max(1, a.steps // 4), a.min_time / 2)
    check(ctx, nres_ev)
    p_ev = max(1, a.steps // 4) * reps_ev
    live_all = {k: kt_live[k] / p_ev for k in kt_live}
    all_events = dict(value=round(world * a.samples / (dt_ev / p_ev) / 1e6, 2), ms_per_step=round(dt_ev / p_ev * 1e3, 4), passes_timed=p_ev)
    ctx.set_timing(1)

    # second figure: the same passes with the constellation (payload symbols after carrier recovery, what the reference
    # publishes per frame at lib/flex_rx_impl.cc:217-221) copied to the host with every block
    with_syms = None
    if not a.no_constellation:
        ctx2 = fx.RxContext(1, device=local, segment_len=a.segment, want_framesyms=True)
        ctx2.reset(); check(ctx2, ctx2.process_raw(ptrs, counts, True))
        ctx2.set_depth(depth)
        dt2, reps2, nres2 = timed(make_runner(ctx2, None), max(1, a.steps // 4), a.min_time / 2)
        check(ctx2, nres2)
        p2 = max(1, a.steps // 4) * reps2
        with_syms = dict(value=round(world * a.samples / (dt2 / p2) / 1e6, 2), ms_per_step=round(dt2 / p2 * 1e3, 4), passes_timed=p2,
                         d2h_bytes_per_step=int(8 * tm["payload_symbols"]))
        ctx2.close()

    # another secondary figure: the same passes fed as consecutive blocks of ONE continuing stream (no reset in between: the
    # state-dependent stretch of every block waits for the block before it)
    as_stream = None
    if not a.no_constellation and not stub and not a.continuous:
        ctx.reset()
        dtc, repsc, nresc = timed(make_runner(ctx, None, cont=True), max(1, a.steps // 4), a.min_time / 2)
        check(ctx, nresc)                                    # (the capture ends in noise: every block of the continuing stream holds the same whole frames)
        pc = max(1, a.steps // 4) * repsc
        as_stream = dict(value=round(world * a.samples / (dtc / pc) / 1e6, 2), ms_per_step=round(dtc / pc * 1e3, 4), passes_timed=pc)
        ctx.reset()

    # third figure ("H2D excluded and included, both reported"): the same passes with the IQ in pinned host memory,
    # uploaded inside every submit -- never `value`
    with_h2d = None
    if not a.no_constellation and not stub:
        xh = torch.from_numpy(x).pin_memory()
        ctx3 = fx.RxContext(1, device=local, segment_len=a.segment)
        ctx3.reset(); check(ctx3, ctx3.process_raw([xh.data_ptr()], counts, False))
        ctx3.set_depth(depth)
        dt3, reps3, nres3 = timed(make_runner(ctx3, None, [xh.data_ptr()], False), max(1, a.steps // 4), a.min_time / 2)
        check(ctx3, nres3)
        p3 = max(1, a.steps // 4) * reps3
        with_h2d = dict(value=round(world * a.samples / (dt3 / p3) / 1e6, 2), ms_per_step=round(dt3 / p3 * 1e3, 4), passes_timed=p3, h2d_bytes_per_step=int(8 * a.samples))
        ctx3.close()

    # fourth figure: what the reference's block really moves -- IQ from host memory in, the constellation of every frame out
    # (lib/flex_rx_impl.cc:208,217-221) -- both PCIe directions inside every step
    with_io = None
    if not a.no_constellation and not stub:
        ctx4 = fx.RxContext(1, device=local, segment_len=a.segment, want_framesyms=True)
        ctx4.reset(); check(ctx4, ctx4.process_raw([xh.data_ptr()], counts, False))
        ctx4.set_depth(depth)
        dt4, reps4, nres4 = timed(make_runner(ctx4, None, [xh.data_ptr()], False), max(1, a.steps // 4), a.min_time / 2)
        check(ctx4, nres4)
        p4 = max(1, a.steps // 4) * reps4
        with_io = dict(value=round(world * a.samples / (dt4 / p4) / 1e6, 2), ms_per_step=round(dt4 / p4 * 1e3, 4), passes_timed=p4,
                       h2d_bytes_per_step=int(8 * a.samples), d2h_bytes_per_step=int(8 * tm["payload_symbols"]))
        ctx4.close()

    # fifth figure: through the boundary the reference has -- the C++ flex_rx block shell fed by work() calls of 4096 items from
    # pageable memory, flexframesync_execute(q, in, 256) inside (lib/flex_rx_impl.cc:212-215), three messages per frame built
    # and checked (csrc/blocks/dropin_feed.cpp)
    dropin = None
    if not a.no_constellation and not stub and world == 1:
        import ctypes as C
        F = fx._ffi.feed_lib()
        st = fx._ffi.DropinStats()
        want_hash = fx._ffi.fnv1a([pl for _, pl in injected])

        def feed(reps):
            if F.dropin_feed(x.ctypes.data, len(x) - len(x) % 256, 4096, reps, C.byref(st)) != 0:
                raise SystemExit("bench: drop-in block could not be created: " + fx.lib().fxrx_last_error().decode())
            return st.seconds
        t1 = feed(1)
        if (st.frames, st.header_valid, st.payload_valid, st.errors) != (len(injected), len(injected), len(injected), 0) or st.payload_hash != want_hash:
            raise SystemExit("bench: drop-in path published %d frames (%d valid) of %d -- refusing to report a throughput" % (st.frames, st.payload_valid, len(injected)))
        rd = max(2, int(math.ceil(a.min_time / max(t1, 1e-3))))
        td = feed(rd)
        if st.frames != rd * len(injected) or st.payload_valid != rd * len(injected) or st.errors:
            raise SystemExit("bench: drop-in path lost frames in the timed run")
        dropin = dict(value=round(rd * (len(x) - len(x) % 256) / td / 1e6, 2), unit="Msamples/s", passes_timed=rd, seconds=round(td, 3), items_per_work=4096, samples_per_execute=256,
                      block_samples=int(os.environ.get("FXRX_SYNC_BLOCK", 1 << 20)), blocks_in_flight=int(os.environ.get("FXRX_SYNC_DEPTH", 3)),
                      first_frame_latency_ms=round(st.first_frame_seconds * 1e3, 2),
                      host_to_device_gbs=round(8.0 * rd * len(x) / td / 1e9, 2), constellation_to_host_gbs=round(8.0 * st.constellation_syms / td / 1e9, 2),
                      note="pageable host IQ -> memcpy into pinned ring -> async upload + kernel chain per 2^20-sample block of one continuing stream; "
                           "frames, payloads and constellations delivered through the callback, one per call")

    if rank == 0:
        # HIP events bracket the stages of a block's kernel chain.  Four stages are a single kernel; the stitch stage is two small
        # ones and the decode stage six (batch Viterbi: front part, forward pass, hand-over check x2, traceback, back part, plus
        # the wave-per-frame decoder for what that path does not take).  The roofline line is quoted for the single kernel that
        # holds the largest share of GPU time in the rocprofv3 summary of this command (profiles/): the PLL, also the longest
        # stage of a block on its own.  (Fixed, not picked per run: walker and PLL are close under load and would swap places.)
        names = dict(walk_ms="fx_walk_kernel", seekverify_ms="fx_seekverify_kernel", chain_ms="fx_chainfast_kernel+fx_plan_kernel", paymf_ms="fx_paymf_kernel",
                     paypll_ms="fx_paypll_kernel", paydec_ms="decode stage (fx_vbpre/vbfwd/vbfix/vbtrace/vbfinish_kernel + fx_paydec_kernel)")
        dom = "paypll_ms"      # (the largest share of all kernel time in the rocprofv3 summary of this command, profiles/README.md; the walker follows)
        alg_bytes = BYTES_PER_SAMPLE * a.samples
        achieved = alg_bytes / (live[dom] * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s through flex_rx (QPSK r1/2 1024B)", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "repeats": reps, "passes_timed": passes, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
            "timed_region_s": round(dt, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "flex_rx single stream per GPU, %d samples (10 Msym), PSK4 r=1/2 (CONV_V27), 1024-B payload, CRC-24, "
                                   "256-sample gaps, CFO/phase/delay + AWGN Es/N0=20 dB" % a.samples,
                       "frames_per_stream": len(injected), "frames_decoded_ok": ok, "streams_per_gpu": 1, "blocks_in_flight": depth, "stage_events": "PLL only (fxrx_set_timing level 1)",
                       "passes_per_step": 1, "steps": a.steps, "repeats": reps,
                       "passes": "consecutive blocks of one continuing stream" if a.continuous else "independent captures (reset between passes)",
                       "segments": int(tm["walk_jobs"]), "repairs": int(tm["repairs"])},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(names[dom]),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(live[dom], 4),
                         "note": "path is latency/VALU-bound, not HBM-bound: see DESIGN.md section 6"},
            "kernels_ms": {names[k]: round(live_all[k], 4) for k in names},
            "value_with_all_stage_events": all_events["value"], "with_all_stage_events": all_events,
            "kernels_ms_one_block_in_flight": {names[k]: round(kt[k], 4) for k in names},
            "device_ms_per_step": round(kt["total_ms"], 4),
            "host_ms_per_step": {"in_submit": round(live["host_submit_ms"], 4), "of_which_waiting_for_walker": round(live["host_walkwait_ms"], 4),
                                 "in_collect_waiting_for_results": round(live["host_collectwait_ms"], 4)},
            "whole_path_hbm_gbs": round(alg_bytes / (dt / passes) / 1e9, 2),
            "whole_path_hbm_frac": round(alg_bytes / (dt / passes) / 1e9 / HBM_PEAK_GBS, 5),
        }
        if with_syms is not None:
            out["value_with_constellation_d2h"] = with_syms["value"]
            out["with_constellation_d2h"] = with_syms
        if with_h2d is not None:
            out["value_with_h2d"] = with_h2d["value"]
            out["with_h2d"] = with_h2d
        if as_stream is not None:
            out["value_one_continuing_stream"] = as_stream["value"]
            out["one_continuing_stream"] = as_stream
        if with_io is not None:
            out["value_with_h2d_and_constellation_d2h"] = with_io["value"]
            out["with_h2d_and_constellation_d2h"] = with_io
        if dropin is not None:
            out["value_through_dropin_abi"] = dropin["value"]
            out["through_dropin_abi"] = dropin
        out["whole_path_hbm_gbs_iq_only"] = round(8.0 * a.samples / (dt / passes) / 1e9, 2)        # the "IQ-only" figure (8 B/sample)
        if world == 1 and not a.no_cpu_baseline:
            one, allc = cpu_baseline(x)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = allc
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
