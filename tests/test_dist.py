"""N>1 path of bench.py on CPU: world_size 2 over gloo.  Streams shard with no data-path collective; the only
cross-rank traffic is the barrier and the max-reduce of the elapsed time (bench.py: shard_streams, reduce_max_time)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    r, w = dist.get_rank(), dist.get_world_size()
    mine = bench.shard_streams(1024, r, w)                 # config 5: 1024 streams over the ranks
    one = bench.shard_streams(w, r, w)                     # headline: one stream per rank
    dist.barrier()
    t = bench.reduce_max_time(1.0 + r, dist)               # rank r pretends to have taken 1+r seconds
    print(json.dumps(dict(rank=r, n=len(mine), lo=mine[0], hi=mine[-1], one=one, tmax=t)), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def test_gloo_world2_sharding_and_time_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29573", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [d["n"] for d in outs] == [512, 512]
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 511, 512, 1023)
    assert outs[0]["one"] == [0] and outs[1]["one"] == [1]
    assert all(d["tmax"] == 2.0 for d in outs)             # max over ranks, identical on every rank


def test_shard_streams_covers_everything_once():
    sys.path.insert(0, ROOT)
    import bench
    for total, world in ((1024, 8), (1000, 8), (7, 4), (8, 8), (3, 8)):
        got = sum((bench.shard_streams(total, r, world) for r in range(world)), [])
        assert got == list(range(total))


def _run_bench(args, env_extra, timeout=240):
    env = dict(os.environ, BENCH_STUB="1", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        if k not in env_extra: env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_spawns_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher: main() starts two fresh rank processes itself (RANK / WORLD_SIZE /
    MASTER_* set), they rendezvous (gloo here; the GPU part is stubbed by BENCH_STUB=1), rank 0 prints the line with
    n_gpus = 2; the timed region is stretched to --min-time by repeating the K-step schedule."""
    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--min-time", "0.2"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                         # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["data"] == "stub" and d["scaling"] == "weak"
    assert d["steps"] == 4 and d["repeats"] >= 2 and d["passes_timed"] == 4 * d["repeats"]
    # max over ranks: rank 1's stub step is twice as slow as rank 0's, and the whole region lasted >= min-time
    assert d["ms_per_step"] * d["passes_timed"] >= 190.0
    assert d["ms_per_step"] >= 0.9


def test_bench_under_a_launcher_checks_world_size():
    """Ranks from the environment (torch.distributed.run style): --gpus must agree with WORLD_SIZE, loudly."""
    r = _run_bench(["--gpus", "8", "--steps", "2"], dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    r = _run_bench(["--gpus", "1", "--steps", "2", "--min-time", "0.01"], dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_repeat_count():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.choose_repeats(0.001, 20, 1.0) == 50
    assert bench.choose_repeats(0.001, 200, 0.1) == 1
    assert bench.choose_repeats(0.0, 20, 1.0) == 1
    e = bench.rank_env(3, 8, 1234, base={})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "1234")


def test_config5_runner_shards_1024_streams_over_the_ranks():
    """BASELINE config 5 as specified: 1024 streams sharded over the ranks, stream s -> modulation {PSK4, QAM16, QAM32, QAM64}
    [s mod 4] x inner code s mod 7.  `tools/bench_configs.py --only 5 --gpus 2` (and `bench.py --config 5 --gpus 2`, which
    forwards to it) start the ranks through bench.py's launcher; with BENCH_STUB=1 the GPU part is skipped and what is
    checked is the launch, the rendezvous (gloo), who takes which streams, and that rank 0 prints one line for the job.
    The same command on a GPU box with BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo runs two ranks' real shares on one GPU."""
    for cmd in ([os.path.join(ROOT, "tools", "bench_configs.py"), "--only", "5", "--gpus", "2"], [os.path.join(ROOT, "bench.py"), "--config", "5", "--gpus", "2"]):
        env = dict(os.environ, BENCH_STUB="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"): env.pop(k, None)
        r = subprocess.run([sys.executable] + cmd, env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["streams"] == 1024 and d["streams_this_rank"] == 512 and d["groups_per_pass"] == 4 and d["streams_per_group"] == 128
        assert d["distinct_streams"] == d["streams_this_rank"]
        ranks = sorted(d["ranks"], key=lambda x: x["rank"])
        assert [(x["first"], x["last"], x["streams"]) for x in ranks] == [(0, 511, 512), (512, 1023, 512)]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    grid = [bench_configs.config5_props(s) for s in range(1024)]
    assert [g["mod"] for g in grid[:4]] == [2, 27, 28, 29] and [g["fec0"] for g in grid[:7]] == [1, 11, 15, 17, 18, 19, 20]
    assert len(set((g["mod"], g["fec0"]) for g in grid)) == 28          # every (modulation, inner code) pair of the sweep occurs
