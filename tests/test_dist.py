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
