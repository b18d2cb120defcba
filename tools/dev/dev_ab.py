"""A/B harness: python tests/dev_ab.py ROUNDS 'ENV1=.. ENV2=..|args' 'ENV..|args' ...  -> alternates variants, prints medians"""
import json, os, subprocess, sys, statistics
rounds = int(sys.argv[1]); variants = sys.argv[2:]
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        envs, args = v.split("|")
        env = dict(os.environ)
        for kv in envs.split():
            k, val = kv.split("="); env[k] = val
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), "--no-cpu-baseline"] + args.split(), capture_output=True, text=True, env=env)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1]); res[v].append(d["value"])
        except Exception as e:
            print("FAILED", v, out.stderr[-500:])
for v in variants:
    xs = res[v]
    if xs: print("%-70s median %7.0f  min %7.0f  max %7.0f  (n=%d)" % (v, statistics.median(xs), min(xs), max(xs), len(xs)))
