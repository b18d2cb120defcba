"""dev helper: python tests/dev_sweep.py <bench args...> -- prints compact result line"""
import json, subprocess, sys
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print(" ".join(sys.argv[1:]), "| segs", d["config"]["segments"], "depth", d["config"]["blocks_in_flight"], "| %.0f Msps  %.3f ms/step |" % (d["value"], d["ms_per_step"]), d["kernels_ms"])
except Exception as e:
    print("FAILED", e, out.stdout[-500:], out.stderr[-1500:])
