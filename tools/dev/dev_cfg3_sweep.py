"""config 3 (detector only) against walker segment length and detector-kernel occupancy build: ms per pass, one at a time and four in flight"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = [("default", None)] + [(os.path.basename(p), p) for p in sys.argv[1:]]
for name, lib in libs:
    for seg in (0, 174763, 149797, 131072, 116509, 104858, 87382, 65536):
        env = dict(os.environ)
        if lib: env["FXRX_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_configs.py"), "--only", "3", "--passes", "5", "--segment-len", str(seg)], capture_output=True, text=True, env=env)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print("%-20s seg %7d jobs %5d: %.2f ms alone (walk %.2f), %.2f ms in flight -> %.0f / %.0f Msamples/s, found %d/%d" % (name, seg, d["walk_jobs"], d["ms_per_pass"],
                  d["kernels_ms_last_group"]["walk_ms"], d["ms_per_pass_4_in_flight"], d["msamples_per_s"], d["msamples_per_s_4_in_flight"], d["found"], d["injected"]), flush=True)
        except Exception as e:
            print("FAILED", name, seg, out.stderr[-400:], flush=True)
