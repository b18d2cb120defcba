"""One continuing stream cut into blocks of 2^k samples, IQ resident in HBM (or in pinned host memory: --host), no callbacks: what the
batched API does at the block lengths the drop-in uses.  python tools/dev/dev_cont_blocks.py [--host]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
host = "--host" in sys.argv
x, inj = fx.synth_stream(20_000_000, stream_id=0)
xd = torch.from_numpy(x).pin_memory() if host else torch.from_numpy(x).cuda()
base = xd.data_ptr()
only = [a for a in sys.argv[1:] if a.startswith("--only=")]
logns, depths = ((18, 19, 20, 21, 22), (3, 4, 8)) if not only else ((int(only[0][7:].split(",")[0]),), (int(only[0][7:].split(",")[1]),))
for logn in logns:
    n = 1 << logn
    nblk = len(x) // n
    for depth in depths:
        ctx = fx.RxContext(1, want_framesyms=host)
        ctx.set_depth(depth)
        acc = [0.0, 0.0, 0, 0]
        def run():
            ctx.reset(); infl = 0; got = 0
            for b in range(nblk):
                if infl == depth:
                    t = time.perf_counter(); got += ctx.collect_raw(); acc[0] += time.perf_counter() - t; acc[2] += 1; infl -= 1
                t = time.perf_counter(); ctx.submit_raw([base + 8 * n * b], [n], not host); acc[1] += time.perf_counter() - t; acc[3] += 1; infl += 1
            while infl: got += ctx.collect_raw(); infl -= 1
            return got
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps): got = run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("block 2^%d depth %d%s: %.0f Msamples/s, %.3f ms per block, %d frames; host: %.3f ms per submit, %.3f ms per collect while full" % (logn, depth, " (pinned host IQ, constellations wanted)" if host else "",
              nblk * n / dt / 1e6, dt / nblk * 1e3, got, acc[1] / max(acc[3], 1) * 1e3, acc[0] / max(acc[2], 1) * 1e3), flush=True)
        ctx.close()
