"""Job timeline of the detector-only walker on config 3 (needs a -DFX_STAMPS build:
make -C gr-liquiddsp_amd/csrc EXTRA=-DFX_STAMPS OUT=libfxrx_stamps.so OBJDIR=build_stamps libfxrx_stamps.so;
FXRX_LIB=gr-liquiddsp_amd/csrc/libfxrx_stamps.so python tools/dev/dev_walk_timeline.py [streams] [segment_len]).
Every walk job records its start and end on the 100 MHz wall clock and the CU it ran on: how many jobs run at any time, how long the
chip takes to fill and to drain, how job durations spread."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fx = importlib.import_module("gr-liquiddsp_amd")
import torch
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x, inj = fx.synth_streams_device(ns, 1 << 20, first_stream_id=3000, snr_db=20.0)
torch.cuda.synchronize()
ptrs = [x[s].data_ptr() for s in range(ns)]
ctx = fx.RxContext(ns, mode=fx.MODE_DETECTOR, threshold=0.45, segment_len=seg)
for it in range(3):
    n = ctx.process_raw(ptrs, [1 << 20] * ns, True); tm = ctx.timing(); ctx.reset()
L = fx.lib()
buf = np.zeros(6 * 65536, dtype=np.uint32)
nj = L.fxrx_debug_walk_jobs(ctx.h, buf.ctypes.data, 65536)
r = buf[:6 * nj].reshape(nj, 6).astype(np.int64)
t0 = r[:, 0].min()
st = (r[:, 0] - t0) / 100e3; en = (r[:, 1] - t0) / 100e3          # ms
dur = en - st
print("walk_ms %.2f, %d jobs, %d hops; first start 0, last start %.2f, first end %.2f, last end %.2f ms" % (tm["walk_ms"], nj, r[:, 4].sum(), st.max(), en.min(), en.max()))
print("job duration ms: min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f;  us per hop (job time x 4 waves / hops): median %.1f" % (dur.min(), np.percentile(dur, 10), np.median(dur),
      np.percentile(dur, 90), dur.max(), np.median(dur * 1e3 / r[:, 4])))
T = en.max()
grid = np.linspace(0, T, 41)
act = [(int(((st <= t) & (en > t)).sum())) for t in grid]
print("jobs running at t (ms):", " ".join("%.1f:%d" % (t, a) for t, a in zip(grid, act)))
cu = r[:, 2]
print("distinct CU ids %d; jobs per CU: min %d max %d" % (len(set(cu.tolist())), np.bincount(np.unique(cu, return_inverse=True)[1]).min(), np.bincount(np.unique(cu, return_inverse=True)[1]).max()))
# by starting order: duration of the jobs of the first wave against those started later
first = st < 0.5
print("jobs started in the first 0.5 ms: %d, median duration %.2f ms; the rest: %d, median duration %.2f ms" % (first.sum(), np.median(dur[first]), (~first).sum(), np.median(dur[~first]) if (~first).any() else 0))
print("work conservation: sum of job durations %.0f ms = %.1f jobs running on average over the kernel's %.2f ms" % (dur.sum(), dur.sum() / T, T))
