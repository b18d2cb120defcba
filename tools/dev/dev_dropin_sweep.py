"""Drop-in boundary (flexframesync_execute in 256-sample calls through the C++ flex_rx shell) over block length x blocks in
flight: python tools/dev/dev_dropin_sweep.py [driver.so]   (driver.so: a libdropin_feed.so built against another libfxrx.so,
e.g. round 2's synchronous drop-in, for the before / after figure)"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
fx = importlib.import_module("gr-liquiddsp_amd")
st = fx._ffi.DropinStats()
if len(sys.argv) > 1:
    # the stream comes from the OTHER library's own generator (the wire format may differ between rounds: Hamming(8,4) table)
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process: torch's first)
    F = C.CDLL(sys.argv[1])
    L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "libfxrx.so"))
    L.flexframegen_create.restype = C.c_void_p; L.flexframegen_create.argtypes = [C.c_void_p]
    L.flexframegen_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    L.flexframegen_getframelen.restype = C.c_uint; L.flexframegen_getframelen.argtypes = [C.c_void_p]
    L.flexframegen_write_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    g = L.flexframegen_create(C.byref(fx._ffi.GenProps(5, 11, 1, 2)))
    rng = np.random.RandomState(1); x = np.zeros(20_000_000, np.complex64); inj = []; p = 0; hdr = np.zeros(14, np.uint8)
    while True:
        pl = rng.randint(0, 256, 1024).astype(np.uint8)
        L.flexframegen_assemble(g, hdr.ctypes.data, pl.ctypes.data, 1024); fl = L.flexframegen_getframelen(g)
        if p + fl > len(x): break
        buf = np.empty(fl, np.complex64); L.flexframegen_write_samples(g, buf.ctypes.data, fl); x[p:p + fl] = buf; inj.append(p); p += fl + 256
    x *= np.exp(1j * (0.02 * np.arange(len(x)) + 0.5)).astype(np.complex64)
    x += (np.float32(np.sqrt(0.5 * 0.01)) * rng.standard_normal(2 * len(x)).astype(np.float32).view(np.complex64))
    n = len(x) - len(x) % 256
    F.dropin_feed.restype = C.c_int
    F.dropin_feed.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint, C.c_uint, C.POINTER(fx._ffi.DropinStats)]
    for blk in (1 << 16, 1 << 18, 1 << 20):
        os.environ["FXRX_SYNC_BLOCK"] = str(blk)      # (round 2 has no such knob: the driver's block shell cannot change it either; printed for the record)
        F.dropin_feed(x.ctypes.data, n, 4096, 1, C.byref(st))
        print("other library %s: %.1f Msamples/s, %d of %d frames valid, first frame after %.1f ms" % (sys.argv[1], n / st.seconds / 1e6, st.payload_valid, len(inj), st.first_frame_seconds * 1e3), flush=True)
        break
    sys.exit(0)
x, inj = fx.synth_stream(20_000_000, stream_id=0)
n = len(x) - len(x) % 256
F = fx._ffi.feed_lib()
print("copy-only ceiling (256-sample memcpy into pinned memory): %.0f Msamples/s" % (F.dropin_copy_ceiling(x.ctypes.data, n, 1 << 20, 3) / 1e6), flush=True)
for blk in (1 << 20, 1 << 21):
    for depth in (3, 4):
        os.environ["FXRX_SYNC_BLOCK"] = str(blk); os.environ["FXRX_SYNC_DEPTH"] = str(depth)
        F.dropin_feed(x.ctypes.data, n, 4096, 1, C.byref(st))
        F.dropin_feed(x.ctypes.data, n, 4096, 3, C.byref(st))
        ok = st.payload_valid == 3 * len(inj) and st.errors == 0
        shell = 3 * n / st.seconds / 1e6
        F.dropin_feed_raw(x.ctypes.data, n, 3, C.byref(st))
        print("block %8d depth %d: %8.1f Msamples/s through the block shell, %8.1f bare callback  %s" % (blk, depth, shell, 3 * n / st.seconds / 1e6, "ok" if ok and st.payload_valid == 3 * len(inj) else "FRAMES MISSING %d/%d" % (st.payload_valid, 3 * len(inj))), flush=True)
