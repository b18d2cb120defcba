"""Frame generator (flex_tx counterpart, lib/flex_tx_impl.cc:191-209) and the synthetic IQ source of
SURVEY.md section 8(d) / BASELINE.md: frames back to back with 256-sample gaps, per-stream CFO / phase /
fractional delay, AWGN at Es/N0 = 20 dB.  Host-side numpy + the C-ABI flexframegen_* of libfxrx.so.
"""
import ctypes as C
import numpy as np
from . import _ffi

CRC_24, CRC_32 = 5, 6
# block-API index -> liquid enum (reference: lib/flex_tx_impl.cc:75-181)
MOD_BY_INDEX = [1, 2, 3, 4, 9, 10, 11, 18, 27, 28, 29]
INNER_BY_INDEX = [1, 11, 15, 17, 18, 19, 20]
OUTER_BY_INDEX = [1, 7, 27, 4, 6, 8, 9, 10]


class FrameGen:
    def __init__(self, mod=2, fec0=11, fec1=1, check=CRC_24):
        self.L = _ffi.lib()
        self.props = _ffi.GenProps(check, fec0, fec1, mod)
        self.h = self.L.flexframegen_create(C.byref(self.props))
        if not self.h:
            raise ValueError("unsupported frame properties")

    def setprops(self, mod=None, fec0=None, fec1=None, check=None):
        p = self.props
        if mod is not None: p.mod_scheme = mod
        if fec0 is not None: p.fec0 = fec0
        if fec1 is not None: p.fec1 = fec1
        if check is not None: p.check = check
        if self.L.flexframegen_setprops(self.h, C.byref(p)) != 0:
            raise ValueError("unsupported frame properties")

    def frame(self, payload, header=None, dt=0.0):
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        hdr = np.zeros(14, np.uint8) if header is None else np.ascontiguousarray(header, dtype=np.uint8)
        self.L.fxrx_gen_set_delay(self.h, float(dt))
        self.L.flexframegen_assemble(self.h, hdr.ctypes.data, payload.ctypes.data, len(payload))
        n = self.L.flexframegen_getframelen(self.h)
        out = np.empty(n, np.complex64)
        if self.L.flexframegen_write_samples(self.h, out.ctypes.data, n) != 1:
            raise RuntimeError("flexframegen_write_samples failed")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.flexframegen_destroy(self.h)
            self.h = None

    __del__ = close


class TxContext:
    """Batched frame generator on the GPU (fxtx_* in include/fxrx.h): many frames, one call, straight into a device
    buffer.  Samples are bit-identical to FrameGen.frame()."""

    def __init__(self, device=0):
        self.L = _ffi.lib()
        self.h = self.L.fxtx_create(int(device))
        if not self.h:
            raise RuntimeError("fxtx_create: " + self.L.fxrx_last_error().decode())

    @staticmethod
    def _desc(fr, keep):
        t = _ffi.TxFrame()
        t.props = _ffi.GenProps(fr.get("check", CRC_24), fr.get("fec0", 11), fr.get("fec1", 1), fr.get("mod", 2))
        pl = np.ascontiguousarray(fr["payload"], dtype=np.uint8); keep.append(pl)
        t.payload = pl.ctypes.data if len(pl) else None; t.payload_len = len(pl)
        if fr.get("header") is not None:
            hd = np.ascontiguousarray(fr["header"], dtype=np.uint8); assert len(hd) == 14; keep.append(hd); t.header = hd.ctypes.data
        t.dt = float(fr.get("dt", 0.0)); t.out_offset = int(fr.get("offset", 0))
        return t

    def frame_len(self, fr):
        keep = []
        t = self._desc(fr, keep)
        return int(self.L.fxtx_frame_len(C.byref(t)))

    def generate(self, frames, out_ptr, out_len):
        """frames: dicts with payload, offset and optionally mod, fec0, fec1, check, header, dt.  out_ptr: device pointer to
        out_len complex64 samples (e.g. torch_tensor.data_ptr())."""
        keep = []
        arr = (_ffi.TxFrame * len(frames))(*[self._desc(fr, keep) for fr in frames])
        r = self.L.fxtx_generate(self.h, arr, len(frames), C.c_void_p(int(out_ptr)), int(out_len))
        if r != 0:
            raise RuntimeError("fxtx_generate: " + self.L.fxrx_last_error().decode())

    def channel(self, iq_ptr, n_streams, n_per_stream, chans):
        """chans: per stream (cfo rad/sample, phase rad, gain, sigma per real dimension, seed) -- applied in place on the device."""
        arr = (_ffi.TxChannel * n_streams)(*[_ffi.TxChannel(float(c[0]), float(c[1]), float(c[2]), float(c[3]), int(c[4])) for c in chans])
        r = self.L.fxtx_apply_channel(self.h, C.c_void_p(int(iq_ptr)), int(n_streams), int(n_per_stream), arr)
        if r != 0:
            raise RuntimeError("fxtx_apply_channel: " + self.L.fxrx_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.fxtx_destroy(self.h)
            self.h = None

    __del__ = close


def synth_streams_device(n_streams, n_samples, first_stream_id=0, props=None, payload_len=1024, gap=256, snr_db=20.0, device=0, tx=None,
                         streams_per_call=64):
    """n_streams DISTINCT synthetic IQ streams of n_samples each, generated on the GPU (fxtx_generate + fxtx_channel) straight
    into one device tensor of shape (n_streams, n_samples), complex64 -- the SURVEY 8(d) workload without the host ever
    holding a stream: frames back to back with `gap` zero samples, per-stream payload bytes MT19937(0x5EED + id), per-stream
    channel draws MT19937(0xC0FFEE + id): CFO ~ U(-0.05, 0.05) rad/sample, phase ~ U(-pi, pi), delay ~ U(-0.5, 0.5) sample,
    AWGN at Es/N0 = snr_db from the device's counter-based generator (seed = id).  props(stream_id) -> dict with any of
    mod, fec0, fec1, check, payload_len, snr_db (per-stream frame properties: the mod/FEC sweep of BASELINE config 5).
    Returns (tensor, injected) with injected[s] = [(start sample, payload bytes), ...]."""
    import torch
    own = tx is None
    tx = TxContext(device) if own else tx
    out = torch.zeros((n_streams, n_samples), dtype=torch.complex64, device="cuda:%d" % device)
    torch.cuda.synchronize(device)
    injected, chans = [], []
    for s0 in range(0, n_streams, streams_per_call):
        frames = []
        for s in range(s0, min(n_streams, s0 + streams_per_call)):
            sid = first_stream_id + s
            pr = dict(mod=2, fec0=11, fec1=1, check=CRC_24, payload_len=payload_len, snr_db=snr_db)
            if props is not None:
                pr.update(props(sid))
            prng = np.random.RandomState((0x5EED + sid) & 0x7FFFFFFF)
            crng = np.random.RandomState((0xC0FFEE + sid) & 0x7FFFFFFF)
            cfo, phase, delay = crng.uniform(-0.05, 0.05), crng.uniform(-np.pi, np.pi), crng.uniform(-0.5, 0.5)
            flen = tx.frame_len(dict(payload=np.zeros(pr["payload_len"], np.uint8), mod=pr["mod"], fec0=pr["fec0"], fec1=pr["fec1"], check=pr["check"]))
            nfr = max(0, (n_samples + gap) // (flen + gap))
            pls = prng.randint(0, 256, (nfr, pr["payload_len"])).astype(np.uint8)
            mine = []
            for k in range(nfr):
                p = k * (flen + gap)
                frames.append(dict(payload=pls[k], mod=pr["mod"], fec0=pr["fec0"], fec1=pr["fec1"], check=pr["check"], dt=delay,
                                   offset=s * n_samples + p))
                mine.append((p, pls[k].tobytes()))
            injected.append(mine)
            chans.append((cfo, phase, 1.0, np.sqrt(0.5 * 10.0 ** (-pr["snr_db"] / 10.0)), sid))
        tx.generate(frames, out.data_ptr(), n_streams * n_samples)
    tx.channel(out.data_ptr(), n_streams, n_samples, chans)
    if own:
        tx.close()
    return out, injected


def synth_stream(n_samples, stream_id=0, mod=2, fec0=11, fec1=1, check=CRC_24, payload_len=1024, gap=256,
                 snr_db=20.0, cfo=None, phase=None, delay=None, gain=1.0, lead=0, return_payloads=True):
    """One synthetic IQ stream of exactly n_samples samples.

    Payload bytes: MT19937(0x5EED + stream_id); channel draws: MT19937(0xC0FFEE + stream_id):
    CFO ~ U(-0.05, 0.05) rad/sample, phase ~ U(-pi, pi), delay ~ U(-0.5, 0.5) sample, AWGN with
    sigma^2 = 10^(-snr/10) per complex sample (unit-power signal), as in BASELINE.md section 3.
    Returns (iq complex64, list of (start_index, payload bytes) of frames that fit entirely)."""
    prng = np.random.RandomState((0x5EED + stream_id) & 0x7FFFFFFF)
    crng = np.random.RandomState((0xC0FFEE + stream_id) & 0x7FFFFFFF)
    cfo = crng.uniform(-0.05, 0.05) if cfo is None else cfo
    phase = crng.uniform(-np.pi, np.pi) if phase is None else phase
    delay = crng.uniform(-0.5, 0.5) if delay is None else delay
    g = FrameGen(mod, fec0, fec1, check)
    x = np.zeros(n_samples, np.complex64)
    frames = []
    p = lead
    while True:
        pl = prng.randint(0, 256, payload_len).astype(np.uint8)
        fr = g.frame(pl, dt=delay)
        if p + len(fr) > n_samples:
            break
        x[p:p + len(fr)] = fr
        frames.append((p, pl.tobytes()))
        p += len(fr) + gap
    g.close()
    n = np.arange(n_samples, dtype=np.float64)
    rot = np.exp(1j * (cfo * n + phase)).astype(np.complex64)
    x *= rot
    if gain != 1.0:
        x *= np.float32(gain)
    sigma = np.sqrt(0.5 * 10.0 ** (-snr_db / 10.0))
    noise = crng.standard_normal(2 * n_samples).astype(np.float32).view(np.complex64)
    x += np.float32(sigma) * noise
    return (x, frames) if return_payloads else x
