"""Batched receive context: many independent IQ streams per call (BASELINE configs 2-5).

The reference has one flexframesync handle per flex_rx block (lib/flex_rx_impl.cc:49) fed 256 samples
at a time (:212-215); here one context owns the carried state of every stream and each process() call
runs whole blocks of all streams through the HIP kernels.
"""
import ctypes as C
import numpy as np
from . import _ffi

MODE_FLEX_RX, MODE_DETECTOR = 0, 1


class RxError(RuntimeError):
    pass


def _frame_to_dict(f, copy_syms):
    d = dict(stream=f.stream, start=f.start, cfo_bin=f.cfo_bin, rxy=f.rxy, tau=f.tau, gamma=f.gamma, dphi=f.dphi,
             phi=f.phi, pfb_index=f.pfb_index, pilot_dphi=f.pilot_dphi, pilot_phi=f.pilot_phi,
             pilot_gain=f.pilot_gain, header_valid=f.header_valid, payload_valid=f.payload_valid,
             header=bytes(f.header), evm_db=f.evm_db, rssi_db=f.rssi_db, cfo=f.cfo, evm_sum=f.evm_sum,
             mod_scheme=f.mod_scheme, mod_bps=f.mod_bps, check=f.check, fec0=f.fec0, fec1=f.fec1,
             num_framesyms=f.num_framesyms)
    d["payload"] = C.string_at(f.payload, f.payload_len) if (f.payload and f.payload_len) else b""
    if copy_syms and f.framesyms and f.num_framesyms:
        buf = C.cast(f.framesyms, C.POINTER(C.c_float * (2 * f.num_framesyms))).contents
        d["framesyms"] = np.frombuffer(buf, dtype=np.complex64).copy()
    else:
        d["framesyms"] = None
    d["soft_bits"] = np.frombuffer(C.string_at(f.soft_bits, f.num_soft_bits), np.uint8) if (copy_syms and f.soft_bits and f.num_soft_bits) else None
    return d


class RxContext:
    """fxrx_ctx wrapper.  Raises RxError when the library or a HIP device is missing (no fallback)."""

    def __init__(self, n_streams=1, mode=MODE_FLEX_RX, device=0, threshold=0.0, segment_len=0, want_framesyms=False, equalizer=False, soft_decision=False):
        self.L = _ffi.lib()
        cfg = _ffi.Config(device, mode, n_streams, threshold, segment_len, 1 if want_framesyms else 0, 1 if equalizer else 0, 1 if soft_decision else 0)
        self.h = self.L.fxrx_create(C.byref(cfg))
        if not self.h:
            raise RxError("fxrx_create failed: %s" % self.L.fxrx_last_error().decode())
        self.n_streams, self.mode, self.want_framesyms = n_streams, mode, want_framesyms
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.fxrx_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        self.L.fxrx_reset(self.h)

    def process_raw(self, ptrs, counts, on_device):
        """ptrs/counts: one address and sample count per stream.  Returns number of results."""
        n = self.n_streams
        a = (C.c_void_p * n)(*ptrs)
        c = (C.c_uint64 * n)(*counts)
        r = self.L.fxrx_process(self.h, a, c, 1 if on_device else 0)
        if r < 0:
            raise RxError("fxrx_process failed (%d): %s" % (r, self.L.fxrx_last_error().decode()))
        return r

    def set_depth(self, depth):
        """Allow up to `depth` blocks in flight (submit/collect pipelining over three HIP streams)."""
        if self.L.fxrx_set_depth(self.h, depth) != 0:
            raise RxError("fxrx_set_depth failed: %s" % self.L.fxrx_last_error().decode())

    def set_timing(self, level):
        """Stage events per block: 2 all stages, 1 the PLL only, 0 none, -1 automatic (see include/fxrx.h: fxrx_set_timing)."""
        if self.L.fxrx_set_timing(self.h, int(level)) != 0:
            raise RxError("fxrx_set_timing: bad level")

    def submit_raw(self, ptrs, counts, on_device):
        n = self.n_streams
        a = (C.c_void_p * n)(*ptrs)
        c = (C.c_uint64 * n)(*counts)
        r = self.L.fxrx_submit(self.h, a, c, 1 if on_device else 0)
        if r < 0:
            raise RxError("fxrx_submit failed (%d): %s" % (r, self.L.fxrx_last_error().decode()))

    def collect_raw(self):
        r = self.L.fxrx_collect(self.h)
        if r < 0:
            raise RxError("fxrx_collect failed (%d): %s" % (r, self.L.fxrx_last_error().decode()))
        return r

    def process(self, streams):
        """streams: list (len n_streams) of numpy complex64 arrays (host) or torch complex64 CUDA tensors."""
        if len(streams) != self.n_streams:
            raise ValueError("expected %d streams" % self.n_streams)
        on_device = hasattr(streams[0], "data_ptr")
        if on_device:
            keep = [s.contiguous() for s in streams]
            ptrs = [s.data_ptr() for s in keep]
            counts = [s.numel() for s in keep]
            import torch
            torch.cuda.synchronize()        # inputs produced on torch's stream must be complete
        else:
            keep = [np.ascontiguousarray(s, dtype=np.complex64) for s in streams]
            ptrs = [s.ctypes.data for s in keep]
            counts = [len(s) for s in keep]
        self._keep = keep
        return self.results(self.process_raw(ptrs, counts, on_device))

    def results(self, n):
        out = []
        f = _ffi.Frame()
        for i in range(n):
            self.L.fxrx_result(self.h, i, C.byref(f))
            out.append(_frame_to_dict(f, self.want_framesyms))
        return out

    def timing(self):
        t = _ffi.Timing()
        self.L.fxrx_last_timing(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in _ffi.Timing._fields_}

    def stream_handle(self):
        return self.L.fxrx_stream(self.h)
