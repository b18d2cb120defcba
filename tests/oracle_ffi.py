"""ctypes binding of the CPU oracle (oracle/libfxref.so).  Test infrastructure only.

The oracle is the checker: nothing under gr-liquiddsp_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libfxref.so")

# numeric enums (oracle/fxref.h)
CRC_NONE, CRC_CHECKSUM, CRC_8, CRC_16, CRC_24, CRC_32 = 1, 2, 3, 4, 5, 6
FEC_NONE, FEC_HAMMING84, FEC_SECDED7264 = 1, 5, 10
FEC_CONV_V27, FEC_CONV_V27P23, FEC_CONV_V27P34, FEC_CONV_V27P45 = 11, 15, 16, 17
FEC_CONV_V27P56, FEC_CONV_V27P67, FEC_CONV_V27P78 = 18, 19, 20
MODEM = dict(PSK2=1, PSK4=2, PSK8=3, PSK16=4, DPSK2=9, DPSK4=10, DPSK8=11, ASK4=18,
             QAM16=27, QAM32=28, QAM64=29, QPSK=40)
# block-API index -> liquid enum (reference: lib/flex_tx_impl.cc:75-181)
MOD_BY_INDEX = [1, 2, 3, 4, 9, 10, 11, 18, 27, 28, 29]
INNER_BY_INDEX = [1, 11, 15, 17, 18, 19, 20]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _ORACLE_DIR], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Stats(C.Structure):
    _fields_ = [("evm", C.c_float), ("rssi", C.c_float), ("cfo", C.c_float),
                ("framesyms", C.c_void_p), ("num_framesyms", C.c_uint),
                ("mod_scheme", C.c_uint), ("mod_bps", C.c_uint), ("check", C.c_uint),
                ("fec0", C.c_uint), ("fec1", C.c_uint)]


class FrameInfo(C.Structure):
    _fields_ = [("start", C.c_int64), ("offset", C.c_int), ("rxy", C.c_float), ("tau", C.c_float),
                ("gamma", C.c_float), ("dphi", C.c_float), ("phi", C.c_float),
                ("pfb_index", C.c_uint), ("mf_counter0", C.c_int),
                ("pilot_dphi", C.c_float), ("pilot_phi", C.c_float), ("pilot_gain", C.c_float),
                ("evm_sum", C.c_float)]


class Detection(C.Structure):
    _fields_ = [("pos", C.c_int64), ("tau", C.c_float), ("gamma", C.c_float), ("dphi", C.c_float),
                ("phi", C.c_float), ("rxy", C.c_float), ("offset", C.c_int)]


class GenProps(C.Structure):
    _fields_ = [("check", C.c_int), ("fec0", C.c_int), ("fec1", C.c_int), ("mod_scheme", C.c_int)]


CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_ubyte), C.c_uint,
                       C.c_int, Stats, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.fxr_init.restype = None
        L.fxr_rad2u32.restype = C.c_uint32; L.fxr_rad2u32.argtypes = [C.c_float]
        L.fxr_sincos_u32.argtypes = [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.fxr_atan2.restype = C.c_float; L.fxr_atan2.argtypes = [C.c_float, C.c_float]
        for f in ("fxr_fft512", "fxr_ifft512"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        for f, t in (("fxr_mf_proto", C.c_float), ("fxr_tx_taps", C.c_float), ("fxr_preamble_pn", C.c_float),
                     ("fxr_template", C.c_float), ("fxr_template_fft", C.c_float), ("fxr_pilots", C.c_float),
                     ("fxr_twiddle512", C.c_float), ("fxr_sincos_table", C.c_float)):
            getattr(L, f).restype = C.POINTER(t)
        L.fxr_template_energy.restype = C.c_float
        L.fxr_firdes_arkaiser.argtypes = [C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_void_p]
        L.fxr_crc_key.restype = C.c_uint32; L.fxr_crc_key.argtypes = [C.c_int, C.c_void_p, C.c_uint]
        L.fxr_scramble.argtypes = [C.c_void_p, C.c_uint]
        L.fxr_interleave.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        L.fxr_fec_enc_len.restype = C.c_uint; L.fxr_fec_enc_len.argtypes = [C.c_int, C.c_uint]
        L.fxr_fec_encode.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        L.fxr_fec_decode.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
        L.fxr_packet_enc_len.restype = C.c_uint; L.fxr_packet_enc_len.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int]
        L.fxr_packet_encode.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fxr_packet_decode.restype = C.c_int
        L.fxr_packet_decode.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fxr_modem_bps.restype = C.c_uint; L.fxr_modem_bps.argtypes = [C.c_int]
        L.fxr_qpm_sym_len.restype = C.c_uint; L.fxr_qpm_sym_len.argtypes = [C.c_uint] + [C.c_int] * 4
        L.fxr_gen_frame_len.restype = C.c_uint; L.fxr_gen_frame_len.argtypes = [C.POINTER(GenProps), C.c_uint]
        L.fxr_gen_frame.restype = C.c_uint
        L.fxr_gen_frame.argtypes = [C.POINTER(GenProps), C.c_void_p, C.c_void_p, C.c_uint, C.c_float, C.c_void_p]
        L.fxr_qdet_create_flexframe.restype = C.c_void_p
        L.fxr_qdet_destroy.argtypes = [C.c_void_p]
        L.fxr_qdet_reset.argtypes = [C.c_void_p]
        L.fxr_qdet_set_threshold.argtypes = [C.c_void_p, C.c_float]
        L.fxr_qdet_run.restype = C.c_uint
        L.fxr_qdet_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, C.c_uint]
        for f in ("tau", "gamma", "dphi", "phi", "rxy"):
            g = getattr(L, "fxr_qdet_" + f); g.restype = C.c_float; g.argtypes = [C.c_void_p]
        L.fxr_qdet_offset.restype = C.c_int; L.fxr_qdet_offset.argtypes = [C.c_void_p]
        L.fxr_qdet_num_hops.restype = C.c_uint64; L.fxr_qdet_num_hops.argtypes = [C.c_void_p]
        L.fxr_sync_create.restype = C.c_void_p; L.fxr_sync_create.argtypes = [CALLBACK, C.c_void_p]
        L.fxr_sync_destroy.argtypes = [C.c_void_p]
        L.fxr_sync_reset.argtypes = [C.c_void_p]
        L.fxr_sync_set_threshold.argtypes = [C.c_void_p, C.c_float]
        L.fxr_sync_set_equalizer.argtypes = [C.c_void_p, C.c_int]
        L.fxr_sync_set_soft.argtypes = [C.c_void_p, C.c_int]
        L.fxr_sync_last_soft.restype = C.c_void_p; L.fxr_sync_last_soft.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        L.fxr_modem_demod_soft.argtypes = [C.c_int, C.c_uint64, C.c_uint, C.c_void_p]
        L.fxr_eq_init_taps.argtypes = [C.c_void_p]
        L.fxr_sync_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.fxr_sync_execute_chunked.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint]
        L.fxr_sync_last_frame.argtypes = [C.c_void_p, C.POINTER(FrameInfo)]
        L.fxr_init()
        _lib = L
    return _lib


def table(name, n, complex_=True):
    p = getattr(lib(), name)()
    a = np.ctypeslib.as_array(p, shape=(n * (2 if complex_ else 1),)).copy()
    return a.view(np.complex64) if complex_ else a


def fft512(x, inverse=False):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty(512, np.complex64)
    (lib().fxr_ifft512 if inverse else lib().fxr_fft512)(x.ctypes.data, out.ctypes.data)
    return out


def gen_frame(payload, mod=2, fec0=FEC_CONV_V27, fec1=FEC_NONE, check=CRC_24, header=None, dt=0.0):
    """One flexframe as complex64 samples (oracle TX)."""
    L = lib()
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    hdr = np.zeros(14, np.uint8) if header is None else np.ascontiguousarray(header, dtype=np.uint8)
    p = GenProps(check, fec0, fec1, mod)
    n = L.fxr_gen_frame_len(C.byref(p), len(payload))
    out = np.empty(n, np.complex64)
    w = L.fxr_gen_frame(C.byref(p), hdr.ctypes.data, payload.ctypes.data, len(payload), C.c_float(dt), out.ctypes.data)
    assert w == n
    return out


class Frame:
    __slots__ = ("header", "header_valid", "payload", "payload_valid", "evm", "rssi", "cfo", "framesyms",
                 "mod_scheme", "mod_bps", "check", "fec0", "fec1", "info", "soft")

    def __repr__(self):
        return "Frame(hv=%d pv=%d len=%d mod=%d start=%d)" % (
            self.header_valid, self.payload_valid, len(self.payload), self.mod_scheme, self.info["start"])


class Sync:
    """Oracle flexframesync driven like the reference block (256-sample execute calls)."""

    def __init__(self, threshold=None, equalizer=False, soft=False):
        self.L = lib()
        self.frames = []
        self._cb = CALLBACK(self._on_frame)
        self.q = self.L.fxr_sync_create(self._cb, None)
        if threshold is not None:
            self.L.fxr_sync_set_threshold(self.q, threshold)
        if equalizer:
            self.L.fxr_sync_set_equalizer(self.q, 1)
        self.soft = bool(soft)
        if soft:
            self.L.fxr_sync_set_soft(self.q, 1)

    def _on_frame(self, header, hv, payload, plen, pv, st, ud):
        f = Frame()
        f.header = bytes(bytearray(header[i] for i in range(14)))
        f.header_valid, f.payload_valid = int(hv), int(pv)
        f.payload = bytes(bytearray(payload[i] for i in range(plen))) if plen else b""
        f.evm, f.rssi, f.cfo = st.evm, st.rssi, st.cfo
        if st.num_framesyms and st.framesyms:
            buf = (C.c_float * (2 * st.num_framesyms)).from_address(st.framesyms)
            f.framesyms = np.frombuffer(buf, dtype=np.complex64).copy()
        else:
            f.framesyms = np.zeros(0, np.complex64)
        f.mod_scheme, f.mod_bps, f.check, f.fec0, f.fec1 = st.mod_scheme, st.mod_bps, st.check, st.fec0, st.fec1
        fi = FrameInfo()
        self.L.fxr_sync_last_frame(self.q, C.byref(fi))
        f.info = {k: getattr(fi, k) for k, _ in FrameInfo._fields_}
        f.soft = None
        if self.soft and hv:
            n = C.c_uint(0)
            p = self.L.fxr_sync_last_soft(self.q, C.byref(n))
            f.soft = np.frombuffer((C.c_ubyte * n.value).from_address(p), np.uint8).copy() if (p and n.value) else np.zeros(0, np.uint8)
        self.frames.append(f)
        return 0

    def execute(self, x, chunk=256):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n = len(x)
        if chunk is None:
            self.L.fxr_sync_execute(self.q, x.ctypes.data, n)
        else:
            self.L.fxr_sync_execute_chunked(self.q, x.ctypes.data, n, chunk)      # the 256-sample loop, in C like the reference's
        return self.frames

    def close(self):
        if self.q:
            self.L.fxr_sync_destroy(self.q)
            self.q = None

    def __del__(self):
        self.close()


class Detector:
    """Oracle qdetector_cccf, per-sample like lib/frame_detector_cc_impl.cc:76-83."""

    def __init__(self, threshold=0.45):
        self.L = lib()
        self.q = self.L.fxr_qdet_create_flexframe()
        self.L.fxr_qdet_set_threshold(self.q, threshold)
        self.consumed = 0

    def run(self, x, max_det=65536):
        """Feeds x sample by sample; returns detections as dicts (pos = absolute index of aligned sample 0)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = (Detection * max_det)()
        n = self.L.fxr_qdet_run(self.q, x.ctypes.data, len(x), self.consumed, out, max_det)
        self.consumed += len(x)
        assert n <= max_det
        return [{k: getattr(out[i], k) for k, _ in Detection._fields_} for i in range(n)]

    @property
    def hops(self):
        return self.L.fxr_qdet_num_hops(self.q)

    def close(self):
        if self.q:
            self.L.fxr_qdet_destroy(self.q)
            self.q = None

    def __del__(self):
        self.close()
