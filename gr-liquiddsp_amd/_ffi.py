"""ctypes binding of libfxrx.so (C ABI in include/fxrx.h).

The library is the product: if it is missing or no HIP device is usable, constructors raise.
There is no Python/CPU implementation of the receive path in this package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FXRX_LIB", os.path.join(CSRC, "libfxrx.so"))   # FXRX_LIB: A/B builds during tuning


class FxComplex(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class FrameSyncStats(C.Structure):          # framesyncstats_s
    _fields_ = [("evm", C.c_float), ("rssi", C.c_float), ("cfo", C.c_float),
                ("framesyms", C.POINTER(FxComplex)), ("num_framesyms", C.c_uint),
                ("mod_scheme", C.c_uint), ("mod_bps", C.c_uint), ("check", C.c_uint),
                ("fec0", C.c_uint), ("fec1", C.c_uint)]


FRAMESYNC_CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_ubyte), C.c_uint,
                                 C.c_int, FrameSyncStats, C.c_void_p)


class GenProps(C.Structure):                # flexframegenprops_s
    _fields_ = [("check", C.c_uint), ("fec0", C.c_uint), ("fec1", C.c_uint), ("mod_scheme", C.c_uint)]


class TxFrame(C.Structure):                 # fxtx_frame
    _fields_ = [("props", GenProps), ("header", C.c_void_p), ("payload", C.c_void_p), ("payload_len", C.c_uint),
                ("dt", C.c_float), ("out_offset", C.c_ulonglong)]


class TxChannel(C.Structure):               # fxtx_channel
    _fields_ = [("cfo", C.c_float), ("phase", C.c_float), ("gain", C.c_float), ("sigma", C.c_float), ("seed", C.c_ulonglong)]


class Config(C.Structure):                  # fxrx_config
    _fields_ = [("device", C.c_int), ("mode", C.c_int), ("n_streams", C.c_uint), ("threshold", C.c_float),
                ("segment_len", C.c_uint), ("want_framesyms", C.c_int), ("equalizer", C.c_int), ("soft_decision", C.c_int)]


class Frame(C.Structure):                   # fxrx_frame
    _fields_ = [("stream", C.c_uint), ("start", C.c_int64), ("cfo_bin", C.c_int),
                ("rxy", C.c_float), ("tau", C.c_float), ("gamma", C.c_float), ("dphi", C.c_float), ("phi", C.c_float),
                ("pfb_index", C.c_uint),
                ("pilot_dphi", C.c_float), ("pilot_phi", C.c_float), ("pilot_gain", C.c_float),
                ("header_valid", C.c_int), ("payload_valid", C.c_int),
                ("header", C.c_ubyte * 20),
                ("payload", C.POINTER(C.c_ubyte)), ("payload_len", C.c_uint),
                ("framesyms", C.POINTER(FxComplex)), ("num_framesyms", C.c_uint),
                ("evm_db", C.c_float), ("rssi_db", C.c_float), ("cfo", C.c_float), ("evm_sum", C.c_float),
                ("mod_scheme", C.c_uint), ("mod_bps", C.c_uint), ("check", C.c_uint), ("fec0", C.c_uint), ("fec1", C.c_uint),
                ("soft_bits", C.POINTER(C.c_ubyte)), ("num_soft_bits", C.c_uint)]


class Timing(C.Structure):                  # fxrx_timing
    _fields_ = [("walk_ms", C.c_double), ("paymf_ms", C.c_double), ("paypll_ms", C.c_double),
                ("paydec_ms", C.c_double), ("total_ms", C.c_double),
                ("hops", C.c_uint64), ("walk_jobs", C.c_uint64), ("repairs", C.c_uint64),
                ("frames", C.c_uint64), ("payload_symbols", C.c_uint64), ("samples", C.c_uint64), ("hops_cheap", C.c_uint64),
                ("host_submit_ms", C.c_double), ("host_walkwait_ms", C.c_double),
                ("seekverify_ms", C.c_double), ("verify_hops", C.c_uint64), ("verify_failures", C.c_uint64),
                ("host_collectwait_ms", C.c_double), ("walk_mode", C.c_uint64), ("chain_ms", C.c_double), ("replays", C.c_uint64), ("vb_blocks", C.c_uint64), ("vb_repairs", C.c_uint64), ("late_decodes", C.c_uint64), ("vb_fallbacks", C.c_uint64)]


# every symbol include/fxrx.h declares (checked by tests/test_cabi.py)
EXPORTS = [
    "flexframesync_create", "flexframesync_destroy", "flexframesync_execute", "flexframesync_reset",
    "fxrx_sync_flush", "fxrx_sync_set_block", "fxrx_sync_set_threshold", "fxrx_sync_set_equalizer", "fxrx_sync_set_soft", "fxrx_sync_pending", "fxrx_sync_errors", "fxrx_qdet_errors",
    "msequence_create", "msequence_advance", "msequence_destroy",
    "qdetector_cccf_create_linear", "qdetector_cccf_destroy", "qdetector_cccf_set_threshold",
    "qdetector_cccf_execute", "qdetector_cccf_get_tau", "qdetector_cccf_get_gamma", "qdetector_cccf_get_dphi",
    "qdetector_cccf_get_phi", "qdetector_cccf_get_buf_len",
    "flexframegenprops_init_default", "flexframegen_create", "flexframegen_destroy", "flexframegen_setprops",
    "flexframegen_assemble", "flexframegen_getframelen", "flexframegen_write_samples", "fxrx_gen_set_delay",
    "fxrx_last_error", "fxrx_version", "fxrx_device_count", "fxrx_create", "fxrx_destroy", "fxrx_reset",
    "fxrx_process", "fxrx_result", "fxrx_set_depth", "fxrx_submit", "fxrx_collect", "fxrx_debug_stamps", "fxrx_debug_chain_stamps", "fxrx_debug_walk_stamps", "fxrx_debug_walk_maxjob", "fxrx_debug_walk_jobs", "fxrx_device_framesyms", "fxrx_last_timing", "fxrx_stream", "fxrx_gen_frame_len",
    "fxrx_mod_from_index", "fxrx_mod_to_index", "fxrx_inner_from_index", "fxrx_inner_to_index",
    "fxrx_outer_from_index", "fxrx_outer_to_index",
    "fxtx_create", "fxtx_destroy", "fxtx_frame_len", "fxtx_generate",
    "fxtx_apply_channel", "fxrx_set_timing", "fxrx_debug_block_times", "fxrx_ready", "fxrx_inflight", "fxrx_debug_fail", "fxrx_pinned_alloc", "fxrx_pinned_free", "fxrx_sync_context",
]


class DropinStats(C.Structure):            # dropin_stats of csrc/blocks/dropin_feed.cpp
    _fields_ = [("seconds", C.c_double), ("frames", C.c_uint64), ("header_valid", C.c_uint64), ("payload_valid", C.c_uint64),
                ("payload_bytes", C.c_uint64), ("constellation_syms", C.c_uint64), ("packet_infos", C.c_uint64),
                ("payload_hash", C.c_uint64), ("errors", C.c_uint64), ("first_frame_seconds", C.c_double)]


_feed = None


def feed_lib():
    """libdropin_feed.so: drives the C++ flex_rx block shell over a host buffer in 256-sample flexframesync_execute calls."""
    global _feed
    if _feed is None:
        lib()
        L = C.CDLL(os.path.join(CSRC, "libdropin_feed.so"))
        L.dropin_feed.restype = C.c_int
        L.dropin_feed.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint, C.c_uint, C.POINTER(DropinStats)]
        L.dropin_feed_raw.restype = C.c_int
        L.dropin_feed_raw.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint, C.POINTER(DropinStats)]
        L.dropin_copy_ceiling.restype = C.c_double
        L.dropin_copy_ceiling.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint, C.c_uint]
        L.dropin_feed_threads.restype = C.c_int
        L.dropin_feed_threads.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_ulonglong), C.c_uint, C.c_uint, C.POINTER(DropinStats)]
        _feed = L
    return _feed


def fnv1a(chunks):
    """FNV-1a (64 bit) over the concatenation of byte strings: what dropin_feed hashes the payload_data messages with."""
    import numpy as np
    h = 14695981039346656037
    for b in chunks:
        for v in b:
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def build(force=False):
    """Compile libfxrx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-j4", "-C", CSRC, "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).  If torch
    # is going to be used in this process it must be loaded first so that libfxrx.so binds to that copy;
    # loading two runtimes leaves the second one without a device ("No HIP GPUs are available").
    if os.environ.get("FXRX_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libfxrx.so is not built (%s); run __graft_entry__.build() -- there is no fallback path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.fxrx_last_error.restype = C.c_char_p
    L.fxrx_version.restype = C.c_char_p
    L.fxrx_device_count.restype = C.c_int
    L.fxrx_create.restype = C.c_void_p; L.fxrx_create.argtypes = [C.POINTER(Config)]
    L.fxrx_destroy.argtypes = [C.c_void_p]; L.fxrx_destroy.restype = None
    L.fxrx_reset.argtypes = [C.c_void_p]; L.fxrx_reset.restype = None
    L.fxrx_process.restype = C.c_int
    L.fxrx_process.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int]
    L.fxrx_result.restype = C.c_int; L.fxrx_result.argtypes = [C.c_void_p, C.c_uint, C.POINTER(Frame)]
    L.fxrx_submit.restype = C.c_int
    L.fxrx_submit.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int]
    L.fxrx_collect.restype = C.c_int; L.fxrx_collect.argtypes = [C.c_void_p]
    L.fxrx_set_depth.restype = C.c_int; L.fxrx_set_depth.argtypes = [C.c_void_p, C.c_uint]
    L.fxrx_set_timing.restype = C.c_int; L.fxrx_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.fxrx_debug_block_times.restype = C.c_int; L.fxrx_debug_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 4)]
    L.fxrx_ready.restype = C.c_int; L.fxrx_ready.argtypes = [C.c_void_p]
    L.fxrx_inflight.restype = C.c_uint; L.fxrx_inflight.argtypes = [C.c_void_p]
    L.fxrx_debug_fail.restype = C.c_int; L.fxrx_debug_fail.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    L.fxrx_pinned_alloc.restype = C.c_void_p; L.fxrx_pinned_alloc.argtypes = [C.c_size_t]
    L.fxrx_pinned_free.restype = None; L.fxrx_pinned_free.argtypes = [C.c_void_p]
    L.fxrx_sync_context.restype = C.c_void_p; L.fxrx_sync_context.argtypes = [C.c_void_p]
    L.fxrx_debug_stamps.restype = C.c_int; L.fxrx_debug_stamps.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint32 * 8)]
    L.fxrx_debug_chain_stamps.restype = C.c_int; L.fxrx_debug_chain_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 8)]
    L.fxrx_debug_walk_stamps.restype = C.c_int; L.fxrx_debug_walk_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 4)]
    L.fxrx_debug_walk_maxjob.restype = C.c_int; L.fxrx_debug_walk_maxjob.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 8)]
    L.fxrx_debug_walk_jobs.restype = C.c_int; L.fxrx_debug_walk_jobs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    L.fxrx_device_framesyms.restype = C.c_void_p; L.fxrx_device_framesyms.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.fxrx_last_timing.restype = C.c_int; L.fxrx_last_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.fxrx_stream.restype = C.c_void_p; L.fxrx_stream.argtypes = [C.c_void_p]
    L.fxrx_gen_frame_len.restype = C.c_uint; L.fxrx_gen_frame_len.argtypes = [C.c_uint] * 5
    L.fxtx_create.restype = C.c_void_p; L.fxtx_create.argtypes = [C.c_int]
    L.fxtx_destroy.restype = None; L.fxtx_destroy.argtypes = [C.c_void_p]
    L.fxtx_frame_len.restype = C.c_uint; L.fxtx_frame_len.argtypes = [C.POINTER(TxFrame)]
    L.fxtx_generate.restype = C.c_int
    L.fxtx_generate.argtypes = [C.c_void_p, C.POINTER(TxFrame), C.c_uint, C.c_void_p, C.c_ulonglong]
    L.fxtx_apply_channel.restype = C.c_int
    L.fxtx_apply_channel.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_ulonglong, C.POINTER(TxChannel)]
    for n in ("mod", "inner", "outer"):
        f = getattr(L, "fxrx_%s_from_index" % n); f.restype = C.c_int; f.argtypes = [C.c_int]
        f = getattr(L, "fxrx_%s_to_index" % n); f.restype = C.c_int; f.argtypes = [C.c_uint]
    # drop-in names
    L.flexframesync_create.restype = C.c_void_p; L.flexframesync_create.argtypes = [FRAMESYNC_CALLBACK, C.c_void_p]
    L.flexframesync_destroy.argtypes = [C.c_void_p]; L.flexframesync_destroy.restype = None
    L.flexframesync_reset.argtypes = [C.c_void_p]; L.flexframesync_reset.restype = None
    L.flexframesync_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]; L.flexframesync_execute.restype = None
    L.fxrx_sync_flush.argtypes = [C.c_void_p]; L.fxrx_sync_flush.restype = None
    L.fxrx_sync_set_block.argtypes = [C.c_void_p, C.c_uint]; L.fxrx_sync_set_block.restype = None
    L.fxrx_sync_set_threshold.argtypes = [C.c_void_p, C.c_float]; L.fxrx_sync_set_threshold.restype = None
    L.fxrx_sync_set_equalizer.argtypes = [C.c_void_p, C.c_int]; L.fxrx_sync_set_equalizer.restype = None
    L.fxrx_sync_set_soft.argtypes = [C.c_void_p, C.c_int]; L.fxrx_sync_set_soft.restype = None
    L.fxrx_sync_pending.argtypes = [C.c_void_p]; L.fxrx_sync_pending.restype = C.c_uint
    L.fxrx_sync_errors.argtypes = [C.c_void_p]; L.fxrx_sync_errors.restype = C.c_uint
    L.fxrx_qdet_errors.argtypes = [C.c_void_p]; L.fxrx_qdet_errors.restype = C.c_uint
    L.msequence_create.restype = C.c_void_p; L.msequence_create.argtypes = [C.c_uint] * 3
    L.msequence_advance.restype = C.c_uint; L.msequence_advance.argtypes = [C.c_void_p]
    L.msequence_destroy.argtypes = [C.c_void_p]; L.msequence_destroy.restype = None
    L.qdetector_cccf_create_linear.restype = C.c_void_p
    L.qdetector_cccf_create_linear.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_float]
    L.qdetector_cccf_destroy.argtypes = [C.c_void_p]; L.qdetector_cccf_destroy.restype = None
    L.qdetector_cccf_set_threshold.argtypes = [C.c_void_p, C.c_float]; L.qdetector_cccf_set_threshold.restype = None
    L.qdetector_cccf_execute.restype = C.c_void_p; L.qdetector_cccf_execute.argtypes = [C.c_void_p, FxComplex]
    for n in ("tau", "gamma", "dphi", "phi"):
        f = getattr(L, "qdetector_cccf_get_" + n); f.restype = C.c_float; f.argtypes = [C.c_void_p]
    L.qdetector_cccf_get_buf_len.restype = C.c_uint; L.qdetector_cccf_get_buf_len.argtypes = [C.c_void_p]
    L.flexframegenprops_init_default.argtypes = [C.POINTER(GenProps)]
    L.flexframegen_create.restype = C.c_void_p; L.flexframegen_create.argtypes = [C.POINTER(GenProps)]
    L.flexframegen_destroy.argtypes = [C.c_void_p]; L.flexframegen_destroy.restype = None
    L.flexframegen_setprops.argtypes = [C.c_void_p, C.POINTER(GenProps)]
    L.flexframegen_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    L.flexframegen_getframelen.restype = C.c_uint; L.flexframegen_getframelen.argtypes = [C.c_void_p]
    L.flexframegen_write_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    L.fxrx_gen_set_delay.argtypes = [C.c_void_p, C.c_float]; L.fxrx_gen_set_delay.restype = None
    _lib = L
    return L
