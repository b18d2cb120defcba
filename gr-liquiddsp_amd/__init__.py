"""gr-liquiddsp_amd -- MI355X-native flexframe receive path behind the gr::liquiddsp block API.

    import importlib; fx = importlib.import_module("gr-liquiddsp_amd")

Layers (bottom up):
  csrc/          HIP kernels + host runtime -> libfxrx.so (C ABI: include/fxrx.h)
  _ffi.py        ctypes binding of that ABI
  rx.py          RxContext: batched multi-stream receive / detect (device or host IQ)
  tx.py          FrameGen + synthetic stream generator (flex_tx counterpart; test/bench signal source)
  blocks.py      flex_rx / frame_detector_cc / flex_tx with the reference's work() + message-port contract
"""
from . import _ffi
from ._ffi import build, lib, LIB_PATH
from .rx import RxContext, MODE_FLEX_RX, MODE_DETECTOR
from .tx import FrameGen, TxContext, synth_stream, synth_streams_device, MOD_BY_INDEX, INNER_BY_INDEX, OUTER_BY_INDEX, CRC_24, CRC_32
from .blocks import flex_rx, frame_detector_cc, flex_tx

__all__ = ["build", "lib", "RxContext", "FrameGen", "TxContext", "synth_stream", "synth_streams_device", "flex_rx", "frame_detector_cc", "flex_tx",
           "MODE_FLEX_RX", "MODE_DETECTOR"]
