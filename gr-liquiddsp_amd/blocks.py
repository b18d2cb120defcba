"""The reference's three blocks with their work()/message contract, hosted on libfxrx.so.

GNU Radio 3.7, pmt and SWIG are absent from this image, so the blocks are plain Python classes that keep
the *shape* of the reference API: same names, same make()/constructor arguments, work() taking
(noutput_items, input_items, output_items) and returning items consumed, and the three message ports of
flex_rx publishing the same payloads (a `message_port_pub(port, msg)` hook receives them; by default
they are appended to `self.messages[port]`).  PMT values map to Python as:
  c32vector -> numpy complex64, u8vector -> bytes, dict of from_long -> dict of int, (PMT_NIL . v) -> (None, v).
"""
import ctypes as C
import numpy as np
from . import _ffi
from .rx import RxContext, MODE_FLEX_RX, MODE_DETECTOR
from .tx import FrameGen, MOD_BY_INDEX, INNER_BY_INDEX, OUTER_BY_INDEX, CRC_24


class _Block:
    def __init__(self, name):
        self.name = name
        self.messages = {}
        self._ports = []

    def message_port_register_out(self, port):
        self._ports.append(port)
        self.messages[port] = []

    def message_port_pub(self, port, msg):
        self.messages[port].append(msg)


class flex_rx(_Block):
    """gr::liquiddsp::flex_rx (include/liquiddsp/flex_rx.h:50, lib/flex_rx_impl.cc:43-64,203-254)."""
    d_inbuf_len = 256                                   # lib/flex_rx_impl.h:47

    def __init__(self, device=0):
        _Block.__init__(self, "flex_rx")
        self.output_multiple = self.d_inbuf_len         # set_output_multiple(256), lib/flex_rx_impl.cc:50
        self.ctx = RxContext(1, MODE_FLEX_RX, device=device, want_framesyms=True)
        for p in ("constellation", "payload_data", "packet_info"):   # lib/flex_rx_impl.cc:61-63
            self.message_port_register_out(p)
        self.L = _ffi.lib()
        self.num_frames = 0

    @classmethod
    def make(cls, *a, **k):
        return cls(*a, **k)

    def work(self, noutput_items, input_items, output_items=None):
        assert noutput_items % self.d_inbuf_len == 0    # lib/flex_rx_impl.cc:210
        x = input_items[0][:noutput_items]
        # one call for the whole block of samples instead of the 256-sample loop of lib/flex_rx_impl.cc:212-215
        for f in self.ctx.process([x]):
            self.num_frames += 1
            syms = f["framesyms"] if f["framesyms"] is not None else np.zeros(0, np.complex64)
            self.message_port_pub("constellation", (None, syms))                 # :217-221, even if invalid
            if f["header_valid"]:                                                # :223
                self.message_port_pub("payload_data", (None, f["payload"]))      # :224-229
                info = dict(header_valid=1, payload_valid=int(f["payload_valid"]),
                            modulation=self.L.fxrx_mod_to_index(f["mod_scheme"]),
                            inner_code=self.L.fxrx_inner_to_index(f["fec0"]),
                            outer_code=self.L.fxrx_outer_to_index(f["fec1"]))    # :232-247
                self.message_port_pub("packet_info", info)
        return noutput_items                                                     # :253


class frame_detector_cc(_Block):
    """gr::liquiddsp::frame_detector_cc (lib/frame_detector_cc_impl.cc:41-56,66-97): pass-through + count."""

    def __init__(self, device=0, verbose=False):
        _Block.__init__(self, "frame_detector_cc")
        self.ctx = RxContext(1, MODE_DETECTOR, device=device, threshold=0.45)    # :55
        self.d_num_frames = 0
        self.verbose = verbose
        self.detections = []

    @classmethod
    def make(cls, *a, **k):
        return cls(*a, **k)

    def work(self, noutput_items, input_items, output_items):
        x = input_items[0][:noutput_items]
        for d in self.ctx.process([x]):
            if self.verbose:
                print("Detected %d frames!" % self.d_num_frames)                 # :79
            self.d_num_frames += 1                                               # :80
            self.detections.append(d)
        output_items[0][:noutput_items] = x                                      # :82
        return noutput_items                                                     # :96


class flex_tx(_Block):
    """gr::liquiddsp::flex_tx (lib/flex_tx_impl.cc:42-65,183-209): message-only frame source."""

    def __init__(self, modulation, inner_code, outer_code):
        _Block.__init__(self, "flex_tx")
        self.message_port_register_out("pdus")
        self.gen = FrameGen(self._mod(modulation), self._inner(inner_code), self._outer(outer_code), CRC_24)  # :52
        self.d_header = np.zeros(14, np.uint8)                                   # :58-59
        self.d_num_frames = 0

    @classmethod
    def make(cls, modulation, inner_code, outer_code):
        return cls(modulation, inner_code, outer_code)

    @staticmethod
    def _mod(i):
        return MOD_BY_INDEX[i] if 0 <= i < 11 else MOD_BY_INDEX[0]               # default PSK2, :112-115

    @staticmethod
    def _inner(i):
        return INNER_BY_INDEX[i] if 0 <= i < 7 else 1                            # default none, :143-146

    @staticmethod
    def _outer(i):
        return OUTER_BY_INDEX[i] if 0 <= i < 8 else 1

    def set_modulation(self, m): self.gen.setprops(mod=self._mod(m))
    def set_inner_code(self, c): self.gen.setprops(fec0=self._inner(c))
    def set_outer_code(self, c): self.gen.setprops(fec1=self._outer(c))

    def configure(self, configuration):                                          # :183-189
        if "modulation" in configuration: self.set_modulation(int(configuration["modulation"]))
        if "inner_code" in configuration: self.set_inner_code(int(configuration["inner_code"]))
        if "outer_code" in configuration: self.set_outer_code(int(configuration["outer_code"]))

    def send_pkt(self, pdu):                                                     # :191-209
        meta, data = pdu
        vec = self.gen.frame(np.frombuffer(bytes(data), dtype=np.uint8), header=self.d_header)
        self.message_port_pub("pdus", (None, vec))
        self.d_num_frames += 1

    def work(self, noutput_items, input_items, output_items):                    # :211-218
        raise RuntimeError("This is not a stream block.")
