#!/usr/bin/env python3
"""Throughput of the batched GPU frame generator (fxtx_generate) on the config-2 frame mix: 1154 PSK4 r1/2 frames of 1024
bytes = 19.7 Msamples per call, generated straight into a device buffer; compared with the host generator behind
flexframegen_* on one core.  Not the headline bench (that is bench.py, the receive path)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("gr-liquiddsp_amd")
import torch

rng = np.random.default_rng(1)
tx = fx.TxContext()
frames, off = [], 0
for i in range(1154):
    fr = dict(mod=2, fec0=11, fec1=1, check=5, payload=rng.integers(0, 256, 1024, dtype=np.uint8), dt=float(rng.uniform(-0.5, 0.5)), offset=off)
    off += tx.frame_len(fr) + 256
    frames.append(fr)
out = torch.zeros(off, dtype=torch.complex64, device="cuda")
tx.generate(frames, out.data_ptr(), out.numel())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): tx.generate(frames, out.data_ptr(), out.numel())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
g = fx.FrameGen(2, 11, 1, 5)
t0 = time.perf_counter()
for fr in frames[:200]: g.frame(fr["payload"], dt=fr["dt"])
dth = (time.perf_counter() - t0) / 200 * len(frames)
rx = fx.RxContext(1)
res = rx.results(rx.process_raw([out.data_ptr()], [out.numel()], True))
ok = sum(1 for g_, fr in zip(res, frames) if g_["payload_valid"] and g_["payload"] == fr["payload"].tobytes())
print(json.dumps({"frames": len(frames), "samples": off, "gpu_generator_ms_per_call": round(dt * 1e3, 2), "gpu_generator_msamples_per_s": round(off / dt / 1e6, 1),
                  "host_generator_one_core_ms": round(dth * 1e3, 1), "host_generator_msamples_per_s": round(off / dth / 1e6, 1),
                  "note": "the GPU call includes descriptor building, uploads and the packet encoding (on the GPU unless FXTX_HOST_ENCODE=1); clean-channel loopback check",
                  "loopback_frames_ok": ok}))
