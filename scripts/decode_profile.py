#!/usr/bin/env python3
"""Per-kernel-class device time of the DECODE path (rollout replay + unmap + scan + reconstruct)
on the cfg3 workload, device-resident buffers.  Companion of bench.py (which times compress)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tezip_amd import _lib  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)   # own stream: torch.cuda.synchronize() wherever torch and the library hand over buffers
cfg = PredNetConfig()
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
frames = bench.turbulence_cuda(80, 512, 512, 3, dev)
payload = torch.empty(80 * 512 * 512 * 3, dtype=torch.int16, device=dev)
key, _ = ctx.rollout(frames, 0, 20)
_, table, _ = ctx.encode("rel", [1e-3], True, payload=payload)
keys = torch.zeros_like(frames)
keys[torch.from_numpy(key).to(dev)] = frames[torch.from_numpy(key).to(dev)]
out = torch.empty_like(frames)
for _ in range(2):
    ctx.rollout_decode(keys, 0)
    ctx.decode(payload, table, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ctx.rollout_decode(keys, 0)
    ctx.decode(payload, table, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
ctx.prof_enable(True)
ctx.prof_reset()
ctx.rollout_decode(keys, 0)
ctx.decode(payload, table, out=out)
prof = ctx.prof_get()
n = 80 * 512 * 512 * 3
res = {"decode_frames_per_s": 80 / dt, "ms_per_sequence": dt * 1e3, "lossless_roundtrip": bool((out == frames).all()),
       "kernel_ms": {k: v[0] for k, v in prof.items() if v[1]},
       "GBps": {"lut_remap(4B/el)": 4 * n / (prof["lut_remap"][0] * 1e-3) / 1e9,
                "undelta_scan(4B/el)": 4 * n / (prof["undelta_scan"][0] * 1e-3) / 1e9,
                "reconstruct(7B/el)": 7 * n / (prof["reconstruct"][0] * 1e-3) / 1e9}}
print(json.dumps(res, indent=1))
