#!/usr/bin/env python3
"""Rollout of B windows of 20 steps at 512x512 (B = 1: one DWP window's worth): best of 5, ms.  TEZIP_EPART=0|1 compares the
fused gate launches with the split ones (tz_prednet.hip "E-part ahead").   python scripts/b1_time.py [B [size]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
SZ = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(SZ, SZ, B)
f = synth.turbulence(21 * B, SZ, SZ)
for _ in range(3):
    ctx.rollout(f, 0, 21)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.rollout(f, 0, 21)
    best = min(best, time.perf_counter() - t0)
print("TEZIP_EPART=%s  B=%d %dx%d rollout of 20 steps: %.2f ms" % (os.environ.get("TEZIP_EPART", "default"), B, SZ, SZ, best * 1e3))
