#!/usr/bin/env python3
"""B = 1 rollout of 20 steps at 512x512 (one DWP window's worth): best of 5, ms.  TEZIP_EPART=0|1 compares the fused gate
launches with the split ones (scripts/gpu_r05_epart.sh)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 1)
f = synth.turbulence(21, 512, 512)
for _ in range(3):
    ctx.rollout(f, 0, 20)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.rollout(f, 0, 20)
    best = min(best, time.perf_counter() - t0)
print("TEZIP_EPART=%s  B=1 512x512 rollout of 20 steps: %.2f ms" % (os.environ.get("TEZIP_EPART", "default"), best * 1e3))
