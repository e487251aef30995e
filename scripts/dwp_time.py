#!/usr/bin/env python3
"""One DWP rollout of 80 frames at 512x512 with a threshold nothing reaches (79 dependent B = 1 steps + the window SSE and
decision per step), best of 5, ms.  TEZIP_DWP_SPEC=0|1, TEZIP_EPART=0|1 for A/B runs on one box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402
import torch  # noqa: E402

cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 1)
f = torch.from_numpy(synth.turbulence(80, 512, 512)).cuda()
for _ in range(2):
    ctx.rollout(f, 0, None, 1e9)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.rollout(f, 0, None, 1e9)
    best = min(best, time.perf_counter() - t0)
print("DWP_SPEC=%s EPART=%s: 80-frame DWP rollout %.2f ms = %.1f us per step" % (os.environ.get("TEZIP_DWP_SPEC", "default"),
      os.environ.get("TEZIP_EPART", "default"), best * 1e3, best * 1e6 / 79))
