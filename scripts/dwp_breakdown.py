#!/usr/bin/env python3
"""cfg5 (512x512 DWP, lossless): where the wall time of rollout + encode goes.  Device time by kernel
class (HIP events around every launch) against the host wall clock of each call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

f = synth.turbulence(80, 512, 512)
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 1)
_, mse = ctx.rollout(f[:40], 0, None, 1e9, want_mse=True)
thr = float(np.sort(mse[1:])[10])
pin = _lib.pinned_copy(f) if hasattr(_lib, "pinned_copy") else f
for name, frames in (("pageable", f), ("pinned", pin)):
    for prof in (False, True):
        ctx.prof_enable(prof)
        best = None
        for _ in range(3):
            ctx.prof_reset()
            t0 = time.perf_counter()
            key, _ = ctx.rollout(frames, 0, None, thr)
            t1 = time.perf_counter()
            payload, table, _ = ctx.encode("abs", [0.0], True)
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1, ctx.prof_get() if prof else None)
        print("%s prof=%d: total %.2f ms (%.0f frames/s), rollout %.2f ms, encode %.2f ms, keys %d" %
              (name, prof, best[0] * 1e3, 80 / best[0], best[1] * 1e3, best[2] * 1e3, int(key.sum())))
        if prof:
            tot = 0.0
            for k, v in sorted(best[3].items(), key=lambda kv: -kv[1][0]):
                if v[1]:
                    print("    %-22s %8.3f ms  %5d launches" % (k, v[0], v[1]))
            print("    (sub-classes of conv3x3_mfma are listed too; device time outside the classes is not counted)")
ctx.close()
