#!/usr/bin/env python3
"""Soak of the DWP decision kernel (k_sse_decide: every workgroup publishes a partial, the last ticket sums): the same
80-frame 512x512 DWP rollout N times -- 79 decisions each, 192 partials per decision -- under TEZIP_POISON=255 if set by
the caller; every repetition must give the first one's key mask and window-MSE log bit for bit.
python scripts/soak_dwp.py [--reps 300]"""
import argparse
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tezip_amd import _lib  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=300)
args = ap.parse_args()
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 1)
img = np.random.default_rng(3).integers(0, 256, (64, 64, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
f = torch.from_numpy(np.stack([np.roll(img, 3 * t, axis=1) for t in range(80)])).cuda()
_, probe = ctx.rollout(f, 0, None, 1e9, want_mse=True)
thr = float(np.sort(probe[1:])[20])
first, bad, t0 = None, 0, time.perf_counter()
for rep in range(args.reps):
    key, mse = ctx.rollout(f, 0, None, thr, want_mse=True)
    d = hashlib.sha256(key.tobytes() + mse.tobytes()).hexdigest()
    if not np.isfinite(mse).all():
        bad += 1
    if first is None:
        first = d
        print("keys", int(key.sum()), "threshold", thr)
    elif d != first:
        bad += 1
print("%d DWP rollouts (%d decisions) in %.1f s, TEZIP_POISON=%s: %d differ from the first" % (
    args.reps, args.reps * 79, time.perf_counter() - t0, os.environ.get("TEZIP_POISON", "unset"), bad))
sys.exit(1 if bad else 0)
