#!/usr/bin/env python3
"""Soak of the split gate launches ("E-part ahead", tz_prednet.hip) over random frame sizes and batch sizes: since round 6 the
split is chosen by measurement and therefore runs at many more shapes than the one it was written for (512x512, one
window).  For every shape the predictor output of a context that never splits (TEZIP_EPART=0) and of one that always does
(TEZIP_EPART=1) must agree bit for bit, under TZ-PA2, ragged tiles and odd level sizes included.
python scripts/soak_epart.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tezip_amd import _lib  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
cfg = PredNetConfig()
w = cfg.init_weights(seed=17, bias_scale=0.1)
ctxs = {}
for mode in ("0", "1"):
    os.environ["TEZIP_EPART"] = mode      # read when the context is made
    ctxs[mode] = _lib.Context(0)
    ctxs[mode].load_model(cfg, w)
os.environ.pop("TEZIP_EPART")
rng = np.random.default_rng(args.seed)
bad, t0, split_seen = 0, time.perf_counter(), 0
for case in range(args.cases):
    hp, wp = 8 * int(rng.integers(1, 52)), 8 * int(rng.integers(1, 52))
    batch = int(rng.integers(1, 6))
    frames = rng.integers(0, 256, (batch + 1, hp, wp, 3)).astype(np.float32) / np.float32(255)
    outs, launches = {}, {}
    for mode, c in ctxs.items():
        c.prepare(hp, wp, max_batch=batch)
        c.set_contract(2)
        c.prof_enable(True)
        c.prof_reset()
        first = c.predict_next(frames)              # a full batch and a batch of one
        launches[mode] = c.prof_get()["wino_pa2"][1]
        c.prof_enable(False)
        outs[mode] = (first, c.predict_next(first[:batch]))   # ... and the recursion fed back
    same = all(np.array_equal(a, b) for a, b in zip(outs["0"], outs["1"]))
    split_seen += launches["1"] > launches["0"]
    if not same:
        bad += 1
        print("MISMATCH %dx%d batch %d (k_wino launches %d fused / %d split)" % (hp, wp, batch, launches["0"], launches["1"]), flush=True)
print("%d random shapes (%d of them with split launches) in %.1f s: %d mismatching" % (args.cases, split_seen, time.perf_counter() - t0, bad))
sys.exit(1 if bad else 0)
