#!/usr/bin/env python3
"""Device time of the decoder's tail (tz_decode after one tz_rollout_decode) at the cfg3 size, HIP events around the
launches (tz_prof_*), everything resident.  python scripts/decode_tail_bench.py [--reps 10]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tezip_amd import _lib  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--mode", default="abs")
ap.add_argument("--bound", type=float, nargs="+", default=[2.0])
args = ap.parse_args()

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)   # own stream: torch.cuda.synchronize() wherever torch and the library hand over buffers
cfg = PredNetConfig()
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
nt, H, W = 80, 512, 512
frames = bench.turbulence_cuda(nt, 0, nt, H, W, 3, dev)
n = nt * H * W * 3
torch.cuda.synchronize()
key, _ = ctx.rollout(frames, 0, 20)
payload = torch.empty(n, dtype=torch.int16, device=dev)
_, table, _ = ctx.encode(args.mode, args.bound, True, payload=payload)
keys = torch.zeros_like(frames)
kidx = torch.from_numpy(key).to(dev)
keys[kidx] = frames[kidx]
out = torch.empty_like(frames)
torch.cuda.synchronize()
ctx.rollout_decode(keys, 0)
ctx.decode(payload, table, out=out)
ctx.synchronize()
err = int((out.to(torch.int16) - frames.to(torch.int16)).abs().max())
ctx.prof_enable(True)
ts = []
for _ in range(args.reps):
    ctx.prof_reset()
    ctx.decode(payload, table, out=out)
    p = ctx.prof_get()
    ts.append((p["undelta_scan"][0], p["reconstruct"][0]))
ts = np.array(ts) * 1e3
print("max |decoded - frame| = %d" % err)
assert err <= max(args.bound[0], 0) or args.mode != "abs", "decode outside the bound"
print("undelta_scan us: min %.1f median %.1f   reconstruct us: min %.1f median %.1f   (G = %s)" % (
    ts[:, 0].min(), np.median(ts[:, 0]), ts[:, 1].min(), np.median(ts[:, 1]), os.environ.get("TEZIP_SCAN_G", "default")))
tot = np.median(ts.sum(1))
print("tail: %.1f us, %.2f TB/s of 7 B/element" % (tot, 7.0 * n / tot / 1e6))
