#!/usr/bin/env python3
"""Device time of every codec kernel at the cfg3 size (80 x 512x512x3 = 62.9 M elements) through the
C ABI's stand-alone operators, HIP events around each launch (tz_prof_*), device-resident buffers.
The iteration tool for the HBM-bound kernels: prints GB/s of ALGORITHMIC bytes per
kernel and the fraction of the 8 TB/s HBM3E peak.

  python scripts/codec_bench.py [--reps 5] [--mode abs --bound 2]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tezip_amd import _lib  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mode", default="abs")
ap.add_argument("--bound", type=float, nargs="+", default=[2.0])
ap.add_argument("--trained", action="store_true", help="train a model first (tezip_amd/train.py) instead of glorot weights")
args = ap.parse_args()

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)   # own stream: torch.cuda.synchronize() wherever torch and the library hand over buffers
cfg = PredNetConfig()
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
nt, H, W = 80, 512, 512
frames = bench.turbulence_cuda(nt, 0, nt, H, W, 3, dev)
n = nt * H * W * 3
key, _ = ctx.rollout(frames, 0, 20)
pred = torch.from_numpy(ctx.get_predictions()).to(dev)
gfirst = key.astype(np.uint8)

delta = torch.empty(n, dtype=torch.int16, device=dev)
quant = torch.empty(n, dtype=torch.int16, device=dev)
sd = torch.empty(n, dtype=torch.int16, device=dev)
pay = torch.empty(n, dtype=torch.int16, device=dev)
back = torch.empty(n, dtype=torch.int16, device=dev)
und = torch.empty(n, dtype=torch.int16, device=dev)
out = torch.empty(n, dtype=torch.uint8, device=dev)
keys = torch.zeros_like(frames)
kidx = torch.from_numpy(key).to(dev)
keys[kidx] = frames[kidx]
torch.cuda.synchronize()


def once():
    ctx.delta_encode(pred, frames, gfirst, out=delta)
    quant.copy_(delta)
    ctx.synchronize()
    ctx.error_bound(frames, quant, args.mode, args.bound, gfirst)
    ctx.spatial_delta(quant, 0, out=sd)                      # no histogram
    hist = np.zeros(_lib.TZ_NBINS, np.uint64)
    ctx.spatial_delta(quant, 1, hist=hist, out=sd)           # + offset + histogram
    table = ctx.build_table(hist)
    ctx.remap(sd, table, out=pay)
    ctx.unmap(pay, table, offset=True, out=back)
    ctx.spatial_undelta(back, out=und)
    ctx.reconstruct(pred, keys, gfirst, und.view(nt, H, W, 3), out=out)
    return table


table = once()
ctx.synchronize()
assert torch.equal(und, quant), "codec round trip broken"
res = {}
ctx.prof_enable(True)
names = ["delta", "quant", "spatial_delta_hist", "lut_remap", "undelta_scan", "reconstruct"]
acc = {k: [] for k in names + ["sdelta_nohist", "sdelta_hist"]}
for _ in range(args.reps):
    ctx.prof_reset()
    ctx.delta_encode(pred, frames, gfirst, out=delta)
    p = ctx.prof_get()
    acc["delta"].append(p["delta"][0])
    quant.copy_(delta)
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.error_bound(frames, quant, args.mode, args.bound, gfirst)
    acc["quant"].append(ctx.prof_get()["quant"][0])
    ctx.prof_reset()
    ctx.spatial_delta(quant, 0, out=sd)
    acc["sdelta_nohist"].append(ctx.prof_get()["spatial_delta_hist"][0])
    ctx.prof_reset()
    hist = np.zeros(_lib.TZ_NBINS, np.uint64)
    ctx.spatial_delta(quant, 1, hist=hist, out=sd)
    acc["sdelta_hist"].append(ctx.prof_get()["spatial_delta_hist"][0])
    ctx.prof_reset()
    ctx.remap(sd, table, out=pay)
    ctx.unmap(pay, table, offset=True, out=back)
    acc["lut_remap"].append(ctx.prof_get()["lut_remap"][0] / 2)
    ctx.prof_reset()
    ctx.spatial_undelta(back, out=und)
    acc["undelta_scan"].append(ctx.prof_get()["undelta_scan"][0])
    ctx.prof_reset()
    ctx.reconstruct(pred, keys, gfirst, und.view(nt, H, W, 3), out=out)
    acc["reconstruct"].append(ctx.prof_get()["reconstruct"][0])
ctx.prof_enable(False)
BYTES = {"delta": 7, "quant": 5, "sdelta_nohist": 4, "sdelta_hist": 4, "lut_remap": 4, "undelta_scan": 4, "reconstruct": 7}
for k, b in BYTES.items():
    ms = float(np.median(acc[k]))
    res[k] = {"ms": round(ms, 4), "GBps_algorithmic": round(b * n / (ms * 1e-3) / 1e9, 1),
              "frac_of_8TBps": round(b * n / (ms * 1e-3) / 8e12, 3), "bytes_per_element": b}
res["table_symbols"] = int(len(table))
res["mode"] = "%s %s" % (args.mode, args.bound)
print(json.dumps(res, indent=1))
ctx.close()
