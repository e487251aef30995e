#!/usr/bin/env python3
"""Runs the five BASELINE.json configurations through the HIP path on ONE MI355X (the 8-GPU
configs are run as their per-GPU share) and prints a table: frames/s for compress
(rollout + encode) and decompress (rollout replay + decode), checks the round trip.
Not the bench line (bench.py is): a parity/throughput sweep for DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402


def run(name, frames, p, window, thr, mode, bound, max_batch, repeat=3):
    nt, h, w = frames.shape[:3]
    cfg = PredNetConfig()
    ctx = _lib.Context(0)
    ctx.load_model(cfg, cfg.init_weights(seed=123))
    ctx.prepare(_lib.pad8(h), _lib.pad8(w), max_batch)
    enc = dec = 1e9
    parts = None
    for _ in range(repeat):
        t0 = time.perf_counter()
        key, _ = ctx.rollout(frames, p, window, thr)
        t1 = time.perf_counter()
        payload, table, _ = ctx.encode(mode, bound, True)
        t2 = time.perf_counter()
        if t2 - t0 < enc:
            enc, parts = t2 - t0, [t1 - t0, t2 - t1]
    key_stack = np.zeros_like(frames)
    key_stack[key] = frames[key]
    for _ in range(repeat):
        t0 = time.perf_counter()
        ctx.rollout_decode(key_stack, p)
        t1 = time.perf_counter()
        out = ctx.decode(payload, table)
        t2 = time.perf_counter()
        if t2 - t0 < dec:
            dec, dparts = t2 - t0, [t1 - t0, t2 - t1]
    err = int(np.abs(out.astype(np.int16) - frames.astype(np.int16)).max())
    ctx.close()
    row = dict(config=name, frames=nt, size="%dx%d" % (h, w), keys=int(key.sum()), compress_fps=nt / enc,
               decompress_fps=nt / dec, max_abs_err=err, table=len(table),
               ms=[round(1e3 * v, 2) for v in parts + dparts])  # rollout, encode, rollout_decode, decode
    print(json.dumps(row), flush=True)
    return row


def main():
    rows = []
    rows.append(run("cfg1 64x64 moving blobs, w=20, lossless", synth.moving_blobs(40, 64, 64), 0, 20, None, "abs", [0.0], 2))
    rows.append(run("cfg2 128x160 KITTI-like, w=10, lossless", synth.translating_scene(40, 128, 160), 0, 10, None, "abs", [0.0], 4))
    f3 = synth.turbulence(80, 512, 512)
    rows.append(run("cfg3 512x512 turbulence, w=20, rel 1e-3", f3, 0, 20, None, "rel", [1e-3], 4))
    rows.append(run("cfg3b 512x512 turbulence, w=20, abs 2", f3, 0, 20, None, "abs", [2.0], 4))
    rows.append(run("cfg4 1024x1024 detector, w=40, abs 2 (one GPU share: 2 windows)", synth.detector(80, 1024, 1024), 0, 40,
                    None, "abs", [2.0], 2, repeat=2))
    # cfg5: DWP with a threshold inside the observed MSE range + an SWP sweep point
    ctx = _lib.Context(0)
    cfg = PredNetConfig()
    ctx.load_model(cfg, cfg.init_weights(seed=123))
    ctx.prepare(512, 512, 1)
    _, mse = ctx.rollout(f3[:40], 0, None, 1e9, want_mse=True)
    ctx.close()
    thr = float(np.sort(mse[1:])[10])
    rows.append(run("cfg5 512x512 DWP -t %.4g, lossless" % thr, f3, 0, None, thr, "abs", [0.0], 1, repeat=2))
    rows.append(run("cfg5 512x512 SWP w=5 (sweep point), lossless", f3, 0, 5, None, "abs", [0.0], 16, repeat=2))
    print("| config | frames | keys | compress frames/s | decompress frames/s | max abs err |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %d | %.0f | %.0f | %d |" % (r["config"], r["frames"], r["keys"], r["compress_fps"],
                                                  r["decompress_fps"], r["max_abs_err"]))


if __name__ == "__main__":
    main()
