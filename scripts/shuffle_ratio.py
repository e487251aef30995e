#!/usr/bin/env python3
"""Does the opt-in byte shuffle pay?  Compression ratio of cfg3-style data (512x512 turbulence,
20-frame windows) with and without --shuffle, for the bench's random weights and for a model trained
here (tezip_amd/train.py, the reference's schedule), lossless / `abs 2` / entropy remap off.
Same libzstd level 9 for every file.  Output: a markdown table."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tezip_amd import _lib, compress, synth, zstd  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402
from scripts.trained_model import trained_weights  # noqa: E402


def sizes(ctx, frames, mode, bound, entropy):
    nt, h, w = frames.shape[:3]
    key, _ = ctx.rollout(frames, 0, 20)
    out = {}
    for shuf in (False, True):
        payload, table, _ = ctx.encode(mode, bound, entropy, shuffle=shuf)
        _, ent = compress.pack_outputs(frames, key, payload, table if entropy else None, 0, shuf)
        out[shuf] = len(ent)
    kb, _ = compress.pack_outputs(frames, key, payload, table if entropy else None, 0, True)
    return out[False], out[True], len(kb)


def main():
    cfg = PredNetConfig()
    frames = synth.turbulence(40, 512, 512, seed=3)
    rows = []
    for name, wts in (("glorot seed 123", cfg.init_weights(seed=123)), ("trained 100 epochs", trained_weights(100))):
        ctx = _lib.Context(0)
        ctx.load_model(cfg, wts)
        ctx.prepare(512, 512, 2)
        for mode, bound, entropy in (("abs", [0.0], True), ("abs", [2.0], True), ("abs", [0.0], False)):
            plain, shuf, kb = sizes(ctx, frames, mode, bound, entropy)
            rows.append((name, "%s %g%s" % (mode, bound[0], "" if entropy else ", -n"), frames.nbytes / (plain + kb),
                         frames.nbytes / (shuf + kb), plain, shuf))
        ctx.close()
    print("| weights | mode | ratio | ratio --shuffle | entropy.dat | entropy.dat --shuffle | change |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.3f | %.3f | %d | %d | %+.1f %% |" % (r + (100.0 * (r[5] - r[4]) / r[4],)))


if __name__ == "__main__":
    main()
