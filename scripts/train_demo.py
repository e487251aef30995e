#!/usr/bin/env python3
"""End-to-end demonstration on one MI355X: build a training set of synthetic turbulence
sequences, train PredNet with the reference's schedule (tezip_amd/train.py), then compress a held
out 512x512 sequence with the trained and with random weights and report the compression ratios
(same libzstd, level 9; pre-zstd streams are what parity is judged on)."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, compress, synth, train, weights, zstd  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402


def ratio(cfg, wts, frames, window, mode, bound):
    nt, h, w = frames.shape[:3]
    ctx = _lib.Context(0)
    ctx.load_model(cfg, wts)
    ctx.prepare(h, w, 4)
    key, _ = ctx.rollout(frames, 0, window)
    payload, table, _ = ctx.encode(mode, bound, True)
    ctx.close()
    stream = compress.build_stream(payload, table, (1, nt, h, w, 3), 0)
    kf = np.zeros_like(frames)
    kf[key] = frames[key]
    size = len(zstd.compress_array(stream, 9)) + len(zstd.compress_array(kf, 9))
    return frames.nbytes / size, len(table)


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    tmp = tempfile.mkdtemp(prefix="tz_train_")
    data = os.path.join(tmp, "set")
    os.makedirs(data)
    seqs = [synth.turbulence(12, 128, 128, seed=100 + s) for s in range(10)]
    np.save(os.path.join(data, "X_train.npy"), np.concatenate(seqs[:9]))
    np.save(os.path.join(data, "sources_train.npy"), np.repeat(["train-%d" % s for s in range(9)], 12))
    np.save(os.path.join(data, "X_val.npy"), seqs[9])
    np.save(os.path.join(data, "sources_val.npy"), np.repeat(["val-9"], 12))
    t0 = time.time()
    hist = train.run(os.path.join(tmp, "model"), data, False, nb_epoch=epochs)
    t_train = time.time() - t0
    cfg, trained, _ = weights.load_model(os.path.join(tmp, "model"))
    test = synth.turbulence(40, 512, 512, seed=3)
    out = {"epochs": epochs, "train_seconds": t_train, "loss_first": hist[0][0], "loss_last": hist[-1][0],
           "val_best": min(h[1] for h in hist)}
    for name, wts in (("random", PredNetConfig().init_weights(123)), ("trained", trained)):
        for mode, bound in (("abs", [0.0]), ("abs", [2.0]), ("rel", [0.01])):
            r, t = ratio(cfg, wts, test, 20, mode, bound)
            out["%s %s %s" % (name, mode, bound[0])] = round(r, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
