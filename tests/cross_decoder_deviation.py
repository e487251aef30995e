#!/usr/bin/env python3
"""Cross-decoder deviation of the HIP predictor (SURVEY.md §7 hard part 1).
A measurement that uses the oracle package as the foreign decoder, hence kept under tests/ (run it as
`python tests/cross_decoder_deviation.py` on the GPU box; pytest does not collect it).

Lossless decoding rebuilds  recon = trunc(pred*255) - (trunc(pred_enc*255) - orig)
(decompress.py:252-253), so a decoder whose predictor sums in another order than the encoder's
reproduces the frames only where trunc(pred*255) agrees.  This script measures, per recursion
depth inside a window, how often and by how much it disagrees between

  A  the HIP path (TZ-PA1 fmaf chains, constants folded, collapsed upsample taps; bit-identical to
     oracle/tz_oracle.c), and
  B  "foreign-order" decoders: oracle/prednet_torch.py on the CPU (MKL-DNN/oneDNN summation order)
     and on the GPU (MIOpen), both evaluating the literal two-step Keras graph with the
     un-collapsed 9-tap convolution over the materialised upsampled tensor -- stand-ins for the
     reference's TensorFlow decoder, which cannot run here (no TF; parity unpinned).

Each side rolls out from the same key frame feeding on ITS OWN predictions, exactly what an
encoder/decoder pair does.  Workloads: cfg2 (128x160, 10-frame windows) and cfg3 (512x512,
20-frame windows) with (i) the bench's random glorot weights and (ii) a model trained here with
the reference's schedule (tezip_amd/train.py), since an untrained PredNet contracts to a constant
and hides the effect.  Output: gpurun_out/deviation.json + deviation.md (table for DESIGN.md §3).
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import prednet_torch  # noqa: E402
from scripts.trained_model import trained_weights  # noqa: E402
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402


def measure(cfg, wts, frames, window, foreign, max_windows):
    nt, h, w = frames.shape[:3]
    hp, wp = _lib.pad8(h), _lib.pad8(w)
    ctx = _lib.Context(0)
    ctx.load_model(cfg, wts)
    ctx.prepare(hp, wp, (nt + window - 1) // window)
    key, _ = ctx.rollout(frames, 0, window)
    hip = ctx.get_predictions()
    ctx.close()
    starts = key.nonzero()[0].tolist()[:max_windows]
    depth = window - 1
    flips = np.zeros(depth)
    maxd = np.zeros(depth, np.int64)
    maxf = np.zeros(depth)
    for s in starts:
        cur = np.zeros((hp, wp, 3), np.float32)
        cur[:h, :w] = frames[s].astype(np.float32) / np.float32(255)
        for d in range(1, depth + 1):
            if s + d >= nt:
                break
            cur = foreign.next(cur)
            a = hip[s + d]
            diff = prednet_torch.trunc255(a[:h, :w]) - prednet_torch.trunc255(cur[:h, :w])
            flips[d - 1] += float((diff != 0).mean()) / len(starts)
            maxd[d - 1] = max(maxd[d - 1], int(np.abs(diff).max()))
            maxf[d - 1] = max(maxf[d - 1], float(np.abs(a - cur).max()))
    return [{"depth": d + 1, "flip_rate": flips[d], "max_pixel_error": int(maxd[d]), "max_abs_pred_diff": maxf[d]}
            for d in range(depth)]


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    cfg = PredNetConfig()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    t0 = time.time()
    models = {"glorot seed 123 (bench weights)": cfg.init_weights(seed=123), "trained %d epochs" % epochs: trained_weights(epochs)}
    print("training done in %.1f s" % (time.time() - t0), flush=True)
    loads = {"cfg2 128x160 w=10": (synth.translating_scene(40, 128, 160, seed=2), 10),
             "cfg3 512x512 w=20": (synth.turbulence(80, 512, 512, seed=3), 20)}
    res = {}
    for mname, wts in models.items():
        for lname, (frames, window) in loads.items():
            h, w = frames.shape[1:3]
            for fname, dev, nwin in (("torch-CPU", "cpu", 1 if h >= 512 else 4), ("torch-GPU (MIOpen)", "cuda", 4)):
                t1 = time.time()
                foreign = prednet_torch.TorchPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, _lib.pad8(h), _lib.pad8(w), device=dev)
                rows = measure(cfg, wts, frames, window, foreign, nwin)
                res["%s | %s | %s" % (mname, lname, fname)] = rows
                print("%s | %s | %s: %.1f s, flip rate depth1 %.3g, last %.3g, max err %d" % (
                    mname, lname, fname, time.time() - t1, rows[0]["flip_rate"], rows[-1]["flip_rate"],
                    max(r["max_pixel_error"] for r in rows)), flush=True)
    json.dump(res, open(os.path.join(out_dir, "deviation.json"), "w"), indent=1)
    with open(os.path.join(out_dir, "deviation.md"), "w") as f:
        for k, rows in res.items():
            f.write("\n**%s**\n\n| depth | " % k + " | ".join(str(r["depth"]) for r in rows) + " |\n")
            f.write("|---|" + "---|" * len(rows) + "\n")
            f.write("| flip rate | " + " | ".join("%.2g" % r["flip_rate"] for r in rows) + " |\n")
            f.write("| max pixel error | " + " | ".join(str(r["max_pixel_error"]) for r in rows) + " |\n")
            f.write("| max abs pred diff | " + " | ".join("%.1e" % r["max_abs_pred_diff"] for r in rows) + " |\n")
    print(open(os.path.join(out_dir, "deviation.md")).read())


if __name__ == "__main__":
    main()
