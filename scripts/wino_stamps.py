#!/usr/bin/env python3
"""Diagnostic (build with TEZIP_DEFINES=TZW_STAMPS): where a k_wino workgroup's life goes.  Runs one cfg3 predictor step
and prints, per launch shape (ncb = column blocks), the mean microseconds between the stamps: entry -> loop start
(prologue: geometry, zero fill, first DMA round trip) -> same-resolution loop -> output transform -> upsampled loop ->
epilogue.   gpurun -- 'TEZIP_DEFINES=TZW_STAMPS python -m tezip_amd.build --force && python scripts/wino_stamps.py'"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
frames = synth.turbulence(8, 512, 512, seed=3)
ctx.rollout(frames, 0, 2)
ctx.rollout(frames, 0, 2)
ctx.synchronize()
buf = np.zeros((8, 4096, 16), np.uint64)
lib = _lib.load()
lib.tz_debug_wino_stamps.argtypes = [C.c_void_p]
assert lib.tz_debug_wino_stamps(buf.ctypes.data) == 0
names = ["prologue", "same-res loop", "output transform", "upsampled loop", "epilogue"]
for slot in range(8):
    st = buf[slot]
    ok = (st[:, 0] > 0) & (st[:, 5] > 0)
    if not ok.any():
        continue
    st = st[ok].astype(np.int64)
    d = np.diff(st[:, :6], axis=1) / 100.0   # 100 MHz -> us
    total = (st[:, 5] - st[:, 0]) / 100.0
    what = {7: "L1 gates", 6: "L2 gates", 4: "L3 gates", 2: "A1", 3: "A2"}.get(slot, "slot %d" % slot)
    extra = ""
    if (st[:, 6] > 0).all():
        extra = "; epilogue: cell-state loads landed after %.2f" % ((st[:, 6] - st[:, 4]).mean() / 100.0)
    # gap between the end of a workgroup and the start of the next one on the same CU is not visible here; the launch's
    # span over its workgroups is
    span = (st[:, 5].max() - st[:, 0].min()) / 100.0
    # the same CU's consecutive workgroups: the gap between one's last stamp and the next one's first
    hw = st[:, 7]
    cu = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    gaps = []
    for c in np.unique(cu):
        q = st[cu == c]
        q = q[np.argsort(q[:, 0])]
        gaps += list((q[1:, 0] - q[:-1, 5]) / 100.0)
    if gaps:
        extra += "; %d CUs, gap between workgroups of a CU %.2f us (min %.2f, max %.2f)" % (len(np.unique(cu)), np.mean(gaps), np.min(gaps), np.max(gaps))
    # core clock during the stage loops: cycles of s_memtime per microsecond of s_memrealtime
    for a_, b_, nm in ((1, 2, "same-res loop"), (3, 4, "upsampled loop")):
        dt = (st[:, b_] - st[:, a_]) / 100.0
        if dt.mean() > 1.0:
            extra += "; core clock in the %s %.0f MHz" % (nm, ((st[:, 8 + b_] - st[:, 8 + a_]) / dt).mean())
    print("%s: %d workgroups stamped, life %.1f us: " % (what, ok.sum(), total.mean()) +
          ", ".join("%s %.2f" % (n, v) for n, v in zip(names, d.mean(0))) + extra + "; launch span %.1f us = %.2f lives" % (span, span / total.mean()))
