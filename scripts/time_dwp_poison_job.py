#!/usr/bin/env python3
"""Where does tests/test_gpu_poison.py's DWP job spend its time?  (diagnostic)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.perf_counter()
def lap(what):
    global t0
    t = time.perf_counter(); print("%-40s %.2f s" % (what, t - t0), flush=True); t0 = t
import numpy as np
from oracle import coracle
from oracle import oracle as O
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
lap("imports")
cfg = PredNetConfig(stack_sizes=(3, 16, 32))
wts = cfg.init_weights(seed=3, bias_scale=0.2)
ctx = _lib.Context(0); lap("context")
ctx.load_model(cfg, wts); lap("load_model")
for (nt, H, W, p) in [(10, 64, 96, 0), (9, 45, 61, 2)]:
    hp, wp = _lib.pad8(H), _lib.pad8(W)
    frames = synth.turbulence(nt, H, W, seed=6)
    net = coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)
    class P:
        def c0(self, a, b): return net.c0()
        def next(self, f): return net.next(np.asarray(f, np.float32))
    probe = O.rollout(frames, p, None, 1e9, P()); lap("oracle probe %dx%d" % (H, W))
    thr = float(np.median(probe["mse"]))
    ref = O.rollout(frames, p, None, thr, P()); lap("oracle ref")
    ctx.prepare(hp, wp, max_batch=4); lap("prepare")
    for rep in range(3):
        key, mse = ctx.rollout(frames, p, None, thr, want_mse=True); lap("gpu rollout")
full = PredNetConfig()
ctx.load_model(full, full.init_weights(seed=123)); lap("load full")
ctx.prepare(512, 512, max_batch=1); lap("prepare 512")
img = np.random.default_rng(3).integers(0, 256, (64, 64, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
f = np.stack([np.roll(img, 3 * t, axis=1) for t in range(14)])
_, probe = ctx.rollout(f, 0, None, 1e9, want_mse=True); lap("rollout probe 512")
for rep in range(3):
    ctx.rollout(f, 0, None, 0.3, want_mse=True); lap("rollout 512")
