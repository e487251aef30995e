import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
frames = synth.turbulence(8, 512, 512)
for rep in range(3):
    ctx.rollout(frames, 0, 2)   # 4 windows of 2 frames: one predictor step at B=4; last conv16 launch = level-1 gates
buf = np.zeros(4096 * 8, np.uint64)
ctx.lib.tz_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
ctx.lib.tz_debug_read(ctx.h, buf.ctypes.data, buf.nbytes)
d = buf.reshape(4096, 8).astype(np.int64)
n = int(d[0, 6]); d = d[:n]
print("grid", n)
t0 = d[:, 0].min()
clk = 2.39e3  # cycles per us (approx; s_memtime at shader clock)
rel = (d[:, :6] - t0) / clk
def pr(name, x): print("%-32s mean %8.1f  p5 %8.1f  p95 %8.1f us" % (name, x.mean(), np.percentile(x, 5), np.percentile(x, 95)))
pr("start (since first WG start)", rel[:, 0])
pr("prologue (entry -> e-phase ready)", rel[:, 1] - rel[:, 0])
pr("e-phase K loop", rel[:, 2] - rel[:, 1])
pr("phase switch load", rel[:, 3] - rel[:, 2])
pr("up-phase K loop", rel[:, 4] - rel[:, 3])
pr("epilogue (incl. store drain)", rel[:, 5] - rel[:, 4])
pr("WG lifetime", rel[:, 5] - rel[:, 0])
print("kernel span %.1f us" % rel[:, 5].max())
order = np.argsort(rel[:, 0])
st = rel[order, 0]
print("round starts (every 768th WG):", [round(float(st[i]), 1) for i in range(0, n, 768)])
print("end times percentiles:", [round(float(np.percentile(rel[:, 5], q)), 1) for q in (1, 25, 50, 75, 99, 100)])
