#!/usr/bin/env python3
"""Soak run for the LDS-DMA convolution kernels: many rollouts with different data, batch sizes
and frame sizes, each compared bit for bit with the general kernel (tz_set_conv_impl).  A DMA /
barrier race would show up as rare differing tiles; the protocol is race-free by construction
(tz_conv_kernels.hip.h), this is the empirical side.  One process, one GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123, bias_scale=0.1))
t0 = time.time()
runs = bad = 0
MULT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# (the small and odd sizes run k_convlat -- register weight ring, asm LDS-DMA -- and its split gate launches)
for (h, w, batch, reps) in ((512, 512, 4, 24), (376, 1248, 2, 8), (128, 160, 4, 16), (1024, 1024, 2, 4), (64, 64, 2, 16),
                            (40, 56, 2, 8), (61, 90, 3, 8), (24, 16, 1, 8), (96, 72, 5, 8), (200, 120, 1, 8), (512, 512, 1, 6)):
    ctx.prepare((h + 7) // 8 * 8, (w + 7) // 8 * 8, max_batch=batch)
    for rep in range(reps * MULT):
        nt = 3 * batch
        frames = synth.turbulence(nt, h, w, seed=1000 + runs)
        # both arithmetic contracts: TZ-PA1 (k_conv16 / k_conv16b / k_convlat against k_conv3x3) and TZ-PA2 (k_wino --
        # asm MFMAs in AGPRs, DMA ring with counted waits -- against the plain k_wino_ref), alternating over the repetitions
        pa = 1 + (rep & 1)
        ctx.set_contract(pa)
        out = []
        for impl, lat in ((1, None), (0, None), (1, None), (1, "always")):
            ctx.set_conv_impl(impl, lat=lat)
            ctx.rollout(frames, 0, 3)
            out.append(ctx.get_predictions())
        ctx.set_conv_impl(1)
        ok = all(np.array_equal(out[0], o) for o in out[1:])
        runs += 1
        bad += not ok
        if not ok:
            d = np.argwhere(out[0] != out[1])
            print("MISMATCH", h, w, batch, rep, "TZ-PA%d" % pa, len(d), d[:4].tolist(), flush=True)
ctx.set_contract(0)
print("soak: %d rollouts (both contracts), %d mismatching, %.1f s" % (runs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
