#!/usr/bin/env python3
"""the "E-part ahead" rule (tz_prednet.hip) outside 512x512.  One process per TEZIP_EPART setting -- the
switch is read when the context is made -- each timing, for every (shape, batch), an SWP rollout of `batch` windows of 20
steps (best of 5) and counting the k_wino launches a step made (5 fused, 7 split at the reference's four levels).  Shapes below
256x256 pixels run under TEZIP_PA=2 (by default they take TZ-PA1, which has no k_wino launches to split).

  python scripts/epart_shapes.py mode        # every shape under the TEZIP_EPART of the environment
  python scripts/epart_shapes.py             # the table: fused (0) vs default vs forced (1)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(64, 64, 1), (64, 64, 4), (128, 160, 1), (128, 160, 4), (128, 160, 8), (192, 192, 1), (192, 192, 4),
          (256, 256, 1), (256, 256, 2), (256, 256, 3), (256, 256, 4), (256, 256, 5), (256, 256, 6), (256, 512, 1), (256, 512, 2),
          (384, 384, 1), (384, 384, 2), (384, 384, 3), (376, 1248, 1), (512, 512, 1), (512, 512, 2), (512, 512, 3), (512, 512, 4),
          (720, 1280, 1), (1024, 1024, 1)]
STEPS = 20


def mode():
    import numpy as np
    import torch
    from tezip_amd import _lib
    from tezip_amd.prednet import PredNetConfig
    cfg = PredNetConfig()
    ctx = _lib.Context(0)
    ctx.load_model(cfg, cfg.init_weights(seed=123))
    rng = np.random.default_rng(5)
    for h, w, batch in SHAPES:
        ctx.prepare(_lib.pad8(h), _lib.pad8(w), batch)
        ctx.set_contract(2)
        nt = STEPS * batch
        f = torch.from_numpy(rng.integers(0, 256, (nt, h, w, 3), dtype=np.uint8)).cuda()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.rollout(f, 0, STEPS)
        launches = ctx.prof_get()["wino_pa2"][1]
        ctx.prof_enable(False)
        for _ in range(2):
            ctx.rollout(f, 0, STEPS)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.rollout(f, 0, STEPS)
            best = min(best, time.perf_counter() - t0)
        print("CELL %d %d %d %.3f %d" % (h, w, batch, best * 1e3, launches), flush=True)
        del f
    ctx.close()


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "mode":
        return mode()
    res = {}
    for m in ("0", None, "1"):
        env = dict(os.environ)
        env.pop("TEZIP_EPART", None)
        if m is not None:
            env["TEZIP_EPART"] = m
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "mode"], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print("TEZIP_EPART=%s failed:\n%s" % (m, r.stderr[-2000:]), flush=True)
            return 1
        for ln in r.stderr.splitlines():   # TEZIP_EPART_LOG=1: what every measurement saw
            if ln.startswith("[tezip] E-part"):
                print(ln, file=sys.stderr, flush=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("CELL"):
                _, h, w, b, ms, launches = ln.split()
                res[(int(h), int(w), int(b), m)] = (float(ms), int(launches))
    print("| frame | windows | fused ms | default ms | default is | default vs fused | forced split ms | forced vs fused | k_wino launches per step fused / default / forced |")
    print("|---|---|---|---|---|---|---|---|---|")
    for h, w, b in SHAPES:
        fused, dflt, forced = res[(h, w, b, "0")], res[(h, w, b, None)], res[(h, w, b, "1")]
        print("| %dx%d | %d | %.3f | %.3f | %s | %+.1f %% | %.3f | %+.1f %% | %d / %d / %d |"
              % (h, w, b, fused[0], dflt[0], "split" if dflt[1] != fused[1] else "fused", (fused[0] / dflt[0] - 1) * 100, forced[0],
                 (fused[0] / forced[0] - 1) * 100, fused[1] // 19, dflt[1] // 19, forced[1] // 19), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
