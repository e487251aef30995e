#!/usr/bin/env python3
"""the "E-part ahead" rule (tz_prednet.hip) outside 512x512.  One process per (shape, batch, TEZIP_EPART)
cell -- the switch is read when the context is made -- each timing an SWP rollout of `batch` windows of 20 steps, best of 5,
and reporting how many k_wino launches a step made (5 fused, 7 split at the reference's four levels).

  python scripts/epart_shapes.py cell H W batch        # one cell (TEZIP_EPART from the environment)
  python scripts/epart_shapes.py                       # the table: every shape x batch x {0, default, 1}
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(1024, 1024, 1), (376, 1248, 1), (256, 256, 1), (256, 256, 2), (256, 256, 3), (256, 256, 4),
          (512, 512, 1), (512, 512, 2), (512, 512, 3), (512, 512, 4), (128, 160, 1), (1024, 1024, 2)]
STEPS = 20


def cell(h, w, batch):
    import numpy as np
    import torch
    from tezip_amd import _lib
    from tezip_amd.prednet import PredNetConfig
    cfg = PredNetConfig()
    ctx = _lib.Context(0)
    ctx.load_model(cfg, cfg.init_weights(seed=123))
    ctx.prepare(_lib.pad8(h), _lib.pad8(w), batch)
    nt = STEPS * batch
    rng = np.random.default_rng(5)
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
    ctx.close()
    print("CELL %d %d %d %s %.3f %d" % (h, w, batch, os.environ.get("TEZIP_EPART", "default"), best * 1e3, launches), flush=True)


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "cell":
        return cell(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    rows = []
    for h, w, b in SHAPES:
        res = {}
        for mode in ("0", None, "1"):
            env = dict(os.environ)
            env.pop("TEZIP_EPART", None)
            if mode is not None:
                env["TEZIP_EPART"] = mode
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "cell", str(h), str(w), str(b)], env=env,
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("CELL")]
            if r.returncode != 0 or not line:
                print("cell %dx%d B=%d EPART=%s failed:\n%s" % (h, w, b, mode, r.stderr[-1500:]), flush=True)
                return 1
            _, _, _, _, _, ms, launches = line[-1].split()
            res[mode] = (float(ms), int(launches))
        per_step = 19 * 1   # predictor steps of one rollout (windows run side by side)
        fused, dflt, forced = res["0"], res[None], res["1"]
        engaged = dflt[1] != fused[1]
        rows.append((h, w, b, fused[0], dflt[0], forced[0], engaged, fused[1] // per_step, forced[1] // per_step))
        print("%4dx%-4d B=%d  fused %8.3f ms  default %8.3f ms (%s, %+5.1f %%)  forced %8.3f ms (%+5.1f %%)   k_wino launches/step fused %d forced %d"
              % (h, w, b, fused[0], dflt[0], "SPLIT" if engaged else "fused", (fused[0] / dflt[0] - 1) * 100, forced[0],
                 (fused[0] / forced[0] - 1) * 100, fused[1] // per_step, forced[1] // per_step), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
