#!/bin/bash
# On the GPU box: kernel trace of a cfg5 DWP rollout (512x512, B = 1): per-kernel durations and the gaps between launches.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/dwp.py <<'PY'
import sys, os, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
f = synth.turbulence(80, 512, 512)
_, mse = ctx.rollout(f[:40], 0, None, 1e9, want_mse=True)
thr = float(np.sort(mse[1:])[10])
for _ in range(2): ctx.rollout(f, 0, None, thr)
t0 = time.perf_counter()
key, _ = ctx.rollout(f, 0, None, thr)
print("DWP rollout of 80 frames: %.2f ms, %d keys" % ((time.perf_counter() - t0) * 1e3, int(key.sum())))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dwp_kt -- python /tmp/dwp.py > gpurun_out/dwp.out 2> gpurun_out/dwp.err
python profiles/summarize.py gpurun_out/dwp_sum gpurun_out/dwp_kt > /dev/null
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/dwp_kt/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-12 * 79:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("last %d launches: busy %.1f us, span %.1f us, gap per launch %.2f us" % (len(rows), busy / 1e3, span / 1e3, (span - busy) / 1e3 / len(rows)))
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap[b["Kernel_Name"].split("(")[0].replace("void ", "")[:40]].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    print("  gap in front of %-42s n=%4d avg %.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
PY
cat gpurun_out/dwp.out
head -16 gpurun_out/dwp_sum/per_shape.csv
rm -rf gpurun_out/dwp_kt
