#!/bin/bash
# On the GPU box: per-shape kernel durations of the cfg1 (64x64, 2 windows) and cfg2 (128x160, 4 windows) rollouts.
# SMALL_LAT=always|never forces / forbids k_convlat (default: the cost model).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in 1 2; do
cat > /tmp/small$C.py <<PY
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123))
if os.environ.get("SMALL_LAT"): ctx.set_conv_impl(True, lat=os.environ["SMALL_LAT"])
if $C == 1:
    f = synth.moving_blobs(40, 64, 64); ctx.prepare(64, 64, 2); w = 20
else:
    f = synth.translating_scene(40, 128, 160); ctx.prepare(128, 160, 4); w = 10
for _ in range(3): ctx.rollout(f, 0, w)
t0 = time.perf_counter()
for _ in range(5): ctx.rollout(f, 0, w)
print("cfg$C rollout ms", (time.perf_counter() - t0) / 5 * 1e3)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small${C}_kt -- python /tmp/small$C.py > gpurun_out/small$C.out 2> gpurun_out/small$C.err
python profiles/summarize.py gpurun_out/small${C}_sum gpurun_out/small${C}_kt > /dev/null
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/small${C}_kt/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].startswith(("void k_", "k_"))][-400:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("cfg$C last 400 launches: busy %.1f us, span %.1f us, gap per launch %.2f us" % (busy / 1e3, span / 1e3, (span - busy) / 1e3 / len(rows)))
PY
cat gpurun_out/small$C.out
head -12 gpurun_out/small${C}_sum/per_shape.csv
rm -rf gpurun_out/small${C}_kt
done
