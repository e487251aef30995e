#!/bin/bash
# On the GPU box: per-shape kernel durations of a B = 1 rollout (DWP is sequential: one window at a time).
# usage: gpu_b1.sh [tag [lat-mode [TEZIP_LAT_WIDE_MIN]]]   lat-mode: model (default) | always | never
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-b1}; export B1_LAT=${2:-model}
if [ -n "$3" ]; then export TEZIP_LAT_WIDE_MIN=$3; fi
cat > /tmp/b1.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
lat = os.environ.get("B1_LAT", "model")
if lat != "model": ctx.set_conv_impl(True, lat=lat)
f = synth.turbulence(21, 512, 512)
for _ in range(3): ctx.rollout(f, 0, 20)
t0 = time.perf_counter()
for _ in range(3): ctx.rollout(f, 0, 20)
print("B=1 512x512 rollout of 20 steps: %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt -- python /tmp/b1.py > gpurun_out/${TAG}.out 2> gpurun_out/${TAG}.err
python profiles/summarize.py gpurun_out/${TAG}_sum gpurun_out/${TAG}_kt > /dev/null
echo "== $TAG lat=$B1_LAT wide_min=$TEZIP_LAT_WIDE_MIN"; cat gpurun_out/${TAG}.out
head -12 gpurun_out/${TAG}_sum/per_shape.csv
rm -rf gpurun_out/${TAG}_kt
