#!/bin/bash
# On the GPU box: per-shape kernel durations of a B = 1 rollout (DWP is sequential: one window at a time).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/b1.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
f = synth.turbulence(21, 512, 512)
for _ in range(3): ctx.rollout(f, 0, 20)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1_kt -- python /tmp/b1.py > /dev/null 2> gpurun_out/b1.err
python profiles/summarize.py gpurun_out/b1_sum gpurun_out/b1_kt > /dev/null
head -12 gpurun_out/b1_sum/per_shape.csv
