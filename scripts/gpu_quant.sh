#!/bin/bash
# On the GPU box: kernel durations of the encode back half (delta, quantiser, spatial delta, remap) at cfg3, abs 2 and rel 1e-3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/quant.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 4)
f = synth.turbulence(80, 512, 512)
ctx.rollout(f, 0, 20)
for _ in range(3):
    ctx.encode("abs", [2.0], True)
    ctx.encode("rel", [1e-3], True)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/quant_kt -- python /tmp/quant.py > /dev/null 2> gpurun_out/quant.err
python profiles/summarize.py gpurun_out/quant_sum gpurun_out/quant_kt > /dev/null
grep "k_q_\|k_delta\|k_sdelta\|k_lut" gpurun_out/quant_sum/per_shape.csv
rm -rf gpurun_out/quant_kt
