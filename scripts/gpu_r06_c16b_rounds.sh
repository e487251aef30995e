#!/bin/bash
# Round 6, item 6 (level-0 whole-round grids): how many k_conv16b workgroups does a CU hold at once?  One 512x512 predictor
# step for B = 1..8 windows (1024 workgroups per window) under the kernel trace; the staircase of the durations over B gives
# the number of workgroup slots on the chip (768 = 3 per CU: rounds 2,3,4,6,7,8,10,11; 1024 = 4 per CU: rounds 1..8).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
cat > /tmp/c16b.py <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123))
rng = np.random.default_rng(1)
for b in range(1, 9):
    ctx.prepare(512, 512, b)
    f = rng.integers(0, 256, (b, 512, 512, 3)).astype(np.float32) / np.float32(255)
    for _ in range(4):
        ctx.predict_next(f)
PY
TEZIP_EPART=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c16b_kt -- python /tmp/c16b.py > /dev/null 2> gpurun_out/c16b.err
python profiles/summarize.py gpurun_out/c16b_sum gpurun_out/c16b_kt > /dev/null
grep -E "^kernel|conv16b|conv_small" gpurun_out/c16b_sum/per_shape.csv | sort -t, -k1,1 -k2,2n | tee gpurun_out/r06_c16b_rounds.txt
rm -rf gpurun_out/c16b_kt
