#!/bin/bash
# Round 5, item 8: the split gate launches -- parity, then the B = 1 rollout with and without them, then the cfg3 step (must not move).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
python -m pytest tests/test_gpu_epart.py tests/test_gpu_wino.py -x -q -m gpu > gpurun_out/r05_epart_tests.log 2>&1 || { tail -40 gpurun_out/r05_epart_tests.log; exit 1; }
tail -2 gpurun_out/r05_epart_tests.log
cat > /tmp/b1t.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
f = synth.turbulence(21, 512, 512)
for _ in range(3): ctx.rollout(f, 0, 20)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); ctx.rollout(f, 0, 20); best = min(best, time.perf_counter() - t0)
print("TEZIP_EPART=%s  B=1 512x512 rollout of 20 steps: %.2f ms" % (os.environ.get("TEZIP_EPART", "default"), best * 1e3))
PY
for rep in 1 2; do
TEZIP_EPART=0 python /tmp/b1t.py
TEZIP_EPART=1 python /tmp/b1t.py
python /tmp/b1t.py
done | tee gpurun_out/r05_epart_b1.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-trained-ratio > gpurun_out/r05_bench_e.json 2> gpurun_out/r05_bench_e.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_e.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'wino ms', d['kernel_ms_per_step']['wino_pa2'], 'cfg5', d['configs']['cfg5_dwp']['frames_per_s'], 'decode', d['configs']['cfg5_dwp']['decode_frames_per_s'])"
