#!/bin/bash
# Round 6: (a) k_conv16b workgroup slots per CU from the duration staircase over B; (b) the bench with FIVE ranks on this one
# GPU over gloo (the pool allows at most six GPU processes; eight are the driver's to run on an 8-GPU node).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
# (staircase: scripts/gpu_r06_c16b_rounds.sh, run separately)
TEZIP_BENCH_SINGLE_DEVICE=1 TEZIP_BENCH_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 5 --steps 2 --warmup 1 --no-cpu-baseline --no-trained-ratio > gpurun_out/r06_bench_5rank.json 2> gpurun_out/r06_bench_5rank.err || { tail -40 gpurun_out/r06_bench_5rank.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_5rank.json"))
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "sharded_check")})
print("config", d["config"])
for k in ("cfg4_sharded", "cfg5_sweep", "replicas"):
    print(k, json.dumps(d.get(k))[:1200])
PY
