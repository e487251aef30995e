#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_contract.py tests/test_gpu_poison.py tests/test_gpu_train.py -x -q -m gpu --durations=8 > gpurun_out/r06_g_tests.log 2>&1 || { tail -60 gpurun_out/r06_g_tests.log; exit 1; }
tail -14 gpurun_out/r06_g_tests.log
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-trained-ratio > gpurun_out/r06_bench_g.json 2> gpurun_out/r06_bench_g.err || { tail -30 gpurun_out/r06_bench_g.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_g.json"))
print("value", d["value"])
print("host_pipeline", json.dumps(d.get("host_pipeline"), indent=1))
PY
