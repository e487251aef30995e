#!/bin/bash
# Round 6: identity shortcut of the quantiser: parity, then the bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 900 python -m pytest tests/test_gpu_qmap.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ref_runs.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_poison.py -x -q -m gpu --durations=12 > gpurun_out/r06_d_tests.log 2>&1 || { tail -60 gpurun_out/r06_d_tests.log; exit 1; }
tail -18 gpurun_out/r06_d_tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r06_bench_d.json 2> gpurun_out/r06_bench_d.err || { tail -30 gpurun_out/r06_bench_d.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_d.json"))
print("value", d["value"], "ms", d["ms_per_step"], "h2h", d["value_host_to_host"], d["value_host_to_host_pipelined"])
print("roofline", d["roofline"]["frac"], "encode_tail", json.dumps(d["roofline_encode_tail"]))
PY
