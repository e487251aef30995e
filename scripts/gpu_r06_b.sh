#!/bin/bash
# Round 6: measured E-part decision + elementwise quantiser map: parity first, then the shape table (default = measured),
# then the driver-style bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 900 python -m pytest tests/test_gpu_epart.py tests/test_gpu_qmap.py tests/test_gpu_parity.py tests/test_gpu_poison.py tests/test_gpu_configs.py tests/test_gpu_ref_runs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r06_b_tests.log 2>&1 || { tail -60 gpurun_out/r06_b_tests.log; exit 1; }
tail -3 gpurun_out/r06_b_tests.log
TEZIP_EPART_LOG=1 timeout -k 10 900 python scripts/epart_shapes.py 2> gpurun_out/r06_epart_measure.log | tee gpurun_out/r06_epart_shapes_measured.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err || { tail -30 gpurun_out/r06_bench_b.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_b.json"))
print("value", d["value"], "ms", d["ms_per_step"], "h2h", d["value_host_to_host"], d["value_host_to_host_pipelined"])
print("roofline", d["roofline"]["frac"], "encode_tail", json.dumps(d["roofline_encode_tail"]))
print("lossy_abs2", json.dumps(d.get("lossy_abs2")))
print("configs", json.dumps(d.get("configs")))
PY
