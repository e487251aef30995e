#!/bin/bash
# Round 6, item 3: "E-part ahead" at the shapes it was never measured at -- fused vs default vs forced, one box -- then its
# parity tests, then the driver-style bench line (host_pipeline / roofline_encode_tail legs included).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 900 python scripts/epart_shapes.py | tee gpurun_out/r06_epart_shapes.txt
timeout -k 10 600 python -m pytest tests/test_gpu_epart.py tests/test_gpu_contract.py -x -q -m gpu > gpurun_out/r06_epart_tests.log 2>&1 || { tail -40 gpurun_out/r06_epart_tests.log; exit 1; }
tail -2 gpurun_out/r06_epart_tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err || { tail -30 gpurun_out/r06_bench_a.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_a.json"))
print("value", d["value"], "ms", d["ms_per_step"], "h2h", d["value_host_to_host"], d["value_host_to_host_pipelined"])
print("roofline", d["roofline"]["frac"], "encode_tail", json.dumps(d["roofline_encode_tail"]))
print("host_pipeline", json.dumps(d.get("host_pipeline"), indent=1))
print("configs", json.dumps(d.get("configs")))
PY
