#!/bin/bash
# Round 5: deferred payload on its own stream (test + bench line), B = 1 per-shape table.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
python -m pytest tests/test_gpu_deferred.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05_c_tests.log 2>&1 || { tail -30 gpurun_out/r05_c_tests.log; exit 1; }
tail -2 gpurun_out/r05_c_tests.log
python bench.py --steps 10 --warmup 2 --no-trained-ratio > gpurun_out/r05_bench_c.json 2> gpurun_out/r05_bench_c.err || { tail -20 gpurun_out/r05_bench_c.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_c.json'))
print('value', d['value'], 'h2h', d['value_host_to_host'], 'pipelined', d.get('value_host_to_host_pipelined'), 'wino ms', d['kernel_ms_per_step']['wino_pa2'], 'frac', d['roofline']['frac'])
print({k: v for k, v in d['host_to_host'].items() if 'note' not in k})
print('cfg5', json.dumps(d.get('configs', {}))[:1500])"
bash scripts/gpu_b1.sh r05_b1
