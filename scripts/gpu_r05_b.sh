#!/bin/bash
# Round 5: deferred payload test, MFMA shape microbenchmark, A/B of the packed output-transform subtractions, bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
python -m pytest tests/test_gpu_deferred.py tests/test_gpu_wino.py tests/test_abi.py -x -q -m "gpu or not gpu" > gpurun_out/r05_b_tests.log 2>&1 || { tail -30 gpurun_out/r05_b_tests.log; exit 1; }
tail -2 gpurun_out/r05_b_tests.log
./scratch/mfma_shape_clock | tee gpurun_out/r05_mfma_shape_clock.txt
FILES="scratch/wino_new.h|scratch/wino_v2.h" bash scripts/gpu_wino_ab_files.sh | tee gpurun_out/r05_v2_ab.txt
python bench.py --steps 10 --warmup 2 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err || { tail -20 gpurun_out/r05_bench_a.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_a.json'))
print('value', d['value'], 'h2h', d['value_host_to_host'], 'pipelined', d.get('value_host_to_host_pipelined'), 'wino ms', d['kernel_ms_per_step']['wino_pa2'], 'frac', d['roofline']['frac'])
print(json.dumps(d['host_to_host'])[:900]); print(json.dumps(d['cpu_baseline'])[:600])"
