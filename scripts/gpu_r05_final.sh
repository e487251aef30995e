#!/bin/bash
# Round 5: what profiles/r05/ is made of -- the driver-style bench line, the rocprofv3 kernel trace + PMC passes of the same
# command, the dynamic instruction mix.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
python bench.py --steps 20 --warmup 2 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err || { tail -20 gpurun_out/r05_bench_n1.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_n1.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'h2h', d['value_host_to_host'], 'pipelined', d.get('value_host_to_host_pipelined'), 'wino ms', d['kernel_ms_per_step']['wino_pa2'], 'frac', d['roofline']['frac'], 'delta frac', d['roofline_delta']['frac'])"
bash profiles/collect.sh r05
python profiles/summarize.py gpurun_out/r05_sum gpurun_out/r05_kt gpurun_out/r05_sq gpurun_out/r05_tcc gpurun_out/r05_fetch gpurun_out/r05_write > /dev/null
bash scripts/gpu_instmix.sh > gpurun_out/r05_sum/instruction_mix.txt
head -14 gpurun_out/r05_sum/per_shape.csv
rm -rf gpurun_out/r05_kt gpurun_out/r05_sq gpurun_out/r05_tcc gpurun_out/r05_fetch gpurun_out/r05_write
