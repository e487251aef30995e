#!/bin/bash
# On the GPU box: A/B of whole k_wino header versions on ONE box (files under scratch/, bench step each, twice; timing only).
#   gpurun -- 'FILES="scratch/wino_old.h|scratch/wino_new.h" bash scripts/gpu_wino_ab_files.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
IFS='|' read -ra V <<< "$FILES"
cp tezip_amd/csrc/tz_wino_kernels.hip.h gpurun_out/.wino_head.h   # the tracked header comes back at the end
for rep in 1 2; do
for v in "${V[@]}"; do
  cp "$v" tezip_amd/csrc/tz_wino_kernels.hip.h
  python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json'))
print('$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['kernel_ms_per_step']['wino_pa2'],3))"
done; done
cp gpurun_out/.wino_head.h tezip_amd/csrc/tz_wino_kernels.hip.h && rm -f gpurun_out/.wino_head.h
python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || echo "WARNING: rebuild of the tracked header failed"
