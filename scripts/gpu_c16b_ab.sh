#!/bin/bash
# On the GPU box: time the bench step's kernel classes with variants of tz_conv_kernels.hip.h (FILES="a.h|b.h|...", ablations
# give wrong results by design: timing only).  Restores the tracked header and rebuilds at the end.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out && cp tezip_amd/csrc/tz_conv_kernels.hip.h gpurun_out/.c16b_head.h
IFS='|' read -ra files <<< "$FILES"
: > gpurun_out/c16b_ab.txt
for rep in $(seq ${REPS:-1}); do
for f in "${files[@]}"; do
  cp "$f" tezip_amd/csrc/tz_conv_kernels.hip.h
  python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $f" | tee -a gpurun_out/c16b_ab.txt; continue; }
  python bench.py --steps ${STEPS:-6} --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "bench failed: $f" | tee -a gpurun_out/c16b_ab.txt; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); k=d['kernel_ms_per_step']
print('$f', 'fps', round(d['value'],1), 'step', round(d['ms_per_step'],3), 'conv16b', round(k['conv16b_level0'],3), 'wino', round(k['wino_pa2'],3), 'small', round(k['conv_small_valu'],3))" | tee -a gpurun_out/c16b_ab.txt
done; done
cp gpurun_out/.c16b_head.h tezip_amd/csrc/tz_conv_kernels.hip.h && rm -f gpurun_out/.c16b_head.h
python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || echo "WARNING: rebuild of the tracked sources failed"
