#!/bin/bash
# On the GPU box: A/B of two whole csrc/ source sets on ONE box (scratch/old_csrc vs scratch/new_csrc), bench step each, REPS times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/.csrc_head && cp tezip_amd/csrc/*.hip tezip_amd/csrc/*.h gpurun_out/.csrc_head/
for rep in $(seq ${REPS:-2}); do
for v in old_csrc new_csrc; do
  cp scratch/$v/* tezip_amd/csrc/
  python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python bench.py --steps ${STEPS:-8} --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json'))
print('$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['kernel_ms_per_step']['wino_pa2'],3))"
done; done
cp gpurun_out/.csrc_head/* tezip_amd/csrc/ && rm -rf gpurun_out/.csrc_head
python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || echo "WARNING: rebuild of the tracked sources failed"
