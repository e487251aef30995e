#!/bin/bash
# On the GPU box: A/B of k_wino build variants on ONE box (TEZIP_DEFINES per variant, bench step each, twice).
#   gpurun -- 'VARIANTS="-DTZW_LEAD=5|-DTZW_LEAD=4" bash scripts/gpu_wino_ab.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TEZIP_ALLOW_DIAGNOSTIC_BUILD=1   # _lib.load() refuses a library built with diagnostic defines otherwise
IFS='|' read -ra V <<< "$VARIANTS"
for rep in $(seq ${REPS:-2}); do
for v in "${V[@]}"; do
  d=$(echo $v | sed 's/-D//g')
  TEZIP_DEFINES="$d" python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python bench.py --steps ${STEPS:-5} --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json'))
print('$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['kernel_ms_per_step']['wino_pa2'],3))"
done; done
# leave the product build behind, whatever ran last (the flag stamp of tezip_amd/build.py would rebuild it anyway)
env -u TEZIP_DEFINES python -c "from tezip_amd import build; build.build(force=True)" > /dev/null 2>&1 || echo "WARNING: clean rebuild failed"
