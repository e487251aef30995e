#!/bin/bash
# Round 5, item 6: parity of the edited k_wino, A/B of header versions, then the stamps of the A1 / A2 shapes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
python -m pytest tests/test_gpu_wino.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r05_a1_parity.log 2>&1 || { tail -30 gpurun_out/r05_a1_parity.log; exit 1; }
tail -2 gpurun_out/r05_a1_parity.log
bash scripts/gpu_wino_ab_files.sh | tee gpurun_out/r05_a1_ab.txt
export TEZIP_ALLOW_DIAGNOSTIC_BUILD=1
TEZIP_DEFINES=TZW_STAMPS python -m tezip_amd.build --force > /dev/null 2>&1
python scripts/wino_stamps.py | tee gpurun_out/r05_a1_stamps.txt
