#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 900 python -m pytest tests/test_gpu_qmap.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_poison.py -x -q -m gpu --durations=5 > gpurun_out/r06_e_tests.log 2>&1 || { tail -60 gpurun_out/r06_e_tests.log; exit 1; }
tail -10 gpurun_out/r06_e_tests.log
bash scripts/gpu_r06_final.sh
