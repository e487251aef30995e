#!/bin/bash
# Round 6 closing run: the whole GPU suite as the driver runs it (with durations), smoke(), then what profiles/r06/ is made of.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/r06_full_gpu.log 2>&1 || { tail -60 gpurun_out/r06_full_gpu.log; exit 1; }
tail -16 gpurun_out/r06_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"
bash scripts/gpu_r06_final.sh
