#!/bin/bash
# Round 6 closing run: DWP soak (with and without poison), the whole GPU suite with durations, smoke.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
TEZIP_POISON=255 python scripts/soak_dwp.py --reps 300 2>/dev/null | tee gpurun_out/r06_soak_dwp.txt
python scripts/soak_dwp.py --reps 300 2>/dev/null | tee -a gpurun_out/r06_soak_dwp.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r06_full_gpu.log 2>&1 || { tail -60 gpurun_out/r06_full_gpu.log; exit 1; }
tail -22 gpurun_out/r06_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"
