#!/bin/bash
# Round 6: the whole GPU suite as the driver runs it, smoke(), then what profiles/r06/ is made of.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r06_full_gpu.log 2>&1 || { tail -60 gpurun_out/r06_full_gpu.log; exit 1; }
tail -3 gpurun_out/r06_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"
