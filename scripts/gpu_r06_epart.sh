#!/bin/bash
# Round 6, item 3: "E-part ahead" at the shapes it was never measured at, fused vs default vs forced, one box; then the DWP
# rollout both ways and the timeline of one DWP step (k_sse_decide with the real ordering: returning exchange + vmcnt(0)).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 900 python scripts/epart_shapes.py | tee gpurun_out/r06_epart_shapes.txt
for rep in 1 2; do
TEZIP_EPART=0 python scripts/dwp_time.py
python scripts/dwp_time.py
done 2>/dev/null | tee gpurun_out/r06_dwp_ab.txt
bash scripts/gpu_dwp_trace.sh | tee gpurun_out/r06_dwp_step_trace.txt
