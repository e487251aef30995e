#!/bin/bash
# On the GPU box: build and run the round-4 Winograd K-loop microbenchmark (scripts/microbench/wino16.hip).
#   gpurun -- 'bash scripts/gpu_wino16.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SRC=${SRC:-wino16}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $EXTRA scripts/microbench/$SRC.hip -o /tmp/$SRC.bin && timeout -k 10 120 /tmp/$SRC.bin $ARGS > gpurun_out/$SRC.log 2>&1
echo "rc=$?" >> gpurun_out/$SRC.log
cat gpurun_out/$SRC.log
