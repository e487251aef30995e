#!/bin/bash
# On the GPU box: build and run one of the microbenchmarks of scripts/microbench
# (MB=mfma_f32_peak | mfma_lds_floor | conv_skeleton; default mfma_lds_floor).
#   gpurun -- 'MB=conv_skeleton bash scripts/gpu_floor.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MB=${MB:-mfma_lds_floor}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/microbench/$MB.hip -o /tmp/mb && /tmp/mb > gpurun_out/$MB.log 2>&1
cat gpurun_out/$MB.log
