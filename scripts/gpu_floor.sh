#!/bin/bash
# On the GPU box: fp32-MFMA ceiling microbenchmark + clock (GRBM_GUI_ACTIVE) pass of the bench.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/microbench/${MB:-mfma_lds_floor}.hip -o /tmp/mb && /tmp/mb > gpurun_out/${MB:-mfma_lds_floor}.log 2>&1; cat gpurun_out/${MB:-mfma_lds_floor}.log; exit 0
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fix.json 2>gpurun_out/bench_fix.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/clk -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/clk.err
cat gpurun_out/floor.log; cat gpurun_out/bench_fix.json
