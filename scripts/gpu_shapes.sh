#!/bin/bash
# On the GPU box: per-shape kernel durations of one bench run (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-shapes}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_kt.err
python profiles/summarize.py gpurun_out/${TAG}_sum gpurun_out/${TAG}_kt > /dev/null
head -12 gpurun_out/${TAG}_sum/per_shape.csv
