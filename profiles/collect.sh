#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + PMC passes of the bench workload.
# Counters are collected in their own passes, with --kernel-trace only (gpurun refuses --pmc
# together with sys/hip/hsa traces).  Usage: profiles/collect.sh <tag>   -> gpurun_out/<tag>_*
set -e
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --profile-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --profile-legs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_kt.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq -- $B > /dev/null 2> gpurun_out/${TAG}_sq.err
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/${TAG}_tcc -- $B > /dev/null 2> gpurun_out/${TAG}_tcc.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -- $B > /dev/null 2> gpurun_out/${TAG}_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -- $B > /dev/null 2> gpurun_out/${TAG}_write.err
echo collected $TAG
