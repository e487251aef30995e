#!/bin/bash
# On the GPU box: SQ counters of the B = 1 512x512 rollout (k_convlat in the throughput regime vs k_conv16).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/b1.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
f = synth.turbulence(6, 512, 512)
ctx.rollout(f, 0, 6)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1p_kt -- python /tmp/b1.py > /dev/null 2> gpurun_out/b1p_kt.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/b1p_sq -- python /tmp/b1.py > /dev/null 2> gpurun_out/b1p_sq.err
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/b1p_tcc -- python /tmp/b1.py > /dev/null 2> gpurun_out/b1p_tcc.err
python profiles/summarize.py gpurun_out/b1p_sum gpurun_out/b1p_kt gpurun_out/b1p_sq gpurun_out/b1p_tcc > /dev/null
rm -rf gpurun_out/b1p_kt gpurun_out/b1p_sq gpurun_out/b1p_tcc
head -8 gpurun_out/b1p_sum/per_shape.csv
