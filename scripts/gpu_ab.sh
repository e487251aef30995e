#!/bin/bash
# On the GPU box: parity tests, then A/B of the two convolution kernels in one session.
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for v in 0 1 0 1; do
  TEZIP_CONV16=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('CONV16=$v conv_ms %.2f  step_ms %.2f  frames/s %.1f frac %.3f' % (d['kernel_ms_per_step']['conv3x3_mfma'], d['ms_per_step'], d['value'], d['roofline']['frac']))"
done
