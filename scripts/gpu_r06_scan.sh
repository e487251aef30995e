#!/bin/bash
# Round 6: inverse scan with a wave's run kept in registers between the two phases (payload read once) vs the two reads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ref_runs.py tests/test_gpu_fuzz.py -x -q -m gpu -k "undelta or scan or decode or decompress or roundtrip or reconstruct or fuzz or ref or cfg" > gpurun_out/r06_scan_tests.log 2>&1 || { tail -40 gpurun_out/r06_scan_tests.log; exit 1; }
tail -2 gpurun_out/r06_scan_tests.log
for rep in 1 2; do
echo "TEZIP_SCAN_KEEP=0"; TEZIP_SCAN_KEEP=0 python scripts/decode_tail_bench.py --reps 20 2>/dev/null | tail -4
echo "default (kept in registers)"; python scripts/decode_tail_bench.py --reps 20 2>/dev/null | tail -4
done | tee gpurun_out/r06_scan_ab.txt
