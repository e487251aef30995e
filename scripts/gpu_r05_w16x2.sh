#!/bin/bash
# Round 5 go / no-go microbenchmark: 16 positions x 2 column tiles per wave, two 4-wave workgroups per CU (wino16x2.hip)
# against the pair-of-waves form the product kernel was built from (wino8.hip).  Binaries prebuilt into scratch/w16/.
set -e
mkdir -p gpurun_out
out=gpurun_out/r05_w16x2.txt
: > $out
for b in wino8 wino16x2_5_4 wino16x2_4_3 wino16x2_3_2; do
  echo "== $b" >> $out
  timeout -k 10 120 scratch/w16/$b.bin >> $out 2>&1
done
echo "== wino16x2_5_4 ablations" >> $out
timeout -k 10 120 scratch/w16/wino16x2_5_4.bin abl >> $out 2>&1
cat $out
