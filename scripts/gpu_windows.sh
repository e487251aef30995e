#!/bin/bash
# On the GPU box: the bench step at several sequence lengths (windows in flight = frames / 20).
cd "$GRAFT_REPO_ROOT"
for nt in 40 80 160 320 640; do
  python bench.py --frames $nt --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d['roofline_all_convolutions']
print('nt=%4d windows=%2d  %7.1f frames/s  k_conv16 %.3f  all convs %.3f of peak' % ($nt, $nt//20, d['value'], r['frac'], a['frac']))"
done
