#!/bin/bash
# Timeline of one DWP predictor step (B = 1, 512x512): every launch with start / end, to see what stands between two steps.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/dwpt.py <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
f = synth.turbulence(12, 512, 512)
for _ in range(2): ctx.rollout(f, 0, None, 1e9)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dwpt -- python /tmp/dwpt.py > /dev/null 2> gpurun_out/dwpt.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dwpt/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void k_", "k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_sse_decide" in r["Kernel_Name"]]
i0 = starts[-4]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0: starts[-3] + 2]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-30s grid %5d  queue %s  start %8.1f  end %8.1f  (%.1f us)" % (name, grid, r.get("Queue_Id", "?"), s, e, e - s))
PY
rm -rf gpurun_out/dwpt
