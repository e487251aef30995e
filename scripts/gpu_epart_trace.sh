#!/bin/bash
# Timeline of one B = 1 predictor step with the split gate launches: do the side launches overlap the critical path at all?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/b1t.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(512, 512, 1)
f = synth.turbulence(6, 512, 512)
for _ in range(2): ctx.rollout(f, 0, 5)
PY
TEZIP_EPART=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ept -- python /tmp/b1t.py > /dev/null 2> gpurun_out/ept.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ept/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void k_", "k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last predictor step: from the last k_conv16b<3,4,false> (A0) on
starts = [i for i, r in enumerate(rows) if "k_conv16b<3" in r["Kernel_Name"]]
i0 = starts[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0: i0 + 11]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print("%-28s grid %5d  queue %s  start %8.1f us  end %8.1f us  (%.1f us)" % (name, grid, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf gpurun_out/ept
