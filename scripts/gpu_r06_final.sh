#!/bin/bash
# Round 6: what profiles/r06/ is made of -- the driver-style bench line, the rocprofv3 kernel trace + PMC passes of the same
# command, the dynamic instruction mix.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
python bench.py --steps 20 --warmup 2 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err || { tail -20 gpurun_out/r06_bench_n1.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_n1.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'h2h', d['value_host_to_host'], 'pipelined', d.get('value_host_to_host_pipelined'), 'wino ms', d['kernel_ms_per_step']['wino_pa2'], 'frac', d['roofline']['frac'], 'delta frac', d['roofline_delta']['frac'])"
bash profiles/collect.sh r06
python profiles/summarize.py gpurun_out/r06_sum gpurun_out/r06_kt gpurun_out/r06_sq gpurun_out/r06_tcc gpurun_out/r06_fetch gpurun_out/r06_write > /dev/null
bash scripts/gpu_instmix.sh > gpurun_out/r06_sum/instruction_mix.txt
head -14 gpurun_out/r06_sum/per_shape.csv
rm -rf gpurun_out/r06_kt gpurun_out/r06_sq gpurun_out/r06_tcc gpurun_out/r06_fetch gpurun_out/r06_write
# ROCTx ranges (TEZIP_ROCTX=1) in a marker + kernel trace of one small job
cat > /tmp/roctx_job.py <<'PY'
import sys, os
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(); ctx = _lib.Context(0); ctx.load_model(cfg, cfg.init_weights(seed=123)); ctx.prepare(256, 256, 2)
f = synth.turbulence(8, 256, 256)
key, _ = ctx.rollout(f, 0, 4)
payload, table, _ = ctx.encode("abs", [2.0], True)
ctx.rollout_decode(np.where(key[:, None, None, None], f, 0).astype(np.uint8), 0)
ctx.decode(payload, table)
PY
export TEZIP_ROCTX=1
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d gpurun_out/r06_roctx -- python /tmp/roctx_job.py > /dev/null 2> gpurun_out/r06_roctx.err
unset TEZIP_ROCTX
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r06_roctx/*/*_marker_api_trace.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
c = collections.Counter()
t = collections.Counter()
for r in rows:
    name = r.get("Function") or r.get("Message") or r.get("Name") or "?"
    c[name] += 1
    t[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open("gpurun_out/r06_sum/roctx_ranges.txt", "w") as out:
    out.write("# rocprofv3 --marker-trace of one 8-frame 256x256 job with TEZIP_ROCTX=1: range name, count, total host us\n")
    for k, v in sorted(c.items(), key=lambda kv: -t[kv[0]]):
        out.write("%-28s %5d %10.1f\n" % (k, v, t[k] / 1e3))
print(open("gpurun_out/r06_sum/roctx_ranges.txt").read())
PY
rm -rf gpurun_out/r06_roctx
