#!/bin/bash
# On the GPU box: dynamic instruction mix per kernel of the bench step (one --pmc pass): VALU (MFMA included), MFMA,
# SALU, LDS, VMEM instructions per wave.  VALU instructions that are not MFMAs take 4 cycles of their SIMD each, which the
# matrix pipe of that SIMD does not get (scripts/microbench/coexec.hip): this is the budget to watch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/imix -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/imix.err
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/imix/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid " + str(int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"],)
    if key not in seen:
        seen.add(key)
        n[k] += 1
print("%-46s %6s %9s %8s %8s %8s %8s %8s %10s" % ("kernel", "calls", "waves", "VALU-M", "MFMA", "SALU", "LDS", "VMEM", "VALU*4/MFMA*32"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0)):
    w = c.get("SQ_WAVES", 0) or 1
    valu = c.get("SQ_INSTS_VALU", 0) - c.get("SQ_INSTS_MFMA", 0)
    mf = c.get("SQ_INSTS_MFMA", 0)
    if not k.startswith("k_"):
        continue
    print("%-46s %6d %9.0f %8.0f %8.0f %8.0f %8.0f %8.0f %10.3f" % (k[:46], n[k], w / n[k], valu / w, mf / w, c.get("SQ_INSTS_SALU", 0) / w,
          c.get("SQ_INSTS_LDS", 0) / w, (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / w, (valu * 4) / (mf * 32) if mf else float("nan")))
PY
rm -rf gpurun_out/imix
