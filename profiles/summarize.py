#!/usr/bin/env python3
"""Turns rocprofv3 output directories (kernel trace + separate --pmc passes, collected as
profiles/collect.sh does) into the small summaries committed under profiles/rNN/:
  kernel_stats.csv   copy of rocprofv3's --stats table
  per_shape.csv      per (kernel, grid) average duration
  traffic.json       per-launch HBM traffic per kernel from FETCH_SIZE / WRITE_SIZE, corrected as
                     MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2 for wide coalesced
                     reads; both counters are in KiB) -- bench.py reads this file for `traffic`.
usage: summarize.py <out_dir> <kernel_trace_dir> [<pmc_dir> ...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def rows(d, pat):
    f = glob.glob(os.path.join(d, "*", pat))
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    out, kt, pmcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.makedirs(out, exist_ok=True)
    st = glob.glob(os.path.join(kt, "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(out, "kernel_stats.csv"))
    agg = collections.defaultdict(list)
    for r in rows(kt, "*_kernel_trace.csv"):
        if r["Kernel_Name"].startswith(("void k_", "k_")):
            grid = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            agg[(short(r["Kernel_Name"]), grid, r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, "per_shape.csv"), "w") as f:
        f.write("kernel,blocks,threads,lds_bytes,vgpr,calls,avg_us,total_ms\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%d,%s,%s,%s,%d,%.1f,%.2f\n" % (*k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
    counters = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in pmcs:
        for r in rows(d, "*_counter_collection.csv"):
            if r["Kernel_Name"].startswith(("void k_", "k_")):
                k = short(r["Kernel_Name"])
                counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]] += 1
    traffic = {}
    for k, c in counters.items():
        per = {n: c[n] / max(launches[k][n], 1) for n in c}
        e = {"counters_per_launch": per, "launches": max(launches[k].values())}
        if "FETCH_SIZE" in per or "WRITE_SIZE" in per:
            e["hbm_read_bytes_per_launch"] = per.get("FETCH_SIZE", 0.0) * 1024 * 2
            e["hbm_write_bytes_per_launch"] = per.get("WRITE_SIZE", 0.0) * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        if "TCC_HIT_sum" in per:
            e["l2_hit_rate"] = per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per.get("TCC_MISS_sum", 0.0), 1.0)
        traffic[k] = e
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
