#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs of one workload (the passes of tools/pmc_sq.sh) as ONE table: a row per dispatch of this repo's kernels
in dispatch order (the passes run the same deterministic workload, so dispatch k of a kernel is the same launch in every pass),
a column per counter, plus the launch's duration from the kernel trace of the first pass.
usage: pmc_dispatch_table.py <dir of pass*/> [kernel substring ...]  > table.csv"""
import collections
import csv
import glob
import os
import sys

root, want = sys.argv[1], sys.argv[2:]


def short_name(name):
    short = name.split("(anonymous namespace)::", 1)[-1]
    depth = 0
    for i, ch in enumerate(short):  # the name ends at the first "(" outside the template argument list
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return short[:i]
    return short


rows = collections.OrderedDict()  # (kernel, ordinal among its dispatches) -> {counter: value}
counters = []
for p in sorted(glob.glob(os.path.join(root, "pass*"))):
    if not os.path.isdir(p):
        continue
    seen = collections.defaultdict(dict)  # dispatch id -> counters
    meta = {}
    for f in glob.glob(os.path.join(p, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "anonymous namespace" not in r["Kernel_Name"] or "at::" in r["Kernel_Name"]:
                continue
            k = short_name(r["Kernel_Name"])
            if want and not any(s in k for s in want):
                continue
            d = int(r["Dispatch_Id"])
            seen[d][r["Counter_Name"]] = float(r["Counter_Value"])
            meta[d] = (k, r["Grid_Size"], r["Workgroup_Size"], r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""),
                       r.get("Scratch_Size", ""))
            if r["Counter_Name"] not in counters:
                counters.append(r["Counter_Name"])
    dur = {}
    for f in glob.glob(os.path.join(p, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    ordinal = collections.Counter()
    for d in sorted(seen):
        k = meta[d][0]
        key = (k, ordinal[k])
        ordinal[k] += 1
        row = rows.setdefault(key, {"_meta": meta[d]})
        row.update(seen[d])
        if d in dur:
            row.setdefault("_ms", []).append(dur[d])
print("kernel,ordinal,grid_threads,workgroup,lds_bytes,vgprs,agprs,scratch,ms_profiled," + ",".join(counters))
for (k, o), row in rows.items():
    m = row["_meta"]
    ms = row.get("_ms", [])
    print(f"\"{k}\",{o},{m[1]},{m[2]},{m[3]},{m[4]},{m[5]},{m[6]},{(sum(ms) / len(ms)) if ms else float('nan'):.4f}," +
          ",".join(f"{row.get(c, float('nan')):.0f}" for c in counters))
