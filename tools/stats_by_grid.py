#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid size): bench.py launches the same kernel at several
batch sizes (10 000-block Monte-Carlo steps, 125 000-block VA, B = 1 by-word calls), which --stats averages together.
usage: stats_by_grid.py <..._kernel_trace.csv> [min_calls]"""
import collections
import csv
import sys

agg = collections.defaultdict(list)
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "anonymous namespace" not in name or "at::" in name:
        continue
    short = name.split("(anonymous namespace)::", 1)[-1]
    depth = 0
    for i, ch in enumerate(short):  # the name ends at the first "(" outside the template argument list
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            short = short[:i]
            break
    gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    gy = int(r.get("Grid_Size_Y", 1) or 1)  # trial-batched launches: gridDim.y = trials
    key = (short, f"{gx}x{gy}" if gy > 1 else gx)
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[key] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size_X"])
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("kernel,workgroups,calls,avg_us,min_us,max_us,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,workgroup_size")
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):  # g: workgroups, or 'x-by-y' for 2-D grids
    if len(v) >= min_calls:
        print(f"\"{k}\",{g},{len(v)},{sum(v)/len(v)/1e3:.2f},{min(v)/1e3:.2f},{max(v)/1e3:.2f}," + ",".join(meta[(k, g)]))
