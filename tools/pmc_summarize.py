#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for this repo's kernels.

bench.py launches some kernels at several sizes (count_errors_kernel: 10 000-row headline steps, a 100-row and a 125 000-row
config; the guard kernel; ...).  A mean over ALL dispatches of a name is a mean over different workloads -- round 2's
"count_errors_kernel reads 98.5 MB for 80 MB" was exactly that: 45 headline dispatches of 625 k requests averaged with one
125 000-row dispatch of 7.8 M.  So the figure reported per kernel is the MEDIAN over the dispatches of its most frequent grid
size (count_errors_kernel caps its grid at 512 workgroups, so the grid alone does not tell its workloads apart): the headline
workload for every kernel of the headline step, the only workload for the others.  The column keeps its name
(`mean_per_dispatch`: the tools downstream read it); the plain mean is printed next to it."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))  # kernel -> grid -> counter -> values
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous namespace" in name and "at::" not in name:
            short = name.split("(anonymous namespace)::", 1)[-1]
            depth = 0
            for i, ch in enumerate(short):  # the name ends at the first "(" outside the template argument list
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    short = short[:i]
                    break
            agg[short][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,mean_per_dispatch,dispatches,grid_threads,other_grids,plain_mean")
for k in sorted(agg):
    grids = agg[k]
    main = max(grids, key=lambda g: max(len(v) for v in grids[g].values()))
    for c in sorted(grids[main]):
        v = sorted(grids[main][c])
        med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
        print(f"\"{k}\",{c},{med:.1f},{len(v)},{main},{len(grids) - 1},{sum(v)/len(v):.1f}")
