#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for this repo's kernels."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "anonymous namespace" in name and "at::" not in name:
            short = name.split("::")[-1].split("(")[0]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,mean_per_dispatch,dispatches")
for k in sorted(agg):
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"\"{k}\",{c},{sum(v)/len(v):.1f},{len(v)}")
