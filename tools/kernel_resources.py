#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / static LDS of every kernel of a gfx950 code object (the metadata notes hipcc writes).
usage: hipcc --offload-arch=gfx950 <flags> --offload-device-only -c csrc/mvn_hip.hip -o mvn.co
       clang-offload-bundler --unbundle --type=o --input=mvn.co --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=dev.elf
       llvm-readelf --notes dev.elf | python3 tools/kernel_resources.py [substring ...]"""
import re
import subprocess
import sys

text = sys.stdin.read()
rows = []
names, vals = [], []
for e in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", e) or [None, "?"])[1]  # noqa: E731
    names.append(re.search(r"\.name:\s+(\S+)", e).group(1))
    vals.append((g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
for d, (v, s, p, l) in zip(dem, vals):
    d = d.split("(anonymous namespace)::", 1)[-1]
    depth = 0
    for i, ch in enumerate(d):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            d = d[:i]
            break
    rows.append(f"{d:62s} vgpr {v:>4s} sgpr {s:>4s} scratch {p:>5s} static_lds {l}")
for r in sorted(rows):
    if not sys.argv[1:] or any(a in r for a in sys.argv[1:]):
        print(r)
