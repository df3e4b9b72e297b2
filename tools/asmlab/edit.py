#!/usr/bin/env python3
"""Edits of the compiler's assembly for vnet16_fusedn_kernel<false, 2> (timing experiments only).
   edit.py in.s out.s nonop        : drop every s_nop inside the two depth-2 loops (the k-loops)  [results may be wrong]
   edit.py in.s out.s nonop_all    : drop every s_nop of the kernel
   edit.py in.s out.s dup:REGEX    : issue every matching instruction of the k-loops twice (marginal cost in situ)
   edit.py in.s out.s kill:REGEX   : drop every instruction of the k-loops whose text matches REGEX (knock-out timing)
"""
import re
import sys
src, dst, what = sys.argv[1:4]
L = open(src).read().split("\n")
beg = next(i for i, l in enumerate(L) if l.startswith("_ZN12_GLOBAL__N_120vnet16_fusedn_kernelILb0ELi2EE"))
end = next(i for i in range(beg, len(L)) if L[i].startswith(".Lfunc_end"))
out, depth2, n = [], False, 0
for i, l in enumerate(L):
    if beg <= i < end:
        if "Inner Loop Header: Depth=2" in l:
            depth2 = True
        if depth2 and l.strip().startswith("s_cbranch"):
            depth2 = False
        if what == "nonop" and depth2 and l.strip().startswith("s_nop"):
            n += 1
            continue
        if what == "nonop_all" and l.strip().startswith("s_nop"):
            n += 1
            continue
        if what.startswith("kill:") and depth2 and re.match(what[5:], l.strip()):
            n += 1
            continue
        if what.startswith("dup:") and depth2 and re.match(what[4:], l.strip()):
            n += 1
            out.append(l)  # issued twice (an in-place accumulate or an idempotent op: only the timing is read)
    out.append(l)
open(dst, "w").write("\n".join(out))
print(f"{what}: removed {n} lines")
