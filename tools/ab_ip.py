#!/usr/bin/env python3
"""A/B of vnet_fused_ip_kernel's workgroup shapes (waves per workgroup x workgroups per CU the kernel is compiled for, per state
count) and of its switches: variant builds of the library in tools/dbg/ (git-ignored; travels to the GPU box), each timed on the same
box by tools/time_vnet_states.py through MVN_LIB_PATH.  usage: ab_ip.py build | run [S,B ...]"""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "tools", "dbg")


def shape(s48, s32, s64, *flags):
    """(waves, workgroups per CU) at S = 4 / 8, 32, 64"""
    w = f"((LB)<=1?{s48[0]}:(LB)==3?{s32[0]}:(LB)==4?{s64[0]}:4)"
    g = f"((LB)<=1?{s48[1]}:(LB)==3?{s32[1]}:(LB)==4?{s64[1]}:1)"
    return (w, g, list(flags))


VARIANTS = {
    "the build in the tree (4x5 4x5 4x3 4x2)": None,
    "priorities rotated by workgroup": shape((4, 5), (4, 3), (4, 2), "-DMVN_IP_PRIO=2"),
    "alias at every S": shape((4, 5), (4, 3), (4, 2), "-DMVN_IP_ALIAS(LB)=1"),
    "no alias": shape((4, 5), (4, 3), (4, 2), "-DMVN_IP_ALIAS(LB)=0"),
}
if sys.argv[1] == "build":
    os.makedirs(DBG, exist_ok=True)
    procs, rc = [], 0
    for k, v in enumerate(VARIANTS.values()):
        if v is None:
            continue
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                                       f"-DMVN_IP_WAVES(LB)={v[0]}", f"-DMVN_IP_WGS(LB)={v[1]}", *v[2],
                                       os.path.join(ROOT, "meta-viterbinet_amd", "csrc", "mvn_hip.hip"), "-o", os.path.join(DBG, f"libmvn_ip{k}.so")],
                                      stderr=subprocess.DEVNULL))
        if len(procs) == 4:
            rc = max([rc] + [p.wait() for p in procs])
            procs = []
    sys.exit(max([rc] + [p.wait() for p in procs]))

for k, (name, v) in enumerate(VARIANTS.items()):
    env = dict(os.environ)
    if v is not None:
        env["MVN_LIB_PATH"] = os.path.join(DBG, f"libmvn_ip{k}.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_vnet_states.py"), *sys.argv[2:]], env=env, capture_output=True, text=True)
    print(f"== {name}", flush=True)
    print("\n".join(l.split("   two kernels")[0] for l in out.stdout.splitlines() if l.startswith("S ")) or out.stderr[-2000:], flush=True)
