#!/usr/bin/env python3
"""What each phase of the training kernels' 32-sample chunk costs IN PRODUCTION TIMING: builds of the library with one phase's
body compiled out (-DMVN_ABLATE=k; its barrier stays; results are wrong by construction) timed against the full kernel --
the difference is what the phase adds to an iteration, overlap with its neighbours included (the s_memtime timeline of
tools/prof_train_phases.py inflates every phase by its stamps).  usage: ablate_train.py build | run"""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "tools", "dbg")
NAMES = ["h1 = sigmoid(W1 y + b1)", "z2 = h1 W2^T", "logits", "softmax / CE / dlogits", "dz2, dW3, db3, loss", "dz1, dW2 (units 0..47)",
         "dW1, db1, db2, dW2 units 48, 49", "Adam"]
if sys.argv[1] == "build":
    os.makedirs(DBG, exist_ok=True)
    procs = []
    for k in range(8):
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                                       f"-DMVN_ABLATE={k}", os.path.join(ROOT, "meta-viterbinet_amd", "csrc", "mvn_hip.hip"), "-o",
                                       os.path.join(DBG, f"libmvn_abl{k}.so")], stderr=subprocess.DEVNULL))
        if len(procs) == 4:
            for p in procs:
                p.wait()
            procs = []
    for p in procs:
        p.wait()
    sys.exit(0)

rows = {}
for k in [None] + list(range(8)):
    env = dict(os.environ)
    if k is not None:
        env["MVN_LIB_PATH"] = os.path.join(DBG, f"libmvn_abl{k}.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_train_kernels.py"), "3", "--no-check"], env=env, capture_output=True, text=True).stdout
    vals = [float(l.split(",")[-2]) for l in out.splitlines() if l.startswith('"')]
    rows[k] = vals
    print(("full kernel" if k is None else f"without {NAMES[k]}").ljust(44), " ".join(f"{v:8.2f}" for v in vals), flush=True)
print("\nus per iteration and trial slot: minibatch | whole word | whole word chunked | second-order step | second-order chunked")
print("what a phase adds (full - without):")
for k in range(8):
    print(f"  {NAMES[k]:40s}", " ".join(f"{a - b:8.2f}" for a, b in zip(rows[None], rows[k])))
