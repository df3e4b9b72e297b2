"""Where a wave's cycles go in the two forms of the fused 16-state ViterbiNet kernel: a -DMVN_DIAG_PHASES build (tools/dbg/
libmvn_phases.so, built here; wrong final metrics by construction) accumulates s_memtime differences per phase and writes them
over the final-metric rows.  usage: diag_phases.py build | run [B]"""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "dbg", "libmvn_phases.so")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                    "-DMVN_DIAG_PHASES", os.path.join(ROOT, "meta-viterbinet_amd", "csrc", "mvn_hip.hip"), "-o", SO], check=True)
    sys.exit(0)
os.environ["MVN_LIB_PATH"] = SO
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import meta_viterbinet_amd as mvn  # noqa: E402

B = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
T = 1000
dev = torch.device("cuda:0")
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [torch.tensor(g7[f"w{i}"], device=dev) for i in range(6)]
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
_, y = mvn.synthetic_words(B, T, 4, 10.0, 0.2, dev, seed=1)
dec = torch.empty(B, T, device=dev)
fm = torch.zeros(B, 16, device=dev)
nb = int(lib.mvn_vnet_workspace_bytes(B, T, 16))
ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
for name, use_ws in (("one wave per block", False), ("dealt", True)):
    for _ in range(30):
        rc = lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *[mvn._lib.ptr(a) for a in w], mvn._lib.ptr(dec), T, None, mvn._lib.ptr(fm),
                                     mvn._lib.ptr(ws) if use_ws else None, nb if use_ws else 0, B, T, 16, st)
        assert rc == 0
    torch.cuda.synchronize()
    o = fm.cpu().numpy().view(np.uint64).reshape(B, 8)
    o = o[o[:, 5] > 0]
    units = o[:, 5].astype(np.float64)
    print(f"{name}: {len(o)} waves, units per wave {units.mean():.1f}; cycles per 32-symbol unit (s_memtime, 100 MHz ticks x 24 = core cycles approx):")
    labels = ["k-loop", "tile phase"] if not use_ws else ["k-loop", "pass A (layer 3)", "wait for metrics", "pass B (sweeps + hand-on)", "pass C (decisions)"]
    tot = 0.0
    for i, lab in enumerate(labels):
        v = (o[:, i].astype(np.float64) / units)
        tot += v.mean()
        print(f"   {lab:28s} mean {v.mean():9.1f}  p10 {np.percentile(v, 10):9.1f}  p90 {np.percentile(v, 90):9.1f}")
    print(f"   {'sum':28s} mean {tot:9.1f}")
    if use_ws:
        print(f"   {'between units':28s} mean {(o[:, 6] / units).mean():9.1f}")
    life_c, life_r = (o[:, 7] >> np.uint64(24)).astype(np.float64), (o[:, 7] & np.uint64(0xffffff)).astype(np.float64)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *[mvn._lib.ptr(x) for x in w], mvn._lib.ptr(dec), T, None, mvn._lib.ptr(fm),
                                mvn._lib.ptr(ws) if use_ws else None, nb if use_ws else 0, B, T, 16, st)
    b.record()
    b.synchronize()
    print(f"   wave lifetime after the prologue: {life_c.mean():.0f} s_memtime ticks = {life_r.mean() / 100:.1f} us (s_memrealtime, 100 MHz) -> "
          f"{life_c.mean() / (life_r.mean() / 100) / 1e3:.3f} GHz; max {life_r.max() / 100:.1f} us; launch (events): {a.elapsed_time(b) / 10 * 1e3:.1f} us")
