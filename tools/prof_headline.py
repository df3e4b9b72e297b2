"""Workload for rocprofv3 (--kernel-trace / --pmc): the headline kernel alone -- mvn_vnet_decode_f32 at 16 states, B blocks x
1000 symbols (BASELINE configs[1]: B = 10 000), a few launches after the clock has settled.  usage: prof_headline.py B [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
T = 1000
g = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
with torch.no_grad():
    for p, i in zip(det.parameters(), range(6)):
        p.copy_(torch.tensor(g[f"w{i}"]))
tx, y = mvn.synthetic_words(B, T, 4, 10.0, 0.2, dev, seed=3450002)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.06:  # clock settle (bench.py does the same)
    det(y, "val")
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    dec = det(y, "val")
b.record()
b.synchronize()
print(f"vnet16 fused decode: {B} blocks x {T}: {a.elapsed_time(b) / reps:.4f} ms per launch, {B * T * reps / (a.elapsed_time(b) * 1e-3):.4g} symbols/s")
