#!/usr/bin/env python3
"""Duration of consecutive launches of the fused ViterbiNet kernel from a cold start: the clock / power state of the GPU
settles over tens of milliseconds of sustained load (so bench.py's warm-up must be that long)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
B, T, S, L = 10000, 1000, 16, 4
g = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [torch.tensor(g[f"w{i}"], device=dev) for i in range(6)]
tx, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=3450002)
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
dec = torch.zeros(B, T, device=dev)
wp = [mvn._lib.ptr(t) for t in w]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *wp, mvn._lib.ptr(dec), T, None, None, None, 0, B, T, S, st)
    ev[i + 1].record()
torch.cuda.synchronize()
d = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for lo in range(0, n, 20):
    print(f"launches {lo:3d}..{lo+19:3d} (t = {sum(d[:lo]):6.1f} ms): mean {np.mean(d[lo:lo+20]):.4f} ms  min {np.min(d[lo:lo+20]):.4f}")
