#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: y arrives in pinned host memory and the decisions go back to it.
(a) serial: H2D copy, decode, D2H copy per batch; (b) two batches in flight on two streams (copies overlap the kernel)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
B, T, S, L = 10000, 1000, 16, 4
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
with torch.no_grad():
    for i, p in enumerate(det.parameters()):
        p.copy_(torch.tensor(g7[f"w{i}"]))
_, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=1)
NB = 8  # batches per measurement
h_y = [y.cpu().pin_memory() for _ in range(2)]
h_dec = [torch.empty(B, T).pin_memory() for _ in range(2)]
d_y = [torch.empty(B, T, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]


def batch(k, stream):
    with torch.cuda.stream(stream):
        d_y[k].copy_(h_y[k], non_blocking=True)
        dec = det(d_y[k], "val")
        h_dec[k].copy_(dec, non_blocking=True)


def run(n_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(NB):
        k = i % n_streams
        batch(k, streams[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / NB * 1e3


for n_streams, name in ((1, "serial (one stream)"), (2, "two batches in flight")):
    run(n_streams)
    ms = min(run(n_streams) for _ in range(3))
    print(f"{name:24s}: {ms:.3f} ms per batch of {B} x {T}  ->  {B*T/ms/1e6:.2f} Gsym/s host to host "
          f"({8*B*T/ms/1e6:.1f} GB/s over PCIe, both directions)")
