#!/usr/bin/env python3
"""Timing of the optional survivor outputs (SURVEY 8 row †): mvn_acs_sweep_surv_f32 / mvn_va_decode_surv_f32 + mvn_traceback_f32
against the sweeps without survivors.  usage: time_survivors.py [B]   (T = 1000)"""
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
T = 1000


def ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


for S, B in ((16, int(sys.argv[1]) if len(sys.argv) > 1 else 10000), (64, 2500), (256, 1000)):
    torch.manual_seed(S)
    cost = torch.randn(B, T, S, device=dev)
    t0 = ms(lambda: mvn.acs_sweep(cost))
    t1 = ms(lambda: mvn.acs_sweep_survivors(cost))
    dec, fm, surv = mvn.acs_sweep_survivors(cost)
    t2 = ms(lambda: mvn.traceback(surv, fm))
    gb0, gb1 = B * T * (4 * S + 4) / 1e9, B * T * (4 * S + 4 + max(1, S // 8)) / 1e9
    print(f"S {S:4d}  {B} x {T}: sweep {t0:7.3f} ms ({gb0 / t0:6.2f} TB/s)   with survivors {t1:7.3f} ms ({gb1 / t1:6.2f} TB/s, {t1 / t0:4.2f} x)   "
          f"traceback {t2:7.3f} ms", flush=True)

CC = {"train": "time_decay", "val": "time_decay"}
for L, B in ((4, 10000), (8, 20000)):
    S = 2 ** L
    tx, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=5)
    va = mvn.VADetector(S, L, T, 1, "ISI_AWGN", 0, False, 1, CC)
    t0 = ms(lambda: va(y, "val", 10.0, 0.2))
    t1 = ms(lambda: va.viterbi_path(y, 10.0, 0.2))
    print(f"VA S {S:4d}  {B} x {T}: forward('val') {t0:7.3f} ms   viterbi_path (sweep with survivors + traceback) {t1:7.3f} ms ({t1 / t0:4.2f} x)", flush=True)

import numpy as np  # noqa: E402

g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
B, S, L = 10000, 16, 4
det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
with torch.no_grad():
    for p, i in zip(det.parameters(), range(6)):
        p.copy_(torch.tensor(g7[f"w{i}"]))
tx, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=5)
t0 = ms(lambda: det(y, "val"))
t1 = ms(lambda: det.viterbi_path(y))
bits, dec = det.viterbi_path(y), det(y, "val")
print(f"ViterbiNet S {S}  {B} x {T}: forward('val') {t0:7.3f} ms   viterbi_path (fused detector with survivors + traceback) {t1:7.3f} ms ({t1 / t0:4.2f} x)   "
      f"bit errors at 10 dB: running argmin {int((dec[:, :T - L] != tx[:, :T - L]).sum())}, traceback {int((bits[:, :T - L] != tx[:, :T - L]).sum())}", flush=True)
