#!/usr/bin/env python3
"""One Monte-Carlo point of the classical Viterbi detector: the fused launch (mvn_va_montecarlo_f32: words generated inside the
detector, nothing but four counters written) against generate_words -> forward('val') -> count_errors.  usage: time_montecarlo.py"""
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
CC = {"train": "time_decay", "val": "time_decay"}


def ms(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


for L, B in ((4, 100), (4, 10000), (4, 100000), (8, 2000), (8, 125000)):
    S, T, snr, gamma = 2 ** L, 1000, 10.0, 0.2
    det = mvn.VADetector(S, L, T, 1, "ISI_AWGN", 0, False, 1, CC)
    h = det._estimate_all(gamma, "val")

    def three():
        tx, y = mvn.generate_words(B, T, h, snr, L, dev, 7)
        return mvn.count_errors(det(y, "val", snr, gamma), tx)

    want, got = three(), mvn.va_monte_carlo(det, B, snr, gamma, dev, 7)
    t3, t1 = ms(three), ms(lambda: mvn.va_monte_carlo(det, B, snr, gamma, dev, 7))
    tx, y = mvn.generate_words(B, T, h, snr, L, dev, 7)
    td = ms(lambda: det(y, "val", snr, gamma))
    print(f"S {S:4d}  {B:7d} x {T}: three launches {t3:8.3f} ms (the decode alone {td:8.3f})   fused {t1:8.3f} ms = {B * T / t1 / 1e6:8.2f} Gsym/s "
          f"({t3 / t1:4.2f} x)   device memory for words and decisions {12 * B * T / 1e6:8.1f} MB -> 0   counters equal {got.tolist() == want.tolist()}  "
          f"ser {got[0].item() / got[1].item():.5f}", flush=True)
    del tx, y
