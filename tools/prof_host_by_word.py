"""Where the HOST time of one sequential harness.eval_by_word goes (cProfile; the GPU work is asynchronous, so what shows up is
Python / launch overhead and the per-block synchronisation).  usage: prof_host_by_word.py [flow number 0|1] [snr]"""
import cProfile
import pstats
import sys
import time

import torch

from trial_setup import FLOWS, L, T, dev, mvn, nsym, sub, w, words
from meta_viterbinet_amd.trials import TrialDraws

name, (coef, kw) = list(FLOWS.items())[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
msg, rx = words(coef, snr, 200)


def make():
    det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, a in zip(det.parameters(), w):
            p.copy_(torch.as_tensor(a))
    return det, mvn.OnlineTrainer(det, L), mvn.META_VNETDetector(16, {"train": T, "val": T}), TrialDraws(200, dev)


def run():
    det, tr, md, dr = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ser = mvn.eval_by_word(det, msg, rx, snr, 0.2, nsym, sub, online_trainer=tr, meta_detector=md, draws=dr, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, ser


run()
ms, ser = run()
print(f"{name} at {snr} dB: {ms:.1f} ms = {ms / len(ser):.3f} ms per block, mean ser {ser.mean():.5f}")
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
