#!/usr/bin/env python3
"""Timing: 200 self-supervised iterations on one word (vnet_trainer.py:49-60) at 16 ... 128 trellis states: the one-launch HIP kernel
(online_train_kernel<SC>; 64 / 128 states: round 5) vs the same loop on stock autograd (OnlineTrainer(use_kernel=False), what
64 / 128 states ran on before).  usage: time_online_states.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
T, N = 136, 200
for L in (4, 5, 6, 7):
    S = 2 ** L
    torch.manual_seed(L)
    tx, rx = torch.randint(0, 2, (1, T)).float().to(dev), torch.randn(1, T, device=dev)
    out = []
    for full_word in (False, True):
        for use_kernel in (True, False):
            det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
            tr = mvn.OnlineTrainer(det, L, use_kernel=use_kernel)
            tr.online_training(tx, rx, iterations=N if use_kernel else 10, full_word=full_word)
            torch.cuda.synchronize()
            reps = 5 if use_kernel else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                tr.online_training(tx, rx, iterations=N, full_word=full_word)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / reps / N * 1e6)
    print(f"S {S:4d}: minibatch of 32: kernel {out[0]:7.1f} us / iteration, autograd {out[1]:7.1f} ({out[1] / out[0]:5.1f} x)   "
          f"whole word: kernel {out[2]:7.1f}, autograd {out[3]:7.1f} ({out[3] / out[2]:5.1f} x)", flush=True)
