#!/usr/bin/env python3
"""Timing: 200 self-supervised iterations on one word (vnet_trainer.py:49-60): one-launch HIP kernel vs the same loop
in eager PyTorch-ROCm (what the reference does on a GPU) and on the host CPU."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
T, L, S, N = 136, 4, 16, 200
torch.manual_seed(0)
tx = torch.randint(0, 2, (1, T)).float()
rx = torch.randn(1, T)


def eager(device, n):
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(device)
    opt = torch.optim.Adam(det.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    gt = mvn.calculate_states(L, tx.to(device))
    r = rx.to(device)
    t0 = time.perf_counter()
    for _ in range(n):
        soft = det(r, "train").reshape(-1, S)
        ind = torch.multinomial(torch.arange(gt.shape[0], device=device).float(), 32).long()
        loss = crit(soft[ind], gt[ind])
        for p in det.parameters():
            p.grad = None
        loss.backward()
        opt.step()
    if device.type == "cuda":
        torch.cuda.synchronize()
    return time.perf_counter() - t0


eager(dev, 20)
t_eager = eager(dev, N)
t_cpu = eager(torch.device("cpu"), N)
det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
tr = mvn.OnlineTrainer(det, L)
tr.online_training(tx.to(dev), rx.to(dev), iterations=N)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    tr.online_training(tx.to(dev), rx.to(dev), iterations=N)
torch.cuda.synchronize()
t_hip = (time.perf_counter() - t0) / reps
print(f"{N} online-training iterations on one word (T={T}, minibatch 32):")
print(f"  HIP one-launch kernel : {t_hip*1e3:8.3f} ms  ({t_hip/N*1e6:.1f} us/iteration)")
print(f"  eager PyTorch-ROCm    : {t_eager*1e3:8.3f} ms  ({t_eager/N*1e6:.1f} us/iteration)  -> {t_eager/t_hip:.1f}x")
print(f"  eager PyTorch CPU     : {t_cpu*1e3:8.3f} ms  ({t_cpu/N*1e6:.1f} us/iteration)  -> {t_cpu/t_hip:.1f}x")

# ---- the MAML meta-learning step (trainer.py:425-453): 200 steps on buffered words
NS, NW = 200, 12
gen = torch.Generator(device=dev).manual_seed(1)
rxw = torch.randn(NW, T, generator=gen, device=dev)
txw = torch.randint(0, 2, (NW, T), generator=gen, device=dev).float()
sup = torch.randint(0, NW, (NS, 1), generator=gen, device=dev)
qry = torch.randint(0, NW, (NS,), generator=gen, device=dev)
meta = mvn.META_VNETDetector(S, {"train": T, "val": T})


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
tr = mvn.OnlineTrainer(det, L)
t_full = timed(lambda: tr.online_training(tx.to(dev), rx.to(dev), iterations=N, full_word=True))
t_maml = timed(lambda: tr.maml_training(rxw, txw, sup, qry, 0.1, True))
t_fo = timed(lambda: tr.maml_training(rxw, txw, sup, qry, 0.1, False))
graphed = mvn.GraphedMetaStep(det, meta, tr, 1, T, 0.1, True)


def run_graphed():
    for k in range(NS):
        graphed(rxw, txw, sup[k], qry[k:k + 1])


def run_eager():
    for k in range(NS):
        mvn.meta_train_loop(det, meta, tr, rxw, txw, sup[k], qry[k:k + 1], 0.1, True)


t_graph, t_eag = timed(run_graphed), timed(run_eager)
print(f"{N} full-word training iterations (T={T} samples each), HIP kernel: {t_full*1e3:.3f} ms ({t_full/N*1e6:.1f} us/iteration)")
print(f"{NS} MAML meta-learning steps (support and query word of {T} symbols):")
print(f"  HIP one-launch kernel, second order : {t_maml*1e3:8.3f} ms  ({t_maml/NS*1e6:.0f} us/step)")
print(f"  HIP one-launch kernel, first order  : {t_fo*1e3:8.3f} ms  ({t_fo/NS*1e6:.0f} us/step)")
print(f"  torch autograd from a hipGraph      : {t_graph*1e3:8.3f} ms  ({t_graph/NS*1e6:.0f} us/step)  -> {t_graph/t_maml:.1f}x")
print(f"  eager torch autograd (ROCm)         : {t_eag*1e3:8.3f} ms  ({t_eag/NS*1e6:.0f} us/step)  -> {t_eag/t_maml:.1f}x")
