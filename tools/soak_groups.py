#!/usr/bin/env python3
"""Soak run of the chunked training launches (not part of the pytest suite; `python tools/soak_groups.py [seconds] [seed]` on a GPU
box): random state counts, word lengths, iteration / step counts and support windows; every case runs on the XCD-aware grid (a
trial's workgroups on one XCD, gradient exchange through its L2), on the (groups, trials) grid (write-through exchange) and on
ONE workgroup, single trials and a few batched ones -- weights, both Adam moments and the reported losses must be IDENTICAL bit
for bit across the three, and no status word may be set."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
FORMS = (("1", "1"), ("1", "0"), ("0", "1"))  # (MVN_TRAIN_GROUPS, MVN_TRAIN_XCD)


def rand_weights(S):
    return [(rng.uniform(-1, 1, (100, 1))).astype(np.float32), rng.uniform(-1, 1, 100).astype(np.float32),
            rng.uniform(-0.1, 0.1, (50, 100)).astype(np.float32), rng.uniform(-0.1, 0.1, 50).astype(np.float32),
            rng.uniform(-0.14, 0.14, (S, 50)).astype(np.float32), rng.uniform(-0.14, 0.14, S).astype(np.float32)]


def detector(w, S, T):
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, a in zip(det.parameters(), w):
            p.copy_(torch.tensor(a))
    return det


n, kinds, t_end = 0, {"online": 0, "maml": 0}, time.time() + budget
while time.time() < t_end:
    S = int(rng.choice([4, 8, 16, 16, 16, 32]))
    L = int(np.log2(S))
    T = int(rng.choice([40, 72, 100, 136, 136, 200, 264, 520]))
    w = rand_weights(S)
    kind = "online" if rng.rand() < 0.5 else "maml"
    results = []
    if kind == "online":
        tx = torch.tensor(rng.randint(0, 2, (1, T)).astype(np.float32), device=dev)
        y = torch.tensor(rng.normal(0, 1.5, (1, T)).astype(np.float32), device=dev)
        its = [int(rng.randint(1, 60)), int(rng.randint(1, 30))]
    else:
        W, MAML = int(rng.choice([1, 1, 2])), bool(rng.rand() < 0.7)
        words = 6
        rxw = torch.tensor(rng.normal(0, 1.5, (words, T)).astype(np.float32), device=dev)
        txw = torch.tensor(rng.randint(0, 2, (words, T)).astype(np.float32), device=dev)
        steps = int(rng.randint(1, 12))
        sup = torch.tensor(rng.randint(0, words, (steps, W)), device=dev)
        qry = torch.tensor(rng.randint(0, words, steps), device=dev)
    for groups, xcd in FORMS:
        os.environ["MVN_TRAIN_GROUPS"], os.environ["MVN_TRAIN_XCD"] = groups, xcd
        mvn._lib.reload_switches()
        det = detector(w, S, T)
        tr = mvn.OnlineTrainer(det, L)
        if kind == "online":
            out = [tr.online_training(tx, y, iterations=k, full_word=True, return_loss=True) for k in its]
        else:
            out = [tr.maml_training(rxw, txw, sup, qry, 0.1, MAML, return_loss=True), tr.maml_training(rxw, txw, sup[:1], qry[:1], 0.1, MAML, return_loss=True)]
        tr.check_status()
        results.append([p.detach().clone() for p in det.parameters()] + [tr.exp_avg.clone(), tr.exp_avg_sq.clone()] + out)
    torch.cuda.synchronize()
    tag = f"{kind} S={S} T={T}"
    for a, b, c in zip(*results):
        assert torch.equal(a, b) and torch.equal(a, c), tag
    assert all(bool(torch.isfinite(t).all()) for t in results[0]), tag
    n += 1
    kinds[kind] += 1
for k in ("MVN_TRAIN_GROUPS", "MVN_TRAIN_XCD"):
    os.environ.pop(k, None)
mvn._lib.reload_switches()
print(f"soak_groups: {n} random cases, three launch forms each, bit-identical in {budget:.0f} s  {kinds}")
