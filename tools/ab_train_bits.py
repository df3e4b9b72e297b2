"""Bit-level A/B of two builds of the library on the training kernels: prints a digest of the weights / Adam moments / losses
after online training (minibatch and full word, one workgroup and one per chunk) and meta-learning steps for several S.
usage: MVN_LIB_PATH=<lib.so> python tools/ab_train_bits.py   (run once per build, compare the lines)"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
h = hashlib.sha256()
for S, T in ((16, 136), (16, 40), (4, 100), (8, 136), (32, 100), (2, 64)):
    L = int(np.log2(S))
    rng = np.random.RandomState(S + T)
    w = [(rng.uniform(-1, 1, (100, 1))).astype(np.float32), rng.uniform(-1, 1, 100).astype(np.float32),
         rng.uniform(-0.1, 0.1, (50, 100)).astype(np.float32), rng.uniform(-0.1, 0.1, 50).astype(np.float32),
         rng.uniform(-0.14, 0.14, (S, 50)).astype(np.float32), rng.uniform(-0.14, 0.14, S).astype(np.float32)]
    gen = torch.Generator(device=dev).manual_seed(S)
    rxw = torch.randn(6, T, generator=gen, device=dev)
    txw = torch.randint(0, 2, (6, T), generator=gen, device=dev).float()
    for groups in ("1", "0"):
        os.environ["MVN_TRAIN_GROUPS"] = groups
        mvn._lib.reload_switches()
        det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
        with torch.no_grad():
            for p, a in zip(det.parameters(), w):
                p.copy_(torch.tensor(a))
        tr = mvn.OnlineTrainer(det, L)
        bi = torch.stack([torch.randperm(T - 1, generator=gen, device=dev)[:32] + 1 for _ in range(30)]).to(torch.int32)
        l1 = tr.online_training(txw[:1], rxw[:1], iterations=30, batch_idx=bi, return_loss=True)
        l2 = tr.online_training(txw[1:2], rxw[1:2], iterations=20, full_word=True, return_loss=True)
        sup = torch.arange(8, device=dev).reshape(-1, 1) % 6
        l3 = tr.maml_training(rxw, txw, sup, (torch.arange(8, device=dev) + 1) % 6, 0.1, True, return_loss=True)
        torch.cuda.synchronize()
        for t in list(det.parameters()) + [tr.exp_avg, tr.exp_avg_sq, l1, l2, l3]:
            h.update(t.detach().cpu().numpy().tobytes())
    print(S, T, h.hexdigest()[:16])
