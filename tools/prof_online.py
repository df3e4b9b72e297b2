#!/usr/bin/env python3
"""Phase breakdown of the online-training kernel (needs a -DMVN_TRAIN_PROFILE build: MVN_LIB_PATH=...)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
T, L, S = 136, 4, 16
torch.manual_seed(0)
det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
tr = mvn.OnlineTrainer(det, L)
tx = torch.randint(0, 2, (1, T), device=dev).float()
rx = torch.randn(1, T, device=dev)
loss = tr.online_training(tx, rx, iterations=20, return_loss=True)
torch.cuda.synchronize()
names = ["load chunk", "h1 sigmoid", "z2 mfma", "logits mfma", "CE", "dz2/dW3/db3", "dz1/dW2/db2", "dW1/db1", "adam"]
vals = loss.tolist() if hasattr(loss, "tolist") else list(loss)
for n, v in zip(names, vals):
    print(f"{n:14s} {v:9.0f} ticks")
print("sum", sum(vals[:len(names)]))
