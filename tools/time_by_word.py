#!/usr/bin/env python3
"""Per-block latency of the sequential online evaluation (BASELINE configs[2]/[4]: 300 blocks, T=136, RS(17,15))."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [g7[f"w{i}"] for i in range(6)]
N, K, nsym, L, snr = 300, 120, 2, 4, 10.0
gen = torch.Generator(device=dev).manual_seed(5)
msg = torch.randint(0, 2, (N, K), generator=gen, device=dev).float()
cw = mvn.rs_encode(msg, nsym)
h = np.concatenate([mvn.estimate_channel(L, 0.2, "time_decay", fading=True, index=i, fading_taps_type=2) for i in range(N)])
y = mvn.transmit(cw, h, snr, L, torch.randn(N, K + 8 * nsym, generator=gen, device=dev))


def make():
    det = mvn.VNETDetector(16, {"train": 136, "val": 136}).to(dev)
    with torch.no_grad():
        for p, a in zip(det.parameters(), w):
            p.copy_(torch.tensor(a))
    return det


for name, kw in (("joint (no updates)", {}), ("self-supervised, 200 it/block", {"self_supervised": True, "self_supervised_iterations": 200})):
    det = make()
    if kw:
        kw["online_trainer"] = mvn.OnlineTrainer(det, L)
    mvn.eval_by_word(det, msg[:5], y[:5], snr, 0.2, nsym, 25, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ser = mvn.eval_by_word(det, msg, y, snr, 0.2, nsym, 25, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"eval_by_word {name}: {dt*1e3:.1f} ms for {N} blocks = {dt/N*1e6:.0f} us/block, mean ser {ser.mean():.4f}")
# BASELINE configs[4]: Meta-ViterbiNet online evaluation with the reference's defaults (every 5 blocks: 20 x 10 MAML
# meta-steps through torch autograd; after every qualifying block: 200 full-word iterations of the HIP training kernel)
NM = 60
for name, kw in (("HIP meta-learning kernel", {}), ("torch autograd replayed from a hipGraph", {"hip_meta": False}),
                 ("eager torch autograd", {"hip_meta": False, "graphed_meta": False})):
    det = make()
    meta = mvn.META_VNETDetector(16, {"train": 136, "val": 136})
    tr = mvn.OnlineTrainer(det, L)
    torch.manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ser = mvn.eval_by_word(det, msg[:NM], y[:NM], snr, 0.2, nsym, 25, self_supervised=True, online_trainer=tr,
                           self_supervised_iterations=200, online_meta=True, meta_detector=meta, meta_train_iterations=20,
                           meta_j_num=10, meta_subframes=5, meta_style_online_training=True, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"eval_by_word online meta, MAML steps by {name}: {dt*1e3:.0f} ms for {NM} blocks = {dt/NM*1e3:.1f} ms/block, "
          f"mean ser {ser.mean():.4f}", flush=True)
det = make()
torch.cuda.synchronize()
t0 = time.perf_counter()
dec = mvn.rs_decode(det(y, "val"), nsym)
ser = (dec != msg).float().mean(dim=1)
torch.cuda.synchronize()
print(f"batched joint variant (one decode + one RS launch for all {N} blocks): {(time.perf_counter()-t0)*1e3:.2f} ms")
