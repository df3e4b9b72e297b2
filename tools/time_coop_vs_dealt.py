"""The 16-state ViterbiNet detector on SMALL batches: the cooperative kernel (a 16-wave workgroup per block, MVN_COOP=1) against
the dealt kernel (MVN_COOP=0: one 8-wave group per block below 768 blocks) -- where the default should switch.
usage: time_coop_vs_dealt.py [T] [B ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [torch.tensor(g7[f"w{i}"], device=dev) for i in range(6)]
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Bs = [int(a) for a in sys.argv[2:]] or [1, 8, 32, 64, 100, 128, 192, 256, 384, 512, 640, 768]
_, ymax = mvn.synthetic_words(max(Bs), T, 4, 10.0, 0.2, dev, seed=1)
dec = torch.empty(max(Bs), T, device=dev)
wp = [mvn._lib.ptr(a) for a in w]
print(f"T = {T}\nblocks   coop ms    dealt ms   coop/dealt")
for B in Bs:
    res = []
    for coop in ("1", "0"):
        os.environ["MVN_COOP"] = coop
        lib.mvn_reload_switches()
        nb = int(lib.mvn_vnet_workspace_bytes(B, T, 16))
        ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)

        def call():
            assert lib.mvn_vnet_decode_f32(mvn._lib.ptr(ymax), T, *wp, mvn._lib.ptr(dec), T, None, None, mvn._lib.ptr(ws), nb, B, T, 16, st) == 0

        ts = []
        for rep in range(4):
            for _ in range(5):
                call()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                call()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) / 20)
        res.append(min(ts))
    print(f"{B:6d}   {res[0]:8.4f}   {res[1]:8.4f}   {res[0] / res[1]:.3f}", flush=True)
