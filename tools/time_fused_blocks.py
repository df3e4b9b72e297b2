"""Launch time of the fused ViterbiNet kernel (mvn_vnet_decode_f32, S = 16, T = 1000) against the number of blocks: where
the end-of-launch tail of BASELINE configs[1] (10 000 blocks = 39.06 per CU on 256 CUs) comes from.  A block is one wave's
indivisible work item (its trellis sweep is a serial chain), so a launch cannot end before the CU with the most blocks has
finished: ceil(B / 256 / 8) * 8 blocks of 1000 symbols.  usage: time_fused_blocks.py [B ...]"""
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
T = 1000
Bs = [int(a) for a in sys.argv[1:]] or [2048, 4096, 6144, 8192, 9216, 10000, 10240, 12288, 14336, 16384, 18432, 20480, 40960]
ymax = torch.randn(max(Bs), T, device=dev)
dec = torch.empty(max(Bs), T, device=dev)
# settle the clocks
for _ in range(60):
    lib.mvn_vnet_decode_f32(mvn._lib.ptr(ymax), T, *[mvn._lib.ptr(a) for a in w], mvn._lib.ptr(dec), T, None, None, None, 0, 10000, T, 16, st)
torch.cuda.synchronize()
print("blocks  blocks/CU  max blocks on a CU   ms     cycles/symbol/SIMD @2.4GHz   ms per (max blocks on a CU)/4")
for B in Bs:
    ts = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            lib.mvn_vnet_decode_f32(mvn._lib.ptr(ymax), T, *[mvn._lib.ptr(x) for x in w], mvn._lib.ptr(dec), T, None, None, None, 0, B, T, 16, st)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    ms = min(ts)
    wgs = (B + 7) // 8
    max_cu = -(-wgs // 256) * 8
    print(f"{B:6d}  {B / 256:8.2f}  {max_cu:6d}            {ms:7.4f}  {ms * 1e-3 * 2.4e9 * 1024 / (B * T):8.1f}                    {ms / (max_cu / 4):.5f}")
