"""Launch time of the 16-state ViterbiNet detector (mvn_vnet_decode_f32, T = 1000) against the number of blocks, for the two forms
of the fused kernel: one wave per block (vnet16_fusedn_kernel: no workspace passed) and dealt in 32-symbol units
(vnet16_dealt_kernel: the workspace mvn_vnet_workspace_bytes asks for).  Fraction = 12.0 kFLOP per symbol / time / 157.3 TFLOP/s.
usage: time_dealt.py [B ...]"""
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
T = int(os.environ.get("MVN_TIME_T", "1000"))
Bs = [int(a) for a in sys.argv[1:]] or [800, 1024, 1250, 1536, 2048, 3072, 4096, 6144, 8192, 9216, 10000, 10240, 12288, 16384, 20480, 40960]
_, ymax = mvn.synthetic_words(max(Bs), T, 4, 10.0, 0.2, dev, seed=1)
dec = torch.empty(max(Bs), T, device=dev)
wp = [mvn._lib.ptr(a) for a in w]


def call(B, ws, nb):
    rc = lib.mvn_vnet_decode_f32(mvn._lib.ptr(ymax), T, *wp, mvn._lib.ptr(dec), T, None, None, mvn._lib.ptr(ws), nb, B, T, 16, st)
    assert rc == 0


for _ in range(60):  # settle the clocks
    call(10000, None, 0)
torch.cuda.synchronize()
print(f"T = {T}\nblocks   one wave per block: ms  frac    dealt: ms  frac   speed-up   (cycles/symbol/SIMD @2.385 GHz: plain, dealt)")
for B in Bs:
    nb = int(lib.mvn_vnet_workspace_bytes(B, T, 16))
    ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
    res = []
    for use_ws in (False, True):
        ts = []
        for rep in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            call(B, ws if use_ws else None, nb if use_ws else 0)
            a.record()
            for _ in range(10):
                call(B, ws if use_ws else None, nb if use_ws else 0)
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) / 10)
        res.append(min(ts))
    f = [12.0e3 * B * T / (ms * 1e-3) / 157.3e12 for ms in res]
    cyc = [ms * 1e-3 * 2.385e9 * 1024 / (B * T) for ms in res]
    print(f"{B:6d}   {res[0]:8.4f}  {f[0]:.3f}    {res[1]:8.4f}  {f[1]:.3f}   {res[0] / res[1]:.3f}     {cyc[0]:6.1f} {cyc[1]:6.1f}", flush=True)
