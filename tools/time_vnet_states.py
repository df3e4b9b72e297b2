"""ViterbiNet 'val' at state counts other than 16: the fused kernel (vnet_fused_ip_kernel<LB>: MLP inside the in-place sweep) against
the two-kernel route it replaces (mlp_kernel -> [B, T, S] logits in HBM -> sweep_inplace_kernel; MVN_UNFUSED=1).
usage: time_vnet_states.py [S,B ...]   (T = 1000)"""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
T = 1000
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, 10000), (8, 10000), (32, 10000), (64, 10000), (128, 4000), (256, 2000)]
lib = mvn._lib.load()
for S, B in cases:
    torch.manual_seed(S)
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    y = torch.randn(B, T, device=dev) * 1.5
    res = {}
    for unfused in ("0", "1"):
        os.environ["MVN_UNFUSED"], os.environ["MVN_FUSED_IP"] = unfused, "1"  # (128 / 256 states: fused on request only)
        lib.mvn_reload_switches()
        for _ in range(3):
            dec = det(y, "val")
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            dec = det(y, "val")
        b.record()
        b.synchronize()
        res[unfused] = (a.elapsed_time(b) / 5, dec.clone())
    same = torch.equal(res["0"][1], res["1"][1])
    print(f"S {S:4d}  {B:6d} blocks x {T}: fused {res['0'][0]:8.3f} ms ({B * T / res['0'][0] / 1e6:7.2f} Gsym/s)   two kernels {res['1'][0]:8.3f} ms "
          f"({B * T / res['1'][0] / 1e6:7.2f} Gsym/s)   ratio {res['1'][0] / res['0'][0]:.2f}   same decisions {same}", flush=True)
