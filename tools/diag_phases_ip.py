"""Where a wave's cycles go in vnet_fused_ip_kernel (S = 4, 8, 32, 64): the -DMVN_DIAG_PHASES build (tools/dbg/libmvn_phases.so:
tools/diag_phases.py build) sums s_memtime differences per phase of a chunk and writes them over the final-metric rows.
usage: diag_phases_ip.py [B]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVN_LIB_PATH"] = os.path.join(ROOT, "tools", "dbg", "libmvn_phases.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import meta_viterbinet_amd as mvn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = 1000
dev = torch.device("cuda:0")
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
for S in (4, 8, 32, 64):
    torch.manual_seed(S)
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    w = [p.detach().contiguous() for p in det.parameters()]
    y = torch.randn(B, T, device=dev) * 1.5
    dec = torch.empty(B, T, device=dev)
    fm = torch.zeros(B, S, device=dev)
    for _ in range(3):
        rc = lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *[mvn._lib.ptr(a) for a in w], mvn._lib.ptr(dec), T, None, mvn._lib.ptr(fm), None, 0, B, T, S, st)
        assert rc == 0
    torch.cuda.synchronize()
    o = fm.cpu().numpy().reshape(B // 2, 2 * S)[:, :8].copy().view(np.uint64).reshape(B // 2, 4).astype(np.float64)
    chunks = (T + 15) // 16
    print(f"S = {S}: {len(o)} waves x {chunks} chunks of 32 symbols; s_memtime ticks per chunk (100 MHz x 24 ~ core cycles):")
    for i, lab in enumerate(["k-loop (layers 1, 2)", "transposes + layer 3 -> image", "sweep (16 steps, 2 blocks)", "decision flush"]):
        v = o[:, i] / chunks
        print(f"   {lab:32s} mean {v.mean():9.1f}  p10 {np.percentile(v, 10):9.1f}  p90 {np.percentile(v, 90):9.1f}")
    print(f"   {'sum':32s} mean {(o.sum(1) / chunks).mean():9.1f}")
