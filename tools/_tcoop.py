import os, sys, ctypes
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn
dev = torch.device("cuda:0")
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
def ev(fn, iters=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for T in (16, 32, 64, 136, 256, 512, 1024):
    det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, i in zip(det.parameters(), range(6)): p.copy_(torch.tensor(g7[f"w{i}"]))
    wl = [mvn._lib.ptr(p) for p in det.parameters()]
    y = torch.randn(1, T, device=dev); dec = torch.empty(1, T, device=dev)
    t = ev(lambda: lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *wl, mvn._lib.ptr(dec), T, None, None, None, 0, 1, T, 16, st))
    print(f"T {T:5d}: coop detect B=1 {t:6.1f} us")
x = torch.zeros(1024, device=dev)
print("trivial torch kernel back to back:", ev(lambda: x.add_(1.0)), "us")
