"""Launch durations (HIP events, back-to-back launches) of the by-word step kernel against its parts: vnet16_coop_kernel
(B = 1 detect), rs_decode / count / rs_encode launches, and the whole 300-block no-update evaluation."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402
from meta_viterbinet_amd.trials import TrialBank  # noqa: E402

dev = torch.device("cuda:0")
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [g7[f"w{i}"] for i in range(6)]
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
N, K, nsym, T = 300, 120, 2, 136


def ev(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for snr in (12.0, 6.0):
    gen = torch.Generator(device=dev).manual_seed(1)
    for R in (1, 64, 256):
        msg = torch.randint(0, 2, (R, K), generator=gen, device=dev).float()
        cw = mvn.rs_encode(msg, nsym)
        rx = mvn.transmit(cw, mvn.estimate_channel(4, 0.2, "time_decay"), snr, 4, torch.randn(R, T, generator=gen, device=dev))
        bank = TrialBank([w] * R, 16, 4, dev)
        wp = [ctypes.c_void_p(bank.theta.data_ptr() + 4 * int(bank.off[a])) for a in range(6)]
        wst = (ctypes.c_int64 * 6)(*([bank.P] * 6))
        dec, enc = torch.empty(R, T, device=dev), torch.empty(R, T, device=dev)
        lab, nerr = torch.empty(R, T, dtype=torch.int32, device=dev), torch.empty(R, dtype=torch.int32, device=dev)

        def step(pilot):
            assert lib.mvn_vnet_byword_step_f32(mvn._lib.ptr(rx), T, mvn._lib.ptr(msg), K, *wp, wst, mvn._lib.ptr(dec), T, None, K,
                                                mvn._lib.ptr(enc), T, None, T, mvn._lib.ptr(lab), T, mvn._lib.ptr(nerr), R, T, nsym,
                                                pilot, 16, st) == 0

        t_data, t_pilot = ev(lambda: step(0)), ev(lambda: step(1))
        det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
        with torch.no_grad():
            for p, a in zip(det.parameters(), w):
                p.copy_(torch.tensor(a))
        wl = [mvn._lib.ptr(p) for p in det.parameters()]
        t_det = ev(lambda: lib.mvn_vnet_decode_f32(mvn._lib.ptr(rx), T, *wl, mvn._lib.ptr(dec), T, None, None, None, 0, R, T, 16, st))
        dm = torch.empty(R, K, device=dev)
        t_rsd = ev(lambda: lib.mvn_rs_decode_bits_f32(mvn._lib.ptr(dec), T, mvn._lib.ptr(dm), K, None, R, T, nsym, st))
        t_rse = ev(lambda: lib.mvn_rs_encode_bits_f32(mvn._lib.ptr(dm), K, mvn._lib.ptr(enc), T, R, K, nsym, st))
        step(0)
        print(f"snr {snr:4.1f} R {R:3d}: step data {t_data:6.1f} us, pilot {t_pilot:5.1f} us | detect {t_det:5.1f}, rs decode {t_rsd:5.1f}, "
              f"rs encode {t_rse:5.1f} us | words with errors {int((nerr > 0).sum())}")

msg = torch.randint(0, 2, (N, K), device=dev).float()
rx = mvn.transmit(mvn.rs_encode(msg, nsym), mvn.estimate_channel(4, 0.2, "time_decay"), 10.0, 4, torch.randn(N, T, device=dev))
det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
for fused in (True, False):
    mvn.eval_by_word(det, msg, rx, 10.0, 0.2, nsym, 25, fused_step=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        mvn.eval_by_word(det, msg, rx, 10.0, 0.2, nsym, 25, fused_step=fused)
    torch.cuda.synchronize()
    print(f"eval_by_word without updates, fused_step={fused}: {(time.perf_counter() - t0) / 5 / N * 1e6:.1f} us per block")
