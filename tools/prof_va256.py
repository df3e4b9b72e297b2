"""Workload for rocprofv3 (--kernel-trace / --pmc): the classical Viterbi kernel of BASELINE configs[3] alone -- mvn_va_decode_f32 at
256 states (va256_wave_kernel), B blocks x 1000 symbols (one GPU's share of 10^6: B = 125 000), a few launches after the clock has
settled.  usage: prof_va256.py [B] [reps]"""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
T, L = 1000, 8
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
tx, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=3450002)
va = mvn.VADetector(256, L, T, 1, "ISI_AWGN", 0, False, 1, "time_decay")
pri = va.compute_state_priors(mvn.estimate_channel(L, 0.2, "time_decay")).to(dev).T.contiguous()
dec = torch.empty_like(y)


def launch():
    rc = lib.mvn_va_decode_f32(mvn._lib.ptr(y), T, mvn._lib.ptr(pri), 1, mvn._lib.ptr(dec), T, None, B, T, 256, st)
    assert rc == 0, rc


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.06:  # clock settle (bench.py does the same)
    launch()
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    launch()
b.record()
b.synchronize()
ms = a.elapsed_time(b) / reps
print(f"va256 decode: {B} blocks x {T}: {ms:.4f} ms per launch, {B * T / (ms * 1e-3):.4g} symbols/s, "
      f"{ms * 1e-3 * 2.4e9 * 1024 / (B * T):.1f} SIMD-cycles per symbol at 2.4 GHz")
