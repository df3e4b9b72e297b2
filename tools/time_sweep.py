#!/usr/bin/env python3
"""Timing of mvn_acs_sweep_f32 (HBM-bound ACS sweep over materialised costs) at S=16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
lib = mvn._lib.load()
st = mvn._lib.current_stream(dev)
T, S = 1000, 16
for B in [int(a) for a in sys.argv[1:]] or [10000, 40000]:
    cost = torch.randn(B, T, S, device=dev)
    dec = torch.zeros(B, T, device=dev)

    def run():
        assert lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st) == 0

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"acs_sweep S=16 B={B} T={T}: {ms:.4f} ms  {68.0*B*T/ms/1e6:.0f} GB/s  ({B*T/ms/1e6:.1f} Gsym/s)")
