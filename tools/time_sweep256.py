#!/usr/bin/env python3
"""mvn_acs_sweep_f32 at S=256 (1028 B/symbol): generic kernel HBM rate."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn
dev = torch.device("cuda:0"); lib = mvn._lib.load(); st = mvn._lib.current_stream(dev)
T = 1000
for S, B in ((256, 2000), (64, 8000), (4, 100000)):
    cost = torch.randn(B, T, S, device=dev); dec = torch.zeros(B, T, device=dev)
    run = lambda: lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); e1.synchronize(); ms = e0.elapsed_time(e1) / 5
    print(f"acs_sweep S={S} B={B}: {ms:.3f} ms  {(4*S+4)*B*T/ms/1e6:.0f} GB/s  {B*T/ms/1e6:.2f} Gsym/s")
    del cost
