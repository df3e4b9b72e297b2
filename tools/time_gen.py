#!/usr/bin/env python3
"""How long does generating one Monte-Carlo batch take next to decoding it?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn
dev = torch.device("cuda:0")
B, T, L = 10000, 1000, 4
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
g = torch.Generator(device=dev); g.manual_seed(1)
print("synthetic_words        :", t(lambda: mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=3)), "ms")
print("  randint bits (int8)  :", t(lambda: torch.randint(0, 2, (B, T), generator=g, device=dev, dtype=torch.int8)), "ms")
print("  randn noise (fp32)   :", t(lambda: torch.randn(B, T, generator=g, device=dev)), "ms")
bits = torch.randint(0, 2, (B, T), generator=g, device=dev).float(); nz = torch.randn(B, T, generator=g, device=dev)
h = mvn.estimate_channel(L, 0.2, "time_decay")
print("  transmit kernel      :", t(lambda: mvn.transmit(bits, h, 10.0, L, nz)), "ms")
