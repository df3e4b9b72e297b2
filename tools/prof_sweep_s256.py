#!/usr/bin/env python3
"""Workload for rocprofv3: mvn_acs_sweep_f32 at S = 256 (6000 blocks x 1000 steps, 1028 B/symbol) and
mvn_va_decode_f32 at S = 256 (20 000 x 1000), a few launches each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
lib, st = mvn._lib.load(), mvn._lib.current_stream(dev)
T, S = 1000, 256
B = 6000
cost = torch.randn(B, T, S, device=dev)
dec = torch.zeros(B, T, device=dev)
for _ in range(6):
    assert lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st) == 0
del cost
B = 20000
y = torch.randn(B, T, device=dev)
pri = torch.randn(1, S, device=dev)
dec = torch.zeros(B, T, device=dev)
for _ in range(6):
    assert lib.mvn_va_decode_f32(mvn._lib.ptr(y), T, mvn._lib.ptr(pri), 1, mvn._lib.ptr(dec), T, None, B, T, S, st) == 0
torch.cuda.synchronize()
print("done")
