#!/usr/bin/env python3
"""Timing of mvn_acs_sweep_f32 (HBM-bound ACS sweep over materialised costs) at S=16, per kernel variant.

usage: time_sweep.py [--variants lds,quad,rows] B [B ...]     (decisions of all variants are compared with the first)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

args = sys.argv[1:]
variants = [os.environ.get("MVN_SWEEP16", "")]
if args and args[0] == "--variants":
    variants = args[1].split(",")
    args = args[2:]
dev = torch.device("cuda:0")
lib = mvn._lib.load()
st = mvn._lib.current_stream(dev)
T, S = 1000, 16
for B in [int(a) for a in args] or [10000, 40000]:
    cost = torch.randn(B, T, S, device=dev)
    ref = None
    for v in variants:
        os.environ.pop("MVN_SWEEP_INPLACE", None)
        if v == "inplace":
            os.environ["MVN_SWEEP_INPLACE"] = "1"
        elif v:
            os.environ["MVN_SWEEP16"] = v
        mvn._lib.reload_switches()
        dec = torch.zeros(B, T, device=dev)

        def run():
            assert lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st) == 0

        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        same = ""
        if ref is None:
            ref = dec.clone()
        else:
            same = "  == first" if torch.equal(ref, dec) else "  MISMATCH vs first"
        print(f"acs_sweep S=16 B={B} T={T} [{v or 'default'}]: {ms:.4f} ms  {68.0*B*T/ms/1e6:.0f} GB/s  "
              f"({B*T/ms/1e6:.1f} Gsym/s){same}", flush=True)
