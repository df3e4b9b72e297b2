#!/usr/bin/env python3
"""mvn_acs_sweep_f32 at S != 16 ((4 S + 4) B/symbol): in-place LDS-DMA kernel vs the generic LDS-exchange kernel."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn
dev = torch.device("cuda:0"); lib = mvn._lib.load(); st = mvn._lib.current_stream(dev)
T = 1000
cases = ((256, 2000), (256, 6000), (128, 4000), (64, 8000), (32, 16000), (8, 50000), (4, 100000))
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for S, B in cases:
    cost = torch.randn(B, T, S, device=dev)
    ref = None
    for variant in ("inplace", "generic"):
        os.environ["MVN_GENERIC_SWEEP"] = "1" if variant == "generic" else "0"
        mvn._lib.reload_switches()
        dec = torch.zeros(B, T, device=dev)
        run = lambda: lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); e1.synchronize(); ms = e0.elapsed_time(e1) / 5
        same = "" if ref is None else ("  == first" if torch.equal(ref, dec) else "  MISMATCH")
        ref = dec if ref is None else ref
        print(f"acs_sweep S={S} B={B} [{variant}]: {ms:.3f} ms  {(4*S+4)*B*T/ms/1e6:.0f} GB/s  {B*T/ms/1e6:.2f} Gsym/s{same}", flush=True)
    del cost
