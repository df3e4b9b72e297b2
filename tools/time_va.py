#!/usr/bin/env python3
"""Timing of mvn_va_decode_f32 / mvn_acs_sweep_f32 at several (L, B) (HIP events, warm)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
lib = mvn._lib.load()
st = mvn._lib.current_stream(dev)
T = 1000
cases = [(4, 100), (4, 10000), (4, 200000), (8, 2000), (8, 20000), (2, 100000), (3, 100000), (5, 50000), (6, 20000), (7, 20000)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for L, B in cases:
    S = 2 ** L
    tx, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=1)
    va = mvn.VADetector(S, L, T, 1, "ISI_AWGN", 0, False, 1, {"train": "time_decay", "val": "time_decay"})
    pri = va.compute_state_priors(mvn.estimate_channel(L, 0.2, "time_decay")).to(dev).T.contiguous()
    dec = torch.zeros_like(y)

    def run():
        rc = lib.mvn_va_decode_f32(mvn._lib.ptr(y), T, mvn._lib.ptr(pri), 1, mvn._lib.ptr(dec), T, None, B, T, S, st)
        assert rc == 0

    ref = None
    for variant in (["rows", "quad", "tile", "split", "tmpl"] if S == 16 else ["wave", "tmpl", "generic"] if S == 256 else ["tmpl", "generic"] if S >= 4 else [""]):
        os.environ.pop("MVN_VA_INPLACE", None)
        os.environ["MVN_VA256"] = "" if variant == "wave" else "inplace"
        os.environ.pop("MVN_GENERIC_SWEEP", None)
        if variant in ("rows", "quad", "tile", "split"):
            os.environ["MVN_VA16"] = variant
        elif variant == "tmpl":
            os.environ["MVN_VA_INPLACE"] = "1"
        elif variant == "generic":
            os.environ["MVN_GENERIC_SWEEP"] = "1"
        mvn._lib.reload_switches()
        dec.zero_()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / n
        c = mvn.count_errors(dec, tx)
        ser, fer = mvn.rates_from_counters(c)
        same = ""
        if ref is None:
            ref = dec.clone()
        else:
            same = "  == first" if torch.equal(ref, dec) else "  MISMATCH vs first"
        print(f"VA L={L} S={S} B={B} T={T} {variant:7s}: {ms:.4f} ms  {B*T/ms/1e6:.3f} Gsym/s  "
              f"({B*T*S/ms/1e6:.1f} G state-steps/s)  ser={ser:.4g}{same}", flush=True)
