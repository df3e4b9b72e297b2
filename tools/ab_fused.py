#!/usr/bin/env python3
"""A/B timing of ABI-level variants selected by environment variables, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24).  Usage: python tools/ab_fused.py VAR=a,b [VAR2=c,d] [--blocks N]"""
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

B, T, S = 10000, 1000, 16
args = [a for a in sys.argv[1:] if "=" in a]
for a in sys.argv[1:]:
    if a.startswith("--blocks"):
        B = int(a.split("=")[1])
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args]
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [torch.tensor(g[f"w{i}"], device=dev) for i in range(6)]
tx, y = mvn.synthetic_words(B, T, 4, 10.0, 0.2, dev, seed=1)
lib = mvn._lib.load()
st = mvn._lib.current_stream(dev)
dec = torch.zeros(B, T, device=dev)
ref = None


COUNT = "--count" in sys.argv
counters = torch.zeros(4, dtype=torch.int64, device=dev)


def run():
    if COUNT:  # decode + fused error counting, no decision store
        rc = lib.mvn_vnet_decode_count_f32(mvn._lib.ptr(y), T, *[mvn._lib.ptr(t) for t in w], mvn._lib.ptr(tx), T, T, None,
                                           mvn._lib.ptr(counters), None, T, None, 0, B, T, S, st)
    else:
        rc = lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *[mvn._lib.ptr(t) for t in w], mvn._lib.ptr(dec), T, None, None, None,
                                     0, B, T, S, st)
    assert rc == 0


combos = list(itertools.product(*[v for _, v in axes]))
times = {c: [] for c in combos}
for rnd in range(7):
    for c in combos:
        for (k, _), v in zip(axes, c):
            os.environ[k] = v
        mvn._lib.reload_switches()
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        e1.synchronize()
        times[c].append(e0.elapsed_time(e1) / 5)
        if not COUNT:
            if ref is None:
                ref = dec.clone()
            assert torch.equal(ref, dec), f"variant {c} changes the decisions"
for c in combos:
    t = sorted(times[c])
    print(dict(zip([k for k, _ in axes], c)), f"median {t[len(t)//2]:.4f} ms  min {t[0]:.4f} ms  -> {B*T/t[len(t)//2]/1e6:.2f} Gsym/s")
