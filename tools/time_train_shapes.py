"""Time of the per-trial training kernels against the word length (how a partial last chunk is priced): R trials x n full-word
iterations / second-order steps for T in a list.  A/B two builds in one gpurun call through MVN_LIB_PATH.
usage: time_train_shapes.py [T ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trial_setup import L, dev, mvn, w  # noqa: E402
from meta_viterbinet_amd import trials as tr_mod  # noqa: E402

Ts = [int(a) for a in sys.argv[1:]] or [128, 136, 144, 160]
lib = mvn._lib.load()
S, NW, R = 16, 8, 256
gen = torch.Generator(device=dev).manual_seed(1)
print("lib", mvn._lib.LIB_PATH)
for T in Ts:
    for kind, n in (("online", 100), ("maml", 20), ("maml1", 20)):  # maml1: first-order steps (no Hessian pass)
        bank = tr_mod.TrialBank([w] * R, S, L, dev)
        rxw = torch.randn(R, NW, T, generator=gen, device=dev)
        txw = torch.randint(0, 2, (R, NW, T), generator=gen, device=dev).float()
        labels = torch.stack([mvn.calculate_states(L, txw[r]).reshape(NW, T) for r in range(R)]).to(torch.int32).contiguous()
        sup = torch.randint(0, NW, (R, n, 1), generator=gen, device=dev).to(torch.int32)
        qry = torch.randint(0, NW, (R, n), generator=gen, device=dev).to(torch.int32)
        d = np.zeros(R, dtype=tr_mod.TRIAL_DTYPE)
        th = bank.pointers(bank.theta)
        for r in range(R):
            d[r]["y"], d[r]["labels"] = rxw[r].data_ptr(), labels[r].data_ptr()
            d[r]["idx"] = sup[r].data_ptr() if kind != "online" else 0
            d[r]["query_idx"] = qry[r].data_ptr() if kind != "online" else 0
            d[r]["w_in"], d[r]["w_out"] = th[r], th[r]
            d[r]["adam_m"], d[r]["adam_v"] = bank.exp_avg[r].data_ptr(), bank.exp_avg_sq[r].data_ptr()
            d[r]["b1pow"], d[r]["b2pow"], d[r]["n"] = 1.0, 1.0, n
        dd = torch.from_numpy(d.view(np.uint8)).to(dev)
        st = mvn._lib.current_stream(dev)

        def launch():
            if kind != "online":
                rc = lib.mvn_vnet_maml_train_trials_f32(mvn._lib.ptr(dd), R, T, 1, 0.1, 1 if kind == "maml" else 0, 1e-3, 0.9, 0.999, 1e-8, S, None, 0, st)
            else:
                rc = lib.mvn_vnet_online_train_trials_f32(mvn._lib.ptr(dd), R, T, 0, 1e-3, 0.9, 0.999, 1e-8, S, None, 0, st)
            assert rc == 0

        launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            launch()
        b.record()
        b.synchronize()
        print(f"  T {T:4d} {kind:6s}: {a.elapsed_time(b) / 3 / n * 1e3:8.2f} us per {'iteration' if kind == 'online' else 'step'} (one workgroup per trial, {R} trials)", flush=True)
