"""Minibatch / full-word online-training launches of R trials: one 1024-thread workgroup per trial (a trial per CU at a time)
against the 512-thread form that puts two trials on a CU (MVN_TRAIN_PAIR=0|1).  usage: time_train_pair.py [R ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trial_setup import L, T, dev, mvn, w  # noqa: E402
from meta_viterbinet_amd import trials as tr_mod  # noqa: E402

Rs = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
lib = mvn._lib.load()
S, n = 16, 200
gen = torch.Generator(device=dev).manual_seed(1)
for R in Rs:
    for M in (32, 0):
        res = {}
        for pair in ("0", "1"):
            os.environ["MVN_TRAIN_PAIR"], os.environ["MVN_TRAIN_GROUPS"] = pair, "0"
            lib.mvn_reload_switches()
            bank = tr_mod.TrialBank([w] * R, S, L, dev)
            g2 = torch.Generator(device=dev).manual_seed(5)
            rxw = torch.randn(R, T, generator=g2, device=dev)
            txw = torch.randint(0, 2, (R, T), generator=g2, device=dev).float()
            labels = mvn.calculate_states(L, txw).reshape(R, T).to(torch.int32).contiguous()
            bidx = (torch.multinomial(torch.arange(T, dtype=torch.float32, device=dev).expand(R * n, T), 32, generator=g2)
                    .to(torch.int32).reshape(R, n, 32)) if M else None
            d = np.zeros(R, dtype=tr_mod.TRIAL_DTYPE)
            th = bank.pointers(bank.theta)
            for r in range(R):
                d[r]["y"], d[r]["labels"] = rxw[r].data_ptr(), labels[r].data_ptr()
                d[r]["idx"] = bidx[r].data_ptr() if M else 0
                d[r]["w_in"], d[r]["w_out"] = th[r], th[r]
                d[r]["adam_m"], d[r]["adam_v"] = bank.exp_avg[r].data_ptr(), bank.exp_avg_sq[r].data_ptr()
                d[r]["b1pow"], d[r]["b2pow"], d[r]["n"] = 1.0, 1.0, n
            dd = torch.from_numpy(d.view(np.uint8)).to(dev)
            st = mvn._lib.current_stream(dev)
            launch = lambda: lib.mvn_vnet_online_train_trials_f32(mvn._lib.ptr(dd), R, T, M, 1e-3, 0.9, 0.999, 1e-8, S, None, 0, st)  # noqa: E731
            assert launch() == 0
            torch.cuda.synchronize()
            first = (bank.theta.clone(), bank.exp_avg.clone(), bank.exp_avg_sq.clone())
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                assert launch() == 0
            b.record()
            b.synchronize()
            res[pair] = (a.elapsed_time(b) / 3, first)
        same = all(torch.equal(x, y) for x, y in zip(res["0"][1], res["1"][1]))
        print(f"R {R:5d} {'minibatch' if M else 'full word'}: 1024 threads/trial {res['0'][0]:8.3f} ms   512 threads/trial, two per CU "
              f"{res['1'][0]:8.3f} ms   ratio {res['0'][0] / res['1'][0]:.3f}   bit-identical {same}", flush=True)
