"""Shared setup of the trial-batched timing / profiling tools (time_trials.py, prof_trials.py, prof_train_kernels.py): the
by-word words of a trial, the reference-trained weights and the two BASELINE flows with online training."""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
N, K, nsym, L, sub, T = 300, 120, 2, 4, 25, 136
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w = [g7[f"w{i}"] for i in range(6)]


def words(coef, snr, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    msg = torch.randint(0, 2, (N, K), generator=gen, device=dev).float()
    cw = mvn.rs_encode(msg, nsym)
    if coef == "cost2100":
        h = np.concatenate([mvn.estimate_channel(L, 0.2, "cost2100", index=i) for i in range(N)])
    else:
        h = np.concatenate([mvn.estimate_channel(L, 0.2, "time_decay", fading=True, index=i, fading_taps_type=2) for i in range(N)])
    return msg, mvn.transmit(cw, h, snr, L, torch.randn(N, T, generator=gen, device=dev))


FLOWS = {"configs[2] self-supervised (200 minibatch iterations / block)": ("cost2100", dict(self_supervised=True, self_supervised_iterations=200)),
         "configs[4] Meta-ViterbiNet (200 / 20 / 10 / 5)": ("time_decay", dict(self_supervised=True, self_supervised_iterations=200, online_meta=True,
                                                                meta_train_iterations=20, meta_j_num=10, meta_subframes=5,
                                                                meta_style_online_training=True))}
