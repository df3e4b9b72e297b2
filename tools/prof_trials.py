"""Workload for rocprofv3 --kernel-trace: one batched evaluation of R trials (trials.eval_by_word_batched) of one of the two
BASELINE flows with online training.  usage: prof_trials.py R "configs[2]"|"configs[4]"   (summarise with stats_by_grid.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trial_setup import FLOWS, L, dev, nsym, sub, w, words  # noqa: E402
from meta_viterbinet_amd.trials import TrialBank, TrialDraws, eval_by_word_batched  # noqa: E402

R, flow = int(sys.argv[1]), sys.argv[2]
name = [k for k in FLOWS if flow in k][0]
coef, kw = FLOWS[name]
ws = [words(coef, 7.0 + (i % 6), 100 + i) for i in range(R)]
msg, rx = torch.stack([a for a, _ in ws]), torch.stack([b for _, b in ws])
bank = TrialBank([w] * R, 16, L, dev)
draws = [TrialDraws(100 + i, dev) for i in range(R)]
ser = eval_by_word_batched(bank, msg, rx, nsym, sub, draws, **kw)
torch.cuda.synchronize()
print(R, name, ser.mean())
