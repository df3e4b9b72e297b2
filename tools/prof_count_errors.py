"""Workload for rocprofv3 --pmc TCC_EA0_RDREQ_sum / FETCH_SIZE: mvn_count_errors over 10 000 rows of K = 1000 symbols with row
strides of 1000 (rows start 0 / 32 / 64 / 96 bytes into a 128-byte line) and 1024 floats (every row starts on a line)."""
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
B, K = 10000, 1000
for ld, reps in ((1000, 3), (1024, 5)):
    dec = (torch.rand(B, ld, device=dev) > 0.5).float()
    tx = (torch.rand(B, ld, device=dev) > 0.5).float()
    c = torch.zeros(4, dtype=torch.int64, device=dev)
    for _ in range(reps):  # the two strides are told apart by their dispatch counts in the summary
        mvn.count_errors(dec[:, :K], tx[:, :K], None, c)
    torch.cuda.synchronize()
    print(ld, c.tolist())
