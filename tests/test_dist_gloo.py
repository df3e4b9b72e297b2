"""CPU-only, world_size 2 over gloo: the block-sharded Monte-Carlo loop gives the same
integer counters as one process.  The decode/count callables are oracle-backed stand-ins here (tests may use
the oracle); on the GPU box the same harness code runs with the HIP detectors and backend='nccl' (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import meta_viterbinet_amd as mvn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_problem():
    rng = np.random.RandomState(0)
    B, T, L = 37, 96, 4  # odd B: ranks get 19 / 18 rows
    tx, y = mvn.synthetic_words(B, T, L, 8.0, 0.2, "cpu", seed=5)
    pri = rng.normal(0, 1, (1, 16)).astype(np.float32)
    h = mvn.estimate_channel(L, 0.2, "time_decay")
    sym = 1 - 2 * ((np.arange(16)[:, None] >> np.arange(L)[::-1]) & 1)
    pri = (sym @ h.T).T.astype(np.float32)
    rows = torch.tensor([i for i in range(B) if i % 5 != 0])
    return tx, y, pri, rows


def _oracle_detector(pri):
    import oracle

    def det(rx, phase, snr=None, gamma=None, count=None):
        return torch.tensor(oracle.va_decode(rx.numpy(), pri, want_final=False))

    return det


def _oracle_counter(detected, tx, rows):
    import oracle

    r = None if rows is None else rows.numpy()
    return torch.tensor(oracle.count_errors(detected.numpy(), tx.numpy(), r))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tx, y, pri, rows = _make_problem()
        ser, fer, counters = mvn.sharded_eval(_oracle_detector(pri), tx, y, 8.0, 0.2, rows, counter=_oracle_counter)
        # rank-local shards + eval_counters (the bench.py pattern: each rank owns its rows) must agree too
        lo, hi = mvn.shard_range(tx.shape[0], rank, world)
        c2 = mvn.eval_counters(_oracle_detector(pri), tx[lo:hi], y[lo:hi], 8.0, 0.2, None, _oracle_counter)
        q.put((rank, counters.tolist(), c2.tolist(), ser, fer))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_counters_equal_single_process():
    tx, y, pri, rows = _make_problem()
    ser1, fer1, c1 = mvn.sharded_eval(_oracle_detector(pri), tx, y, 8.0, 0.2, rows, counter=_oracle_counter,
                                      rank=0, world=1)
    call = mvn.eval_counters(_oracle_detector(pri), tx, y, 8.0, 0.2, None, _oracle_counter)
    assert c1[1] == rows.numel() * tx.shape[1] and c1[0] > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, counters, c2, ser, fer in res:
        assert counters == c1.tolist()  # identical integers on every rank
        assert c2 == call.tolist()
        assert (ser, fer) == (ser1, fer1)
