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


def _oracle_rs(detected, n_symbols):
    import oracle

    return torch.tensor(oracle.rs_decode_bits(detected.numpy(), n_symbols))


def _coded_problem():
    """24 message bytes-of-bits words, RS(5,3)-encoded, sent through the ISI channel at low noise."""
    import oracle

    rng = np.random.RandomState(3)
    msg = rng.randint(0, 2, (21, 24)).astype(np.float32)
    cw = oracle.rs_encode_bits(msg, 2)  # [21, 40]
    L = 4
    s = 1.0 - 2.0 * np.concatenate([cw, np.zeros((21, L), np.float32)], axis=1)
    h = mvn.estimate_channel(L, 0.2, "time_decay")[0]
    y = sum(h[L - 1 - k] * s[:, k:k + 40] for k in range(L)) + 0.25 * rng.normal(size=(21, 40))
    return torch.tensor(msg), torch.tensor(y.astype(np.float32))


def _replica_trial(i):
    """stand-in for one eval_by_word trial: a deterministic ser_by_word[300] of trial i"""
    return np.random.RandomState(100 + i).rand(300).astype(np.float32)


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
        # coded path (use_ecc): RS-decode the detected words before counting, sharded the same way
        msg, cw_y = _coded_problem()
        _, _, c3 = mvn.sharded_eval(_oracle_detector(pri), msg, cw_y, 8.0, 0.2, None, counter=_oracle_counter, n_symbols=2,
                                    rs_decoder=_oracle_rs)
        # replica mode (configs that do not shard within a trial): 5 trials over 2 ranks, one all_gather
        rep = mvn.replica_eval(_replica_trial, 5)
        # ... and with each rank's trials run as one batch (trials.eval_by_word_batched's calling convention)
        repb = mvn.replica_eval(lambda ids: np.stack([_replica_trial(i) for i in ids]), 5, batched=True)
        assert np.array_equal(rep, repb)
        q.put((rank, counters.tolist(), c2.tolist(), ser, fer, c3.tolist(), rep.tolist()))
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
    msg, cw_y = _coded_problem()
    _, _, coded1 = mvn.sharded_eval(_oracle_detector(pri), msg, cw_y, 8.0, 0.2, None, counter=_oracle_counter, n_symbols=2,
                                    rs_decoder=_oracle_rs, rank=0, world=1)
    assert coded1[1] == 21 * 24
    rep1 = mvn.replica_eval(_replica_trial, 5)  # no process group: every trial on this process
    assert rep1.shape == (5, 300) and np.array_equal(rep1[3], _replica_trial(3))
    assert np.array_equal(rep1, mvn.replica_eval(lambda ids: np.stack([_replica_trial(i) for i in ids]), 5, batched=True))
    for rank, counters, c2, ser, fer, c3, rep in res:
        assert np.array_equal(np.asarray(rep, np.float32), rep1)  # same [trial, block] table on every rank
        assert counters == c1.tolist()  # identical integers on every rank
        assert c2 == call.tolist()
        assert (ser, fer) == (ser1, fer1)
        assert c3 == coded1.tolist()
