"""-m gpu: the multi-rank path on hardware.  (i) backend='nccl' (= RCCL) with world_size 1: init_process_group with a
device id, harness.sharded_eval / eval_counters with the real HIP detectors, all_reduce of the device int64[4] counters;
(ii) `python bench.py --gpus 2` started WITHOUT a launcher: bench.py spawns its own ranks (gloo rehearsal, both ranks on
the one GPU of this box)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

import meta_viterbinet_amd as mvn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def rccl_world1():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_world1_sharded_eval_hip_detectors(golden, rccl_world1):
    dev = rccl_world1
    assert dist.get_backend() == "nccl"
    L, S, B, T = 4, 16, 203, 136
    tx, y = mvn.synthetic_words(B, T, L, 9.0, 0.2, dev, seed=11)
    rows = torch.tensor([i for i in range(B) if i % 5 != 0], device=dev)
    g = golden("g7_by_word")
    vnet = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, i in zip(vnet.parameters(), range(6)):
            p.copy_(torch.tensor(g[f"w{i}"]))
    va = mvn.VADetector(S, L, T, 1, "ISI_AWGN", 0, False, 1, {"train": "time_decay", "val": "time_decay"})
    for det in (vnet, va):
        # single-process integers, no collective: decode + the stand-alone counting kernel
        dec = det(y, "val", 9.0, 0.2)
        want = mvn.count_errors(dec, tx, rows).cpu()
        assert want[1] == rows.numel() * T and want[0] > 0
        # the harness: all_reduce(SUM) of the device int64[4] over RCCL
        ser, fer, c = mvn.sharded_eval(det, tx, y, 9.0, 0.2, rows)
        assert c.is_cuda and torch.equal(c.cpu(), want)
        c2 = mvn.eval_counters(det, tx, y, 9.0, 0.2, rows)  # VNET: fused decode+count epilogue
        assert torch.equal(c2.cpu(), want)
        assert (ser, fer) == mvn.rates_from_counters(want)
    # the collective really ran on a device tensor (sum over 1 rank = identity, MAX for the timing reduce)
    t = torch.tensor([1.5, -2.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert t.tolist() == [1.5, -2.0]


@pytest.mark.timeout(600)
def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns both ranks before touching the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MVN_BENCH_BACKEND"] = "gloo"  # two ranks share this box's one GPU; the 8-GPU node runs nccl
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--blocks", "512", "--no-cpu-baseline", "--no-configs"], env=env, capture_output=True, text=True,
                       timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and np.isfinite(out["value"])
    _check_collective_fields(out, 2, 512, 2)


def _check_collective_fields(out, world, blocks, steps):
    """The fields that let a reader of the JSON line confirm that N ranks took part in the collective."""
    c = out["collective"]
    assert c["world_size"] == world and c["backend"] == "gloo"
    assert c["symbols_by_rank"] == [blocks * 1000 * steps] * world
    assert c["frames_all_ranks"] == c["frames_expected"] == world * blocks * steps  # the all-reduced counters saw every rank
    assert out["value"] == pytest.approx(sum(c["symbols_by_rank"]) / (out["ms_per_step"] * 1e-3 * steps), rel=1e-6)


@pytest.mark.timeout(900)
def test_bench_self_launch_four_ranks_with_configs():
    """The multi-rank control flow of the whole bench line (headline + every per-config entry: block-sharded configs,
    replica configs with R trials per rank, the cpu baseline on rank 0) with four gloo ranks sharing this box's GPU -- the
    rehearsal of the 8-GPU run the driver launches (the GPU box admits at most 6 processes on the card at once, so 8 ranks
    cannot be rehearsed here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MVN_BENCH_BACKEND="gloo", MVN_BENCH_TRIALS_SELFSUP="6", MVN_BENCH_TRIALS_META="3", MVN_BENCH_TRIALS_META_MORE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--blocks", "512", "--sustained-seconds", "0.2"], env=env, capture_output=True, text=True, timeout=860)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4
    _check_collective_fields(out, 4, 512, 2)
    # the CPU baselines are timed at N = 1 only (the other ranks would sit in a barrier for ~30 s): at N > 1 the entry says so
    assert out["cpu_baseline"]["value"] is None and "N=1" in out["cpu_baseline"]["sample"] and out["sustained"]["symbols_per_s"] > 0
    strong = out["strong_scaling"]  # the same 512 blocks split over the 4 ranks, next to the weak-scaling headline
    assert strong["blocks_per_gpu"] == 128 and strong["frames_counted"] == strong["frames_expected"] == 512 * out["steps"]
    cfg = {c["config"].split(":")[0]: c for c in out["configs"]}
    assert cfg["BASELINE configs[3]"]["frames"] == 4 * 125000  # all ranks' blocks in the all-reduced counters
    assert cfg["BASELINE configs[2]"]["self_supervised_trials"]["trials_per_gpu"] == 6
    assert cfg["BASELINE configs[4]"]["trials_per_gpu"] == 3 and cfg["BASELINE configs[4]"]["blocks_per_s"] > 0
