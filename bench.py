#!/usr/bin/env python3
"""bench.py -- decoded symbols/sec of the ViterbiNet 'val' hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of VNETDetector.forward(y,'val') + on-device error counting over this rank's batch of
synthetic words (BASELINE.json configs[1]: ViterbiNet L=4, 10 000 blocks x 1000 symbols per GPU, inputs
resident in HBM).  Blocks are independent: every rank owns its own 10 000 blocks (weak scaling) and the only
collective is one all-reduce of the int64[4] error counters at the end.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import meta_viterbinet_amd as mvn  # noqa: E402

L, S, T = 4, 16, 1000
SNR_DB, GAMMA = 10.0, 0.2
FLOP_PER_SYMBOL = 12.0e3  # 2*(100 + 100*50 + 50*16) + biases/activations + ACS
ACS_BYTES_PER_SYMBOL = 68.0  # 16 fp32 costs read + 1 fp32 decision written
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec peak


def golden_weights(device):
    """Briefly trained ViterbiNet weights (raw arrays captured from the reference, tests/golden)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
    return [torch.tensor(g[f"w{i}"], device=device) for i in range(6)]


def measured_traffic(kernel, B, prefix=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, collected with
    tools/pmc_traffic.sh on this same command); None when the workload differs from the profiled one."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        w = t["workload"]
        if (w["blocks"], w["block_length"], w["n_states"]) == (B, T, S):
            if prefix:  # template arguments vary (e.g. <0, true>): match on the kernel's base name
                kernel = next(k for k in t if k.startswith(kernel + "<"))
            return t[kernel]["bytes_per_launch"]
    except (OSError, KeyError, ValueError, StopIteration):
        pass
    return None


def event_time_ms(fn, iters, stream_device):
    """Average duration of fn() measured with HIP events on the stream the kernels are launched on
    (the ABI is called with torch's current stream, so torch.cuda.Event brackets exactly those launches)."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    for _ in range(3):  # warm: first launches of a variant pay cache / clock ramp effects
        fn()
    torch.cuda.synchronize(stream_device)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def cpu_baseline(weights_np, seed):
    """The CPU oracle (a port of the reference's algorithm, oracle/mvn_oracle.c) on a bounded sample of the
    same workload, all host cores.  Reported baseline only -- never part of `value`."""
    import oracle

    oracle.build()
    _, y = mvn.synthetic_words(64, T, L, SNR_DB, GAMMA, "cpu", seed=seed)
    y = y.numpy()
    t0 = time.perf_counter()
    oracle.vnet_decode(y, weights_np)
    dt = time.perf_counter() - t0
    blocks = int(min(200000, max(64, 15.0 / max(dt / 64, 1e-9))))  # ~15 s of CPU work
    _, y = mvn.synthetic_words(blocks, T, L, SNR_DB, GAMMA, "cpu", seed=seed)
    y = y.numpy()
    t0 = time.perf_counter()
    oracle.vnet_decode(y, weights_np)
    dt = time.perf_counter() - t0
    return {"value": blocks * T / dt, "unit": "symbols/s", "cores": oracle.max_threads(), "kind": "port",
            "sample": f"oracle.vnet_decode on {blocks} blocks x {T} symbols (same generator/weights), {dt:.1f} s"}


def cpu_baseline_torch_path(weights_np, seed):
    """The reference's own op sequence on torch-CPU (oracle/torch_path.py: per stage the same ATen calls as
    trellis_utils.py:16-30 inside vnet_detector.py:53-59), all host cores, bounded sample."""
    from oracle import torch_path

    blocks = 1000
    _, y = mvn.synthetic_words(blocks, T, L, SNR_DB, GAMMA, "cpu", seed=seed)
    torch_path.vnet_val(y[:50], weights_np)
    t0 = time.perf_counter()
    torch_path.vnet_val(y, weights_np)
    dt = time.perf_counter() - t0
    return {"value": blocks * T / dt, "unit": "symbols/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/torch_path.vnet_val (op-for-op PyTorch-CPU restatement of the reference's forward('val')) on "
                      f"{blocks} blocks x {T} symbols, {dt:.1f} s"}


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher: start the N ranks ourselves (one process per GPU),
    BEFORE anything touches the GPU in this process.  The parent never initialises HIP and never exec()s: it waits
    for its children, relays their output (rank 0 prints the JSON line) and exits non-zero if any rank failed."""
    import subprocess

    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:  # one rank died: the others would wait in a collective for ever
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:  # exact PIDs we started, nothing else
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=10000, help="blocks per GPU (BASELINE configs[1]: 10000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config entries (BASELINE configs[0],[2],[3],[4])")
    ap.add_argument("--skip-fused-count", action="store_true", help="(profiling) keep every fused-kernel dispatch decode-only: no fused-count timing, no FER curve")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        sys.exit("bench.py: no GPU visible; the hot path has no CPU fallback")
    if world > ndev and os.environ.get("MVN_BENCH_BACKEND", "nccl") == "nccl":
        sys.exit(f"bench.py: {world} ranks but {ndev} GPU(s): one process per GPU is the contract "
                 "(MVN_BENCH_BACKEND=gloo rehearses the control flow with ranks sharing a GPU)")
    local_dev = local_rank % ndev  # (ranks > devices only happens in the gloo rehearsal)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # MVN_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks;
    # the real run uses nccl (= RCCL over xGMI).
    backend = os.environ.get("MVN_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:  # gloo: stage through the host
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    B = args.blocks
    weights = golden_weights(dev)
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, w in zip(det.parameters(), weights):
            p.copy_(w)
    tx, y = mvn.synthetic_words(B, T, L, SNR_DB, GAMMA, dev, seed=3450002 + rank)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)

    def step():
        dec = det(y, "val", SNR_DB, GAMMA)
        mvn.count_errors(dec, tx, None, counters)

    for _ in range(args.warmup):
        step()
    if world > 1:
        all_reduce(counters)  # untimed: the first int64 all-reduce sets up RCCL's channels / kernels for this shape
    counters.zero_()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        all_reduce(counters)  # the single collective: int64[4] error counters over RCCL/xGMI
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    total_symbols = float(world) * B * T * args.steps
    ser, fer = mvn.rates_from_counters(counters)

    # FER@SNR / SER@SNR curve (the second half of BASELINE's metric), outside the timed region: fresh words per SNR
    # point on every rank, fused decode+count, one all-reduce of the int64[4] counters per point.
    fer_curve = []
    for snr_db in (() if args.skip_fused_count else (7.0, 8.0, 9.0, 10.0, 11.0, 12.0)):  # plotter_main.py:117-122: 7..12 dB
        txs, ys = mvn.synthetic_words(B, T, L, snr_db, GAMMA, dev, seed=7860002 + 100 * int(snr_db) + rank)
        c = det.val_count(ys, txs)
        if world > 1:
            all_reduce(c)
        s_, f_ = mvn.rates_from_counters(c)
        fer_curve.append({"snr_db": snr_db, "ser": s_, "fer": f_, "frames": int(c[3].item())})
        del txs, ys

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel, timed alone with HIP events on its launch stream
        lib = mvn._lib.load()
        st = mvn._lib.current_stream(dev)
        wp = [mvn._lib.ptr(w) for w in weights]
        dec = torch.zeros(B, T, device=dev)
        ms_fused = event_time_ms(lambda: lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *wp, mvn._lib.ptr(dec), T, None, None,
                                                                 None, 0, B, T, S, st), 5, dev)
        # secondary: the HBM-bound ACS sweep over materialised costs (mvn_acs_sweep_f32), same B x T x S
        cost = torch.randn(B, T, S, device=dev)
        ms_acs = event_time_ms(lambda: lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st),
                               5, dev)
        ms_step = event_time_ms(step, 3, dev)
        cfused = torch.zeros(4, dtype=torch.int64, device=dev)
        ms_fused_count = float("nan") if args.skip_fused_count else event_time_ms(lambda: det.val_count(y, tx, None, cfused), 10, dev)
        del cost
        mlp_tflops = FLOP_PER_SYMBOL * B * T / (ms_fused * 1e-3) / 1e12
        acs_gbps = ACS_BYTES_PER_SYMBOL * B * T / (ms_acs * 1e-3) / 1e9
        name_buf = ctypes.create_string_buffer(64)
        assert lib.mvn_acs_sweep_kernel_name(B, T, S, name_buf, 64) == 0
        acs_kernel = name_buf.value.decode()
        out = {
            "metric": "decoded symbols/sec, ViterbiNet L=4 ISI (16 states)",
            "value": total_symbols / elapsed,
            "unit": "symbols/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"ViterbiNet L=4, {B} blocks x {T} symbols per GPU (BASELINE configs[1])",
                       "blocks_per_gpu": B, "block_length": T, "n_states": S, "snr_db": SNR_DB,
                       "weights": "tests/golden/g7_by_word.npz (trained on the reference)",
                       "parallelism": f"block-sharded x{world}, one all-reduce of int64[4] counters"},
            "ser_at_snr": ser,
            "fer_at_snr": fer,
            "fer_curve": fer_curve,
            "roofline": {"kernel": "vnet16_fused4_kernel<false> (ViterbiNet MLP on f32 MFMA 16x16x4 + 4x4x1, in-place DPP trellis sweep)", "bound": "mfma",
                         "achieved": mlp_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": mlp_tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": measured_traffic("vnet16_fused4_kernel<false>", B), "traffic_unit": "HBM bytes/launch",
                         "algorithmic_hbm_bytes": 8.0 * B * T,
                         "ms_per_launch": ms_fused, "flop_per_symbol": FLOP_PER_SYMBOL},
            "roofline_acs_sweep": {"kernel": acs_kernel + "<COST> (mvn_acs_sweep_f32, LDS-DMA streamed costs)", "bound": "hbm",
                                   "achieved": acs_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                   "frac": acs_gbps / PEAK_HBM_GBPS,
                                   "traffic": measured_traffic(acs_kernel, B, prefix=True),
                                   "traffic_unit": "HBM bytes/launch", "algorithmic_hbm_bytes": ACS_BYTES_PER_SYMBOL * B * T,
                                   "ms_per_launch": ms_acs,
                                   "bytes_per_symbol": ACS_BYTES_PER_SYMBOL},
            "ms_per_step_events": ms_step,
            "fused_decode_count": {"what": "mvn_vnet_decode_count_f32: decode + error counting in one launch, no decision "
                                           "store (4 B/symbol of HBM traffic); same counters as `value`'s two-launch step",
                                   "ms_per_step": ms_fused_count, "symbols_per_s": B * T / (ms_fused_count * 1e-3)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline([w.cpu().numpy() for w in weights], 3450002)
            out["cpu_baseline_torch_path"] = cpu_baseline_torch_path([w.cpu().numpy() for w in weights], 3450002)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
