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


def csrc_sha16():
    """sha256 over the kernel sources: profiles/*.json measurements record the build they were taken on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "meta-viterbinet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profiled(path, kernel, field, workload):
    """A per-launch counter value (HBM bytes, VALU instructions) from a committed rocprofv3 PMC summary
    (profiles/traffic.json, profiles/valu_insts.json; collected with tools/pmc_traffic.sh on this same command).
    Returns (value, source): value is None when the file is for another workload or kernel, or was taken on another
    build of the kernels (the file records the sha of csrc/ it was measured on) -- a stale number is not reported."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", path)))
        if workload is not None and [t["workload"][k] for k in ("blocks", "block_length", "n_states")] != list(workload):
            return None, f"profiles/{path}: profiled workload differs"
        if t.get("csrc_sha16") != csrc_sha16():
            return None, f"profiles/{path}: taken on another build of csrc/ ({t.get('csrc_sha16')}), not reported"
        if kernel not in t and kernel.endswith(">") and kernel[:-1] + ", false>" in t:
            kernel = kernel[:-1] + ", false>"  # (the profiler prints a defaulted template argument the name query leaves out: sweep16_quad_kernel's SURV)
        return t[kernel][field], f"profiles/{path} (offline rocprofv3 --pmc passes on this command, csrc {t['csrc_sha16']})"
    except (OSError, KeyError, ValueError):
        return None, f"profiles/{path}: no entry for {kernel}"


def event_time_ms(fn, iters, stream_device):
    """Average duration of fn() measured with HIP events on the stream the kernels are launched on
    (the ABI is called with torch's current stream, so torch.cuda.Event brackets exactly those launches)."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    for _ in range(5):  # warm: first launches of a variant pay cache effects; after idle stretches also the clock ramp
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
    trellis_utils.py:16-30 inside vnet_detector.py:53-59) on a bounded sample, at the thread count that serves it best:
    [1000, 16]-sized ops do not scale to a 128-thread host (oversubscription), so the count is swept and the best reported."""
    from oracle import torch_path

    blocks = 1000
    _, y = mvn.synthetic_words(blocks, T, L, SNR_DB, GAMMA, "cpu", seed=seed)
    all_threads = torch.get_num_threads()
    sweep = {}
    try:
        for n in sorted({t for t in (4, 8, 16, 32, all_threads) if t <= all_threads}):
            torch.set_num_threads(n)
            torch_path.vnet_val(y[:50], weights_np)
            t0 = time.perf_counter()
            torch_path.vnet_val(y, weights_np)
            sweep[n] = blocks * T / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(all_threads)
    best = max(sweep, key=sweep.get)
    return {"value": sweep[best], "unit": "symbols/s", "cores": best, "kind": "port",
            "symbols_per_s_by_threads": {str(k): v for k, v in sweep.items()},
            "sample": f"oracle/torch_path.vnet_val (op-for-op PyTorch-CPU restatement of the reference's forward('val')) on "
                      f"{blocks} blocks x {T} symbols, best of torch.set_num_threads {sorted(sweep)} (host has {all_threads})"}


VALU_PEAK_GIPS = 1024 * 2.4 / 2.0  # wave-instructions/ns: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU op (MI355X_MICROARCH.md)


def kernel_name(fn, *a):
    buf = ctypes.create_string_buffer(96)
    assert fn(*a, buf, 96) == 0
    return buf.value.decode()


def valu_roofline(kernel, B, T_, S_, ms):
    """VALU-issue roofline of a VALU-bound kernel: wave-instructions per launch (SQ_INSTS_VALU, committed PMC pass) /
    launch time, against 1024 SIMDs x one wave64 instruction per 2 cycles."""
    insts, src = profiled("valu_insts.json", kernel, "valu_insts_per_launch", None)
    if insts is not None:
        w = json.load(open(os.path.join(ROOT, "profiles", "valu_insts.json")))[kernel]
        if [w["blocks"], w["block_length"], w["n_states"]] != [B, T_, S_]:
            insts, src = None, "profiles/valu_insts.json: profiled workload differs"
    ach = None if insts is None else insts / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "bound": "valu", "achieved": ach, "peak": VALU_PEAK_GIPS * 1e9 / 1e9, "unit": "G wave-instr/s",
            "frac": None if ach is None else ach / VALU_PEAK_GIPS, "valu_insts_per_launch": insts, "insts_source": src,
            "state_steps_per_s": B * T_ * S_ / (ms * 1e-3), "algorithmic_hbm_bytes": 8.0 * B * T_, "ms_per_launch": ms}


def wall_ms(fn, dev, reps=1):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps * 1e3, r


def run_configs(dev, rank, world, all_reduce, backend, weights, by_word=True):
    """The other BASELINE.json configs, outside the headline timed region; every rank takes part (collectives), rank 0
    reports.  [0] VA L=4 100 x 1000 (one GPU's worth; rank 0), [2] ViterbiNet over the COST2100 taps, 300 blocks by word
    (block-sharded when no update runs between blocks, replicas otherwise), [3] VA L=8 at the per-GPU share of 10^6
    blocks, [4] the Meta-ViterbiNet online flow with the reference's default counts (replicas)."""
    lib = mvn._lib.load()
    st = mvn._lib.current_stream(dev)
    out = []
    coef = {"train": "time_decay", "val": "time_decay"}

    def max_over_ranks(ms):
        if world == 1:
            return ms
        t = torch.tensor([ms], dtype=torch.float64, device=dev)
        all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- configs[0]: classical VA, L=4, 100 blocks x 1000 symbols (the reference's CPU-runnable case)
    B0, T0 = 100, 1000
    tx0, y0 = mvn.synthetic_words(B0, T0, 4, SNR_DB, GAMMA, dev, seed=3450002)
    va = mvn.VADetector(16, 4, T0, 1, "ISI_AWGN", 0, False, 1, coef)
    pri = va.compute_state_priors(mvn.estimate_channel(4, GAMMA, "time_decay")).to(dev).T.contiguous()
    dec0 = torch.zeros_like(y0)
    ms0 = event_time_ms(lambda: lib.mvn_va_decode_f32(mvn._lib.ptr(y0), T0, mvn._lib.ptr(pri), 1, mvn._lib.ptr(dec0), T0, None,
                                                      B0, T0, 16, st), 20, dev)
    ms0_fwd = event_time_ms(lambda: va(y0, "val", SNR_DB, GAMMA), 10, dev)
    ser0, fer0 = mvn.rates_from_counters(mvn.count_errors(va(y0, "val", SNR_DB, GAMMA), tx0))
    k0 = kernel_name(lib.mvn_va_decode_kernel_name, B0, T0, 16)
    out.append({"config": "BASELINE configs[0]: classical VA, L=4 (16 states), 100 blocks x 1000 symbols", "n_gpus": 1,
                "ms": ms0, "symbols_per_s": B0 * T0 / (ms0 * 1e-3), "ms_detector_forward": ms0_fwd, "ser": ser0, "fer": fer0,
                "kernel": k0, "roofline": valu_roofline(k0, B0, T0, 16, ms0),
                "note": "100 four-wave workgroups on a 4096-slot chip (va16_split_kernel: a block's recurrence on one wave, its decisions "
                        "on three others): bound by the one recurrence wave's instruction stream, ~48 cycles per trellis step"})

    # ---- configs[3]: VA L=8 (256 states), 10^6 blocks over 8 GPUs = 125 000 blocks x 1000 symbols per GPU (weak scaling)
    B3, T3, L3 = 125000, 1000, 8
    tx3, y3 = mvn.synthetic_words(B3, T3, L3, SNR_DB, GAMMA, dev, seed=3450002 + 17 * rank)
    va8 = mvn.VADetector(256, L3, T3, 1, "ISI_AWGN", 0, False, 1, coef)
    pri8 = va8.compute_state_priors(mvn.estimate_channel(L3, GAMMA, "time_decay")).to(dev).T.contiguous()
    dec3 = torch.empty_like(y3)
    ms3 = max_over_ranks(event_time_ms(lambda: lib.mvn_va_decode_f32(mvn._lib.ptr(y3), T3, mvn._lib.ptr(pri8), 1, mvn._lib.ptr(dec3),
                                                                     T3, None, B3, T3, 256, st), 5, dev))
    c3 = mvn.count_errors(dec3, tx3)
    if world > 1:
        all_reduce(c3)
    ser3, fer3 = mvn.rates_from_counters(c3)
    k3 = kernel_name(lib.mvn_va_decode_kernel_name, B3, T3, 256)
    # the same Monte-Carlo point in ONE launch, words generated inside the detector (mvn_va_montecarlo_f32: no tx, y, decisions in HBM)
    mc_seed = 3450002 + 17 * rank
    c3f = mvn.va_monte_carlo(va8, B3, SNR_DB, GAMMA, dev, mc_seed)
    ms3f = max_over_ranks(event_time_ms(lambda: mvn.va_monte_carlo(va8, B3, SNR_DB, GAMMA, dev, mc_seed, counters=c3f), 3, dev))
    out.append({"config": "BASELINE configs[3]: VA L=8 (256 states), 10^6 blocks x 1000 symbols over 8 GPUs = 125 000 blocks per GPU",
                "n_gpus": world, "blocks_per_gpu": B3, "ms": ms3, "symbols_per_s": world * B3 * T3 / (ms3 * 1e-3),
                "ser": ser3, "fer": fer3, "frames": int(c3[3].item()), "kernel": k3, "roofline": valu_roofline(k3, B3, T3, 256, ms3),
                "fused_monte_carlo": {"kernel": "va256_mc_kernel", "ms": ms3f, "symbols_per_s": world * B3 * T3 / (ms3f * 1e-3),
                                      "hbm_bytes_per_symbol": 0, "vs_generate_decode_count_bytes_per_symbol": 12,
                                      "what": "generate + detect + count in one launch (words never in memory); counters equal the three launches' "
                                              "on the same seed (tests/test_gpu_parity.py::test_va_monte_carlo_equals_the_three_launches)"},
                "parallelism": f"block-sharded x{world}, one all-reduce of int64[4] counters"})
    del tx3, y3, dec3

    # ---- ViterbiNet at the other trellis sizes (vnet_detector.py:35-61 takes any n_states): 10 000 blocks x 1000 symbols per GPU,
    # weights at random initialisation.  FLOP per symbol as for 16 states: 2 (100 + 100 x 50 + 50 S) + biases / activations + ACS
    states = []
    for S_ in (4, 8, 32, 64, 128):
        Bs, Ts = 10000, 1000
        torch.manual_seed(S_)
        det_s = mvn.VNETDetector(S_, {"train": Ts, "val": Ts}).to(dev)
        ws_ = [p.detach().contiguous() for p in det_s.parameters()]
        ys = torch.randn(Bs, Ts, device=dev) * 1.5
        decs = torch.empty_like(ys)
        nws = int(lib.mvn_vnet_workspace_bytes(Bs, Ts, S_))
        wss = torch.empty(max(nws, 4), dtype=torch.uint8, device=dev)
        ms_s = max_over_ranks(event_time_ms(lambda: lib.mvn_vnet_decode_f32(mvn._lib.ptr(ys), Ts, *[mvn._lib.ptr(a) for a in ws_], mvn._lib.ptr(decs), Ts,
                                                                           None, None, mvn._lib.ptr(wss), nws, Bs, Ts, S_, st), 10, dev))
        flop_s = FLOP_PER_SYMBOL + 2 * 50 * (S_ - 16) + 4 * (S_ - 16)
        ach = flop_s * Bs * Ts / (ms_s * 1e-3) / 1e12
        states.append({"n_states": S_, "ms": ms_s, "symbols_per_s": world * Bs * Ts / (ms_s * 1e-3), "kernel": kernel_name(lib.mvn_vnet_decode_kernel_name, Bs, Ts, S_, 0),
                       "workspace_bytes": nws,
                       "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                                    "flop_per_symbol": flop_s}})
        del ys, decs, wss
    out.append({"config": "ViterbiNet at 4 / 8 / 32 / 64 / 128 states, 10 000 blocks x 1000 symbols per GPU (the fused kernel of every other trellis size)",
                "n_gpus": world, "blocks_per_gpu": 10000, "by_states": states, "parallelism": f"block-sharded x{world}"})

    if not by_word:  # (profiling runs: the two VA configs only)
        return out
    # ---- configs[2] / [4]: 300 blocks by word (T = 120 + 8*2, RS(17,15)), ViterbiNet weights trained on the reference
    N, K, nsym, L, sub = 300, 120, 2, 4, 25
    g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))

    def make_det():
        det = mvn.VNETDetector(16, {"train": K + 8 * nsym, "val": K + 8 * nsym}).to(dev)
        with torch.no_grad():
            for p_, i in zip(det.parameters(), range(6)):
                p_.copy_(torch.tensor(g7[f"w{i}"]))
        return det

    def words(coefficients, snr, seed):
        gen = torch.Generator(device=dev).manual_seed(seed)
        msg = torch.randint(0, 2, (N, K), generator=gen, device=dev).float()
        cw = mvn.rs_encode(msg, nsym)
        if coefficients == "cost2100":
            h = np.concatenate([mvn.estimate_channel(L, GAMMA, "cost2100", index=i) for i in range(N)])
        else:
            h = np.concatenate([mvn.estimate_channel(L, GAMMA, "time_decay", fading=True, index=i, fading_taps_type=2)
                                for i in range(N)])
        return msg, mvn.transmit(cw, h, snr, L, torch.randn(N, K + 8 * nsym, generator=gen, device=dev))

    rows = mvn.data_indices(N // sub, sub).to(dev)
    msg, rx = words("cost2100", SNR_DB, 5)
    det = make_det()
    # (a) no update between blocks: blocks are independent -> ONE batched, block-sharded call (decode, RS decode, count)
    mvn.sharded_eval(det, msg, rx, SNR_DB, GAMMA, rows, n_symbols=nsym, rank=rank, world=world)
    ms2a, (ser2a, fer2a, c2a) = wall_ms(lambda: mvn.sharded_eval(det, msg, rx, SNR_DB, GAMMA, rows, n_symbols=nsym, rank=rank,
                                                                 world=world), dev, reps=5)
    # (b) the reference's call pattern: 300 sequential B=1 calls, no update (rank 0)
    mvn.eval_by_word(det, msg[:5], rx[:5], SNR_DB, GAMMA, nsym, sub)
    ms2b, ser_seq = wall_ms(lambda: mvn.eval_by_word(det, msg, rx, SNR_DB, GAMMA, nsym, sub), dev)
    kfused = "byword_step_kernel<2>"

    # (c) self-supervised online training after every block: does not shard within a trial -> replicas (SNR x seed grid),
    # R trials per GPU advancing together (trials.eval_by_word_batched); one trial alone for comparison
    from meta_viterbinet_amd.trials import TrialBank, TrialDraws, eval_by_word_batched

    T2 = K + 8 * nsym
    w_np = [g7[f"w{i}"] for i in range(6)]
    n_cu = ctypes.c_int()
    lib.mvn_device_info(ctypes.byref(n_cu), None, None, 0)

    def one_trial(coefficients, i, seed0, **kw):
        """A run of harness.eval_by_word alone, ready to be timed: words, detector, trainer and the trial's drawn minibatch
        table are inputs, built here like trial_batch builds them before ITS clock starts."""
        m_, r_ = words(coefficients, 7.0 + (i % 6), seed0 + i)  # plotter_main.py:117-122: 7..12 dB
        d_ = make_det()
        tr_, md_, dr_ = mvn.OnlineTrainer(d_, L), mvn.META_VNETDetector(16, {"train": T2, "val": T2}), TrialDraws(seed0 + i, dev)
        if not kw.get("meta_style_online_training"):
            dr_.batches(0, N, T2, kw["self_supervised_iterations"], 32)
        return lambda: mvn.eval_by_word(d_, m_, r_, 7.0 + (i % 6), GAMMA, nsym, sub, online_trainer=tr_, meta_detector=md_, draws=dr_, **kw)

    def trial_batch(coefficients, seed0, stats, **kw):
        def run(ids):
            ws_ = [words(coefficients, 7.0 + (i % 6), seed0 + i) for i in ids]
            bank = TrialBank([w_np] * len(ids), 16, L, dev)
            draws = [TrialDraws(seed0 + i, dev) for i in ids]
            if not kw.get("meta_style_online_training"):
                for d_ in draws:  # the minibatch tables are inputs of the run, like the words
                    d_.batches(0, N, T2, kw["self_supervised_iterations"], 32)
            rec = {}
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ser = eval_by_word_batched(bank, torch.stack([a for a, _ in ws_]), torch.stack([b for _, b in ws_]), nsym, sub, draws,
                                       record=rec, **kw)
            torch.cuda.synchronize(dev)
            stats.update(ms=(time.perf_counter() - t0) * 1e3, trained=int(rec["trained"].sum()), meta=int(rec["meta"].sum()),
                         adam_steps=int(bank.step.sum()))
            return ser
        return run

    def training_roofline(stats, trials, iters, samples, maml, groups):
        """Utilisation of the chip by the training launches: algorithmic MFMA FLOPs of all trials' training (35 kFLOP per
        sample of a CrossEntropy forward + backward pass of the 6 066-parameter MLP; a second-order meta-learning step =
        support + query gradient + a Hessian-vector pass of ~3 gradient passes over the support word) over the wall time of the
        whole evaluation (detection, codec, host decisions included), against the f32 MFMA peak; cu_occupancy = CUs holding
        a training workgroup while a training launch runs."""
        online_steps = stats["trained"] * iters
        maml_steps = stats["adam_steps"] - online_steps
        flop = 35e3 * samples * online_steps + (35e3 * 2 * T2 + 105e3 * T2) * maml_steps
        ach = flop / (stats["ms"] * 1e-3) / 1e12
        # the form the library itself picks for this many trials (mvn_vnet_train_kernel_name: chunked passes finish a trial
        # soonest, one workgroup per trial gets the most trials through a CU per second)
        ws_b = int(lib.mvn_vnet_train_trials_workspace_bytes(16, T2, 1, trials))
        # the launcher decides per call from the trials that are ACTIVE in it: priced at the mean active count per block step
        act_online = max(1, min(trials, round(stats["trained"] / N)))
        act_meta = max(1, min(trials, round(stats["meta"] / max(1, N // 5)))) if maml else 0
        form_online = kernel_name(lib.mvn_vnet_train_kernel_name, 0, act_online, T2, 0 if samples == T2 else samples, 16, ws_b)
        form_maml = kernel_name(lib.mvn_vnet_train_kernel_name, 2, act_meta, T2, 1, 16, ws_b) if maml else None
        form = form_maml if maml else form_online
        groups, per_launch = (int(v) for v in form.split("> ")[1].split(" ")[0].split("x"))
        groups_online = int(form_online.split("> ")[1].split(" ")[0].split("x")[0])
        # the same flow priced by the MFMA instructions its training launches EXECUTE (SQ_INSTS_MFMA per iteration and trial from
        # the committed PMC passes of tools/prof_train_kernels.py, profiles/train_pmc.json; x 2048 FLOP), and what those passes say
        # the dominant training kernel waits for while it runs
        executed, pmc_src, in_kernel = None, None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "train_pmc.json")))
            if pmc.get("csrc_sha16") != csrc_sha16():
                pmc_src = f"profiles/train_pmc.json: taken on another build of csrc/ ({pmc.get('csrc_sha16')}), not reported"
            else:
                c = pmc["cases"]
                online_case = c["online_minibatch" if samples != T2 else ("online_full_word" if groups_online == 1 else "online_full_word_chunked")]
                maml_case = c["maml_second_order" if groups == 1 else "maml_second_order_chunked"]
                mfma = online_case["mfma_per_iteration"] * online_steps + maml_case["mfma_per_iteration"] * maml_steps
                executed = mfma * 2048.0 / (stats["ms"] * 1e-3) / 1e12
                dom = maml_case if maml else online_case
                in_kernel = {k: dom[k] for k in ("kernel", "mfma_tflops_in_kernel", "mfma_busy", "lds_busy", "lds_bank_conflict_share_of_lds_cycles",
                                                 "wait_barrier_or_waitcnt", "wait_issue", "issuing")}
                pmc_src = f"profiles/train_pmc.json (rocprofv3 --pmc passes of tools/prof_train_kernels.py, csrc {pmc['csrc_sha16']})"
        except (OSError, KeyError, ValueError) as e:
            pmc_src = f"profiles/train_pmc.json: {type(e).__name__}"
        return {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "achieved_executed_mfma": executed, "frac_executed_mfma": None if executed is None else executed / PEAK_F32_MFMA_TFLOPS,
                "dominant_kernel_while_running": in_kernel, "pmc_source": pmc_src,
                "cu_occupancy": min(1.0, per_launch * groups / n_cu.value), "training_kernel": form, "online_training_kernel": form_online,
                "forms_priced_at": {"active_trials_per_online_call": act_online, "active_trials_per_meta_call": act_meta,
                                    "note": "the launcher picks the form per call from the trials active in it; mean counts of this run"},
                "workgroups_per_trial": groups,
                "trials_per_launch": per_launch,
                "adam_steps": stats["adam_steps"], "algorithmic_flop": flop}

    def reference_grid(run, stats, iters, samples, maml, groups):
        """The reference's OWN grid -- 6 SNR points, 7..12 dB, one seed (plotter_main.py:117-122), which it walks one after the
        other -- as 6 trials stepping together on one GPU (rank 0; no collective)."""
        ser = run(list(range(6)))
        return {"trials": 6, "ms": stats["ms"], "ms_per_block_step": stats["ms"] / N, "ms_per_block_per_trial": stats["ms"] / (6 * N),
                "blocks_per_s": 6 * N / (stats["ms"] * 1e-3), "mean_ser_by_snr_db": {str(7 + k): float(np.mean(ser[k])) for k in range(6)},
                "roofline": training_roofline(stats, 6, iters, samples, maml, groups),
                "what": "the reference's 6-point SNR grid (7..12 dB, one seed) as 6 trials stepping together on ONE GPU"}

    R2 = int(os.environ.get("MVN_BENCH_TRIALS_SELFSUP", "256"))  # one workgroup per trial: a trial per CU
    kw2 = dict(self_supervised=True, self_supervised_iterations=200)
    one_trial("cost2100", 0, 100, **kw2)()  # warm
    ms2c1, _ = wall_ms(one_trial("cost2100", 0, 100, **kw2), dev)
    st2 = {}
    run2 = trial_batch("cost2100", 100, st2, **kw2)
    run2(list(range(rank, min(R2 * world, 4 * world), world)))  # warm
    grid2 = reference_grid(run2, st2, 200, 32, False, 1) if rank == 0 else None
    rep2 = mvn.replica_eval(run2, R2 * world, rank=rank, world=world, device=dev if backend == "nccl" else "cpu", batched=True)
    ms2c = max_over_ranks(st2["ms"])
    out.append({"config": "BASELINE configs[2]: ViterbiNet L=4, COST2100 taps, 300-block evaluation by word (T=136, RS(17,15))",
                "n_gpus": world, "kernel": kfused,
                "joint_batched_block_sharded": {"ms": ms2a, "symbols_per_s": N * T2 / (ms2a * 1e-3), "coded_ser": ser2a,
                                                "what": "no update between blocks => independent: one decode + one RS + one count launch "
                                                        f"for all 300 blocks, rows sharded x{world}, one all-reduce of int64[4]"},
                "joint_sequential_b1": {"ms": ms2b, "us_per_block": ms2b * 1e3 / N, "symbols_per_s": N * T2 / (ms2b * 1e-3),
                                        "mean_ser": float(np.mean(ser_seq)),
                                        "what": "the reference's sequential pattern without updates: per data block ONE launch of "
                                                "byword_step_kernel (detect B=1 + RS decode + error count + re-encode); the 12 pilot "
                                                "blocks are skipped (their detection is never used), per-block error counts stay on the "
                                                "device and are read back once at the end"},
                "self_supervised_one_trial": {"ms": ms2c1, "blocks_per_s": N / (ms2c1 * 1e-3),
                                              "what": "harness.eval_by_word alone: one fused step launch + one host sync per block, 200 "
                                                      "CE+Adam iterations per qualifying block in one launch of online_train_kernel"},
                "self_supervised_reference_grid": grid2,
                "self_supervised_trials": {"trials_per_gpu": R2, "ms": ms2c, "us_per_block_step": ms2c * 1e3 / N,
                                           "blocks_per_s": world * R2 * N / (ms2c * 1e-3), "symbols_per_s": world * R2 * N * T2 / (ms2c * 1e-3),
                                           "speedup_vs_one_trial_at_a_time": (R2 * N / ms2c) / (N / ms2c1),
                                           "mean_ser_by_snr_db": {str(7 + k): float(np.nanmean(rep2[k::6])) for k in range(6)},
                                           "roofline": training_roofline(st2, R2, 200, 32, False, 1),
                                           "what": f"{R2} independent trials per GPU (SNR 7..12 dB x seeds, plotter_main.py:117-149) stepping "
                                                   "together: per block step one byword_step_kernel launch for all trials, one host sync, one "
                                                   "online_train_kernel launch (gridDim.y = trial) for the trials that train; per trial "
                                                   "bit-identical to the one-trial run; one all_gather of ser_by_word[300] per trial"},
                "roofline": None,
                "note": "the joint variants are one wave of work per launch (launch-latency-bound); the self-supervised variant is bound by "
                        "the training launches, see self_supervised_trials.roofline"})

    # ---- configs[4]: Meta-ViterbiNet online retrain + decode, reference defaults (200 / 20 / 10 / 5), replicas
    R4 = int(os.environ.get("MVN_BENCH_TRIALS_META", "256"))  # a trial per CU: the library then runs one workgroup per trial
    kw4 = dict(self_supervised=True, self_supervised_iterations=200, online_meta=True, meta_train_iterations=20, meta_j_num=10,
               meta_subframes=5, meta_style_online_training=True)
    one_trial("time_decay", 0, 200, **kw4)()  # warm (first launches of the meta-learning kernels, workspace), as for configs[2]
    ms41, _ = wall_ms(one_trial("time_decay", 0, 200, **kw4), dev)
    st4 = {}
    run4 = trial_batch("time_decay", 200, st4, **kw4)
    grid4 = reference_grid(run4, st4, 200, T2, True, 5) if rank == 0 else None
    rep4 = mvn.replica_eval(run4, R4 * world, rank=rank, world=world, device=dev if backend == "nccl" else "cpu", batched=True)
    ms4 = max_over_ranks(st4["ms"])
    roof4 = training_roofline(st4, R4, 200, T2, True, 5)
    # more trials than CUs: the launcher then runs the whole-word iterations as two 512-thread workgroups per CU
    R4m = int(os.environ.get("MVN_BENCH_TRIALS_META_MORE", str(2 * n_cu.value)))
    more4 = None
    if R4m > R4:
        mvn.replica_eval(run4, R4m * world, rank=rank, world=world, device=dev if backend == "nccl" else "cpu", batched=True)
        ms4m = max_over_ranks(st4["ms"])
        more4 = {"trials_per_gpu": R4m, "ms": ms4m, "ms_per_block_step": ms4m / N, "blocks_per_s": world * R4m * N / (ms4m * 1e-3),
                 "roofline": training_roofline(st4, R4m, 200, T2, True, 5),
                 "what": "the same flow with two trials per CU: the whole-word iterations run as two 512-thread workgroups per CU "
                         "(online_train_kernel<16, true, 512>), the meta-learning steps as two rounds of one workgroup per CU"}
    out.append({"config": "BASELINE configs[4]: Meta-ViterbiNet online retrain + decode, L=4, pilot-aided, 300 blocks (reference defaults: "
                          "200 full-word iterations per block, every 5 blocks 20 x <=10 MAML steps)",
                "n_gpus": world, "trials_per_gpu": R4, "ms": ms4, "ms_per_block_step": ms4 / N, "blocks_per_s": world * R4 * N / (ms4 * 1e-3),
                "symbols_per_s": world * R4 * N * T2 / (ms4 * 1e-3),
                "one_trial": {"ms": ms41, "ms_per_block": ms41 / N, "blocks_per_s": N / (ms41 * 1e-3)},
                "reference_grid": grid4,
                "more_trials_than_cus": more4,
                "speedup_vs_one_trial_at_a_time": (R4 * N / ms4) / (N / ms41),
                "mean_ser_by_snr_db": {str(7 + k): float(np.nanmean(rep4[k::6])) for k in range(6)},
                "kernel": "maml_train(_groups)_kernel + online_train(_groups)_kernel + byword_step_kernel",
                "what": f"{R4} independent trial(s) per GPU stepping together (replicas: block k's weights depend on the blocks before it); "
                        "per block step one byword_step_kernel launch, one host sync, and for the trials that train one launch sequence of "
                        "the meta-learning and the online-training kernel (gridDim.y = trial): one workgroup per 32-sample chunk and trial "
                        "(5 workgroups, never more per launch than CUs) for few trials, one workgroup per trial from ~160 trials on "
                        "(more trials through a CU per second; same bits); one all_gather of ser_by_word[300] per trial",
                "roofline": roof4,
                "note": "training passes run one workgroup per 32-sample chunk and trial, gradients exchanged through a per-trial workspace "
                        "with one device-wide barrier per pass (DESIGN.md 5.6, 5.7)"})
    return out



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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=10000, help="blocks per GPU (BASELINE configs[1]: 10000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained-rate loop")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling entry (the same 10 000 blocks split over the ranks)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0)
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config entries (BASELINE configs[0],[2],[3],[4])")
    ap.add_argument("--skip-fused-count", action="store_true", help="(profiling) keep every fused-kernel dispatch decode-only: no fused-count timing, no FER curve, no by-word configs")
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

    # The GPU's clock / power state settles over ~30 ms of sustained load (tools/time_ramp.py: the first 20 launches of the
    # fused kernel run 1.52 ms, every later one 1.37): bring the device to its sustained state first, so that the W warm-up
    # and K timed steps below measure the same machine whatever W is.  Untimed, bounded (60 ms).
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.06:
        step()
        torch.cuda.synchronize(dev)
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

    ser, fer = mvn.rates_from_counters(counters)
    # what every rank decoded inside the timed region, gathered so that the line shows N ranks took part (and `value` is
    # their sum over the max-over-ranks time); the counters' frame count says the same through the all-reduce
    mine = torch.tensor([B * T * args.steps], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    symbols_by_rank = [int(t.item()) for t in per_rank]
    total_symbols = float(sum(symbols_by_rank))
    collective = {"backend": (dist.get_backend() if world > 1 else None), "world_size": (dist.get_world_size() if world > 1 else 1),
                  "rccl_version": None, "symbols_by_rank": symbols_by_rank, "frames_all_ranks": int(counters[3].item()),
                  "frames_expected": world * B * args.steps}
    try:
        v = torch.cuda.nccl.version()
        collective["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001 (a torch build without the binding: say so instead of failing the bench)
        collective["rccl_version"] = f"unavailable ({type(e).__name__})"

    # sustained rate: the same step() for >= 2 s (the timed region above is tens of milliseconds), outside `value`
    sustained = None
    if rank == 0 and not args.no_sustained:
        n_sus, t_sus = 0, time.perf_counter()
        while time.perf_counter() - t_sus < args.sustained_seconds:
            for _ in range(20):
                step()
            torch.cuda.synchronize(dev)
            n_sus += 20
        dt_sus = time.perf_counter() - t_sus
        sustained = {"seconds": dt_sus, "steps": n_sus, "symbols_per_s": n_sus * B * T / dt_sus, "ms_per_step": dt_sus / n_sus * 1e3,
                     "what": "rank 0's step() back to back for the stated time, one host sync per 20 steps"}
    if world > 1:
        dist.barrier()

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

    # ---- strong scaling of the same workload: BASELINE configs[1]'s 10 000 blocks SPLIT over the ranks (the reference calls the
    # detector with whatever its evaluation holds, trainer.py:232), next to the weak-scaling headline above.  Outside `value`.
    strong = None
    if not args.no_strong:
        lo_s, hi_s = mvn.shard_range(args.blocks, rank, world)
        ys, txs_ = y[lo_s:hi_s], tx[lo_s:hi_s]  # (rank r's share of ITS 10 000 words: same distribution on every rank)
        cs = torch.zeros(4, dtype=torch.int64, device=dev)

        def step_strong():
            mvn.count_errors(det(ys, "val", SNR_DB, GAMMA), txs_, None, cs)

        for _ in range(max(args.warmup, 5)):
            step_strong()
        cs.zero_()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0s = time.perf_counter()
        for _ in range(args.steps):
            step_strong()
        if world > 1:
            all_reduce(cs)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        el_s = time.perf_counter() - t0s
        if world > 1:
            te = torch.tensor([el_s], dtype=torch.float64, device=dev)
            all_reduce(te, op=dist.ReduceOp.MAX)
            el_s = float(te.item())
        nb_s = hi_s - lo_s
        kname = ctypes.create_string_buffer(96)
        mvn._lib.load().mvn_vnet_decode_kernel_name(nb_s, T, S, 0, kname, 96)
        strong = {"scaling": "strong", "total_blocks": args.blocks, "blocks_per_gpu": nb_s, "steps": args.steps,
                  "ms_per_step": el_s / args.steps * 1e3, "symbols_per_s": args.blocks * T * args.steps / el_s,
                  "frames_counted": int(cs[3].item()), "frames_expected": args.blocks * args.steps, "kernel": kname.value.decode(),
                  "mfma_frac_of_this_rank": FLOP_PER_SYMBOL * nb_s * T / (el_s / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                  "what": f"the SAME {args.blocks} blocks x {T} symbols split over {world} rank(s) (contiguous shares, harness.shard_range), "
                          "forward('val') + on-device counting per step, one all-reduce of int64[4] at the end, max over ranks; the "
                          "per-rank fraction includes the counting launch"}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel, timed alone with HIP events on its launch stream
        lib = mvn._lib.load()
        st = mvn._lib.current_stream(dev)
        wp = [mvn._lib.ptr(w) for w in weights]
        dec = torch.zeros(B, T, device=dev)
        ws_bytes = int(lib.mvn_vnet_workspace_bytes(B, T, S))  # the dealt kernel's hand-off lines (what VNETDetector.forward passes)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        ms_fused = event_time_ms(lambda: lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *wp, mvn._lib.ptr(dec), T, None, None,
                                                                 mvn._lib.ptr(ws), ws_bytes, B, T, S, st), 5, dev)
        # secondary: the HBM-bound ACS sweep over materialised costs (mvn_acs_sweep_f32), same B x T x S
        cost = torch.randn(B, T, S, device=dev)
        ms_acs = event_time_ms(lambda: lib.mvn_acs_sweep_f32(mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, None, B, T, S, st),
                               5, dev)
        ms_step = event_time_ms(step, 3, dev)
        cfused = torch.zeros(4, dtype=torch.int64, device=dev)
        ms_fused_count = float("nan") if args.skip_fused_count else event_time_ms(lambda: det.val_count(y, tx, None, cfused), 10, dev)
        mlp_tflops = FLOP_PER_SYMBOL * B * T / (ms_fused * 1e-3) / 1e12
        acs_gbps = ACS_BYTES_PER_SYMBOL * B * T / (ms_acs * 1e-3) / 1e9
        acs_kernel = kernel_name(lib.mvn_acs_sweep_kernel_name, mvn._lib.ptr(cost), mvn._lib.ptr(dec), T, B, T, S)
        fused_kernel = kernel_name(lib.mvn_vnet_decode_kernel_name, B, T, S, 0)
        # (rocprofv3 names the kernel without the launcher's " rings of N" suffix)
        traffic_fused, traffic_fused_src = profiled("traffic.json", fused_kernel.split(" rings of")[0], "bytes_per_launch", (B, T, S))
        traffic_acs, traffic_acs_src = profiled("traffic.json", acs_kernel, "bytes_per_launch", (B, T, S))
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
            "collective": collective,
            "sustained": sustained,
            "ser_at_snr": ser,
            "fer_at_snr": fer,
            "fer_curve": fer_curve,
            "roofline": {"kernel": fused_kernel + " (ViterbiNet MLP on f32 MFMA 16x16x4, in-place DPP trellis sweep)", "bound": "mfma",
                         "achieved": mlp_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": mlp_tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic_fused, "traffic_unit": "HBM bytes/launch", "traffic_source": traffic_fused_src,
                         "algorithmic_hbm_bytes": 8.0 * B * T,
                         "ms_per_launch": ms_fused, "flop_per_symbol": FLOP_PER_SYMBOL},
            "roofline_acs_sweep": {"kernel": acs_kernel + " (mvn_acs_sweep_f32, LDS-DMA streamed costs)", "bound": "hbm",
                                   "achieved": acs_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                   "frac": acs_gbps / PEAK_HBM_GBPS,
                                   "traffic": traffic_acs, "traffic_unit": "HBM bytes/launch", "traffic_source": traffic_acs_src, "algorithmic_hbm_bytes": ACS_BYTES_PER_SYMBOL * B * T,
                                   "ms_per_launch": ms_acs,
                                   "bytes_per_symbol": ACS_BYTES_PER_SYMBOL},
            "ms_per_step_events": ms_step,
            "strong_scaling": strong,
            "fused_decode_count": {"what": "mvn_vnet_decode_count_f32: decode + error counting in one launch, no decision "
                                           "store (4 B/symbol of HBM traffic); same counters as `value`'s two-launch step",
                                   "ms_per_step": ms_fused_count, "symbols_per_s": B * T / (ms_fused_count * 1e-3)},
        }
    # the other BASELINE configs: after the headline kernels were timed (the by-word configs are latency-bound and let the
    # clocks drop), every rank takes part
    configs = [] if args.no_configs else run_configs(dev, rank, world, all_reduce, backend, weights, by_word=not args.skip_fused_count)
    if rank == 0:
        out["configs"] = configs
        if sustained is not None:
            sustained["ratio_to_value"] = sustained["symbols_per_s"] * world / out["value"]  # per-GPU sustained rate x N vs the timed region
        if args.no_cpu_baseline:
            pass
        elif world == 1:  # rank 0's host cores, at N = 1 only: at N > 1 the other ranks would sit in the barrier below for ~30 s
            out["cpu_baseline"] = cpu_baseline([w.cpu().numpy() for w in weights], 3450002)
            out["cpu_baseline_torch_path"] = cpu_baseline_torch_path([w.cpu().numpy() for w in weights], 3450002)
        else:
            out["cpu_baseline"] = {"value": None, "unit": "symbols/s", "cores": None, "kind": "port",
                                   "sample": "timed at N=1 only (the other ranks would wait for it); see the N=1 line"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
