"""Teacher-forced replay of the two BASELINE flows with online training (pytest -m gpu): every block of the HIP run is
re-executed on stock PyTorch autograd from the HIP run's own state.

test_gpu_parity.py compares a free-running HIP evaluation with a free-running torch-autograd one; the two diverge as soon as
one block's coded ser lands on the other side of ser_thresh, so that comparison is statistical after the first few dozen
blocks.  Here the HIP run (harness.eval_by_word with the training kernels) records, per block, what it decided (buffer push,
meta-learning step indices, minibatch draws) and its state afterwards (weights, saved weights, both Adam moments, step
count); the torch side is put into the HIP state of block k - 1, performs block k's updates with the recorded decisions and
draws -- meta.meta_train_loop (second-order autograd, pinned to the reference by golden G11) and
OnlineTrainer._online_training_autograd (run_train_loop + torch-style Adam, pinned by G10) -- and must land on the HIP state
of block k within the per-25-iterations tolerance of the kernel tests, |dw| <= 2e-5 + 1e-3 |w|, scaled by the block's
iteration count.  All 300 blocks, at the reference's default counts; the error cannot compound (re-synchronised per block)."""
import numpy as np
import pytest
import torch

import meta_viterbinet_amd as mvn
from meta_viterbinet_amd.trials import TrialDraws

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _vnet_with(w, T, dev):
    det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, a in zip(det.parameters(), w):
            p.copy_(torch.as_tensor(a))
    return det


def _words(dev, coefficients, snr, seed, N=300, K=120, nsym=2, L=4):
    gen = torch.Generator(device=dev).manual_seed(seed)
    msg = torch.randint(0, 2, (N, K), generator=gen, device=dev).float()
    cw = mvn.rs_encode(msg, nsym)
    if coefficients == "cost2100":
        h = np.concatenate([mvn.estimate_channel(L, 0.2, "cost2100", index=i) for i in range(N)])
    else:
        h = np.concatenate([mvn.estimate_channel(L, 0.2, "time_decay", fading=True, index=i, fading_taps_type=2) for i in range(N)])
    return msg, mvn.transmit(cw, h, snr, L, torch.randn(N, K + 8 * nsym, generator=gen, device=dev))


def _state(det, saved, tr):
    return dict(w=[p.detach().clone() for p in det.parameters()],
                saved=None if saved is None else [p.detach().clone() for p in saved.parameters()],
                m=tr.exp_avg.clone(), v=tr.exp_avg_sq.clone(), step=tr.step)


def _load(det, saved, tr, st):
    with torch.no_grad():
        for p, a in zip(det.parameters(), st["w"]):
            p.copy_(a)
        if saved is not None:
            for p, a in zip(saved.parameters(), st["saved"] if st["saved"] is not None else st["w"]):
                p.copy_(a)
        tr.exp_avg.copy_(st["m"])
        tr.exp_avg_sq.copy_(st["v"])
    tr.step = st["step"]


def _ratio(det, tr, ref_w, ref_m, ref_v, scale, moments=True):
    """Largest deviation of (weights, exp_avg, exp_avg_sq) from the reference state, in units of the kernel tests' tolerance
    for `scale` x 25 iterations: weights 2e-5 + 1e-3 |w| (test_online_training_golden), moments 1e-6 + 1e-3 |m|."""
    worst = 0.0
    for a, b in zip(det.parameters(), ref_w):
        worst = max(worst, float(((a.detach() - b).abs() / (scale * (2e-5 + 1e-3 * b.abs()))).max()))
    if moments:
        for a, b in ((tr.exp_avg, ref_m), (tr.exp_avg_sq, ref_v)):
            worst = max(worst, float(((a - b).abs() / (scale * (1e-6 + 1e-3 * b.abs()))).max()))
    return worst


SHARP = 25  # iterations over which the two implementations must agree to the kernel tests' tolerance, in EVERY segment


def _replay(dev, w0, msg, rx, coefficients_kw, iterations, meta_lr=0.1, MAML=True):
    """Runs the HIP flow, then replays every update segment (a meta-learning update, a block's online training) on torch
    autograd FROM THE HIP STATE the segment started in.  Two checks per segment:
      sharp -- the first 25 iterations / meta-learning steps, HIP kernel (re-executed on a scratch trainer from the same
               state: the kernels are deterministic) against torch: within the kernel tests' tolerance, on the real states
               of the flow (carried Adam moments, restored weights, grown buffers);
      whole -- all of the segment's iterations against the WEIGHTS the flow recorded after it, in units of the tolerance of
               the segment's iteration count, (n / 25) x (2e-5 + 1e-3 |w|).  Two fp32 implementations of ~200 chained Adam
               steps through ReLUs are not always one trajectory: a unit whose pre-activation crosses zero one iteration apart
               changes the gradient discretely (and exp_avg, a 10-iteration memory of the gradient, then differs by the size
               of a gradient, which is why the moments are compared in the sharp check only).  Measured over the 242
               segments of configs[4] (full-word iterations restarted from the saved weights, meta-learning updates with an
               inner SGD step of 0.1): 219 end within their tolerance (186 within a tenth of it), 19 between 1 x and 3 x,
               4 between 9 x and 67 x; all 278 minibatch segments of configs[2] end within 0.01 of theirs.  The caller
               asserts the fraction of segments within tolerance; the ratios are returned.
    Returns (segments replayed, worst sharp ratio, list of whole-segment ratios, ser_by_word, Adam steps)."""
    T = rx.shape[1]
    det = _vnet_with(w0, T, dev)
    tr = mvn.OnlineTrainer(det, 4)
    log = []

    def observer(seen):
        log.append(dict(stage=seen["stage"], count=seen["count"], n_buf=seen["buffer_rx"].shape[0], meta=seen["meta"],
                        trained=seen["trained"], batch_idx=seen["batch_idx"], buffers=(seen["buffer_rx"], seen["buffer_tx"]),
                        state=_state(seen["detector"], seen["saved_detector"], tr)))

    ser = mvn.eval_by_word(det, msg, rx, 10.0, 0.2, 2, 25, online_trainer=tr, self_supervised_iterations=iterations,
                           meta_detector=mvn.META_VNETDetector(16, {"train": T, "val": T}), draws=TrialDraws(int(__import__("os").environ.get("MVN_REPLAY_SEED", "17")), dev),
                           observer=observer, meta_lr=meta_lr, MAML=MAML, **coefficients_kw)
    meta_style = coefficients_kw.get("meta_style_online_training", False)
    det_t, saved_t = _vnet_with(w0, T, dev), _vnet_with(w0, T, dev)   # torch side
    tr_t = mvn.OnlineTrainer(det_t, 4, use_kernel=False)
    det_k = _vnet_with(w0, T, dev)                                     # scratch HIP side (sharp check)
    tr_k = mvn.OnlineTrainer(det_k, 4)
    meta_det = mvn.META_VNETDetector(16, {"train": T, "val": T})
    prev = dict(w=[torch.as_tensor(a, device=dev) for a in w0], saved=[torch.as_tensor(a, device=dev) for a in w0],
                m=torch.zeros_like(tr.exp_avg), v=torch.zeros_like(tr.exp_avg), step=0)
    sharp, whole, segments, crossings = 0.0, [], 0, []

    def torch_meta(sup, qry):
        mvn.copy_model(source_model=saved_t, dest_model=det_t)  # trainer.py:331-343 with weights_init = 'last_frame'
        for k in range(qry.shape[0]):
            mvn.meta_train_loop(det_t, meta_det, tr_t, brx, btx, sup[k], qry[k:k + 1], meta_lr, MAML)

    def torch_online(n, batch_idx):
        if meta_style:
            mvn.copy_model(source_model=saved_t, dest_model=det_t)  # metavnet_trainer.py:59
        tr_t._online_training_autograd(btx[-1].reshape(1, -1), brx[-1].reshape(1, -1), n, None if batch_idx is None else batch_idx[:n],
                                       meta_style, False)

    for rec in log:
        st = rec["state"]
        brx, btx = rec["buffers"]
        brx, btx = brx[:rec["n_buf"]], btx[:rec["n_buf"]]
        is_meta = rec["stage"] == "meta"
        if not is_meta and not rec["trained"]:
            assert st["step"] == prev["step"] and all(torch.equal(a, b) for a, b in zip(st["w"], prev["w"]))  # nothing ran
            prev = st
            continue
        segments += 1
        n_all = int(rec["meta"][1].shape[0]) if is_meta else iterations
        n_sharp = min(SHARP, n_all)
        # ---- sharp: the segment's first iterations, HIP (scratch) against torch, both from the flow's state
        sup, qry = rec["meta"] if is_meta else (None, None)

        def first_iterations(n):
            _load(det_k, None, tr_k, prev)
            _load(det_t, saved_t, tr_t, prev)
            if is_meta:
                with torch.no_grad():
                    for p, a in zip(det_k.parameters(), prev["saved"]):
                        p.copy_(a)
                tr_k.maml_training(brx, btx, sup[:n], qry[:n], meta_lr, MAML)
                torch_meta(sup[:n], qry[:n])
            else:
                if meta_style:
                    with torch.no_grad():
                        for p, a in zip(det_k.parameters(), prev["saved"]):
                            p.copy_(a)
                bi = rec["batch_idx"]
                tr_k.online_training(btx[-1].reshape(1, -1), brx[-1].reshape(1, -1), iterations=n,
                                     batch_idx=None if bi is None else bi[:n], full_word=meta_style)
                torch_online(n, bi)
            return _ratio(det_t, tr_t, [p.detach() for p in det_k.parameters()], tr_k.exp_avg, tr_k.exp_avg_sq, 1)

        r = first_iterations(n_sharp)
        if r > 1.0:
            # The one way two correct fp32 implementations part inside 25 iterations: a hidden-2 unit's pre-activation crosses
            # zero for some sample an iteration apart; the unit's gradient row then differs by a gradient's worth, exp_avg carries
            # that for ~10 iterations and the weights follow (tools/_dbg_replay_seg.py: deviations <= 0.02 through iteration 20,
            # then 4 x on exp_avg of one unit's bias at 25).  Its signature is the suddenness -- an arithmetic error grows from the
            # first iteration on --, so such a segment must AGREE WELL (a quarter of the tolerance, moments included) over a
            # shorter window, at the very least over its first iteration; counted, and rare.
            agreed = next((n for n in (20, 15, 10, 5, 2, 1) if n < n_sharp and first_iterations(n) <= 0.25), 0)
            assert agreed >= 1, f"block {rec['count']} ({rec['stage']}): first {n_sharp} iterations deviate {r:.2f} x the tolerance, " \
                                f"and no shorter window agrees"
            crossings.append((rec["count"], rec["stage"], round(r, 2), agreed))
            r = 0.25
        sharp = max(sharp, r)
        # ---- whole: every iteration of the segment against the state the flow recorded
        _load(det_t, saved_t, tr_t, prev)
        if is_meta:
            torch_meta(sup, qry)
        else:
            torch_online(n_all, rec["batch_idx"])
        assert tr_t.step == st["step"], rec["count"]
        r = _ratio(det_t, tr_t, st["w"], st["m"], st["v"], -(-n_all // 25), moments=False)
        whole.append(r)
        if __import__("os").environ.get("MVN_REPLAY_VERBOSE"):
            print(f"  block {rec['count']:3d} {rec['stage']:4s} n {n_all:3d}  whole-segment weights deviation / tolerance {r:7.2f}")
        prev = st
    assert len(crossings) <= max(2, segments // 50), crossings  # (one in 238 segments of configs[4] as measured)
    if crossings:
        print(f"  ReLU crossings inside the sharp window (block, stage, deviation / tolerance at 25, iterations in agreement): {crossings}")
    return segments, sharp, whole, ser, tr.step


@pytest.mark.timeout(1500)
def test_config2_self_supervised_replayed_block_by_block(golden, dev):
    """BASELINE configs[2] with updates: ViterbiNet over the COST2100 taps, 300 blocks, 200 CE + Adam minibatch iterations
    (online_train_kernel) after every qualifying block, each block replayed on torch autograd from the HIP state."""
    g7 = golden("g7_by_word")
    msg, rx = _words(dev, "cost2100", 10.0, 5)
    segs, sharp, whole, ser, steps = _replay(dev, [g7[f"w{i}"] for i in range(6)], msg, rx, dict(self_supervised=True), 200)
    whole = np.asarray(whole)
    print(f"configs[2]: {segs} of 300 blocks trained ({steps} Adam steps); deviation / tolerance: first 25 iterations {sharp:.3f}, "
          f"whole blocks: worst {whole.max():.3f}, within tolerance {np.mean(whole <= 1.0):.3f}; mean ser {ser.mean():.5f}")
    assert segs >= 150 and steps == 200 * segs
    # 32-sample minibatch iterations: with the default draws every block's 200 iterations end within the tolerance (worst 0.011);
    # other draw seeds (MVN_REPLAY_SEED) meet a ReLU crossing in a block or two of the ~270 (seed 5: one block at 2.7 x)
    assert np.mean(whole <= 1.0) >= 0.98 and np.median(whole) <= 0.05


@pytest.mark.timeout(1500)
def test_config4_meta_viterbinet_replayed_block_by_block(golden, dev):
    """BASELINE configs[4] at the reference's defaults (200 full-word iterations per block from the saved weights, every 5
    blocks 20 x <= 10 second-order meta-learning steps): maml_train_groups_kernel and online_train_groups_kernel replayed per
    block on meta.meta_train_loop (torch double backward) and run_train_loop."""
    g7 = golden("g7_by_word")
    msg, rx = _words(dev, "time_decay", 10.0, 9)
    kw = dict(self_supervised=True, online_meta=True, meta_train_iterations=20, meta_j_num=10, meta_subframes=5,
              meta_style_online_training=True)
    segs, sharp, whole, ser, steps = _replay(dev, [g7[f"w{i}"] for i in range(6)], msg, rx, kw, 200)
    whole = np.asarray(whole)
    print(f"configs[4]: {segs} update segments ({steps} Adam steps); deviation / tolerance: first 25 iterations {sharp:.3f}, "
          f"whole segments: median {np.median(whole):.4f}, within tolerance {np.mean(whole <= 1.0):.3f}, worst {whole.max():.1f}; "
          f"mean ser {ser.mean():.5f}")
    assert segs >= 200 and steps > 200 * 150
    assert np.mean(whole <= 1.0) >= 0.85 and np.median(whole) <= 0.1  # (measured 0.905 / 0.003; see _replay on the outliers)
