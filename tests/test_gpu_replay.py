"""Teacher-forced replay of the two BASELINE flows with online training (pytest -m gpu): every UPDATE of the HIP run -- each
of a block's 200 training iterations, each step of a meta-learning update -- is re-executed on stock PyTorch autograd from the
HIP run's own state, and a float64 referee decides the cases where two fp32 implementations may legitimately part.

test_gpu_parity.py compares a free-running HIP evaluation with a free-running torch-autograd one; the two diverge as soon as
one block's coded ser lands on the other side of ser_thresh, so that comparison is statistical after the first few dozen
blocks.  Here the HIP run (harness.eval_by_word with the training kernels) records, per block, what it decided (buffer push,
meta-learning step indices, minibatch draws) and its state afterwards (weights, saved weights, both Adam moments, step
count).  Every recorded segment (a block's online training, a meta-learning update) is then walked ONE ITERATION AT A TIME:

  * the HIP kernel (n = 1, a scratch trainer put into the segment's starting state; the kernels are deterministic and take
    any n) and torch (OnlineTrainer._online_training_autograd = run_train_loop + Adam, pinned to the reference by golden
    G10; meta.meta_train_loop = second-order autograd, pinned by G11) perform the SAME iteration FROM THE SAME STATE, and the
    update must agree within the per-iteration tolerance below -- weights AND both Adam moments;
  * the walk then advances on the HIP state, so nothing compounds: an error that appears at iteration k of a segment is seen
    at iteration k, at its own size;
  * after the segment's last iteration the chain of n = 1 launches must have reproduced the state the 200-iteration launch of
    the flow left behind BIT FOR BIT (weights, moments): the iteration loop of the kernels (sample prefetch, first-chunk
    flags, bias-correction powers, the trial's second weight copy) carries state exactly like 200 separate launches.

The one way two correct fp32 implementations of an iteration differ by more than rounding: a hidden-2 pre-activation z2 lies
within rounding of 0, so that ReLU's derivative is 1 in one summation order and 0 in the other; that sample's whole
contribution to one unit's gradient row is then in or out.  An iteration where HIP and torch differ by more than the tolerance
is not excused, it is DECIDED by a float64 restatement of the iteration (explicit ReLU masks), whose rule (`_explained`) is
    |HIP - f64| <= max(2 |torch32 - f64| + the tolerance, ROUNDING_BAND tolerances),   with torch32 itself within NOISE_CAP of f64
(one iteration's fp32 rounding can exceed the tolerance when a gradient is a small sum of large per-sample terms -- Adam divides by
the element's own sqrt(v), so an element with a history of tiny gradients turns rounding-sized gradient differences into
tolerance-sized steps; torch's own distance from float64 is the yardstick for that, and it is bounded.  Either fp32 implementation
is seen up to 1.8 tolerances from the float64 update in such an iteration while the other sits at 0.1: torch32 with the default
draws, HIP at 1.35 with MVN_REPLAY_SEED=19, block 295 -- hence the band, the same for both; the number of such iterations is
bounded by the tests, <= 1 %).  If the natural float64 update does not explain both, it
must show entries with |z2| < 1e-5 among the iteration's samples, and the rule must hold against the float64 update for SOME
assignment of ReLU's derivative at those entries.  A deviation without such an entry, or one that no assignment explains,
fails.  On every 8th iteration the referee also runs unprovoked, same rule.

Per-iteration tolerance (both implementations start from identical fp32 state, so only one iteration's rounding is in it),
with g = the gradient implied by the first moment, (m' - beta1 m) / (1 - beta1), and G = max |g| over the parameter tensor:
    exp_avg:     |dm| <= (1 - beta1) TOL_G G          exp_avg_sq:  |dv| <= 2 (1 - beta2) TOL_G G^2 + 2e-6 |v|
    weights:     |dw| <= TOL_W (1 + |w|)
TOL_G = 2e-4 and TOL_W = 2e-6 (25 iterations of the kernel tests' 2e-5 + 1e-3 |w| would allow 10-500 x more per iteration);
measured on an MI355X (gpurun_out/r04_t*.txt): minibatch iterations stay below 0.07 of them; full-word iterations late in a
block's 200 (converged: small sums of large terms) reach 1.0-1.5 in ~0.05 % of the iterations, HIP and torch alike against
float64; a ReLU crossing (3 in 23 402 iterations) is 80-530 x the tolerance."""
import itertools
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import meta_viterbinet_amd as mvn
from meta_viterbinet_amd.trials import TrialDraws

pytestmark = pytest.mark.gpu

TOL_G, TOL_W = 2e-4, 2e-6
Z2_NEAR_ZERO = 1e-5   # |z2| below which fp32 summation order can decide ReLU's derivative (z2 = sum of 100 terms of size <~ 1)
MAX_FLIPS = 6         # entries the referee will enumerate (2^k float64 updates)
NOISE_CAP = 4.0       # torch's fp32 update further than this many tolerances from float64 is not rounding any more
ROUNDING_BAND = 2.0   # tolerances either fp32 implementation may be from the float64 update by one iteration's rounding alone
LR, BETAS, EPS = 1e-3, (0.9, 0.999), 1e-8


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


# ------------------------------------------------------------------------------------------------------------------
# states: dict(w = six tensors, m, v = flat moments, step); "saved" only in the flow's records
# ------------------------------------------------------------------------------------------------------------------
def _state(det, saved, tr):
    return dict(w=[p.detach().clone() for p in det.parameters()],
                saved=None if saved is None else [p.detach().clone() for p in saved.parameters()],
                m=tr.exp_avg.clone(), v=tr.exp_avg_sq.clone(), step=tr.step)


def _load(det, tr, w, m, v, step):
    with torch.no_grad():
        for p, a in zip(det.parameters(), w):
            p.copy_(a)
        tr.exp_avg.copy_(m)
        tr.exp_avg_sq.copy_(v)
    tr.step = step


def _flat(ws):
    return torch.cat([a.detach().reshape(-1) for a in ws])


class Deviation:
    """Largest deviation of an update (w', m', v') from a reference update, both taken from the same (w, m, v), in units of the
    per-iteration tolerance of the module docstring; `sizes` = the six parameter tensors' lengths."""

    def __init__(self, sizes, dev):
        self.seg = torch.repeat_interleave(torch.arange(6, device=dev), torch.as_tensor(sizes, device=dev))

    def tensor_max(self, x):  # max |x| per parameter tensor, broadcast back to the elements
        mx = torch.zeros(6, dtype=x.dtype, device=x.device).scatter_reduce(0, self.seg, x.abs(), "amax")
        return mx[self.seg]

    def __call__(self, m0, got, ref):
        """got / ref: (w' flat, m', v'), any float dtype; m0: the first moment both started from.  Returns (ratio, G)."""
        w, m, v = (t.double() for t in got)
        wr, mr, vr = (t.double() for t in ref)
        g_ref = (mr - BETAS[0] * m0.double()) / (1.0 - BETAS[0])
        G = self.tensor_max(g_ref)
        r_m = ((m - mr).abs() / ((1.0 - BETAS[0]) * TOL_G * G + 1e-30)).max()
        r_v = ((v - vr).abs() / (2.0 * (1.0 - BETAS[1]) * TOL_G * G * G + 2e-6 * vr.abs() + 1e-30)).max()
        r_w = ((w - wr).abs() / (TOL_W * (1.0 + wr.abs()))).max()
        return float(torch.stack([r_m, r_v, r_w]).max()), (float(r_m), float(r_v), float(r_w))


# ------------------------------------------------------------------------------------------------------------------
# the float64 referee: one update with explicit ReLU masks
# ------------------------------------------------------------------------------------------------------------------
def _fwd64(params, y, mask):
    """net(y) in float64 (vnet_detector.py:27-33) with ReLU written as z2 * mask; mask None = (z2 > 0).  Returns logits, z2."""
    W1, b1, W2, b2, W3, b3 = params
    h1 = torch.sigmoid(y.reshape(-1, 1) * W1.reshape(1, -1) + b1)
    z2 = h1 @ W2.t() + b2
    mk = (z2 > 0).double() if mask is None else mask
    return (z2 * mk) @ W3.t() + b3, z2


def _adam64(w, m, v, grads, step):
    """torch.optim.Adam's update (amsgrad off, no weight decay) in float64: the `step`-th step; returns flat (w', m', v')."""
    g = _flat(grads)
    m1 = BETAS[0] * m + (1.0 - BETAS[0]) * g
    v1 = BETAS[1] * v + (1.0 - BETAS[1]) * g * g
    bc1, bc2 = 1.0 - BETAS[0] ** step, 1.0 - BETAS[1] ** step
    return w - (LR / bc1) * m1 / (v1.sqrt() / bc2 ** 0.5 + EPS), m1, v1


def _online64(st, y, labels, idx, mask=None):
    """One CE + Adam iteration (trainer.py:492-505) on the samples idx of the word (None: all of it), float64."""
    params = [a.double().requires_grad_() for a in st["w"]]
    ys, ls = (y, labels) if idx is None else (y[idx], labels[idx])
    logits, z2 = _fwd64(params, ys.double(), mask)
    grads = torch.autograd.grad(F.cross_entropy(logits, ls), params)
    return _adam64(_flat(params).detach(), st["m"].double(), st["v"].double(), grads, st["step"] + 1), [z2.detach()]


def _maml64(st, ys, ls, yq, lq, meta_lr, MAML, masks=None):
    """One meta-learning step (trainer.py:425-453 = meta.meta_train_loop) in float64; masks = (support, query) or None."""
    params = [a.double().requires_grad_() for a in st["w"]]
    logits_s, z2s = _fwd64(params, ys.double(), None if masks is None else masks[0])
    local = torch.autograd.grad(F.cross_entropy(logits_s, ls), params, create_graph=MAML)
    updated = [p - meta_lr * g for p, g in zip(params, local)]
    logits_q, z2q = _fwd64(updated, yq.double(), None if masks is None else masks[1])
    grads = torch.autograd.grad(F.cross_entropy(logits_q, lq), params)
    return _adam64(_flat(params).detach(), st["m"].double(), st["v"].double(), grads, st["step"] + 1), [z2s.detach(), z2q.detach()]


def _decide(run64, z2s, m0, hip, t32, dv):
    """An iteration the natural float64 update does not explain: both fp32 updates must be float64 updates for SOME assignment
    of ReLU's derivative at the near-zero pre-activations.  Returns (number of such entries, HIP's and torch's deviations from
    their best assignments, in units of the tolerance)."""
    near = [(k, tuple(int(i) for i in ij)) for k, z in enumerate(z2s) for ij in (z.abs() < Z2_NEAR_ZERO).nonzero().tolist()]
    if not 1 <= len(near) <= MAX_FLIPS:
        raise AssertionError(f"{len(near)} pre-activations within {Z2_NEAR_ZERO} of zero (smallest |z2| {min(float(z.abs().min()) for z in z2s):.3g})")
    best = [np.inf, np.inf]
    for bits in itertools.product((0.0, 1.0), repeat=len(near)):
        masks = [(z > 0).double() for z in z2s]
        for (k, ij), b in zip(near, bits):
            masks[k][ij] = b
        ref, _ = run64(masks)
        for j, got in enumerate((hip, t32)):
            best[j] = min(best[j], dv(m0, got, ref)[0])
    return len(near), best


def _explained(r_hip, r_t32):
    """The referee's rule, in units of the per-iteration tolerance: HIP may be as far from the float64 update as the tolerance
    allows, or -- where one iteration's fp32 rounding exceeds it (gradients that are small sums of large per-sample terms) --
    twice as far as torch's own fp32 arithmetic is, or inside the band either implementation is seen in (ROUNDING_BAND); torch
    itself must stay within NOISE_CAP tolerances of float64, else the yardstick is not rounding and the assignment search has to
    explain both."""
    return r_t32 <= NOISE_CAP and r_hip <= max(2.0 * r_t32 + 1.0, ROUNDING_BAND)


# ------------------------------------------------------------------------------------------------------------------
def _replay(dev, w0, msg, rx, flow_kw, iterations, meta_lr=0.1, MAML=True, stride=1):
    """Runs the HIP flow, then walks every `stride`-th update segment iteration by iteration (module docstring).
    Returns a dict of counts and worst ratios, and the flow's ser_by_word."""
    T = rx.shape[1]
    det = _vnet_with(w0, T, dev)
    tr = mvn.OnlineTrainer(det, 4)
    log = []

    def observer(seen):
        log.append(dict(stage=seen["stage"], count=seen["count"], n_buf=seen["buffer_rx"].shape[0], meta=seen["meta"],
                        trained=seen["trained"], batch_idx=seen["batch_idx"], buffers=(seen["buffer_rx"], seen["buffer_tx"]),
                        state=_state(seen["detector"], seen["saved_detector"], tr)))

    ser = mvn.eval_by_word(det, msg, rx, 10.0, 0.2, 2, 25, online_trainer=tr, self_supervised_iterations=iterations,
                           meta_detector=mvn.META_VNETDetector(16, {"train": T, "val": T}),
                           draws=TrialDraws(int(os.environ.get("MVN_REPLAY_SEED", "17")), dev),
                           observer=observer, meta_lr=meta_lr, MAML=MAML, **flow_kw)
    meta_style = flow_kw.get("meta_style_online_training", False)
    det_t = _vnet_with(w0, T, dev)   # torch side
    tr_t = mvn.OnlineTrainer(det_t, 4, use_kernel=False)
    det_k = _vnet_with(w0, T, dev)   # scratch HIP side
    tr_k = mvn.OnlineTrainer(det_k, 4)
    meta_det = mvn.META_VNETDetector(16, {"train": T, "val": T})
    dv = Deviation([p.numel() for p in det.parameters()], dev)
    prev = dict(w=[torch.as_tensor(a, device=dev) for a in w0], saved=[torch.as_tensor(a, device=dev) for a in w0],
                m=torch.zeros_like(tr.exp_avg), v=torch.zeros_like(tr.exp_avg), step=0)
    out = dict(segments=0, walked=0, iterations=0, worst=0.0, worst_parts=(0.0, 0.0, 0.0), crossings=[], noisy=[], referee_runs=0,
               referee_worst=0.0, chain_inexact=0)
    verbose = bool(os.environ.get("MVN_REPLAY_VERBOSE"))

    for rec in log:
        st = rec["state"]
        is_meta = rec["stage"] == "meta"
        if not is_meta and not rec["trained"]:
            assert st["step"] == prev["step"] and all(torch.equal(a, b) for a, b in zip(st["w"], prev["w"]))  # nothing ran
            prev = st
            continue
        out["segments"] += 1
        if (out["segments"] - 1) % stride:
            prev = st
            continue
        out["walked"] += 1
        brx, btx = (b[:rec["n_buf"]] for b in rec["buffers"])
        labels = mvn.calculate_states(4, btx).reshape(btx.shape[0], T).long()
        sup, qry = rec["meta"] if is_meta else (None, None)
        n_all = int(qry.shape[0]) if is_meta else iterations
        # where the segment starts: a meta-learning update and the Meta-ViterbiNet online training restart from the saved
        # weights (trainer.py:331-343 with weights_init = 'last_frame'; metavnet_trainer.py:59), moments and step carried over
        start_w = prev["saved"] if (is_meta or meta_style) else prev["w"]
        _load(det_k, tr_k, start_w, prev["m"], prev["v"], prev["step"])
        seg_worst = 0.0
        for it in range(n_all):
            cur = dict(w=[p.detach().clone() for p in det_k.parameters()], m=tr_k.exp_avg.clone(), v=tr_k.exp_avg_sq.clone(), step=tr_k.step)
            _load(det_t, tr_t, cur["w"], cur["m"], cur["v"], cur["step"])
            if is_meta:
                s_i, q_i = torch.remainder(sup[it], brx.shape[0]), torch.remainder(qry[it:it + 1], brx.shape[0])
                tr_k.maml_training(brx, btx, sup[it:it + 1], qry[it:it + 1], meta_lr, MAML)
                mvn.meta_train_loop(det_t, meta_det, tr_t, brx, btx, s_i, q_i, meta_lr, MAML)
                ys, ls, yq, lq = brx[s_i].reshape(-1), labels[s_i].reshape(-1), brx[q_i].reshape(-1), labels[q_i].reshape(-1)
                run64 = lambda masks=None: _maml64(cur, ys, ls, yq, lq, meta_lr, MAML, masks)  # noqa: E731
            else:
                bi = rec["batch_idx"]
                b_it = None if bi is None else bi[it:it + 1]
                tr_k.online_training(btx[-1].reshape(1, -1), brx[-1].reshape(1, -1), iterations=1, batch_idx=b_it, full_word=meta_style)
                tr_t._online_training_autograd(btx[-1].reshape(1, -1), brx[-1].reshape(1, -1), 1, b_it, meta_style, False)
                idx = None if b_it is None else b_it[0].long()
                run64 = lambda masks=None: _online64(cur, brx[-1], labels[-1], idx, None if masks is None else masks[0])  # noqa: E731
            hip = (_flat(det_k.parameters()), tr_k.exp_avg, tr_k.exp_avg_sq)
            t32 = (_flat(det_t.parameters()), tr_t.exp_avg, tr_t.exp_avg_sq)
            assert tr_k.step == tr_t.step == cur["step"] + 1
            r, parts = dv(cur["m"], hip, t32)
            out["iterations"] += 1
            if r <= 1.0:
                if r > out["worst"]:
                    out["worst"], out["worst_parts"] = r, parts
                seg_worst = max(seg_worst, r)
            if r > 1.0 or out["iterations"] % 8 == 0:  # the float64 referee: when HIP and torch disagree, and unprovoked
                ref, z2s = run64()
                r_h, r_t = dv(cur["m"], hip, ref)[0], dv(cur["m"], t32, ref)[0]
                where = f"block {rec['count']} ({rec['stage']}) iteration {it} (Adam step {cur['step'] + 1})"
                if r <= 1.0:
                    out["referee_runs"] += 1
                if _explained(r_h, r_t):
                    if r <= 1.0:
                        out["referee_worst"] = max(out["referee_worst"], r_h)
                    else:  # one iteration's rounding above the tolerance, in both fp32 implementations alike
                        out["noisy"].append((rec["count"], rec["stage"], it, round(r, 2), round(r_h, 2), round(r_t, 2)))
                else:
                    # not rounding: a ReLU derivative decided by summation order -- HIP against torch (r > 1), or both fp32
                    # implementations alike against float64 (seen unprovoked) -- or an error: the assignments decide
                    try:
                        n_near, best = _decide(run64, z2s, cur["m"], hip, t32, dv)
                    except AssertionError as e:
                        raise AssertionError(f"{where}: HIP and torch differ by {r:.2f} x the tolerance {parts}; against the float64 "
                                             f"update HIP is at {r_h:.2f} and torch at {r_t:.2f}; {e}") from None
                    assert _explained(*best), (f"{where}: HIP and torch differ by {r:.1f} x the tolerance {parts}; {n_near} near-zero "
                                               f"pre-activation(s), but the best float64 assignments leave HIP at {best[0]:.2f} and torch "
                                               f"at {best[1]:.2f} of the tolerance")
                    out["crossings"].append((rec["count"], rec["stage"], it, round(max(r, r_h), 1), n_near))
        # the chain of n = 1 launches against the state the flow's own launch (all n_all iterations at once) left behind
        chain_w = _flat(det_k.parameters())
        exact = (torch.equal(chain_w, _flat(st["saved"] if is_meta else st["w"])) and torch.equal(tr_k.exp_avg, st["m"])
                 and torch.equal(tr_k.exp_avg_sq, st["v"]) and tr_k.step == st["step"])
        if not exact:  # (beta^step as pow() per launch against the kernel's running product: an ulp of a double, once in ~1e7 steps)
            out["chain_inexact"] += 1
            ref_w = _flat(st["saved"] if is_meta else st["w"])
            assert tr_k.step == st["step"] and bool(((chain_w - ref_w).abs() <= TOL_W * (1.0 + ref_w.abs())).all())
            assert torch.allclose(tr_k.exp_avg, st["m"], rtol=1e-4, atol=1e-9) and torch.allclose(tr_k.exp_avg_sq, st["v"], rtol=1e-4, atol=1e-12)
        if verbose:
            print(f"  block {rec['count']:3d} {rec['stage']:4s} n {n_all:3d}  worst iteration {seg_worst:.3f} of the tolerance, chain "
                  f"{'exact' if exact else 'INEXACT'}")
        prev = st
    return out, ser, tr.step


def _report(name, out, ser, steps):
    print(f"{name}: {out['walked']} of {out['segments']} update segments walked, {out['iterations']} iterations (of {steps} Adam steps); "
          f"HIP vs torch per iteration: worst {out['worst']:.3f} of the tolerance (exp_avg {out['worst_parts'][0]:.3f}, exp_avg_sq "
          f"{out['worst_parts'][1]:.3f}, weights {out['worst_parts'][2]:.3f}) in the iterations within it; {len(out['noisy'])} iterations "
          f"above it by rounding alone (HIP within max(2 x torch's own distance from float64 + 1, 2) tolerances of it: (block, stage, iteration, HIP-torch, HIP-f64, "
          f"torch-f64) {out['noisy'][:8]}); {len(out['crossings'])} ReLU-crossing iterations decided by the float64 assignments "
          f"{out['crossings'][:12]}; referee unprovoked {out['referee_runs']} times, HIP vs float64 worst {out['referee_worst']:.3f}; "
          f"chains not bit-exact: {out['chain_inexact']}; mean ser {ser.mean():.5f}")


@pytest.mark.timeout(1700)
def test_config2_self_supervised_replayed_iteration_by_iteration(golden, dev):
    """BASELINE configs[2] with updates: ViterbiNet over the COST2100 taps, 300 blocks, 200 CE + Adam minibatch iterations
    (online_train_kernel) after every qualifying block; EVERY block's 200 iterations walked one by one (MVN_REPLAY_STRIDE=1)."""
    g7 = golden("g7_by_word")
    msg, rx = _words(dev, "cost2100", 10.0, 5)
    stride = int(os.environ.get("MVN_REPLAY_STRIDE", "1"))
    out, ser, steps = _replay(dev, [g7[f"w{i}"] for i in range(6)], msg, rx, dict(self_supervised=True), 200, stride=stride)
    _report("configs[2]", out, ser, steps)
    assert out["segments"] >= 150 and steps == 200 * out["segments"] and out["iterations"] == 200 * out["walked"]
    assert out["chain_inexact"] <= 1 and len(out["crossings"]) + len(out["noisy"]) <= out["iterations"] // 100


@pytest.mark.timeout(1700)
def test_config4_meta_viterbinet_replayed_iteration_by_iteration(golden, dev):
    """BASELINE configs[4] at the reference's defaults (200 full-word iterations per block from the saved weights, every 5
    blocks 20 x <= 10 second-order meta-learning steps): the full-word online training kernel and the meta-learning kernel
    walked step by step on run_train_loop and meta.meta_train_loop (torch double backward), EVERY segment (MVN_REPLAY_STRIDE=1)."""
    g7 = golden("g7_by_word")
    msg, rx = _words(dev, "time_decay", 10.0, 9)
    kw = dict(self_supervised=True, online_meta=True, meta_train_iterations=20, meta_j_num=10, meta_subframes=5,
              meta_style_online_training=True)
    stride = int(os.environ.get("MVN_REPLAY_STRIDE", "1"))
    out, ser, steps = _replay(dev, [g7[f"w{i}"] for i in range(6)], msg, rx, kw, 200, stride=stride)
    _report("configs[4]", out, ser, steps)
    assert out["segments"] >= 200 and steps > 200 * 150 and out["iterations"] > 150 * out["walked"]
    assert out["chain_inexact"] <= 1 and len(out["crossings"]) + len(out["noisy"]) <= out["iterations"] // 100
