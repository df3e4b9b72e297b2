"""The reference's own `eval_by_word` WITH its update branches, replayed draw for draw (goldens G12, G13).

tests/golden/make_golden.py:g12_by_word_with_updates ran the unmodified reference (trainer.py:267-354) over 75 blocks from the
reference-trained weights of G7, once as VNETTrainer (self-supervised minibatch training, vnet_trainer.py:49-60) and once as
METAVNETTrainer (online meta-learning every 5 blocks + whole-word training from the saved weights, metavnet_trainer.py:52-64),
recording every torch.multinomial / torch.randint draw it made.  Here the same words go through mvn.eval_by_word with a draws
object that hands those recorded values back in call order, on the HIP kernels and on the torch-autograd path of the same host
code.  What must come out: the reference's ser_by_word, block for block (the ser decides which blocks are buffered and trained
on, so one differing block would also misalign the recorded draws — the draws object raises when that happens), every recorded
draw consumed, and the final (and, in the meta flow, saved) weights within 5e-5 of the reference's.
G13 (g13_by_word_switches) holds one such run per remaining switch of the reference's evaluation (FLOWS below)."""
import numpy as np
import pytest
import torch

import meta_viterbinet_amd as mvn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


class RecordedDraws:
    """trials.TrialDraws' interface over the draws the reference made (in its call order)."""

    lenient = False  # True: past the end of the recording (a run that has parted from the reference) the last draws repeat

    def __init__(self, multinomial, randint_high, randint, device, init_weights=None):
        self.multinomial, self.randint_high, self.randint, self.device = multinomial, randint_high, randint, device
        self.inits = np.zeros((0, 0), np.float32) if init_weights is None else init_weights
        self.m_at = self.r_at = self.i_at = 0

    def batches(self, count, n_blocks, T, iterations, M):
        if self.lenient and self.m_at + iterations > len(self.multinomial):
            self.m_at = len(self.multinomial) - iterations
        rows = self.multinomial[self.m_at:self.m_at + iterations]
        if rows.shape != (iterations, M):
            raise AssertionError(f"block {count} trains, the reference's recorded minibatches are used up ({self.m_at} of {len(self.multinomial)})")
        assert rows.max() < T
        self.m_at += iterations
        return torch.as_tensor(rows, dtype=torch.int32, device=self.device)

    def j_hat(self, high, size):
        return self.j_hat_update(high, 1, size)

    def j_hat_update(self, high, iterations, size):
        out = []
        for _ in range(iterations):
            if self.r_at >= len(self.randint):
                if not self.lenient:
                    raise AssertionError("a meta-learning update the reference did not make")
                self.r_at = len(self.randint) - 1
            if self.lenient and self.randint_high[self.r_at] != high:  # (the runs have parted: another buffer length)
                out.append(np.unique(self.randint[self.r_at] % high))
                self.r_at += 1
                continue
            assert self.randint_high[self.r_at] == high, (self.r_at, int(self.randint_high[self.r_at]), high)  # same buffer length
            assert self.randint.shape[1] == size
            out.append(np.unique(self.randint[self.r_at]))  # torch.unique(torch.randint(...)) (trainer.py:337)
            self.r_at += 1
        return np.concatenate(out)

    def init_weights(self, n_states):
        """the weights the reference's initialize_detector() produced at this re-initialisation (trainer.py:356-359)"""
        if self.i_at >= len(self.inits):
            raise AssertionError("a re-initialisation the reference did not make")
        flat, out, at = torch.as_tensor(self.inits[self.i_at]), [], 0
        self.i_at += 1
        for shape in ((100, 1), (100,), (50, 100), (50,), (n_states, 50), (n_states,)):
            n = int(np.prod(shape))
            out.append(flat[at:at + n].reshape(shape))
            at += n
        assert at == flat.numel()
        return out

    def used_up(self):
        return self.m_at == len(self.multinomial) and self.r_at == len(self.randint) and self.i_at == len(self.inits)


# the harness arguments of every recorded flow beyond its iteration counts (tests/golden/make_golden.py: g12 / g13)
FLOWS = {"selfsup": {}, "meta": dict(online_meta=True), "c2_selfsup": {}, "c4_meta": dict(online_meta=True),
         "L3_selfsup": {}, "L3_meta": dict(online_meta=True), "L5_selfsup": {}, "L5_meta": dict(online_meta=True), "L6_selfsup": {}, "L7_selfsup": {}, "fomaml": dict(online_meta=True, MAML=False),
         "window": dict(online_meta=True, window=True), "random": dict(online_meta=True, weights_init="random"),
         "metatrain": dict(online_meta=True, weights_init="meta_training"), "support2": dict(online_meta=True, window_size=2),
         "rmsprop": dict(optimizer_type="RMSprop"), "sgd": dict(optimizer_type="SGD", lr=0.05)}


def _flow_kwargs(g, g7, tag, dev):
    f = dict(FLOWS[tag])
    ss_it, meta_it, j_num, meta_sub, _, subframes, nsym = [int(v) for v in g[f"{tag}_meta"]]
    kw = dict(self_supervised=True, self_supervised_iterations=ss_it, ser_thresh=0.02)
    if f.pop("online_meta", False):
        kw.update(online_meta=True, meta_lr=0.1, MAML=f.pop("MAML", True), window_size=f.pop("window_size", 1),
                  meta_train_iterations=meta_it, meta_j_num=j_num, meta_subframes=meta_sub, meta_style_online_training=True)
        if "weights_init" in f:
            kw["weights_init"] = f.pop("weights_init")
            if kw["weights_init"] == "meta_training":  # the checkpoint the reference reloads = the weights the run started from
                kw["meta_training_weights"] = [g7[f"w{i}"] for i in range(6)]
        if f.pop("window", False):  # buffer_empty=False: the words the reference drew from its training channel
            kw["initial_buffer"] = (torch.tensor(g[f"{tag}_buffer_tx"], device=dev).float(), torch.tensor(g[f"{tag}_buffer_rx"], device=dev))
    return kw, f, nsym, subframes  # f: what is left is the optimizer's (optimizer_type, lr)


def _start_weights(g, g7, tag):
    """the weights the reference's run started from: G7's at 16 states, the run's own (w0) at other state counts"""
    return [g[f"{tag}_w0_{i}"] if f"{tag}_w0_{i}" in g.files else g7[f"w{i}"] for i in range(6)]


def _recorded(g, tag, dev):
    return RecordedDraws(g[f"{tag}_multinomial"], g[f"{tag}_randint_high"], g[f"{tag}_randint"], dev,
                         g[f"{tag}_init_weights"] if f"{tag}_init_weights" in g.files else None)


def _run(g, g7, tag, dev, hip, lenient=False):
    tx = torch.tensor(g[f"{tag}_tx"], device=dev).float()
    rx = torch.tensor(g[f"{tag}_rx"], device=dev)
    kw, opt, nsym, subframes = _flow_kwargs(g, g7, tag, dev)
    T = rx.shape[1]
    w0 = _start_weights(g, g7, tag)
    S = w0[5].shape[0]
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, a in zip(det.parameters(), w0):
            p.copy_(torch.as_tensor(a))
    tr = mvn.OnlineTrainer(det, int(np.log2(S)), use_kernel=hip, **opt)
    draws = _recorded(g, tag, dev)
    draws.lenient = lenient
    if kw.get("online_meta"):
        kw["meta_detector"] = mvn.META_VNETDetector(S, {"train": T, "val": T})
    last = {}

    def observer(seen):  # the saved weights (the reference's saved_detector, trainer.py:275/:343) as the last block leaves them
        if seen["stage"] == "end" and seen["saved_detector"] is not None:
            last["saved"] = [p.detach().cpu().numpy().copy() for p in seen["saved_detector"].parameters()]

    ser = mvn.eval_by_word(det, tx, rx, 9.0, 0.2, nsym, subframes, online_trainer=tr, draws=draws, hip_meta=hip, graphed_meta=False,
                           observer=observer, **kw)
    return ser, draws, [p.detach().cpu().numpy() for p in det.parameters()], last.get("saved")


@pytest.mark.parametrize("tag", ["selfsup", "meta"])
@pytest.mark.parametrize("hip", [True, False], ids=["hip_kernels", "torch_autograd"])
def test_reference_by_word_flow_with_updates(golden, dev, tag, hip):
    g, g7 = golden("g12_by_word_with_updates"), golden("g7_by_word")
    ser, draws, w, saved = _run(g, g7, tag, dev, hip)
    ref = g[f"{tag}_ser_by_word"]
    assert ser.shape == ref.shape == (75,)
    assert np.array_equal(ser, ref), (np.flatnonzero(ser != ref), ser[ser != ref], ref[ser != ref])
    assert draws.used_up(), (draws.m_at, len(draws.multinomial), draws.r_at, len(draws.randint))
    # the weights the reference ended with: 59 x 12 minibatch Adam steps / 14 meta-learning updates (28 steps of <= 4 second-order
    # meta-gradients) + 8 whole-word steps from the saved weights, its CPU kernels against these (different summation orders from
    # the first matmul on).  Measured: 1.6e-6 / 3.6e-7 on weights that moved 0.89 / 0.13; the bound leaves a factor of ~30.
    moved = max(float(np.abs(g[f"{tag}_w1_{i}"] - g7[f"w{i}"]).max()) for i in range(6))
    worst = max(float(np.abs(w[i] - g[f"{tag}_w1_{i}"]).max()) for i in range(6))
    msg = f"g12 {tag} {'hip' if hip else 'torch'}: ser identical on 75 blocks; weights moved {moved:.4f} from the start, end {worst:.2e} from the reference's"
    assert moved > 0.05 and worst <= 5e-5
    if tag == "meta":  # the weights saved by the last meta-learning update (block 70)
        moved_s = max(float(np.abs(g[f"meta_saved_{i}"] - g7[f"w{i}"]).max()) for i in range(6))
        worst_s = max(float(np.abs(saved[i] - g[f"meta_saved_{i}"]).max()) for i in range(6))
        msg += f"; saved weights moved {moved_s:.4f}, end {worst_s:.2e} from the reference's"
        assert moved_s > 0.05 and worst_s <= 5e-5
    print(msg)


G13 = ["fomaml", "window", "random", "metatrain", "support2", "rmsprop", "sgd"]


@pytest.mark.parametrize("tag", G13)
@pytest.mark.parametrize("hip", [True, False], ids=["hip_kernels", "torch_autograd"])
def test_reference_by_word_switches(golden, dev, tag, hip):
    """The reference's remaining eval_by_word switches, one recorded 50-block run each (golden G13): first-order meta-learning,
    the pre-filled fixed-length buffer, meta_weights_init 'random' / 'meta_training', two support words, RMSprop, SGD (the last
    two: in the online-training kernel since round 5, hip parametrisation; on torch.optim in the other)."""
    g, g7 = golden("g13_by_word_switches"), golden("g7_by_word")
    ser, draws, w, saved = _run(g, g7, tag, dev, hip)
    ref = g[f"{tag}_ser_by_word"]
    assert ser.shape == ref.shape == (50,)
    assert np.array_equal(ser, ref), (np.flatnonzero(ser != ref), ser[ser != ref], ref[ser != ref])
    assert draws.used_up(), (draws.m_at, len(draws.multinomial), draws.r_at, len(draws.randint), draws.i_at, len(draws.inits))
    moved = max(float(np.abs(g[f"{tag}_w1_{i}"] - g7[f"w{i}"]).max()) for i in range(6))
    worst = max(float(np.abs(w[i] - g[f"{tag}_w1_{i}"]).max()) for i in range(6))
    msg = f"g13 {tag} {'hip' if hip else 'torch'}: ser identical on 50 blocks; weights moved {moved:.4f}, end {worst:.2e} from the reference's"
    assert moved > 0.01 and worst <= 5e-5, msg
    if saved is not None:
        worst_s = max(float(np.abs(saved[i] - g[f"{tag}_saved_{i}"]).max()) for i in range(6))
        msg += f"; saved weights end {worst_s:.2e} from the reference's"
        assert worst_s <= 5e-5, msg
    print(msg)


G16 = ["L3_selfsup", "L3_meta", "L5_selfsup", "L5_meta"]
G17 = ["L6_selfsup", "L7_selfsup"]


@pytest.mark.parametrize("tag", G16 + G17)
@pytest.mark.parametrize("route", ["hip_kernels", "torch_autograd", "batched_trials"])
def test_reference_by_word_flow_other_state_counts(golden, dev, tag, route):
    """Golden G16: the self-supervised and the meta-learning flow at 8 and 32 states (channel memory 3 and 5), recorded runs of the
    unmodified reference from weights its own trainer produced: the run-time-n_states instantiations of the training kernels
    (online_train / maml_train _kernel<0, ...>, their chunked forms) and the block step as separate launches (the one-launch step
    serves 16 states).  Golden G17 (round 5): the self-supervised flow at 64 and 128 states (memory 6 and 7) on
    online_train_kernel<64 | 128>.  ser_by_word identical on all 50 blocks, final weights within 5e-5 of the reference's, on
    every route."""
    from meta_viterbinet_amd.trials import TrialBank, eval_by_word_batched

    g, g7 = golden("g17_by_word_64_128_states" if tag in G17 else "g16_by_word_other_state_counts"), golden("g7_by_word")
    ref, w0 = g[f"{tag}_ser_by_word"], _start_weights(g, g7, tag)
    S = w0[5].shape[0]
    if route == "batched_trials":
        R = 3
        kw, opt, nsym, subframes = _flow_kwargs(g, g7, tag, dev)
        tx = torch.tensor(g[f"{tag}_tx"], device=dev).float().unsqueeze(0).repeat(R, 1, 1)
        rx = torch.tensor(g[f"{tag}_rx"], device=dev).unsqueeze(0).repeat(R, 1, 1)
        bank = TrialBank([w0] * R, S, int(np.log2(S)), dev)
        draws = [RecordedTableDraws(g, tag, kw["self_supervised_iterations"], subframes, dev) for _ in range(R)]
        ser_all = eval_by_word_batched(bank, tx, rx, nsym, subframes, draws, **kw)
        ser, w = ser_all[0], [t.cpu().numpy() for t in bank.weights(0)]
        assert np.array_equal(ser_all[1], ser) and np.array_equal(ser_all[2], ser) and torch.equal(bank.theta[0], bank.theta[2])
        assert all(d.used_up() for d in draws)
    else:
        ser, draws, w, _ = _run(g, g7, tag, dev, route == "hip_kernels")
        assert draws.used_up(), (draws.m_at, len(draws.multinomial), draws.r_at, len(draws.randint))
    assert ser.shape == ref.shape == (50,)
    assert np.array_equal(ser, ref), (np.flatnonzero(ser != ref), ser[ser != ref], ref[ser != ref])
    moved = max(float(np.abs(g[f"{tag}_w1_{i}"] - w0[i]).max()) for i in range(6))
    worst = max(float(np.abs(w[i] - g[f"{tag}_w1_{i}"]).max()) for i in range(6))
    print(f"g16/17 {tag} ({S} states) {route}: ser identical on 50 blocks; weights moved {moved:.4f}, end {worst:.2e} from the reference's")
    assert moved > 0.005 and worst <= 5e-5


def _flat(params):
    return torch.cat([p.detach().reshape(-1) for p in params]).cpu().numpy()


def _set_flat(params, flat):
    at = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(torch.as_tensor(flat[at:at + n]).reshape(p.shape))
            at += n
    assert at == len(flat)


# |ours - reference's| allowed in the weights one block of configs[2]'s updates leaves (<= 200 minibatch CE + Adam steps from the
# SAME weights: different summation orders from the first matmul on); measured worst 6.7e-6 (hip) on steps that move the weights by
# up to 0.85
G15_BLOCK_TOL = 3e-5


def _resynchronised_run(g, g7, tag, dev, hip):
    """eval_by_word over the 50 recorded blocks of a G15 flow, re-synchronised on the reference after every block: returns
    (ser_by_word, draws, per-block stats).  After block k's updates the detector's (and the saved detector's) weights are compared
    with the state the reference held at the start of block k + 1 and then REPLACED by it, so every block is detected with, and
    every update starts from, the reference's weights; the optimizer's moments are compared where recorded (and replaced there)."""
    blk_w = g[f"{tag}_blk_w"]
    w_end = np.concatenate([g[f"{tag}_w1_{i}"].reshape(-1) for i in range(6)])
    saved_at, saved_w = g[f"{tag}_blk_saved_at"], g[f"{tag}_blk_saved"]
    adam = {int(a): (g[f"{tag}_blk_adam_m"][i], g[f"{tag}_blk_adam_v"][i], int(g[f"{tag}_blk_adam_step"][i]))
            for i, a in enumerate(g[f"{tag}_blk_adam_at"])}
    assert blk_w.shape[0] == 50 and np.array_equal(blk_w[0], np.concatenate([g7[f"w{i}"].reshape(-1) for i in range(6)]))
    tx = torch.tensor(g[f"{tag}_tx"], device=dev).float()
    rx = torch.tensor(g[f"{tag}_rx"], device=dev)
    kw, opt, nsym, subframes = _flow_kwargs(g, g7, tag, dev)
    T = rx.shape[1]
    det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
    _set_flat(list(det.parameters()), blk_w[0])
    tr = mvn.OnlineTrainer(det, 4, use_kernel=hip, **opt)
    draws = _recorded(g, tag, dev)
    if kw.get("online_meta"):
        kw["meta_detector"] = mvn.META_VNETDetector(16, {"train": T, "val": T})
    st = {"max": np.zeros(50), "median": np.zeros(50), "moved": np.zeros(50), "saved_max": np.zeros(50), "adam": []}

    def observer(seen):
        if seen["stage"] != "end":
            return
        k = seen["count"]
        nxt = blk_w[k + 1] if k + 1 < 50 else w_end  # the reference's detector at the start of block k + 1
        d = np.abs(_flat(seen["detector"].parameters()) - nxt)
        st["max"][k], st["median"][k], st["moved"][k] = d.max(), np.median(d), np.abs(nxt - blk_w[k]).max()
        _set_flat(list(seen["detector"].parameters()), nxt)
        if seen["saved_detector"] is not None and len(saved_at) and k + 1 < 50:
            j = int(np.searchsorted(saved_at, k + 1, side="right")) - 1  # the saved weights the reference held at block k + 1
            st["saved_max"][k] = np.abs(_flat(seen["saved_detector"].parameters()) - saved_w[j]).max()
            _set_flat(list(seen["saved_detector"].parameters()), saved_w[j])
        if k + 1 in adam:  # the optimizer's state at the start of block k + 1 (k + 1 = 50: the state the run ends with)
            m, v, step = adam[k + 1]
            assert tr.step == step, (k, tr.step, step)
            st["adam"].append((k, float(np.abs(tr.exp_avg.cpu().numpy() - m).max() / max(np.abs(m).max(), 1e-30)),
                               float(np.abs(tr.exp_avg_sq.cpu().numpy() - v).max() / max(np.abs(v).max(), 1e-30))))
            tr.exp_avg.copy_(torch.as_tensor(m))
            tr.exp_avg_sq.copy_(torch.as_tensor(v))

    ser = mvn.eval_by_word(det, tx, rx, 9.0, 0.2, nsym, subframes, online_trainer=tr, draws=draws, hip_meta=hip, graphed_meta=False,
                           observer=observer, **kw)
    return ser, draws, st


@pytest.mark.timeout(900)
@pytest.mark.parametrize("hip", [True, False], ids=["hip_kernels", "torch_autograd"])
def test_config2_reference_defaults_resynchronised_block_by_block(golden, dev, hip):
    """Golden G15 with the state the reference held at the START of every block (its detector's weights; at seven blocks the
    optimizer's exp_avg / exp_avg_sq / step): BASELINE configs[2] at the reference's own hyperparameters, all 50 blocks, 7 800 Adam
    steps, the replay RE-SYNCHRONISED on the reference after every block -- what a free run cannot offer once two fp32
    trajectories have parted at a word near the decision threshold:
      * every block is detected with the reference's weights -> ser_by_word is the reference's on ALL 50 blocks, bit for bit, and
        with it the buffer, the blocks that train and every recorded draw (the non-lenient draws object raises otherwise);
      * every block's 200 minibatch iterations start from the reference's weights -> what they leave is the reference's next state
        within G15_BLOCK_TOL, block by block, and both Adam moments agree with the reference's where they were recorded."""
    g, g7 = golden("g15_by_word_reference_defaults"), golden("g7_by_word")
    ser, draws, st = _resynchronised_run(g, g7, "c2_selfsup", dev, hip)
    ref = g["c2_selfsup_ser_by_word"]
    assert np.array_equal(ser, ref), (np.flatnonzero(ser != ref), ser[ser != ref], ref[ser != ref])
    assert draws.used_up(), (draws.m_at, len(draws.multinomial))
    trained = int((st["moved"] > 0).sum())
    print(f"g15 c2_selfsup {'hip' if hip else 'torch'} re-synchronised: ser identical on all 50 blocks; after each of the {trained} training "
          f"blocks' 200 iterations the weights are within {st['max'].max():.2e} of the reference's (a block moves them by up to "
          f"{st['moved'].max():.2f}); Adam moments (block, exp_avg, exp_avg_sq; relative to their largest entry): {st['adam']}")
    assert trained >= 35 and st["moved"].max() > 0.3 and len(st["adam"]) == 7
    assert st["max"].max() <= G15_BLOCK_TOL
    assert all(em <= 1e-3 and ev <= 1e-3 for _, em, ev in st["adam"])


@pytest.mark.timeout(900)
def test_config4_reference_defaults_resynchronised_block_by_block(golden, dev):
    """The same for BASELINE configs[4] (Meta-ViterbiNet at the reference's defaults: per block 200 WHOLE-WORD iterations from the
    saved weights, every 5 blocks 20 x <= 10 second-order meta-learning steps; 9 619 Adam steps).  Detection: ser_by_word is the
    reference's on ALL 50 blocks, every draw consumed, on the HIP kernels and on stock PyTorch autograd alike.  Updates: a block's
    200 whole-word Adam iterations are a chaotic map at fp32 -- from the SAME weights and a fresh optimizer (block 0) stock
    PyTorch on this GPU ends 1e-3 from the CPU reference, at later blocks single weights up to 0.3 away (they move by ~0.1 per
    block) -- so no implementation can be held to a rounding-sized bound per block here; what is asserted is that the HIP kernels
    stand where stock PyTorch-GPU stands: per block the median |weight - reference's| of the two routes within a factor of 3 of
    each other in the geometric mean over the training blocks, and below 1e-3 on every block (measured: medians 1e-8 ... 8e-4, the
    ratio's geometric mean 1.1).  The iteration-by-iteration statement (every Adam step of this flow against torch and a float64
    referee) is tests/test_gpu_replay.py."""
    g, g7 = golden("g15_by_word_reference_defaults"), golden("g7_by_word")
    ref = g["c4_meta_ser_by_word"]
    stats = {}
    for hip in (True, False):
        ser, draws, st = _resynchronised_run(g, g7, "c4_meta", dev, hip)
        assert np.array_equal(ser, ref), (hip, np.flatnonzero(ser != ref), ser[ser != ref], ref[ser != ref])
        assert draws.used_up(), (draws.r_at, len(draws.randint))
        stats[hip] = st
    both = (stats[True]["median"] > 0) & (stats[False]["median"] > 0)
    ratio = stats[True]["median"][both] / stats[False]["median"][both]
    geo = float(np.exp(np.mean(np.log(ratio))))
    print(f"g15 c4_meta re-synchronised: ser identical on all 50 blocks on both routes; per-block median |w - reference's| hip "
          f"{stats[True]['median'][both].min():.1e} ... {stats[True]['median'].max():.1e}, stock autograd {stats[False]['median'][both].min():.1e} ... "
          f"{stats[False]['median'].max():.1e}; hip / autograd geometric mean {geo:.2f} over {int(both.sum())} training blocks (range "
          f"{ratio.min():.2f} ... {ratio.max():.2f}); worst single weight hip {stats[True]['max'].max():.2f}, autograd {stats[False]['max'].max():.2f}; "
          f"block 0 (same weights, fresh optimizer): hip {stats[True]['max'][0]:.1e}, autograd {stats[False]['max'][0]:.1e}")
    assert int(both.sum()) >= 35
    assert 1 / 3 <= geo <= 3
    assert stats[True]["median"].max() <= 1e-3 and stats[False]["median"].max() <= 1e-3


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tag", ["c2_selfsup", "c4_meta"])
@pytest.mark.parametrize("route", ["hip_kernels", "torch_autograd", "batched_trials"])
def test_reference_by_word_flow_at_reference_defaults(golden, dev, tag, route):
    """Golden G15: BASELINE configs[2] (ViterbiNet over the COST2100 taps, 200 self-supervised minibatch iterations per block) and
    configs[4] (Meta-ViterbiNet: 200 whole-word iterations per block from the saved weights, every 5 blocks 20 x <= 10 second-order
    meta-learning steps) at the reference's OWN hyperparameters, 50 blocks at 9 dB, run by the unmodified reference with every draw
    recorded -- 7 800 and ~11 000 Adam steps on weights that move by 0.8.
    This is the FREE run (no re-synchronisation: test_reference_defaults_resynchronised_block_by_block above is the parity
    statement over all 50 blocks).  A free run of that length is where fp32 trajectories of two implementations part: the weights
    end 0.1 apart, and at some block a word near the decision threshold decodes with one or two bit errors more or less -- after
    which the runs also train on different blocks.  Stock PyTorch on this GPU ('torch_autograd': torch's own autograd and Adam,
    no kernel of this repo in the training) parts from the CPU reference the same way, at the same or an earlier block (measured
    first differing block: 34-35 of 50 for configs[2], 44-50 for configs[4]).  Asserted here: the first differing block (if
    any) within two bit errors, the mean ser of the 50 blocks within half of the reference's; the block is printed."""
    from meta_viterbinet_amd.trials import TrialBank, eval_by_word_batched

    g, g7 = golden("g15_by_word_reference_defaults"), golden("g7_by_word")
    ref = g[f"{tag}_ser_by_word"]
    if route == "batched_trials":
        R = 2
        kw, opt, nsym, subframes = _flow_kwargs(g, g7, tag, dev)
        tx = torch.tensor(g[f"{tag}_tx"], device=dev).float().unsqueeze(0).repeat(R, 1, 1)
        rx = torch.tensor(g[f"{tag}_rx"], device=dev).unsqueeze(0).repeat(R, 1, 1)
        bank = TrialBank([[g7[f"w{i}"] for i in range(6)]] * R, 16, 4, dev)
        draws = [RecordedTableDraws(g, tag, kw["self_supervised_iterations"], subframes, dev, lenient=True) for _ in range(R)]
        ser_all = eval_by_word_batched(bank, tx, rx, nsym, subframes, draws, **kw)
        ser, w = ser_all[0], [t.cpu().numpy() for t in bank.weights(0)]
        assert np.array_equal(ser_all[1], ser_all[0]) and torch.equal(bank.theta[0], bank.theta[1])
    else:
        ser, draws, w, _ = _run(g, g7, tag, dev, route == "hip_kernels", lenient=True)
    assert ser.shape == ref.shape == (50,)
    differ = np.flatnonzero(ser != ref)
    first = int(differ[0]) if len(differ) else 50
    moved = max(float(np.abs(g[f"{tag}_w1_{i}"] - g7[f"w{i}"]).max()) for i in range(6))
    worst = max(float(np.abs(w[i] - g[f"{tag}_w1_{i}"]).max()) for i in range(6))
    print(f"g15 {tag} {route}: ser identical on the first {first} of 50 blocks"
          + (f" (block {first}: {ser[first]:.4f} vs the reference's {ref[first]:.4f})" if first < 50 else "")
          + f"; mean ser {ser.mean():.5f} vs {ref.mean():.5f}; weights moved {moved:.3f}, end {worst:.2e} from the reference's")
    if first < 50:
        assert abs(ser[first] - ref[first]) <= 2.5 / 120
    assert abs(ser.mean() - ref.mean()) <= 0.5 * ref.mean()  # (50 blocks: a handful of bit errors either way after the runs part)


class RecordedTableDraws(RecordedDraws):
    """The same recorded draws for trials.eval_by_word_batched, which reads a trial's minibatches from a [blocks, iterations, M]
    table by block number: the reference's rows (call order) are laid at the blocks it trained on — the pilots and the data blocks
    whose ser it reported <= ser_thresh (trainer.py:345).  A run that trains on any other block reads rows of -1 and fails."""

    def __init__(self, g, tag, iterations, subframes, device, lenient=False):
        super().__init__(g[f"{tag}_multinomial"], g[f"{tag}_randint_high"], g[f"{tag}_randint"], device,
                         g[f"{tag}_init_weights"] if f"{tag}_init_weights" in g.files else None)
        self.lenient = lenient
        ser = g[f"{tag}_ser_by_word"]
        trained = np.flatnonzero((np.arange(len(ser)) % subframes == 0) | (ser <= 0.02))
        table = np.full((len(ser), iterations, 32), -1, np.int32)
        if lenient:  # a run that parts from the reference trains on other blocks: valid samples there (1..32), not the -1 trap
            table[:] = np.arange(1, 33, dtype=np.int32)
        if self.multinomial.size:
            assert len(self.multinomial) == iterations * len(trained)
            table[trained] = self.multinomial.reshape(len(trained), iterations, 32)
        self._table = torch.as_tensor(table, device=device)
        self.m_at = len(self.multinomial)

    def batches(self, count, n_blocks, T, iterations, M):
        assert self._table.shape == (n_blocks, iterations, M)
        return self._table[count]


@pytest.mark.parametrize("tag", ["selfsup", "meta"] + G13)
def test_reference_by_word_flow_batched_trials(golden, dev, tag):
    """The same reference runs as R = 3 identical trials of eval_by_word_batched (the trial-batched training kernels, RMSprop / SGD included
    since round 5): every row must reproduce the reference's ser_by_word and end on its weights."""
    from meta_viterbinet_amd.trials import TrialBank, eval_by_word_batched

    g, g7 = golden("g12_by_word_with_updates" if tag in ("selfsup", "meta") else "g13_by_word_switches"), golden("g7_by_word")
    R = 3
    kw, opt, nsym, subframes = _flow_kwargs(g, g7, tag, dev)
    tx = torch.tensor(g[f"{tag}_tx"], device=dev).float().unsqueeze(0).repeat(R, 1, 1)
    rx = torch.tensor(g[f"{tag}_rx"], device=dev).unsqueeze(0).repeat(R, 1, 1)
    bank = TrialBank([[g7[f"w{i}"] for i in range(6)]] * R, 16, 4, dev, **opt)
    draws = [RecordedTableDraws(g, tag, kw["self_supervised_iterations"], subframes, dev) for _ in range(R)]
    if opt and kw.get("online_meta"):  # trial after trial through harness.eval_by_word: the draws are asked for in call order
        draws = [_recorded(g, tag, dev) for _ in range(R)]
    ser = eval_by_word_batched(bank, tx, rx, nsym, subframes, draws, **kw)
    ref = g[f"{tag}_ser_by_word"]
    for r in range(R):
        assert np.array_equal(ser[r], ref), (r, np.flatnonzero(ser[r] != ref))
        assert draws[r].used_up()
        worst = max(float(np.abs(w.cpu().numpy() - g[f"{tag}_w1_{i}"]).max()) for i, w in enumerate(bank.weights(r)))
        assert worst <= 5e-5, (r, worst)
        if kw.get("online_meta"):
            worst_s = max(float(np.abs(w.cpu().numpy() - g[f"{tag}_saved_{i}"]).max()) for i, w in enumerate(bank.weights(r, saved=True)))
            assert worst_s <= 5e-5, (r, worst_s)
    assert torch.equal(bank.theta[0], bank.theta[1]) and torch.equal(bank.theta[0], bank.theta[2])


@pytest.mark.parametrize("tag", ["va_coded", "va_plain", "vnet_coded", "vnet_plain", "meta_coded"])
def test_reference_aggregated_evaluate(golden, dev, tag):
    """Golden G14: the reference's evaluate() in 'aggregated' mode (trainer.py:368-381 -> :254-265 -> :243-252 -> :222-241) over
    three SNR points as VATrainer / VNETTrainer / METAVNETTrainer, coded and uncoded.  harness.single_eval_at_point on the words the
    reference drew must return its ser at every point (the reference takes a float32 mean of the bit errors, the harness divides
    integer counters: the same number of bit errors, the ratio equal to float32 rounding), and the generated-words route must reproduce those words."""
    g, g7 = golden("g14_aggregated_evaluate"), golden("g7_by_word")
    ecc, nsym, L, words, T = [int(v) for v in g[f"{tag}_meta"]]
    src = "vnet_coded" if tag == "meta_coded" else tag
    if tag.startswith("va"):
        det = mvn.VADetector(16, L, T, words, "ISI_AWGN", 0, True, 2, {"train": "time_decay", "val": "time_decay"})
    else:
        det = (mvn.META_VNETDetector if tag.startswith("meta") else mvn.VNETDetector)(16, {"train": T, "val": T}).to(dev)
        w = [torch.tensor(g7[f"w{i}"], device=dev) for i in range(6)]
        if tag.startswith("meta"):  # the reference's trainer calls its plain detector either way (metavnet_trainer.py:30-39)
            meta, det = det, (lambda rx, phase, snr, gamma: meta(rx, phase, w))
        else:
            with torch.no_grad():
                for p, a in zip(det.parameters(), w):
                    p.copy_(a)
    rows = torch.tensor(g[f"{tag}_data_indices"], device=dev)
    for k, snr in enumerate((8, 9, 10)):
        tx = torch.tensor(g[f"{src}_tx{snr}"], device=dev).float()
        rx = torch.tensor(g[f"{src}_rx{snr}"], device=dev)
        ser, fer, c = mvn.single_eval_at_point(det, tx, rx, float(snr), 0.2, rows, n_symbols=nsym if ecc else 0)
        ref = float(g[f"{tag}_ser"][k])
        assert c.tolist()[1] == len(rows) * 120 and c.tolist()[3] == len(rows)
        assert ser == pytest.approx(ref, rel=2e-5, abs=1e-7), (tag, snr, ser, ref)  # (the reference's mean carries float32 rounding)
        assert int(c[0]) == int(round(ref * len(rows) * 120)), (tag, snr)  # the integer behind the reference's mean
