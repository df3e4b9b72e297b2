"""CPU-only: host-side mirror of the reference interface (channel taps, state priors, labels, 'train'
phase, exceptions, sharding arithmetic) against the golden vectors."""
import numpy as np
import pytest
import torch

import meta_viterbinet_amd as mvn

CC = {"train": "time_decay", "val": "time_decay"}


def test_estimate_channel_tables(golden):
    g = golden("g6_channels")
    got = np.concatenate([mvn.estimate_channel(4, 0.2, "cost2100", index=i) for i in range(300)])
    assert np.array_equal(got, g["cost2100"])
    assert got[0].tolist() == pytest.approx([1, 0.44173482, 0.29808132, 0.04954621], rel=1e-7)
    for ttype in (1, 2):
        got = np.concatenate([mvn.estimate_channel(4, 0.2, "time_decay", fading=True, index=i, fading_taps_type=ttype)
                              for i in range(300)])
        assert np.array_equal(got, g[f"time_decay_fading{ttype}"])
    for L in (2, 3, 4, 8):
        assert np.array_equal(mvn.estimate_channel(L, 0.2, "time_decay"), g[f"time_decay_L{L}"])
    assert np.array_equal(mvn.estimate_channel(4, 0.5, "time_decay"), g["time_decay_gamma05"])
    with pytest.raises(ValueError):
        mvn.estimate_channel(4, 0.2, "nope")
    with pytest.raises(ValueError):
        mvn.estimate_channel(4, 0.2, "time_decay", fading=True, fading_taps_type=3)
    with pytest.raises(ValueError):  # fading hard-codes 4 taps
        mvn.estimate_channel(8, 0.2, "time_decay", fading=True)


@pytest.mark.parametrize("name", ["L4_static", "L4_fading1", "L4_fading2", "L4_cost2100", "L2_static", "L3_static",
                                  "L8_static"])
def test_state_priors_match_reference(golden, name):
    g = golden("g2_va")
    L, frames, sub, T, snr, fdec, ttype = [int(v) for v in g[f"{name}_meta"]]
    coef = str(g[f"{name}_coef"])
    det = mvn.VADetector(2 ** L, L, T, frames * sub, "ISI_AWGN", 0, bool(fdec), ttype,
                         {"train": "time_decay", "val": coef})
    h = det._estimate_all(0.2, "val")
    assert np.array_equal(h, g[f"{name}_h"])
    pri = det.compute_state_priors(h)
    assert pri.dtype == torch.float32 and tuple(pri.shape) == (2 ** L, frames * sub)
    assert np.array_equal(pri.cpu().numpy(), g[f"{name}_state_priors"])
    assert np.array_equal(mvn.data_indices(frames, sub).numpy(), g[f"{name}_data_indices"])
    # materialised costs (API-parity helper) are the reference's, bit for bit (torch CPU both sides)
    y = torch.tensor(g[f"{name}_rx"])
    head = g[f"{name}_cost_head"]
    cost = det.compute_likelihood_priors(y, snr, 0.2, "val")
    assert np.array_equal(cost[: head.shape[0], :8].numpy(), head)


def test_va_errors():
    det = mvn.VADetector(16, 4, 8, 1, "ISI_AWGN", 0, False, 1, CC)
    with pytest.raises(NotImplementedError):
        det(torch.zeros(1, 8), "train")
    bad = mvn.VADetector(16, 4, 8, 1, "OTHER", 0, False, 1, CC)
    with pytest.raises(Exception, match="No such channel defined"):
        bad.compute_state_priors(np.ones((1, 4)))
    assert det.transition_table_array.tolist() == [[(2 * s) % 16, (2 * s + 1) % 16] for s in range(16)]
    assert det.transition_table.dtype == torch.float32  # kept as a float tensor like the reference (:40)


def test_transition_table(golden):
    g = golden("g1_acs_block")
    for S in (2, 4, 8, 16, 256):
        assert np.array_equal(mvn.create_transition_table(S), g[f"table_S{S}"])


def test_calculate_states(golden):
    g = golden("g5_kats")
    assert mvn.calculate_states(4, torch.tensor(g["states_kat_in"])).tolist() == [13, 6, 3, 9, 4, 2, 1]
    for L in (2, 3, 4, 8):
        got = mvn.calculate_states(L, torch.tensor(g[f"states_L{L}_in"]))
        assert got.dtype == torch.int64 and np.array_equal(got.numpy(), g[f"states_L{L}_out"])


@pytest.mark.parametrize("name", ["S16_trained_exact", "S4_trained_exact", "S256_init_exact"])
def test_vnet_train_phase_and_checkpoint_keys(golden, name):
    """'train' phase = plain torch with autograd; parameter names/order as the reference (a6)."""
    g = golden("g3_vnet")
    S, B, T, _ = [int(v) for v in g[f"{name}_meta"]]
    torch.set_num_threads(1)
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to("cpu")
    keys = list(det.state_dict().keys())
    assert keys == [str(k) for k in g["state_dict_keys"]]
    sd = {k: torch.tensor(g[f"{name}_w{i}"]) for i, k in enumerate(keys)}
    det.load_state_dict(sd)
    assert [tuple(p.shape) for p in det.parameters()] == [(100, 1), (100,), (50, 100), (50,), (S, 50), (S,)]
    y = torch.tensor(g[f"{name}_y"])
    logits = det(y, "train")
    assert logits.requires_grad and tuple(logits.shape) == (B, T, S)
    assert np.array_equal(logits.detach().numpy(), g[f"{name}_logits"])
    meta = mvn.META_VNETDetector(S, {"train": T, "val": T})
    assert len(list(meta.parameters())) == 0
    lm = meta(y, "train", list(det.parameters()))
    assert torch.equal(lm, logits)
    g0 = torch.autograd.grad(lm.sum(), list(det.parameters()))  # MAML differentiates through var
    assert all(t is not None for t in g0)
    import copy

    clone = copy.deepcopy(det)  # trainer.py:275
    assert torch.equal(clone(y, "train"), logits)


def test_shard_range_partitions():
    for n in (0, 1, 7, 100, 10000, 10001):
        for world in (1, 2, 3, 4, 8):
            spans = [mvn.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_rates_from_counters():
    assert mvn.rates_from_counters(torch.tensor([3, 100, 1, 4])) == (0.03, 0.25)
    ser, fer = mvn.rates_from_counters(torch.tensor([0, 0, 0, 0]))
    assert np.isnan(ser) and np.isnan(fer)


@pytest.mark.parametrize("tag,maml", [("maml", True), ("fo", False)])
def test_meta_train_loop_golden(golden, tag, maml):
    """G11: four Trainer.meta_train_loop steps of the reference (second-order MAML and first-order), CPU torch.
    Same autograd graph; the Adam update runs through OnlineTrainer.adam_step.  Tolerance 1e-6 abs (op order of Adam)."""
    g = golden("g11_meta_train_loop")
    torch.set_num_threads(1)
    det = mvn.VNETDetector(16, {"train": 136, "val": 136}).to("cpu")
    with torch.no_grad():
        for i, p in enumerate(det.parameters()):
            p.copy_(torch.tensor(g[f"{tag}_w0_{i}"]))
    meta = mvn.META_VNETDetector(16, {"train": 136, "val": 136})
    tr = mvn.OnlineTrainer(det, 4, lr=float(g["lr"]))
    tx, rx = torch.tensor(g["tx"].astype(np.float32)), torch.tensor(g["rx"])
    losses = []
    for sup, qry in g["pairs"]:
        losses.append(float(mvn.meta_train_loop(det, meta, tr, rx, tx, torch.tensor([int(sup)]), torch.tensor([int(qry)]),
                                                float(g["meta_lr"]), maml)))
    assert np.allclose(losses, g[f"{tag}_loss"], rtol=1e-6)
    for i, p in enumerate(det.parameters()):
        assert np.allclose(p.detach().numpy(), g[f"{tag}_w1_{i}"], rtol=0, atol=1e-6), i
    assert tr.step == 4


def test_online_training_falls_back_to_autograd_above_128_states():
    """memory_length 8 (256 states): the parameter set does not fit the training kernel's LDS image, so
    OnlineTrainer.online_training runs run_train_loop (trainer.py:492-505) on stock autograd -- same draws, same shared
    Adam state -- instead of raising.  Checked against torch.optim.Adam on a copy (CPU tensors: no GPU needed)."""
    import copy

    import torch.nn.functional as F

    L, S, T, iters = 8, 256, 96, 7
    torch.manual_seed(3)
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to("cpu")
    det.net.to("cpu")
    ref = copy.deepcopy(det)
    tx = torch.randint(0, 2, (1, T)).float()
    rx = torch.randn(1, T)
    tr = mvn.OnlineTrainer(det, L)
    idx = tr.select_batches(T, iters)
    loss = tr.online_training(tx, rx, iterations=iters, batch_idx=idx, return_loss=True)
    opt = torch.optim.Adam(ref.parameters(), lr=0.001)
    labels = mvn.calculate_states(L, tx).reshape(-1).long()
    want = []
    for it in range(iters):
        out = ref(rx, "train").reshape(-1, S)
        lo = F.cross_entropy(out[idx[it].long()], labels[idx[it].long()])
        opt.zero_grad()
        lo.backward()
        opt.step()
        want.append(float(lo))
    assert tr.step == iters and np.allclose(loss.numpy(), want, rtol=1e-5)
    for a, b in zip(det.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("opt", ["RMSprop", "SGD", "Adam"])
def test_optimizer_variants_match_torch_optim(opt):
    """deep_learning_setup's three optimizers (trainer.py:163-175): OnlineTrainer.optimizer_step against torch.optim with
    its defaults, three steps on random gradients (CPU tensors)."""
    import copy

    torch.manual_seed(4)
    det = mvn.VNETDetector(16, {"train": 8, "val": 8}).to("cpu")
    det.net.to("cpu")
    ref = copy.deepcopy(det)
    tr = mvn.OnlineTrainer(det, 4, optimizer_type=opt)
    o = {"RMSprop": torch.optim.RMSprop, "SGD": torch.optim.SGD, "Adam": torch.optim.Adam}[opt](ref.parameters(), lr=0.001)
    for _ in range(3):
        grads = [torch.randn_like(p) for p in det.parameters()]
        tr.optimizer_step(grads)
        for p, g in zip(ref.parameters(), grads):
            p.grad = g.clone()
        o.step()
    for a, b in zip(det.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    with pytest.raises(NotImplementedError):
        mvn.OnlineTrainer(det, 4, optimizer_type="Adagrad")


def test_trial_draws_and_descriptor_helpers():
    """Host side of trials.eval_by_word_batched (no GPU): a trial's draws are a pure function of its seed -- the meta-learning
    update's j_hat values drawn in one call equal the reference's per-iteration torch.unique(randint) pattern on the same
    stream --, beta powers are libm pow of the float-rounded beta like the C entry points compute them, and the flat
    parameter layout is parameters() order."""
    from meta_viterbinet_amd import trials

    a, b = trials.TrialDraws(7, "cpu"), trials.TrialDraws(7, "cpu")
    one = a.j_hat_update(37, 20, 10)
    per_iteration = np.concatenate([np.unique(b.rng.randint(0, 37, size=10)) for _ in range(20)])
    assert np.array_equal(one, per_iteration) and one.min() >= 0 and one.max() < 37
    assert not np.array_equal(one, trials.TrialDraws(8, "cpu").j_hat_update(37, 20, 10))
    t1 = trials.TrialDraws(3, "cpu").batches(4, 6, 40, 5, 8)
    t2 = trials.TrialDraws(3, "cpu").batches(4, 6, 40, 5, 8)
    assert t1.shape == (5, 8) and t1.dtype == torch.int32 and torch.equal(t1, t2)
    assert int(t1.min()) >= 1 and int(t1.max()) < 40  # select_batch's weights arange(T): sample 0 is never drawn
    assert all(len(set(row.tolist())) == 8 for row in t1)  # without replacement
    with pytest.raises(ValueError):
        a.batches(0, 6, 40, 5, 8)
        a.batches(0, 7, 40, 5, 8)  # a trial's table is drawn once, for one shape
    steps = np.array([0, 1, 200, 15315], dtype=np.int64)
    want = [float(np.float32(0.999)) ** int(s) for s in steps]
    assert trials.beta_powers(0.999, steps).tolist() == want and trials.beta_power(0.999, 200) == want[2]
    off = trials.param_offsets(16)
    assert off.tolist() == [0, 100, 200, 5200, 5250, 6050, 6066]
    det = mvn.VNETDetector(16, {"train": 8, "val": 8}).to("cpu")
    assert sum(p.numel() for p in det.parameters()) == off[-1]
    assert [p.numel() for p in det.parameters()] == np.diff(off).tolist()
    assert trials.TRIAL_DTYPE.itemsize == 232
    # the reference's per-block ser from an error count (metrics.py:11-16), vectorised
    assert mvn.metrics.ser_from_errors(np.array([0, 3, 120]), 120).tolist() == [0.0, 1.0 - float(np.float32(117) / np.float32(120)), 1.0]
