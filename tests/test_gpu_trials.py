"""GPU tests (pytest -m gpu) of the by-word step kernel, of R evaluations advancing together on one GPU, and of the training
kernels' failure reporting and behaviour under concurrency.  Bit-exact comparisons throughout: the batched forms run the
same arithmetic as the sequential ones, which are pinned to the reference elsewhere (test_gpu_parity.py, G7-G11)."""
import ctypes

import numpy as np
import pytest
import torch

import meta_viterbinet_amd as mvn
from meta_viterbinet_amd.trials import TrialBank, TrialDraws, eval_by_word_batched

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    assert mvn._lib.load().mvn_device_info(None, None, None, 0) == 0
    return torch.device("cuda:0")


def _np(t):
    return t.detach().cpu().numpy()


def _vnet_with(w, T, dev):
    det = mvn.VNETDetector(16, {"train": T, "val": T}).to(dev)
    with torch.no_grad():
        for p, a in zip(det.parameters(), w):
            p.copy_(torch.as_tensor(a))
    return det


def _trial_weights(golden, R, seed=0, spread=0.05):
    """R weight sets: the reference-trained weights of G7, each perturbed a little so that the trials differ."""
    g7 = golden("g7_by_word")
    rng = np.random.RandomState(seed)
    base = [g7[f"w{i}"] for i in range(6)]
    return [[(a * (1.0 + spread * rng.standard_normal(a.shape))).astype(np.float32) if r else a.astype(np.float32) for a in base]
            for r in range(R)]


def _words(dev, R, N, K, nsym, snrs, seed, L=4):
    """tx [R, N, K] message bits and rx [R, N, K + 8 nsym]: trial r at snrs[r] over the fading time-decay channel."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    T = K + 8 * nsym
    msg = torch.randint(0, 2, (R, N, K), generator=gen, device=dev).float()
    cw = mvn.rs_encode(msg.reshape(R * N, K), nsym)
    h = np.concatenate([mvn.estimate_channel(L, 0.2, "time_decay", fading=True, index=i, fading_taps_type=2) for i in range(N)])
    rx = torch.stack([mvn.transmit(cw[r * N:(r + 1) * N], h, float(snrs[r]), L,
                                   torch.randn(N, T, generator=gen, device=dev)) for r in range(R)])
    return msg, rx.contiguous()


def _step(dev, rx, tx, bank_theta, off, nsym, pilot, stride=True):
    """mvn_vnet_byword_step_f32 with every output requested."""
    R, T = rx.shape
    K = T - 8 * nsym
    out = dict(dec=torch.full((R, T), 7.0, device=dev), msg=torch.full((R, K), 7.0, device=dev), enc=torch.full((R, T), 7.0, device=dev),
               lw=torch.full((R, T), 7.0, device=dev), labels=torch.full((R, T), -1, dtype=torch.int32, device=dev),
               nerr=torch.full((R,), -1, dtype=torch.int32, device=dev))
    P = bank_theta.shape[1]
    wp = [ctypes.c_void_p(bank_theta.data_ptr() + 4 * int(off[a])) for a in range(6)]
    ws = (ctypes.c_int64 * 6)(*([P] * 6)) if stride else None
    rc = mvn._lib.load().mvn_vnet_byword_step_f32(mvn._lib.ptr(rx), T, mvn._lib.ptr(tx), K, *wp, ws, mvn._lib.ptr(out["dec"]), T,
                                                  mvn._lib.ptr(out["msg"]), K, mvn._lib.ptr(out["enc"]), T, mvn._lib.ptr(out["lw"]), T,
                                                  mvn._lib.ptr(out["labels"]), T, mvn._lib.ptr(out["nerr"]), R, T, nsym,
                                                  1 if pilot else 0, 16, mvn._lib.current_stream(dev))
    assert rc == 0
    return out


@pytest.mark.parametrize("R,K,nsym,snr", [(1, 120, 2, 10.0), (33, 120, 2, 6.0), (5, 120, 8, 4.0), (3, 984, 5, 5.0), (40, 8, 1, 3.0)])
def test_byword_step_equals_the_separate_launches(golden, dev, R, K, nsym, snr):
    """One launch = detect + RS decode + error count + RS encode + label word + trellis states, for R words with R weight
    sets, against VNETDetector.forward('val'), mvn.rs_decode, a plain comparison, mvn.rs_encode and calculate_states run
    word by word.  The SNRs are low enough that some words exceed the code's correction capacity (the reference's
    uncorrected / mis-corrected outputs included), and one word is sent clean so that the no-error branch is taken too."""
    T = K + 8 * nsym
    w = _trial_weights(golden, R, seed=R + K)
    if R >= 5:  # one trial with a NaN in its last layer: the step kernel's detector follows torch.min's NaN rule like the others
        w[2][4][1, 7] = np.nan
    bank = TrialBank(w, 16, 4, dev)
    msg, rx = _words(dev, R, 1, K, nsym, [snr] * (R - 1) + [40.0], seed=7 * R + nsym)
    msg, rx = msg[:, 0].contiguous(), rx[:, 0].contiguous()
    out = _step(dev, rx, msg, bank.theta, bank.off, nsym, pilot=False)
    n_err_words = 0
    for r in range(R):
        det = _vnet_with(w[r], T, dev)
        dec = det(rx[r:r + 1], "val")
        dmsg = mvn.rs_decode(dec, nsym)
        enc = mvn.rs_encode(dmsg, nsym)
        nerr = int((dmsg != msg[r:r + 1]).sum().item())
        lw = dec if nerr > 0 else enc
        assert torch.equal(out["dec"][r:r + 1], dec), r
        assert torch.equal(out["msg"][r:r + 1], dmsg), r
        assert torch.equal(out["enc"][r:r + 1], enc), r
        assert int(out["nerr"][r].item()) == nerr, r
        assert torch.equal(out["lw"][r:r + 1], lw), r
        assert torch.equal(out["labels"][r].long(), mvn.calculate_states(4, lw)), r
        n_err_words += nerr > 0
    if nsym >= 2:  # (a lone parity byte corrects nothing, and symbol 0 of every word is decided 0: quirk Q1)
        assert int(out["nerr"][R - 1].item()) == 0
    if R > 1:
        assert n_err_words > 0  # the detected-word branch of the label rule ran
    # pilot step: the known word is encoded, nothing is detected (dec / msg untouched)
    pil = _step(dev, rx, msg, bank.theta, bank.off, nsym, pilot=True)
    enc = mvn.rs_encode(msg, nsym)
    assert torch.equal(pil["enc"], enc) and torch.equal(pil["lw"], enc) and int(pil["nerr"].abs().sum().item()) == 0
    assert torch.equal(pil["labels"].long().reshape(-1), mvn.calculate_states(4, enc))
    assert bool((pil["dec"] == 7.0).all()) and bool((pil["msg"] == 7.0).all())
    # one weight set for all words (w_stride NULL): every word decoded with trial 0's weights
    shared = _step(dev, rx, msg, bank.theta, bank.off, nsym, pilot=False, stride=False)
    assert torch.equal(shared["dec"], _vnet_with(w[0], T, dev)(rx, "val"))


@pytest.mark.parametrize("R,K,nsym,snr,Bp", [(1, 120, 2, 9.0, 1), (24, 120, 2, 5.0, 24), (6, 120, 8, 3.0, 3), (3, 984, 5, 4.0, 1), (9, 8, 1, 2.0, 9)])
def test_va_byword_step_equals_the_separate_launches(dev, R, K, nsym, snr, Bp):
    """mvn_va_byword_step_f32: one launch = VADetector.forward('val') + RS decode + error count + RS encode + label word + trellis
    states for R words (word r with row r % Bp of the state priors: its own channel), against mvn_va_decode_f32, mvn.rs_decode, a
    plain comparison, mvn.rs_encode and calculate_states word by word; SNRs low enough that words exceed the code's capacity."""
    T, L = K + 8 * nsym, 4
    msg, rx = _words(dev, R, 1, K, nsym, [snr] * (R - 1) + [40.0], seed=3 * R + nsym)
    msg, rx = msg[:, 0].contiguous(), rx[:, 0].contiguous()
    va = mvn.VADetector(16, L, T, 1, "ISI_AWGN", 0, False, 1, {"train": "time_decay", "val": "time_decay"})
    h = np.concatenate([mvn.estimate_channel(L, 0.2, "time_decay", fading=True, index=i, fading_taps_type=2) for i in range(Bp)])
    if Bp > 1 and R >= 6:
        h[1, 2] = np.nan  # one word's channel estimate is broken: torch.min's NaN rule inside the step's detector too
    pri = va.compute_state_priors(h).to(dev).T.contiguous()  # [Bp, 16]
    lib = mvn._lib.load()
    out = dict(dec=torch.full((R, T), 7.0, device=dev), msg=torch.full((R, K), 7.0, device=dev), enc=torch.full((R, T), 7.0, device=dev),
               lw=torch.full((R, T), 7.0, device=dev), labels=torch.full((R, T), -1, dtype=torch.int32, device=dev),
               nerr=torch.full((R,), -1, dtype=torch.int32, device=dev))

    def step(pilot, o):
        rc = lib.mvn_va_byword_step_f32(mvn._lib.ptr(rx), T, mvn._lib.ptr(msg), K, mvn._lib.ptr(pri), Bp, mvn._lib.ptr(o["dec"]), T,
                                        mvn._lib.ptr(o["msg"]), K, mvn._lib.ptr(o["enc"]), T, mvn._lib.ptr(o["lw"]), T,
                                        mvn._lib.ptr(o["labels"]), T, mvn._lib.ptr(o["nerr"]), R, T, nsym, 1 if pilot else 0, 16,
                                        mvn._lib.current_stream(dev))
        assert rc == 0

    step(False, out)
    dec_ref = torch.empty((R, T), device=dev)
    assert lib.mvn_va_decode_f32(mvn._lib.ptr(rx), T, mvn._lib.ptr(pri), Bp, mvn._lib.ptr(dec_ref), T, None, R, T, 16,
                                 mvn._lib.current_stream(dev)) == 0
    assert torch.equal(out["dec"], dec_ref)
    dmsg = mvn.rs_decode(dec_ref, nsym)
    enc = mvn.rs_encode(dmsg, nsym)
    nerr = (dmsg != msg).sum(dim=1).to(torch.int32)
    lw = torch.where((nerr > 0).reshape(-1, 1), dec_ref, enc)
    assert torch.equal(out["msg"], dmsg) and torch.equal(out["enc"], enc) and torch.equal(out["nerr"], nerr)
    assert torch.equal(out["lw"], lw) and torch.equal(out["labels"].long().reshape(-1), mvn.calculate_states(L, lw))
    if R > 1:
        assert int((nerr > 0).sum()) > 0  # the detected-word branch of the label rule ran
    pil = {k: v.clone().fill_(7 if v.dtype.is_floating_point else -1) for k, v in out.items()}
    step(True, pil)
    enc_tx = mvn.rs_encode(msg, nsym)
    assert torch.equal(pil["enc"], enc_tx) and torch.equal(pil["lw"], enc_tx) and int(pil["nerr"].abs().sum()) == 0
    assert torch.equal(pil["labels"].long().reshape(-1), mvn.calculate_states(L, enc_tx))
    assert bool((pil["dec"] == 7.0).all()) and bool((pil["msg"] == 7.0).all())


def test_eval_by_word_fused_step_equals_separate_launches(golden, dev):
    """harness.eval_by_word with one launch per block against the four-launch route: the no-update loop of G7 (which is
    also pinned to the reference's ser_by_word there) and a self-supervised run (same draws): identical ser and weights."""
    g7 = golden("g7_by_word")
    w = [g7[f"w{i}"] for i in range(6)]
    msg, rx = _words(dev, 1, 100, 120, 2, [8.0], seed=21)
    msg, rx = msg[0], rx[0]
    a = mvn.eval_by_word(_vnet_with(w, 136, dev), msg, rx, 8.0, 0.2, 2, 25)
    b = mvn.eval_by_word(_vnet_with(w, 136, dev), msg, rx, 8.0, 0.2, 2, 25, fused_step=False)
    assert np.array_equal(a, b) and a.max() > 0
    out = []
    for fused in (True, False):
        det = _vnet_with(w, 136, dev)
        tr = mvn.OnlineTrainer(det, 4)
        ser = mvn.eval_by_word(det, msg, rx, 8.0, 0.2, 2, 25, self_supervised=True, online_trainer=tr, self_supervised_iterations=20,
                               draws=TrialDraws(5, dev), fused_step=fused)
        out.append((ser, [p.detach().clone() for p in det.parameters()], tr.exp_avg.clone(), tr.step))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][3] == out[1][3] > 0
    for p, q in zip(out[0][1], out[1][1]):
        assert torch.equal(p, q)
    assert torch.equal(out[0][2], out[1][2])


FLOWS = {
    # BASELINE configs[2] with updates: 32-sample minibatch iterations after every qualifying block (vnet_trainer.py:49-60)
    "self_supervised_minibatch": dict(self_supervised=True, self_supervised_iterations=25),
    # BASELINE configs[4]: meta-learning every 5 blocks + full-word iterations from the saved weights (metavnet_trainer.py:52-64)
    "meta_viterbinet": dict(self_supervised=True, self_supervised_iterations=12, online_meta=True, meta_train_iterations=3,
                            meta_j_num=4, meta_subframes=5, meta_style_online_training=True),
    "meta_first_order_window2": dict(self_supervised=False, online_meta=True, MAML=False, window_size=2, meta_train_iterations=2,
                                     meta_j_num=3, meta_subframes=5),
}


MANY_TRIALS_FLOWS = {  # the configs[4] flow with short iteration counts, for the 176-trial test
    "meta_viterbinet_short": dict(self_supervised=True, self_supervised_iterations=5, online_meta=True, meta_train_iterations=2,
                                  meta_j_num=3, meta_subframes=5, meta_style_online_training=True),
}


def _train_kernel_name(kind, R, T, M_or_W, S, ws_bytes):
    name = ctypes.create_string_buffer(128)
    assert mvn._lib.load().mvn_vnet_train_kernel_name(kind, R, T, M_or_W, S, ws_bytes, name, 128) == 0
    return name.value.decode()


# The launcher's forms of the trial-batched training kernels (mvn_hip.hip: plan_online_groups / plan_maml_groups):
#   chunked   -- one workgroup per 32-sample chunk AND trial: *_train_groups_kernel<SC, true> (the launcher's choice for few trials),
#                on the XCD-aware grid: a trial's workgroups on one XCD, gradient exchange through its L2 ("one XCD per trial")
#   chunked_spread -- the same kernels on the (groups, trials) grid: a trial's workgroups on different XCDs, write-through exchange
#                (what the launcher keeps where the XCD-aware grid would need more launches); MVN_TRAIN_XCD=0
#   per_trial -- one workgroup per trial: online_train_kernel<SC, true> with M = 0, maml_train_kernel<SC, true> -- the launcher's
#                choice from ~154 trials on, i.e. what bench.py times at 256 trials; forced here with MVN_TRAIN_GROUPS=0
#   pair      -- one 512-thread workgroup per trial, two of them per CU (online_train_kernel<SC, true, 512>: the launcher's choice
#                for more trials than CUs; Adam moments in global memory, every wave two roles per phase); MVN_TRAIN_PAIR=1
FORMS = {"chunked": {"MVN_TRAIN_GROUPS": "1", "MVN_TRAIN_PAIR": "0"}, "per_trial": {"MVN_TRAIN_GROUPS": "0", "MVN_TRAIN_PAIR": "0"},
         "pair": {"MVN_TRAIN_GROUPS": "0", "MVN_TRAIN_PAIR": "1"},
         "chunked_spread": {"MVN_TRAIN_GROUPS": "1", "MVN_TRAIN_PAIR": "0", "MVN_TRAIN_XCD": "0"}}
_SEQUENTIAL = {}


def _sequential_runs(golden, dev, flow, R, N, K, nsym, sub, snrs, w, msg, rx, seed0):
    """The R trials one after the other through harness.eval_by_word (the single-trial entry points), once per flow."""
    key = (flow, R, N)
    if key not in _SEQUENTIAL:
        out = []
        for r in range(R):
            det = _vnet_with(w[r], K + 8 * nsym, dev)
            tr = mvn.OnlineTrainer(det, 4)
            ser = mvn.eval_by_word(det, msg[r], rx[r], snrs[r], 0.2, nsym, sub, online_trainer=tr,
                                   meta_detector=mvn.META_VNETDetector(16, {"train": K + 8 * nsym, "val": K + 8 * nsym}),
                                   draws=TrialDraws(seed0 + r, dev), **{**FLOWS, **MANY_TRIALS_FLOWS}[flow])
            out.append((ser, [p.detach().clone() for p in det.parameters()], tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.step))
        _SEQUENTIAL[key] = out
    return _SEQUENTIAL[key]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("flow", sorted(FLOWS))
def test_batched_trials_equal_sequential_runs(golden, dev, monkeypatch, flow, form):
    """R trials stepping together (trials.eval_by_word_batched: one step launch, one sync, one launch sequence per training
    kind and block for ALL trials) against the same trials run one after the other through harness.eval_by_word with the
    same per-trial draws: ser_by_word, the final weights, the saved weights' effect, both Adam moments and the step counts
    are IDENTICAL per trial.  Trials sit at different SNRs, so they push, train and meta-learn at different blocks.
    Both forms of the trial kernels (FORMS) against the same sequential runs, which use the default single-trial kernels."""
    R, N, K, nsym, sub = 7, 58, 120, 2, 25
    T = K + 8 * nsym
    kw = FLOWS[flow]
    snrs = [6.0, 7.0, 8.0, 9.0, 10.0, 11.0, 12.0]
    w = _trial_weights(golden, R, seed=3)
    msg, rx = _words(dev, R, N, K, nsym, snrs, seed=11)
    for name in ("MVN_TRAIN_GROUPS", "MVN_TRAIN_PAIR", "MVN_TRAIN_XCD"):
        monkeypatch.delenv(name, raising=False)
    seq = _sequential_runs(golden, dev, flow, R, N, K, nsym, sub, snrs, w, msg, rx, 100)
    for name, value in FORMS[form].items():
        monkeypatch.setenv(name, value)
    W = kw.get("window_size", 1)
    ws_bytes = int(mvn._lib.load().mvn_vnet_train_trials_workspace_bytes(16, T, W, R))
    tag = "_groups_kernel<16, true>" if form.startswith("chunked") else "_kernel<16, true> 1x"
    online_tag = "_kernel<16, true, 512> 1x" if form == "pair" else tag
    if kw.get("meta_style_online_training"):  # full-word iterations
        assert online_tag in _train_kernel_name(0, R, T, 0, 16, ws_bytes)
    elif kw.get("self_supervised"):  # minibatch iterations are one chunk: one workgroup per trial, 1024 or 512 threads
        assert ("_kernel<16, true, 512> 1x" if form == "pair" else "_kernel<16, true> 1x") in _train_kernel_name(0, R, T, 32, 16, ws_bytes)
    if kw.get("online_meta"):
        assert tag in _train_kernel_name(2 if kw.get("MAML", True) else 1, R, T, W, 16, ws_bytes)
    bank = TrialBank(w, 16, 4, dev)
    rec = {}
    ser_b = eval_by_word_batched(bank, msg, rx, nsym, sub, [TrialDraws(100 + r, dev) for r in range(R)], record=rec, **kw)
    # the same trials as three cohorts stepping alternately on one stream (host work of one behind the GPU work of another)
    bank3 = TrialBank(w, 16, 4, dev)
    ser_3 = eval_by_word_batched(bank3, msg, rx, nsym, sub, [TrialDraws(100 + r, dev) for r in range(R)], cohorts=3, **kw)
    assert np.array_equal(ser_3, ser_b) and torch.equal(bank3.theta, bank.theta) and torch.equal(bank3.exp_avg_sq, bank.exp_avg_sq)
    assert np.array_equal(bank3.step, bank.step)
    trained_blocks = 0
    for r in range(R):
        ser, wr, m, v, step = seq[r]
        assert np.array_equal(ser, ser_b[r]), (flow, r)
        for a, b in zip(wr, bank.weights(r)):
            assert torch.equal(a, b), (flow, r)
        assert torch.equal(m, bank.exp_avg[r]) and torch.equal(v, bank.exp_avg_sq[r]), (flow, r)
        assert step == int(bank.step[r]), (flow, r)
        trained_blocks += int(rec["trained"][r].sum() + rec["meta"][r].sum())
    assert trained_blocks > R  # the flows did train
    assert len({tuple(row) for row in ser_b}) > 1  # and the trials are not copies of each other
    if kw.get("self_supervised"):
        per_trial = rec["trained"].sum(axis=1)
        assert per_trial.min() < per_trial.max()  # different trials trained on different blocks


SWITCH_FLOWS = {
    # meta_weights_init('random') (trainer.py:356-359): fresh weights from the trial's own stream + a fresh optimizer per update
    "random_init": dict(self_supervised=True, self_supervised_iterations=8, online_meta=True, meta_train_iterations=2, meta_j_num=4,
                        meta_subframes=5, meta_style_online_training=True, weights_init="random"),
    # buffer_empty=False (trainer.py:278-286, :325-328): the buffer starts with 6 words and stays 6 long
    "window_buffer": dict(self_supervised=True, self_supervised_iterations=6, online_meta=True, meta_train_iterations=3, meta_j_num=4,
                          meta_subframes=5, window_size=2, initial_buffer=6),
    # the optimizers the kernels do not implement (trainer.py:163-175): trial after trial on stock autograd
    "rmsprop": dict(self_supervised=True, self_supervised_iterations=3, online_meta=True, meta_train_iterations=1, meta_j_num=2,
                    meta_subframes=5, optimizer_type="RMSprop"),
    "sgd": dict(self_supervised=True, self_supervised_iterations=3, optimizer_type="SGD"),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("flow", sorted(SWITCH_FLOWS))
def test_batched_evaluation_switches_equal_sequential_runs(golden, dev, flow):
    """The reference's remaining eval_by_word switches inside trials.eval_by_word_batched -- weights_init='random',
    buffer_empty=False, RMSprop / SGD -- per trial identical to harness.eval_by_word with the same draws: ser_by_word, weights,
    optimizer state, step counts."""
    kw = dict(SWITCH_FLOWS[flow])
    opt = kw.pop("optimizer_type", "Adam")
    R, N, K, nsym, sub = (3, 24, 120, 2, 25) if opt != "Adam" else (6, 46, 120, 2, 25)
    T = K + 8 * nsym
    snrs = [8.0 + r for r in range(R)]
    w = _trial_weights(golden, R, seed=5)
    msg, rx = _words(dev, R, N, K, nsym, snrs, seed=19)
    ib = kw.pop("initial_buffer", None)
    if ib:  # W0 words from the (training) channel: transmitted codewords and received words, shared by the trials
        m0, r0 = _words(dev, 1, ib, K, nsym, [10.0], seed=77)
        ib = (mvn.rs_encode(m0[0], nsym), r0[0])
    bank = TrialBank(w, 16, 4, dev, optimizer_type=opt)
    rec = {}
    ser_b = eval_by_word_batched(bank, msg, rx, nsym, sub, [TrialDraws(300 + r, dev) for r in range(R)], record=rec,
                                 initial_buffer=ib, **kw)
    for r in range(R):
        det = _vnet_with(w[r], T, dev)
        tr = mvn.OnlineTrainer(det, 4, optimizer_type=opt)
        ser = mvn.eval_by_word(det, msg[r], rx[r], snrs[r], 0.2, nsym, sub, online_trainer=tr,
                               meta_detector=mvn.META_VNETDetector(16, {"train": T, "val": T}), draws=TrialDraws(300 + r, dev),
                               initial_buffer=ib, **kw)
        assert np.array_equal(ser, ser_b[r]), (flow, r)
        for a, b in zip(det.parameters(), bank.weights(r)):
            assert torch.equal(a.detach(), b), (flow, r)
        assert torch.equal(tr.exp_avg, bank.exp_avg[r]) and torch.equal(tr.exp_avg_sq, bank.exp_avg_sq[r]), (flow, r)
        assert tr.step == int(bank.step[r]), (flow, r)
    assert int(rec["trained"].sum()) > R
    if kw.get("online_meta"):
        assert int(rec["meta"].sum()) >= R
    if kw.get("weights_init") == "random":  # every update restarted the step count: far fewer steps than the blocks trained
        assert int(bank.step.max()) <= 8 * 5 + 2 * 4


@pytest.mark.timeout(900)
def test_many_trials_take_the_one_workgroup_per_trial_form_by_themselves(golden, dev, monkeypatch):
    """The launcher's OWN choice at the trial counts bench.py quotes (256 per GPU): from ~154 training trials on, one
    workgroup per trial -- online_train_kernel<16, true> on whole words and maml_train_kernel<16, true> -- instead of one per
    chunk and trial.  176 trials x 12 blocks of the configs[4] flow (short iteration counts), no switch set: the kernel the
    library names for the steps' active-trial counts is that form, and every trial is bit-identical to its sequential run."""
    R, N, K, nsym, sub = 176, 12, 120, 2, 25
    T = K + 8 * nsym
    flow = "meta_viterbinet_short"
    kw = MANY_TRIALS_FLOWS[flow]
    monkeypatch.delenv("MVN_TRAIN_GROUPS", raising=False)
    snrs = [10.0 + 0.0125 * r for r in range(R)]
    w = _trial_weights(golden, R, seed=8, spread=0.02)
    msg, rx = _words(dev, R, N, K, nsym, snrs, seed=13)
    bank = TrialBank(w, 16, 4, dev)
    rec = {}
    ser_b = eval_by_word_batched(bank, msg, rx, nsym, sub, [TrialDraws(500 + r, dev) for r in range(R)], record=rec, **kw)
    ws_bytes = int(mvn._lib.load().mvn_vnet_train_trials_workspace_bytes(16, T, 1, R))
    kernel = lambda kind, active, W: " ".join(_train_kernel_name(kind, int(active), T, W, 16, ws_bytes).split(" ")[:2])  # noqa: E731
    online_names = {kernel(0, a, 0) for a in rec["trained"].sum(axis=0) if a}  # per block: the trials that trained together
    meta_names = {kernel(2, a, 1) for a in rec["meta"].sum(axis=0) if a}
    assert "online_train_kernel<16, true>" in online_names, online_names
    assert "maml_train_kernel<16, true>" in meta_names, meta_names
    seq = _sequential_runs(golden, dev, flow, R, N, K, nsym, sub, snrs, w, msg, rx, 500)
    for r in range(R):
        ser, wr, m, v, step = seq[r]
        assert np.array_equal(ser, ser_b[r]), r
        for a, b in zip(wr, bank.weights(r)):
            assert torch.equal(a, b), r
        assert torch.equal(m, bank.exp_avg[r]) and torch.equal(v, bank.exp_avg_sq[r]) and step == int(bank.step[r]), r
    assert int(rec["meta"].sum()) > R and int(rec["trained"].sum()) > 5 * R


@pytest.fixture
def hooks_lib(monkeypatch):
    """The -DMVN_TEST_HOOKS build of the library (libmvn_hip_hooks.so: same sources + mvn_test_hooks) in place of the
    shipped one for the duration of a test."""
    import __graft_entry__ as ge

    lib = mvn._lib.load_variant(ge.build_hip_hooks())
    lib.mvn_test_hooks.restype, lib.mvn_test_hooks.argtypes = None, [ctypes.c_int64, ctypes.c_int32]
    mvn._lib.load()
    monkeypatch.setattr(mvn._lib, "_lib", lib)
    lib.mvn_reload_switches()
    yield lib
    lib.mvn_test_hooks(1 << 22, 0)


def test_abandoned_group_barrier_is_reported_not_silent(golden, dev, hooks_lib):
    """A training launch whose device-wide barrier cannot complete (forced here: every barrier waits for one workgroup that
    does not exist, with a short spin limit) must not hand back NaN weights with rc == 0 and nothing else: the status word
    is set, OnlineTrainer.check_status() / harness.eval_by_word / trials.eval_by_word_batched raise MvnError."""
    _hooks = hooks_lib.mvn_test_hooks
    g7 = golden("g7_by_word")
    w = [g7[f"w{i}"] for i in range(6)]
    T = 136
    gen = torch.Generator(device=dev).manual_seed(2)
    y = torch.randn(1, T, generator=gen, device=dev)
    tx = torch.randint(0, 2, (1, T), generator=gen, device=dev).float()
    try:
        _hooks(2000, 1)
        det = _vnet_with(w, T, dev)
        tr = mvn.OnlineTrainer(det, 4)
        tr.online_training(tx, y, iterations=3, full_word=True)  # one workgroup per chunk: 5 workgroups, barrier of "6"
        torch.cuda.synchronize()
        assert bool(torch.isnan(next(det.parameters())).all())
        with pytest.raises(mvn._lib.MvnError, match="barrier"):
            tr.check_status()
        tr.check_status()  # reported once, then clear
        det = _vnet_with(w, T, dev)
        tr = mvn.OnlineTrainer(det, 4)
        rxw = torch.randn(6, T, generator=gen, device=dev)
        txw = torch.randint(0, 2, (6, T), generator=gen, device=dev).float()
        tr.maml_training(rxw, txw, torch.tensor([[0], [1]], device=dev), torch.tensor([1, 2], device=dev), 0.1, True)
        with pytest.raises(mvn._lib.MvnError, match="barrier"):
            tr.check_status()
        # the batched evaluation reads the trials' status words at its per-step sync
        msg, rx = _words(dev, 3, 8, 120, 2, [12.0] * 3, seed=4)
        bank = TrialBank([w] * 3, 16, 4, dev)
        with pytest.raises(mvn._lib.MvnError, match="barrier"):
            eval_by_word_batched(bank, msg, rx, 2, 25, [TrialDraws(r, dev) for r in range(3)], self_supervised=True,
                                 self_supervised_iterations=2, meta_style_online_training=True)
    finally:
        _hooks(1 << 22, 0)
    # and with the hook off the same calls succeed
    det = _vnet_with(w, T, dev)
    tr = mvn.OnlineTrainer(det, 4)
    tr.online_training(tx, y, iterations=3, full_word=True)
    tr.check_status()
    assert bool(torch.isfinite(next(det.parameters())).all())


def test_abandoned_one_xcd_tag_barrier_is_reported(golden, dev, hooks_lib, monkeypatch):
    """The give-up path of the barrier a trial's workgroups use when they share an XCD (tags in its L2, train_groups.inc): one
    workgroup of the trial never publishes its tag (test hook), the others abandon the wait after the lowered spin limit -- status
    word set, NaN weights, MvnError from check_status -- exactly as for the device-wide counter barrier."""
    g7 = golden("g7_by_word")
    w = [g7[f"w{i}"] for i in range(6)]
    T = 136
    gen = torch.Generator(device=dev).manual_seed(5)
    y = torch.randn(1, T, generator=gen, device=dev)
    tx = torch.randint(0, 2, (1, T), generator=gen, device=dev).float()
    monkeypatch.setenv("MVN_TRAIN_XCD", "1")  # a single trial's chunk workgroups on one XCD
    mvn._lib.reload_switches()
    name = ctypes.create_string_buffer(128)
    assert hooks_lib.mvn_vnet_train_kernel_name(0, 0, T, 0, 16, 1 << 22, name, 128) == 0 and b"one XCD per trial" in name.value, name.value
    hooks_lib.mvn_test_hooks_skip_tag.argtypes, hooks_lib.mvn_test_hooks_skip_tag.restype = [ctypes.c_int32], None
    try:
        hooks_lib.mvn_test_hooks(4000, -1)
        hooks_lib.mvn_test_hooks_skip_tag(2)
        det = _vnet_with(w, T, dev)
        tr = mvn.OnlineTrainer(det, 4)
        tr.online_training(tx, y, iterations=3, full_word=True)
        torch.cuda.synchronize()
        assert bool(torch.isnan(next(det.parameters())).all())
        with pytest.raises(mvn._lib.MvnError, match="barrier"):
            tr.check_status()
    finally:
        hooks_lib.mvn_test_hooks_skip_tag(-1)
        hooks_lib.mvn_test_hooks(1 << 22, 0)
    det = _vnet_with(w, T, dev)
    tr = mvn.OnlineTrainer(det, 4)
    tr.online_training(tx, y, iterations=3, full_word=True)
    tr.check_status()
    assert bool(torch.isfinite(next(det.parameters())).all())


def test_two_training_launches_in_flight(golden, dev):
    """Two one-workgroup-per-chunk training calls in flight at once on two streams (each with its own workspace and barrier
    counter; 5 + 5 workgroups on a 256-CU device): results identical to running them one after the other."""
    g7 = golden("g7_by_word")
    w = [g7[f"w{i}"] for i in range(6)]
    T = 136
    gen = torch.Generator(device=dev).manual_seed(6)
    y = torch.randn(2, T, generator=gen, device=dev)
    tx = torch.randint(0, 2, (2, T), generator=gen, device=dev).float()
    rxw = torch.randn(6, T, generator=gen, device=dev)
    txw = torch.randint(0, 2, (6, T), generator=gen, device=dev).float()
    sup, qry = torch.arange(40, device=dev).reshape(-1, 1) % 6, (torch.arange(40, device=dev) + 1) % 6

    def run(concurrent):
        dets = [_vnet_with(w, T, dev) for _ in range(2)]
        trs = [mvn.OnlineTrainer(d, 4) for d in dets]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)] if concurrent else [torch.cuda.current_stream(dev)] * 2
        for rep in range(3):
            with torch.cuda.stream(streams[0]):
                trs[0].online_training(tx[:1], y[:1], iterations=150, full_word=True)
            with torch.cuda.stream(streams[1]):
                trs[1].maml_training(rxw, txw, sup, qry, 0.1, True)
        torch.cuda.synchronize()
        for t in trs:
            t.check_status()
        return [p.detach().clone() for d in dets for p in d.parameters()] + [t.exp_avg.clone() for t in trs]

    serial, overlapped = run(False), run(True)
    for a, b in zip(serial, overlapped):
        assert torch.equal(a, b)
    assert all(bool(torch.isfinite(a).all()) for a in serial)


@pytest.mark.parametrize("T,n_streams", [(136, 8), (544, 2), (544, 9)])
def test_single_trial_launches_on_many_streams_do_not_starve_one_xcd(golden, dev, T, n_streams):
    """Single-trial chunked training calls in flight on several streams: 8 x 5 workgroups (T = 136) would not fit the ONE
    XCD (32 CUs) an XCD-aware grid starts on if every launch took the same slot; two (or nine) launches of 17 workgroups
    (T = 544) could each hold a part of it and wait for each other until the spin limit.  The launcher rotates the slot from
    launch to launch and keeps the (groups, trials) grid above 8 workgroups per trial: every call returns finite weights with
    status 0, identical to the serial run."""
    g7 = golden("g7_by_word")
    w = [g7[f"w{i}"] for i in range(6)]
    gen = torch.Generator(device=dev).manual_seed(T)
    y = torch.randn(n_streams, T, generator=gen, device=dev)
    tx = torch.randint(0, 2, (n_streams, T), generator=gen, device=dev).float()

    def run(concurrent):
        dets = [_vnet_with(w, T, dev) for _ in range(n_streams)]
        trs = [mvn.OnlineTrainer(d, 4) for d in dets]
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(dev) if concurrent else cur for _ in range(n_streams)]
        for rep in range(2):
            for k in range(n_streams):
                with torch.cuda.stream(streams[k]):
                    trs[k].online_training(tx[k:k + 1], y[k:k + 1], iterations=120, full_word=True)
        torch.cuda.synchronize()
        for t in trs:
            t.check_status()
        return [p.detach().clone() for d in dets for p in d.parameters()]

    serial, overlapped = run(False), run(True)
    assert all(bool(torch.isfinite(a).all()) for a in overlapped)
    for a, b in zip(serial, overlapped):
        assert torch.equal(a, b)


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("S,T", [(4, 100), (32, 136), (8, 40)])
def test_trial_entry_points_for_other_state_counts(dev, monkeypatch, S, T, form):
    """mvn_vnet_online_train_trials_f32 / mvn_vnet_maml_train_trials_f32 through hand-filled descriptors (what a C caller
    does) for S != 16 -- the run-time-S instantiations of the trial kernels -- with trials of different iteration counts,
    one inactive trial, separate input / output / second-copy weights: per trial bit-identical to mvn.OnlineTrainer (the
    single-trial entry points), minibatch and full-word iterations, second-order meta-learning steps."""
    from meta_viterbinet_amd import trials as tr_mod

    for name, value in FORMS[form].items():
        monkeypatch.setenv(name, value)
    lib, L, R = mvn._lib.load(), int(np.log2(S)), 4
    rng = np.random.RandomState(S)
    gen = torch.Generator(device=dev).manual_seed(S)

    def rand_w():
        return [(rng.uniform(-1, 1, (100, 1))).astype(np.float32), rng.uniform(-1, 1, 100).astype(np.float32),
                rng.uniform(-0.1, 0.1, (50, 100)).astype(np.float32), rng.uniform(-0.1, 0.1, 50).astype(np.float32),
                rng.uniform(-0.14, 0.14, (S, 50)).astype(np.float32), rng.uniform(-0.14, 0.14, S).astype(np.float32)]

    ws = [rand_w() for _ in range(R)]
    rxw = torch.randn(R, 6, T, generator=gen, device=dev)
    txw = torch.randint(0, 2, (R, 6, T), generator=gen, device=dev).float()
    labels = torch.stack([mvn.calculate_states(L, txw[r]).reshape(6, T) for r in range(R)]).to(torch.int32).contiguous()
    n_iter = [7, 0, 12, 5]  # trial 1 is skipped
    bidx = torch.stack([torch.stack([torch.randperm(T - 1, generator=gen, device=dev)[:32] + 1 for _ in range(12)]) for _ in range(R)]).to(torch.int32)
    n_steps = [3, 5, 0, 4]
    sup = torch.randint(0, 6, (R, 5, 1), generator=gen, device=dev).to(torch.int32)
    qry = torch.randint(0, 6, (R, 5), generator=gen, device=dev).to(torch.int32)
    for mode in ("minibatch", "full_word", "maml"):
        bank = tr_mod.TrialBank(ws, S, L, dev)
        bank.exp_avg.normal_(0, 1e-3, generator=gen)  # a carried-over optimizer state, different per trial
        bank.exp_avg_sq.uniform_(1e-8, 1e-5, generator=gen)
        bank.step[:] = [0, 3, 200, 41]
        m0, v0, step0 = bank.exp_avg.clone(), bank.exp_avg_sq.clone(), bank.step.copy()
        out2 = torch.zeros_like(bank.theta)
        d = np.zeros(R, dtype=tr_mod.TRIAL_DTYPE)
        th, sv = bank.pointers(bank.theta), bank.pointers(bank.saved)
        o2 = bank.pointers(out2)
        status = torch.zeros(R, dtype=torch.int32, device=dev)
        for r in range(R):
            n = n_steps[r] if mode == "maml" else n_iter[r]
            d[r]["y"] = rxw[r].data_ptr() if mode == "maml" else rxw[r, 2].data_ptr()
            d[r]["labels"] = labels[r].data_ptr() if mode == "maml" else labels[r, 2].data_ptr()
            d[r]["idx"] = sup[r].data_ptr() if mode == "maml" else (bidx[r].data_ptr() if mode == "minibatch" else 0)
            d[r]["query_idx"] = qry[r].data_ptr() if mode == "maml" else 0
            d[r]["w_in"], d[r]["w_out"], d[r]["w_out2"] = sv[r], th[r], o2[r]  # read the saved copy, write two others
            d[r]["adam_m"], d[r]["adam_v"] = bank.exp_avg[r].data_ptr(), bank.exp_avg_sq[r].data_ptr()
            d[r]["status"] = status[r].data_ptr()
            d[r]["b1pow"], d[r]["b2pow"] = tr_mod.beta_power(0.9, step0[r]), tr_mod.beta_power(0.999, step0[r])
            d[r]["n"] = n
        dd = torch.from_numpy(d.view(np.uint8)).to(dev)
        nb = int(lib.mvn_vnet_train_trials_workspace_bytes(S, T, 1, R))
        wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        if mode != "minibatch":  # the run-time-S instantiations, in the form asked for
            named = _train_kernel_name(2 if mode == "maml" else 0, R, T, 1 if mode == "maml" else 0, S, nb)
            # (the two-trials-per-CU form exists for 16 states only: asked for here, it must not be taken)
            assert ("_groups_kernel<0, true>" if form.startswith("chunked") else "_kernel<0, true> 1x") in named, named
            assert ("one XCD per trial" in named) == (form == "chunked"), named
        if mode == "maml":
            rc = lib.mvn_vnet_maml_train_trials_f32(mvn._lib.ptr(dd), R, T, 1, 0.1, 1, 1e-3, 0.9, 0.999, 1e-8, S, mvn._lib.ptr(wsb), nb,
                                                    mvn._lib.current_stream(dev))
        else:
            rc = lib.mvn_vnet_online_train_trials_f32(mvn._lib.ptr(dd), R, T, 32 if mode == "minibatch" else 0, 1e-3, 0.9, 0.999, 1e-8, S,
                                                      mvn._lib.ptr(wsb), nb, mvn._lib.current_stream(dev))
        assert rc == 0
        torch.cuda.synchronize()
        assert int(status.abs().sum()) == 0
        for r in range(R):
            n = n_steps[r] if mode == "maml" else n_iter[r]
            if n == 0:  # untouched: weights, moments, second copy
                assert torch.equal(bank.theta[r], bank.saved[r]) and torch.equal(bank.exp_avg[r], m0[r]) and not bool(out2[r].any())
                continue
            det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
            with torch.no_grad():
                for p, a in zip(det.parameters(), ws[r]):
                    p.copy_(torch.tensor(a))
            one = mvn.OnlineTrainer(det, L)
            one.exp_avg.copy_(m0[r])
            one.exp_avg_sq.copy_(v0[r])
            one.step = int(step0[r])
            if mode == "maml":
                one.maml_training(rxw[r], txw[r], sup[r, :n].long(), qry[r, :n].long(), 0.1, True)
            else:
                one.online_training(txw[r, 2:3], rxw[r, 2:3], iterations=n, batch_idx=bidx[r, :n] if mode == "minibatch" else None,
                                    full_word=mode == "full_word")
            flat = torch.cat([p.detach().reshape(-1) for p in det.parameters()])
            assert torch.equal(flat, bank.theta[r]) and torch.equal(flat, out2[r]), (mode, r)
            assert torch.equal(one.exp_avg, bank.exp_avg[r]) and torch.equal(one.exp_avg_sq, bank.exp_avg_sq[r]), (mode, r)
            assert torch.equal(bank.saved[r], torch.cat([torch.tensor(a).reshape(-1) for a in ws[r]]).to(dev))  # the input copy is not written
