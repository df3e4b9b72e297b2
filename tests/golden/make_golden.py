#!/usr/bin/env python3
"""Golden-vector generator: imports the UNMODIFIED reference (tomerraviv95/meta-viterbinet,
mounted read-only at /root/reference) on CPU and writes small input/output fixtures to
tests/golden/*.npz.  Run in the build container only:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

The reference never travels to the GPU box; the .npz files (data only: inputs, weights as raw
arrays, expected outputs) do.  Fixture ids follow SURVEY.md section 8c (G1..G7).
"""
import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = os.environ.get("MVN_REFERENCE", "/root/reference")
if REF not in sys.path:
    sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TMP = tempfile.mkdtemp(prefix="mvn_golden_")

from python_code.utils.trellis_utils import create_transition_table, acs_block, calculate_states  # noqa: E402
from python_code.utils.metrics import calculate_error_rates  # noqa: E402
from python_code.detectors.VA.va_detector import VADetector  # noqa: E402
from python_code.detectors.VNET.vnet_detector import VNETDetector  # noqa: E402
from python_code.detectors.META_VNET.meta_vnet_detector import META_VNETDetector  # noqa: E402
import python_code.channel.channel_estimation as chest  # noqa: E402
from python_code.trainers.VA.va_trainer import VATrainer  # noqa: E402
from python_code.trainers.VNET.vnet_trainer import VNETTrainer  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def patch_cost2100():
    """the loader wants combined_h_{i}.mat, the repo ships h_{i}.mat."""
    d = os.path.join(TMP, "cost2100")
    os.makedirs(d, exist_ok=True)
    for i in range(4):
        dst = os.path.join(d, f"combined_h_{i}.mat")
        if not os.path.exists(dst):
            os.symlink(os.path.join(REF, "resources", "cost2100_channel", f"h_{i}.mat"), dst)
    chest.COST2100_DIR = d


# ----------------------------------------------------------------------------- G1
def g1_acs():
    out = {}
    rng = np.random.RandomState(11)
    for S in (2, 4, 8, 16, 256):
        table = create_transition_table(S)
        out[f"table_S{S}"] = table.astype(np.int64)
        tt = torch.Tensor(table)
        for B in (1, 3, 64):
            ip = rng.normal(0, 3, (B, S)).astype(np.float32)
            c = rng.normal(0, 3, (B, S)).astype(np.float32)
            if B == 3:  # force exact ties and a few repeated values
                ip[1] = 0.0
                c[1] = 1.5
                c[2, ::2] = c[2, 1::2]
                ip[2, ::2] = ip[2, 1::2]
            v, j = acs_block(torch.tensor(ip), torch.tensor(c), tt, S)
            out[f"in_S{S}_B{B}"] = ip
            out[f"llr_S{S}_B{B}"] = c
            out[f"out_S{S}_B{B}"] = v.numpy()
            out[f"argj_S{S}_B{B}"] = j.numpy()
    save("g1_acs_block", **out)


# ----------------------------------------------------------------------------- G2
def hand_step_va(det, y, phase, snr, gamma, count=None):
    """Re-run VADetector.forward's own statements to expose costs and final metrics."""
    in_prob = torch.zeros([y.shape[0], det.n_states])
    priors = det.compute_likelihood_priors(y, snr, gamma, phase, count)
    dec = torch.zeros(y.shape)
    for i in range(det.transmission_length):
        dec[:, i] = torch.argmin(in_prob, dim=1) % 2
        in_prob, _ = acs_block(in_prob, priors[:, i], det.transition_table, det.n_states)
    return priors, dec, in_prob


def g2_va():
    cases = [
        # name, L, frames, subframes, T, snr, coeffs, fading_ch, fading_dec, taps_type
        ("L4_static", 4, 1, 25, 256, 10, "time_decay", False, False, 1),
        ("L4_fading1", 4, 1, 25, 256, 10, "time_decay", True, True, 1),
        ("L4_fading2", 4, 2, 10, 128, 7, "time_decay", True, True, 2),
        ("L4_cost2100", 4, 1, 25, 136, 12, "cost2100", False, False, 1),
        ("L2_static", 2, 1, 12, 96, 10, "time_decay", False, False, 1),
        ("L3_static", 3, 1, 12, 96, 8, "time_decay", False, False, 1),
        ("L8_static", 8, 1, 4, 128, 10, "time_decay", False, False, 1),
        ("L4_config1", 4, 4, 25, 1000, 10, "time_decay", False, False, 1),
    ]
    out = {}
    names = []
    for (name, L, frames, sub, T, snr, coef, fch, fdec, ttype) in cases:
        tr = VATrainer(use_ecc=False, memory_length=L, val_block_length=T, val_frames=frames,
                       subframes_in_frame=sub, channel_coefficients=coef, fading_in_channel=fch,
                       fading_in_decoder=fdec, fading_taps_type=ttype, noisy_est_var=0,
                       val_SNR_start=snr, val_SNR_end=snr, gamma=0.2, weights_dir=TMP,
                       noise_seed=3450002, word_seed=7860002)
        tx, rx = tr.channel_dataset["val"].__getitem__(snr_list=[snr], gamma=0.2)
        det = tr.detector
        with torch.no_grad():
            dec = det(rx, "val", snr, 0.2)
            cost, dec2, final = hand_step_va(det, rx, "val", snr, 0.2)
        assert torch.equal(dec, dec2)
        W = frames * sub
        h = np.concatenate([chest.estimate_channel(L, 0.2, noisy_est_var=0, fading=fdec, index=i,
                                                   fading_taps_type=ttype, channel_coefficients=coef)
                            for i in range(W)], axis=0)
        priors = det.compute_state_priors(h).numpy()  # [S, W]
        ser, fer, err_idx = calculate_error_rates(dec[tr.data_indices], tx[tr.data_indices])
        ser_all, fer_all, _ = calculate_error_rates(dec, tx)
        names.append(name)
        big = name == "L4_config1"
        out[f"{name}_meta"] = np.array([L, frames, sub, T, snr, int(fdec), ttype], np.int64)
        out[f"{name}_coef"] = np.array(coef)
        out[f"{name}_tx"] = tx.numpy().astype(np.uint8)
        out[f"{name}_rx"] = rx.numpy()
        out[f"{name}_h"] = h
        out[f"{name}_state_priors"] = priors
        out[f"{name}_cost_head"] = cost[: (2 if big else 4), :8].numpy()
        out[f"{name}_decoded"] = dec.numpy().astype(np.uint8)
        out[f"{name}_final"] = final.numpy()
        out[f"{name}_data_indices"] = tr.data_indices.numpy()
        out[f"{name}_rates"] = np.array([ser, fer, ser_all, fer_all], np.float64)
        out[f"{name}_err_idx"] = err_idx.numpy()
        # by-word call (count given): one word, its own channel row (trainer.py:295)
        if name in ("L4_fading1", "L4_cost2100"):
            cnt = 7
            with torch.no_grad():
                dw = det(rx[cnt].reshape(1, -1), "val", snr, 0.2, cnt)
            out[f"{name}_count{cnt}_decoded"] = dw.numpy().astype(np.uint8)
        print(name, "ser", ser, "fer", fer)
    out["names"] = np.array(names)
    save("g2_va", **out)


# ----------------------------------------------------------------------------- G3/G4
def export_weights(det):
    return [p.detach().numpy().copy() for p in det.parameters()]


def train_vnet(L, snr):
    """Briefly train the reference ViterbiNet with the reference's own trainer (unseeded
    internals, SURVEY 8c caveat): only the resulting weights are kept, as raw arrays."""
    torch.manual_seed(1234)
    tr = VNETTrainer(use_ecc=False, memory_length=L, val_block_length=120, val_frames=1,
                     subframes_in_frame=25, train_block_length=120, train_frames=4,
                     train_minibatch_num=12, channel_coefficients="time_decay",
                     fading_in_channel=False, fading_in_decoder=False, noisy_est_var=0,
                     train_SNR_start=snr, train_SNR_end=snr, val_SNR_start=snr, val_SNR_end=snr,
                     gamma=0.2, weights_dir=os.path.join(TMP, f"w_L{L}"), self_supervised=False,
                     online_meta=False, eval_mode="aggregated", noise_seed=3450002, word_seed=7860002)
    os.makedirs(tr.weights_dir, exist_ok=True)
    tr.train()
    ck = torch.load(os.path.join(tr.weights_dir, f"snr_{snr}_gamma_0.2.pt"))
    tr.detector.load_state_dict(ck["model_state_dict"])
    return tr


def margins(logits, T):
    """per-(b,t) gap between the two smallest DISTINCT path metrics before stage t."""
    B, _, S = logits.shape
    inp = torch.zeros(B, S)
    tt = torch.Tensor(create_transition_table(S))
    out = np.zeros((B, T), np.float32)
    for i in range(T):
        u = torch.sort(inp[:, : max(S // 2, 1)], dim=1).values if i > 0 else inp
        out[:, i] = (u[:, 1] - u[:, 0]).numpy() if u.shape[1] > 1 else 0
        inp, _ = acs_block(inp, -torch.tensor(logits[:, i]), tt, S)
    return out


def g3_g4_vnet():
    out = {}
    names = []
    torch.set_num_threads(1)
    trained = {}
    for L in (4, 2):
        trained[L] = train_vnet(L, 10)
    cases = [
        # name, S, B, T, weights-source, threads
        ("S16_init_exact", 16, 8, 64, "init", 1),
        ("S16_trained_exact", 16, 25, 120, "trained", 1),
        ("S16_trained_mt", 16, 25, 120, "trained", 8),
        ("S16_trained_odd", 16, 7, 45, "trained", 8),
        ("S4_trained_exact", 4, 8, 120, "trained", 1),
        ("S256_init_exact", 256, 2, 64, "init", 1),
        ("S2_init_exact", 2, 4, 32, "init", 1),
    ]
    for (name, S, B, T, src, nthreads) in cases:
        L = int(np.log2(S))
        torch.set_num_threads(nthreads)
        if src == "trained":
            tr = trained[L]
            det = VNETDetector(S, {"train": T, "val": T})
            det.load_state_dict(tr.detector.state_dict())
            # fresh channel draw through the reference's generator at this size
            t2 = VNETTrainer(use_ecc=False, memory_length=L, val_block_length=T, val_frames=1,
                             subframes_in_frame=B, channel_coefficients="time_decay",
                             fading_in_channel=False, fading_in_decoder=False, noisy_est_var=0,
                             val_SNR_start=10, val_SNR_end=10, gamma=0.2, weights_dir=TMP,
                             noise_seed=3450002, word_seed=7860002)
            tx, y = t2.channel_dataset["val"].__getitem__(snr_list=[10], gamma=0.2)
        else:
            torch.manual_seed(100 + S)
            det = VNETDetector(S, {"train": T, "val": T})
            g = torch.Generator().manual_seed(7 + S)
            y = torch.randn(B, T, generator=g) * 1.5
            tx = torch.zeros(B, T)
        with torch.no_grad():
            logits = det(y, "train")
            dec = det(y, "val")
            meta = META_VNETDetector(S, {"train": T, "val": T})
            var = list(det.parameters())
            dec_meta = meta(y, "val", var)
            logits_meta = meta(y, "train", var)
        W = export_weights(det)
        names.append(name)
        out[f"{name}_meta"] = np.array([S, B, T, nthreads], np.int64)
        for i, w in enumerate(W):
            out[f"{name}_w{i}"] = w
        out[f"{name}_y"] = y.numpy()
        out[f"{name}_tx"] = tx.numpy().astype(np.uint8)
        out[f"{name}_logits"] = logits.numpy()
        out[f"{name}_decoded"] = dec.numpy().astype(np.uint8)
        out[f"{name}_margin"] = margins(logits.numpy(), T)
        out[f"{name}_meta_equal"] = np.array([bool(torch.equal(dec, dec_meta)),
                                              bool(torch.equal(logits, logits_meta))])
        print(name, "meta==vnet", out[f"{name}_meta_equal"], "state_dict", list(det.state_dict().keys()))
    out["names"] = np.array(names)
    out["state_dict_keys"] = np.array(list(det.state_dict().keys()))
    torch.set_num_threads(1)
    save("g3_vnet", **out)
    return trained[4]


# ----------------------------------------------------------------------------- G5
def g5_kats():
    out = {}
    out["states_kat_in"] = np.array([[1, 0, 1, 1, 0, 0, 1]], np.float32)
    out["states_kat_out"] = calculate_states(4, torch.tensor(out["states_kat_in"])).numpy()
    rng = np.random.RandomState(5)
    for L in (2, 3, 4, 8):
        w = rng.randint(0, 2, (3, 40)).astype(np.float32)
        out[f"states_L{L}_in"] = w
        out[f"states_L{L}_out"] = calculate_states(L, torch.tensor(w)).numpy()
    pred = rng.randint(0, 2, (12, 30)).astype(np.float32)
    tgt = pred.copy()
    tgt[2, 5] = 1 - tgt[2, 5]
    tgt[2, 9] = 1 - tgt[2, 9]
    tgt[7, 0] = 1 - tgt[7, 0]
    tgt[11] = 1 - tgt[11]
    ser, fer, idx = calculate_error_rates(torch.tensor(pred), torch.tensor(tgt))
    out["er_pred"], out["er_tgt"] = pred, tgt
    out["er_rates"] = np.array([ser, fer], np.float64)
    out["er_idx"] = idx.numpy()
    ser0, fer0, idx0 = calculate_error_rates(torch.tensor(pred), torch.tensor(pred))
    out["er_rates_clean"] = np.array([ser0, fer0], np.float64)
    save("g5_kats", **out)


# ----------------------------------------------------------------------------- G6
def g6_channels():
    out = {}
    out["cost2100"] = np.concatenate([chest.estimate_channel(4, 0.2, "cost2100", index=i) for i in range(300)])
    for ttype in (1, 2):
        out[f"time_decay_fading{ttype}"] = np.concatenate(
            [chest.estimate_channel(4, 0.2, "time_decay", fading=True, index=i, fading_taps_type=ttype)
             for i in range(300)])
    for L in (2, 3, 4, 8):
        out[f"time_decay_L{L}"] = chest.estimate_channel(L, 0.2, "time_decay")
    out["time_decay_gamma05"] = chest.estimate_channel(4, 0.5, "time_decay")
    save("g6_channels", **out)
    # the same [300,4] float64 COST2100 tap table, as the data file the package's channel.py reads
    data_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)), "meta-viterbinet_amd", "data")
    os.makedirs(data_dir, exist_ok=True)
    np.save(os.path.join(data_dir, "cost2100_taps.npy"), out["cost2100"])


# ----------------------------------------------------------------------------- G7
def g7_by_word(trained):
    """eval_by_word without online updates ("joint" variant): deterministic given weights."""
    wdir = os.path.join(TMP, "w_byword")
    os.makedirs(wdir, exist_ok=True)
    torch.save({"model_state_dict": trained.detector.state_dict(), "optimizer_state_dict": {}, "loss": 0.0},
               os.path.join(wdir, "snr_10_gamma_0.2.pt"))
    out = {}
    for coef in ("time_decay", "cost2100"):
        tr = VNETTrainer(eval_mode="by_word", use_ecc=True, n_symbols=2, memory_length=4,
                         val_block_length=120, val_frames=12, subframes_in_frame=25,
                         channel_coefficients=coef, fading_in_channel=(coef == "time_decay"),
                         fading_in_decoder=False, fading_taps_type=2, noisy_est_var=0,
                         self_supervised=False, online_meta=False, val_SNR_start=10, val_SNR_end=10,
                         gamma=0.2, weights_dir=wdir, noise_seed=3450002, word_seed=7860002)
        rec = {"y": [], "dec": [], "count": []}
        orig_forward = tr.detector.forward

        def spy(y, phase, snr=None, gamma=None, count=None, _f=orig_forward):
            r = _f(y, phase, snr, gamma, count)
            if phase == "val":
                rec["y"].append(y.detach().numpy().copy())
                rec["dec"].append(r.detach().numpy().copy())
                rec["count"].append(-1 if count is None else count)
            return r

        tr.detector.forward = spy
        ser_by_word = tr.evaluate()
        out[f"{coef}_ser_by_word"] = np.asarray(ser_by_word, np.float64)
        out[f"{coef}_y"] = np.concatenate(rec["y"]).astype(np.float32)
        out[f"{coef}_detected"] = np.concatenate(rec["dec"]).astype(np.uint8)
        out[f"{coef}_count"] = np.array(rec["count"], np.int64)
        out[f"{coef}_data_indices"] = tr.data_indices.numpy()
        print(coef, "mean ser_by_word", float(np.mean(ser_by_word)), "blocks", len(rec["y"]))
    for i, w in enumerate(export_weights(trained.detector)):
        out[f"w{i}"] = w
    save("g7_by_word", **out)


# ----------------------------------------------------------------------------- G9
def g9_by_word_va():
    """eval_by_word (trainer.py:267-354) end to end with the deterministic VA detector and the RS(17,15) outer
    code: per-block ser, transmitted/received/detected words."""
    out = {}
    for coef, fading in (("time_decay", True), ("cost2100", False)):
        tr = VATrainer(eval_mode="by_word", use_ecc=True, n_symbols=2, memory_length=4, val_block_length=120,
                       val_frames=12, subframes_in_frame=25, channel_coefficients=coef, fading_in_channel=fading,
                       fading_in_decoder=fading, fading_taps_type=2, noisy_est_var=0, self_supervised=False,
                       online_meta=False, val_SNR_start=9, val_SNR_end=9, gamma=0.2, weights_dir=TMP,
                       noise_seed=3450002, word_seed=7860002)
        rec = {"y": [], "dec": [], "count": []}
        orig_forward = tr.detector.forward

        def spy(y, phase, snr=None, gamma=None, count=None, _f=orig_forward):
            r = _f(y, phase, snr, gamma, count)
            rec["y"].append(y.detach().numpy().copy())
            rec["dec"].append(r.detach().numpy().copy())
            rec["count"].append(-1 if count is None else count)
            return r

        tr.detector.forward = spy
        # the trainer draws the words inside eval_by_word; replay the same draw first to capture tx
        t2 = VATrainer(eval_mode="by_word", use_ecc=True, n_symbols=2, memory_length=4, val_block_length=120,
                       val_frames=12, subframes_in_frame=25, channel_coefficients=coef, fading_in_channel=fading,
                       fading_in_decoder=fading, fading_taps_type=2, noisy_est_var=0, self_supervised=False,
                       online_meta=False, val_SNR_start=9, val_SNR_end=9, gamma=0.2, weights_dir=TMP,
                       noise_seed=3450002, word_seed=7860002)
        tx, rx = t2.channel_dataset["val"].__getitem__(snr_list=[9], gamma=0.2)
        ser_by_word = tr.evaluate()
        assert np.array_equal(np.concatenate(rec["y"]), rx.numpy())
        out[f"{coef}_ser_by_word"] = np.asarray(ser_by_word, np.float64)
        out[f"{coef}_tx"] = tx.numpy().astype(np.uint8)
        out[f"{coef}_y"] = rx.numpy()
        out[f"{coef}_detected"] = np.concatenate(rec["dec"]).astype(np.uint8)
        out[f"{coef}_count"] = np.array(rec["count"], np.int64)
        out[f"{coef}_data_indices"] = tr.data_indices.numpy()
        out[f"{coef}_meta"] = np.array([4, 12, 25, 136, 9, int(fading), 2, 2], np.int64)  # L,frames,sub,T,snr,fading,taps,nsym
        print(coef, "mean ser_by_word", float(np.mean(ser_by_word)))
    save("g9_by_word_va", **out)


# ----------------------------------------------------------------------------- G10
def g10_online_training():
    """VNETTrainer.online_training (vnet_trainer.py:49-60) for a few iterations, with the minibatch indices that
    select_batch drew (torch.multinomial is wrapped to record them): initial/final weights and per-iteration loss."""
    torch.manual_seed(77)
    torch.set_num_threads(1)
    tr = VNETTrainer(use_ecc=True, n_symbols=2, memory_length=4, val_block_length=120, val_frames=1,
                     subframes_in_frame=3, train_block_length=120, train_frames=1, train_minibatch_num=1,
                     channel_coefficients="time_decay", fading_in_channel=False, fading_in_decoder=False,
                     noisy_est_var=0, val_SNR_start=10, val_SNR_end=10, gamma=0.2, weights_dir=TMP,
                     self_supervised=True, self_supervised_iterations=25, eval_mode="by_word",
                     noise_seed=3450002, word_seed=7860002)
    tr.deep_learning_setup()
    from python_code.ecc.rs_main import encode

    tx_msg, rx = tr.channel_dataset["val"].__getitem__(snr_list=[10], gamma=0.2)
    tx = torch.Tensor(encode(tx_msg[1].int().numpy(), 2).reshape(1, -1))
    rxw = rx[1].reshape(1, -1)
    w0 = export_weights(tr.detector)
    rec, losses = [], []
    real_multinomial = torch.multinomial

    def spy(weights, n, *a, **k):
        r = real_multinomial(weights, n, *a, **k)
        rec.append(r.numpy().copy())
        return r

    real_loop = tr.run_train_loop

    def loop_spy(soft_estimation, transmitted_words):
        v = real_loop(soft_estimation=soft_estimation, transmitted_words=transmitted_words)
        losses.append(v)
        return v

    torch.multinomial = spy
    tr.run_train_loop = loop_spy
    try:
        tr.online_training(tx, rxw)
    finally:
        torch.multinomial = real_multinomial
    w1 = export_weights(tr.detector)
    out = {"tx": tx.numpy().astype(np.uint8), "rx": rxw.numpy(), "idx": np.array(rec, np.int32),
           "loss": np.array(losses, np.float64), "lr": np.array(tr.lr)}
    for i in range(6):
        out[f"w0_{i}"], out[f"w1_{i}"] = w0[i], w1[i]
    print("online training: iterations", len(rec), "loss", losses[0], "->", losses[-1])
    save("g10_online_training", **out)


# ----------------------------------------------------------------------------- G11
def g11_meta_train_loop():
    """Trainer.meta_train_loop (trainer.py:425-453) on a few buffered words: MAML and first-order variants."""
    from python_code.trainers.META_VNET.metavnet_trainer import METAVNETTrainer
    from python_code.ecc.rs_main import encode

    out = {}
    for maml in (True, False):
        torch.manual_seed(5)
        torch.set_num_threads(1)
        tr = METAVNETTrainer(use_ecc=True, n_symbols=2, memory_length=4, val_block_length=120, val_frames=1,
                             subframes_in_frame=6, train_block_length=120, train_frames=1, train_minibatch_num=1,
                             channel_coefficients="time_decay", fading_in_channel=True, fading_in_decoder=False,
                             fading_taps_type=2, noisy_est_var=0, val_SNR_start=10, val_SNR_end=10, gamma=0.2,
                             weights_dir=TMP, self_supervised=True, online_meta=True, MAML=maml, meta_lr=0.1,
                             window_size=1, eval_mode="by_word", noise_seed=3450002, word_seed=7860002)
        tr.deep_learning_setup()
        tx_msg, rx = tr.channel_dataset["val"].__getitem__(snr_list=[10], gamma=0.2)
        tx = torch.cat([torch.Tensor(encode(w.int().numpy(), 2).reshape(1, -1)) for w in tx_msg], dim=0)
        w0 = export_weights(tr.detector)
        losses = []
        pairs = [(0, 1), (2, 3), (4, 5), (1, 2)]
        for (sup, qry) in pairs:
            losses.append(float(tr.meta_train_loop(rx, tx, torch.tensor([sup]), torch.tensor([qry]))))
        tag = "maml" if maml else "fo"
        out[f"{tag}_loss"] = np.array(losses)
        for i, w in enumerate(export_weights(tr.detector)):
            out[f"{tag}_w1_{i}"] = w
        for i, w in enumerate(w0):
            out[f"{tag}_w0_{i}"] = w
        out["tx"], out["rx"] = tx.numpy().astype(np.uint8), rx.numpy()
        out["pairs"] = np.array(pairs, np.int64)
        out["lr"], out["meta_lr"] = np.array(tr.lr), np.array(tr.meta_lr)
        print("meta_train_loop", tag, losses)
    save("g11_meta_train_loop", **out)


# ----------------------------------------------------------------------------- G12, G13
ADAM_SNAPSHOT_BLOCKS = (1, 5, 6, 10, 25, 49)  # blocks at whose START the optimizer's state is recorded too (per_block runs)


def _by_word_flow(out, tag, cls, kw, g7, blocks_frames=3, weights=None, per_block=False):
    """One run of the unmodified reference's evaluate() -> eval_by_word (trainer.py:267-354, :368-381) from the reference-trained
    weights of G7 (saved as the checkpoint load_weights reads), with every random draw it makes recorded in call order:
    torch.multinomial (select_batch, :542), torch.randint (j_hat, :337) and -- weights_init='random' -- the weights
    initialize_detector() produces (:356-359).  buffer_empty=False: the words eval_by_word draws from the training channel
    (:282-286) are reproduced from a twin trainer with the same seeds (the two datasets share the trainer's RandomStates and
    eval_by_word draws 'val' first, then 'train')."""
    from python_code.ecc.rs_main import encode

    torch.manual_seed(2024)
    torch.set_num_threads(1)
    wdir = os.path.join(TMP, "w_g12_" + tag)
    os.makedirs(wdir, exist_ok=True)
    base = dict(eval_mode="by_word", use_ecc=True, n_symbols=2, memory_length=4, val_block_length=120, val_frames=blocks_frames,
                subframes_in_frame=25, channel_coefficients="time_decay", fading_in_channel=True, fading_in_decoder=False,
                fading_taps_type=2, noisy_est_var=0, val_SNR_start=9, val_SNR_end=9, gamma=0.2, weights_dir=wdir, buffer_empty=True,
                ser_thresh=0.02, noise_seed=3450002, word_seed=7860002)
    base.update(kw)
    tr = cls(**base)
    sd = tr.detector.state_dict()
    for i, k in enumerate(["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias"]):
        sd[k] = torch.tensor(g7[f"w{i}"] if weights is None else weights[i])
        if weights is not None:  # (another state count than G7's: the run's starting weights travel with it)
            out[f"{tag}_w0_{i}"] = np.asarray(weights[i], np.float32)
    torch.save({"model_state_dict": sd, "optimizer_state_dict": {}, "loss": 0.0}, os.path.join(wdir, "snr_9_gamma_0.2.pt"))
    tx_msg, rx = tr.channel_dataset["val"].__getitem__(snr_list=[9], gamma=0.2)  # same seeds -> the words evaluate() will draw
    if not base["buffer_empty"]:
        btx, brx = tr.channel_dataset["train"].__getitem__(snr_list=[9], gamma=0.2)
        out[f"{tag}_buffer_tx"] = np.stack([encode(w.int().numpy(), 2).reshape(-1) for w in btx]).astype(np.uint8)
        out[f"{tag}_buffer_rx"] = brx.numpy().astype(np.float32)
    tr2 = cls(**base)
    multinomials, randints, inits = [], [], []
    real_multinomial, real_randint, real_init = torch.multinomial, torch.randint, tr2.initialize_detector

    def mspy(weights, n, *a, **k):
        r = real_multinomial(weights, n, *a, **k)
        multinomials.append(r.numpy().copy())
        return r

    def rspy(*a, **k):
        r = real_randint(*a, **k)
        randints.append((int(k["high"] if "high" in k else (a[1] if len(a) > 1 else a[0])), r.numpy().copy()))
        return r

    def ispy():
        real_init()
        inits.append(np.concatenate([w.reshape(-1) for w in export_weights(tr2.detector)]))

    # per_block: the state the reference holds at the START of every block -- eval_by_word calls the detector with phase 'val'
    # exactly once per block, first thing (trainer.py:295) -- so that a replay can be re-synchronised block by block: the
    # detector's weights (every block), the saved detector's (when they changed: trainer.py:275/:343) and, at a few blocks, the
    # optimizer's exp_avg / exp_avg_sq / step.
    blk_w, blk_saved_at, blk_saved, blk_adam_at, blk_adam_m, blk_adam_v, blk_adam_step = [], [], [], [], [], [], []
    det_cls = type(tr2.detector)
    real_forward = det_cls.forward

    def flat(det):
        return np.concatenate([w.reshape(-1) for w in export_weights(det)]).astype(np.float32)

    def adam_state():
        opt = getattr(tr2, "optimizer", None)
        ps = list(tr2.detector.parameters())
        m = np.concatenate([(opt.state[p_]["exp_avg"].numpy() if opt is not None and p_ in opt.state and "exp_avg" in opt.state[p_]
                             else np.zeros(tuple(p_.shape), np.float32)).reshape(-1) for p_ in ps])
        v = np.concatenate([(opt.state[p_]["exp_avg_sq"].numpy() if opt is not None and p_ in opt.state and "exp_avg_sq" in opt.state[p_]
                             else np.zeros(tuple(p_.shape), np.float32)).reshape(-1) for p_ in ps])
        step = int(opt.state[ps[0]]["step"]) if opt is not None and ps[0] in opt.state and "step" in opt.state[ps[0]] else 0
        return m.astype(np.float32), v.astype(np.float32), step

    def fspy(self_, y, phase, *a, **k):
        if per_block and phase == "val" and self_ is tr2.detector:
            count = len(blk_w)
            blk_w.append(flat(tr2.detector))
            sd_ = getattr(tr2, "saved_detector", None)
            if sd_ is not None:
                fs = flat(sd_)
                if not blk_saved or not np.array_equal(fs, blk_saved[-1]):
                    blk_saved_at.append(count)
                    blk_saved.append(fs)
            if count in ADAM_SNAPSHOT_BLOCKS:
                m_, v_, st_ = adam_state()
                blk_adam_at.append(count)
                blk_adam_m.append(m_)
                blk_adam_v.append(v_)
                blk_adam_step.append(st_)
        return real_forward(self_, y, phase, *a, **k)

    torch.multinomial, torch.randint, tr2.initialize_detector = mspy, rspy, ispy
    det_cls.forward = fspy
    try:
        ser_by_word = tr2.evaluate()
    finally:
        torch.multinomial, torch.randint = real_multinomial, real_randint
        det_cls.forward = real_forward
    if per_block:
        assert len(blk_w) == len(ser_by_word)
        m_, v_, st_ = adam_state()  # ... and the optimizer's state the run ends with
        out[f"{tag}_blk_w"] = np.stack(blk_w)
        out[f"{tag}_blk_saved_at"] = np.array(blk_saved_at, np.int64)
        out[f"{tag}_blk_saved"] = np.stack(blk_saved) if blk_saved else np.zeros((0, 0), np.float32)
        out[f"{tag}_blk_adam_at"] = np.array(blk_adam_at + [len(ser_by_word)], np.int64)
        out[f"{tag}_blk_adam_m"] = np.stack(blk_adam_m + [m_])
        out[f"{tag}_blk_adam_v"] = np.stack(blk_adam_v + [v_])
        out[f"{tag}_blk_adam_step"] = np.array(blk_adam_step + [st_], np.int64)
    out[f"{tag}_tx"] = tx_msg.numpy().astype(np.uint8)
    out[f"{tag}_rx"] = rx.numpy().astype(np.float32)
    out[f"{tag}_ser_by_word"] = np.asarray(ser_by_word, np.float64)
    out[f"{tag}_multinomial"] = np.array(multinomials, np.uint8 if rx.shape[1] <= 256 else np.int32).reshape(len(multinomials), -1 if multinomials else 0)
    out[f"{tag}_randint_high"] = np.array([h for h, _ in randints], np.int64)
    out[f"{tag}_randint"] = np.array([v for _, v in randints], np.int64).reshape(len(randints), -1 if randints else 0)
    if inits:
        out[f"{tag}_init_weights"] = np.stack(inits).astype(np.float32)
    for i, w in enumerate(export_weights(tr2.detector)):
        out[f"{tag}_w1_{i}"] = w
    if base.get("online_meta"):
        for i, w in enumerate(export_weights(tr2.saved_detector)):
            out[f"{tag}_saved_{i}"] = w
    out[f"{tag}_meta"] = np.array([kw["self_supervised_iterations"], kw.get("meta_train_iterations", 0), kw.get("meta_j_num", 0),
                                   kw.get("meta_subframes", 0), 9, 25, 2], np.int64)
    print("g12/13", tag, "blocks", len(ser_by_word), "mean ser", float(np.mean(ser_by_word)), "multinomial draws", len(multinomials),
          "randint draws", len(randints), "re-initialisations", len(inits))


def g12_by_word_with_updates():
    """Trainer.eval_by_word WITH its update branches: (a) VNETTrainer, self-supervised minibatch training after every qualifying
    block (vnet_trainer.py:49-60); (b) METAVNETTrainer, online meta-learning every 5 blocks + whole-word training from the saved
    weights (metavnet_trainer.py:52-64).  Stored: words, draws in call order, ser_by_word, final (and saved) weights."""
    from python_code.trainers.META_VNET.metavnet_trainer import METAVNETTrainer

    g7 = np.load(os.path.join(HERE, "g7_by_word.npz"))
    out = {}
    _by_word_flow(out, "selfsup", VNETTrainer, dict(self_supervised=True, self_supervised_iterations=12, online_meta=False), g7)
    _by_word_flow(out, "meta", METAVNETTrainer, dict(self_supervised=True, self_supervised_iterations=8, online_meta=True, MAML=True,
                                                     meta_lr=0.1, window_size=1, meta_train_iterations=2, meta_j_num=4,
                                                     meta_subframes=5, weights_init="last_frame"), g7)
    save("g12_by_word_with_updates", **out)


def g13_by_word_switches():
    """The reference's remaining eval_by_word switches, one 50-block run each (same recording as G12): first-order meta-learning
    (MAML=False, trainer.py:441-447), the pre-filled fixed-length buffer (buffer_empty=False, :278-286, :325-328), the three
    meta_weights_init modes (:356-366: 'random' re-initialises the detector and the optimizer, 'meta_training' reloads the
    checkpoint), a window of two support words, and the RMSprop / SGD optimizers (:163-175)."""
    from python_code.trainers.META_VNET.metavnet_trainer import METAVNETTrainer

    g7 = np.load(os.path.join(HERE, "g7_by_word.npz"))
    out = {}
    meta = dict(self_supervised=True, self_supervised_iterations=6, online_meta=True, MAML=True, meta_lr=0.1, window_size=1,
                meta_train_iterations=2, meta_j_num=4, meta_subframes=5, weights_init="last_frame")
    _by_word_flow(out, "fomaml", METAVNETTrainer, dict(meta, MAML=False), g7, 2)
    _by_word_flow(out, "window", METAVNETTrainer, dict(meta, buffer_empty=False, train_block_length=120, train_frames=1), g7, 2)
    _by_word_flow(out, "random", METAVNETTrainer, dict(meta, weights_init="random", meta_train_iterations=3), g7, 2)
    _by_word_flow(out, "metatrain", METAVNETTrainer, dict(meta, weights_init="meta_training"), g7, 2)
    _by_word_flow(out, "support2", METAVNETTrainer, dict(meta, window_size=2), g7, 2)
    _by_word_flow(out, "rmsprop", VNETTrainer, dict(self_supervised=True, self_supervised_iterations=6, online_meta=False,
                                                    optimizer_type="RMSprop"), g7, 2)
    _by_word_flow(out, "sgd", VNETTrainer, dict(self_supervised=True, self_supervised_iterations=6, online_meta=False,
                                                optimizer_type="SGD", lr=0.05), g7, 2)
    save("g13_by_word_switches", **out)


def g15_by_word_reference_defaults():
    """BASELINE configs[2] and configs[4] WITH their updates at the reference's own hyperparameters (config.yaml: 200 self-supervised
    iterations per block; meta-learning every 5 blocks, 20 iterations x 10 drawn pairs, meta_lr 0.1, second order), 50 blocks each,
    run by the unmodified reference and recorded like G12: (a) VNETTrainer over the COST2100 taps, 200 minibatch CE + Adam
    iterations after every qualifying block (9 600 Adam steps); (b) METAVNETTrainer over the fading synthetic channel, 200
    whole-word iterations per block from the saved weights + 9 meta-learning updates of up to 200 second-order steps."""
    from python_code.trainers.META_VNET.metavnet_trainer import METAVNETTrainer

    g7 = np.load(os.path.join(HERE, "g7_by_word.npz"))
    out = {}
    _by_word_flow(out, "c2_selfsup", VNETTrainer, dict(self_supervised=True, self_supervised_iterations=200, online_meta=False,
                                                       channel_coefficients="cost2100", fading_in_channel=False), g7, 2, per_block=True)
    _by_word_flow(out, "c4_meta", METAVNETTrainer, dict(self_supervised=True, self_supervised_iterations=200, online_meta=True, MAML=True,
                                                        meta_lr=0.1, window_size=1, meta_train_iterations=20, meta_j_num=10,
                                                        meta_subframes=5, weights_init="last_frame"), g7, 2, per_block=True)
    save("g15_by_word_reference_defaults", **out)


def g16_by_word_other_state_counts():
    """eval_by_word with updates at other channel memories than 4: ViterbiNet with 8 states (memory 3) and 32 states (memory 5),
    briefly trained by the reference's own trainer (train_vnet), then 50 blocks of the self-supervised and the meta-learning flow each,
    recorded like G12 / G13 -- the run-time-n_states instantiations of the training kernels and the block step's separate launches."""
    from python_code.trainers.META_VNET.metavnet_trainer import METAVNETTrainer

    out = {}
    meta = dict(self_supervised=True, self_supervised_iterations=6, online_meta=True, MAML=True, meta_lr=0.1, window_size=1,
                meta_train_iterations=2, meta_j_num=4, meta_subframes=5, weights_init="last_frame")
    selfsup = dict(self_supervised=True, self_supervised_iterations=8, online_meta=False)
    for L in (3, 5):
        import contextlib
        import io

        with contextlib.redirect_stdout(io.StringIO()):
            w = export_weights(train_vnet(L, 10).detector)
        # (the reference's fading taps exist for memory 4 only, channel_estimation.py: a static channel here)
        _by_word_flow(out, f"L{L}_selfsup", VNETTrainer, dict(selfsup, memory_length=L, fading_in_channel=False), None, 2, weights=w)
        _by_word_flow(out, f"L{L}_meta", METAVNETTrainer, dict(meta, memory_length=L, fading_in_channel=False), None, 2, weights=w)
    save("g16_by_word_other_state_counts", **out)


def g17_by_word_64_128_states():
    """eval_by_word with self-supervised updates at channel memories 6 and 7 (64 and 128 states; round 5: online_train_kernel<64 | 128>,
    parameters + gradient in LDS, the optimizer's moments in global memory): ViterbiNet briefly trained by the reference's own
    trainer, then 50 blocks of the self-supervised flow, recorded like G16.  (The meta-learning kernel stops at 32 states.)"""
    out = {}
    selfsup = dict(self_supervised=True, self_supervised_iterations=8, online_meta=False)
    for L in (6, 7):
        import contextlib
        import io

        with contextlib.redirect_stdout(io.StringIO()):
            w = export_weights(train_vnet(L, 10).detector)
        _by_word_flow(out, f"L{L}_selfsup", VNETTrainer, dict(selfsup, memory_length=L, fading_in_channel=False), None, 2, weights=w)
    save("g17_by_word_64_128_states", **out)


# ----------------------------------------------------------------------------- G14
def g14_aggregated_evaluate():
    """Trainer.evaluate() in 'aggregated' mode (trainer.py:368-381 -> evaluate_at_point :254-265 -> gamma_eval :243-252 ->
    single_eval_at_point :222-241) over three SNR points, run by the unmodified reference as VATrainer, VNETTrainer and
    METAVNETTrainer, coded (use_ecc=True: RS(17,15) decoding of every detected word before the error rates) and uncoded.
    Stored: the words of every SNR point (a twin trainer with the same seeds draws them in the same order), the weights the
    run loaded (G7's, saved as the checkpoint of every SNR point), the data rows and the ser vector evaluate() returned."""
    from python_code.trainers.META_VNET.metavnet_trainer import METAVNETTrainer

    g7 = np.load(os.path.join(HERE, "g7_by_word.npz"))
    out, names = {}, []
    torch.set_num_threads(1)
    for tag, cls, ecc in (("va_coded", VATrainer, True), ("va_plain", VATrainer, False), ("vnet_coded", VNETTrainer, True),
                          ("vnet_plain", VNETTrainer, False), ("meta_coded", METAVNETTrainer, True)):
        wdir = os.path.join(TMP, "w_g14_" + tag)
        os.makedirs(wdir, exist_ok=True)
        kw = dict(eval_mode="aggregated", use_ecc=ecc, n_symbols=2, memory_length=4, val_block_length=120, val_frames=2,
                  subframes_in_frame=25, channel_coefficients="time_decay", fading_in_channel=True, fading_in_decoder=cls is VATrainer,
                  fading_taps_type=2, noisy_est_var=0, val_SNR_start=8, val_SNR_end=10, val_SNR_step=1, gamma=0.2, weights_dir=wdir,
                  self_supervised=False, online_meta=False, noise_seed=3450002, word_seed=7860002)
        twin, tr = cls(**kw), cls(**kw)
        if cls is not VATrainer:
            sd = tr.detector.state_dict()
            for i, k in enumerate(["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias"]):
                sd[k] = torch.tensor(g7[f"w{i}"])
            for snr in (8, 9, 10):
                torch.save({"model_state_dict": sd, "optimizer_state_dict": {}, "loss": 0.0}, os.path.join(wdir, f"snr_{snr}_gamma_0.2.pt"))
        for snr in (8, 9, 10):  # gamma_eval's order
            tx, rx = twin.channel_dataset["val"].__getitem__(snr_list=[snr], gamma=0.2)
            if tag == "meta_coded":  # same seeds, same class of dataset: the words of vnet_coded (not stored twice)
                assert np.array_equal(out[f"vnet_coded_rx{snr}"], rx.numpy()) and np.array_equal(out[f"vnet_coded_tx{snr}"], tx.numpy())
                continue
            out[f"{tag}_tx{snr}"] = tx.numpy().astype(np.uint8)
            out[f"{tag}_rx{snr}"] = rx.numpy().astype(np.float32)
        ser = tr.evaluate()
        out[f"{tag}_ser"] = np.asarray(ser, np.float64)
        out[f"{tag}_data_indices"] = tr.data_indices.numpy()
        out[f"{tag}_meta"] = np.array([int(ecc), 2, 4, 50, 136 if ecc else 120], np.int64)
        names.append(tag)
        print("g14", tag, "ser", ser)
    out["names"] = np.array(names)
    save("g14_aggregated_evaluate", **out)


# ----------------------------------------------------------------------------- G8
def g8_rs():
    """RS(n,k) KATs through the reference's own encode/decode (rs_main.py:9-37), incl. patterns beyond the
    correction capacity (the reference then returns the uncorrected word or mis-corrects: both are pinned)."""
    from python_code.ecc.rs_main import encode, decode

    rng = np.random.RandomState(88)
    out = {}
    for (kbits, nsym, ncase) in ((120, 2, 160), (120, 8, 160), (480, 8, 60), (8, 2, 24), (1976, 8, 6)):
        msgs, cws, rxs, decs = [], [], [], []
        for c in range(ncase):
            b = rng.randint(0, 2, kbits)
            cw = encode(b, nsym)
            rx = cw.copy()
            mode = c % 8
            nbytes = len(cw) // 8
            if mode in (1, 2, 3, 4, 5):  # 1..nsym/2+2 corrupted symbols (random byte values)
                nerr = min(nbytes, max(1, (mode * (nsym // 2 + 2)) // 5))
                for pos in rng.choice(nbytes, nerr, replace=False):
                    flip = rng.randint(1, 256)
                    bits = np.unpackbits(np.array([flip], np.uint8))
                    rx[8 * pos: 8 * pos + 8] ^= bits
            elif mode == 6:  # scattered single-bit errors
                for pos in rng.choice(len(cw), min(len(cw), nsym), replace=False):
                    rx[pos] ^= 1
            elif mode == 7:  # heavy corruption
                rx = rng.randint(0, 2, len(cw))
            d = decode(rx, nsym)
            msgs.append(b)
            cws.append(cw)
            rxs.append(rx)
            decs.append(d)
        tag = f"k{kbits}_n{nsym}"
        out[tag + "_msg"] = np.array(msgs, np.uint8)
        out[tag + "_cw"] = np.array(cws, np.uint8)
        out[tag + "_rx"] = np.array(rxs, np.uint8)
        out[tag + "_dec"] = np.array(decs, np.uint8)
        print(tag, "cases", ncase, "decoded==msg:", int(np.sum(np.all(np.array(decs) == np.array(msgs), axis=1))))
    save("g8_rs", **out)


if __name__ == "__main__":
    import contextlib
    import io

    patch_cost2100()
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8_rs()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g11":
        g11_meta_train_loop()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14_aggregated_evaluate()
        g15_by_word_reference_defaults()
        g16_by_word_other_state_counts()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("g12", "g13", "g15", "g16", "g17"):
        import contextlib
        import io

        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):  # (the reference prints every block)
            {"g12": g12_by_word_with_updates, "g13": g13_by_word_switches, "g15": g15_by_word_reference_defaults,
             "g16": g16_by_word_other_state_counts, "g17": g17_by_word_64_128_states}[sys.argv[1]]()
        print("\n".join(l for l in buf.getvalue().splitlines() if l.startswith(("g12", "wrote"))))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g10":
        g10_online_training()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        import contextlib
        import io

        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g9_by_word_va()
        print("\n".join(l for l in buf.getvalue().splitlines() if l.startswith(("time_decay", "cost2100", "wrote"))))
        sys.exit(0)
    g8_rs()
    g1_acs()
    g5_kats()
    g6_channels()
    g2_va()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):  # the reference trainers print per-step progress
        trained4 = g3_g4_vnet()
    print("\n".join(l for l in buf.getvalue().splitlines() if l.startswith(("S", "wrote", "best"))))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g9_by_word_va()
        g10_online_training()
        g11_meta_train_loop()
        g7_by_word(trained4)
        g12_by_word_with_updates()
        g13_by_word_switches()
        g14_aggregated_evaluate()
    print("\n".join(l for l in buf.getvalue().splitlines() if l.startswith(("time_decay", "cost2100", "wrote", "Final"))))
    print("torch", torch.__version__, "numpy", np.__version__)
