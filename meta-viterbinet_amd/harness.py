"""Evaluation harness for the detection hot path: counterpart of Trainer.single_eval_at_point /
gamma_eval / eval_by_word's detector calls (python_code/trainers/trainer.py:222-265, 292-295), plus the
block-parallel multi-GPU Monte-Carlo loop the reference does not have.

Blocks (words) are independent, so the batch axis is split contiguously across ranks; every rank
decodes its rows, counts errors on the device as integers, and ONE all_reduce(SUM) of an int64[4]
tensor {bit_errors, bits, frame_errors, frames} (RCCL over xGMI when backend='nccl') ends the run.
Decisions are never gathered.  Integer counters make 1/2/4/8-GPU results identical."""
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import metrics as _metrics
from .channel import estimate_channel, transmit
from .ecc import rs_decode, rs_encode
from .trellis import calculate_states


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n rows: rank r owns [lo, hi); sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def data_indices(val_frames: int, subframes_in_frame: int) -> torch.Tensor:
    """Non-pilot rows: every row whose index is not a multiple of subframes_in_frame (trainer.py:100-102)."""
    idx = [i for i in range(val_frames * subframes_in_frame) if i % subframes_in_frame != 0]
    return torch.tensor(idx, dtype=torch.int64)


def synthetic_words(n_words: int, block_length: int, memory_length: int, snr: float, gamma: float,
                    device, seed: int, channel_coefficients: str = "time_decay", fused: bool = True):
    """At-scale synthetic inputs generated on `device`: bits ~ Bernoulli(1/2), zero-padded by
    L, BPSK 1-2c, anti-causal ISI y[t] = sum_k h[L-1-k] s[t+k] + w[t] (channel.py:25-27), w ~ N(0, 10^(-snr/10)).
    Same distribution as ChannelModelDataset (channel_dataset.py:55-95), different RNG stream.
    On a GPU the words come from ONE kernel (channel.generate_words: Philox bits + Box-Muller noise + the L-tap channel);
    fused=False keeps the older route (ATen randint / randn + mvn_isi_awgn_transmit).
    Returns (tx [n,block_length] fp32 {0,1}, y [n,block_length] fp32)."""
    L = memory_length
    h = estimate_channel(L, gamma, channel_coefficients)
    if fused and torch.device(device).type == "cuda":
        from .channel import generate_words

        return generate_words(n_words, block_length, h, snr, L, device, seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    tx = torch.randint(0, 2, (n_words, block_length), generator=g, device=device, dtype=torch.int8).to(torch.float32)
    noise = torch.randn(n_words, block_length, generator=g, device=device)
    if torch.device(device).type == "cuda":
        return tx, transmit(tx, h, snr, L, noise)  # mvn_isi_awgn_transmit
    # CPU (host-side tests only): the same arithmetic in torch float64
    s = 1.0 - 2.0 * torch.cat([tx, torch.zeros(n_words, L)], dim=1).double()
    y = torch.zeros(n_words, block_length, dtype=torch.float64)
    for i in range(L):
        y += float(h[0, L - 1 - i]) * s[:, i:i + block_length]
    y += ((10 ** (snr / 10)) ** (-0.5)) * noise.double()
    return tx, y.float()


def va_monte_carlo(detector, n_words: int, snr: float, gamma: float, device, seed: int, counters: Optional[torch.Tensor] = None):
    """One uncoded Monte-Carlo point of the classical Viterbi detector in ONE launch (mvn_va_montecarlo_f32): what
    Trainer.single_eval_at_point does (trainer.py:222-241: draw the words, detector(rx, 'val'), calculate_error_rates), with the words
    of synthetic_words(..., seed) generated inside the detector and compared there -- no tx, y or decisions in device memory (at
    BASELINE configs[3], 10^6 blocks x 1000 symbols, those are 12 GB).  16 or 256 states; the words' channel is the detector's
    (`_estimate_all`: row b % val_words for word b, like its state priors).  Returns int64[4] counters {bit_errors, bits,
    frame_errors, frames} on `device` (accumulated into `counters` when given): the same numbers as
    count_errors(detector(y, 'val', snr, gamma), tx) over (tx, y) = synthetic_words(n_words, T, L, snr, gamma, device, seed)."""
    from . import _lib

    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.MvnError("va_monte_carlo runs on an MI355X (ROCm) device only")
    T, L, S = detector.transmission_length, detector.memory_length, detector.n_states
    h = torch.as_tensor(np.ascontiguousarray(detector._estimate_all(gamma, "val"), dtype=np.float64).reshape(-1, L), device=dev)
    pri = detector._priors_table(torch.empty(0, device=dev), gamma, "val", None)
    if counters is None:
        counters = torch.zeros(4, dtype=torch.int64, device=dev)
    sigma = (10 ** (snr / 10)) ** (-0.5)  # channel.py:23,31
    with torch.cuda.device(dev):
        rc = _lib.load().mvn_va_montecarlo_f32(_lib.ptr(h), h.shape[0], float(sigma), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(pri),
                                               pri.shape[0], _lib.ptr(counters), n_words, T, L, S, _lib.current_stream(dev))
    _lib.check(rc, "mvn_va_montecarlo_f32")
    return counters


def _gpu_counter(detected: torch.Tensor, tx: torch.Tensor, rows: Optional[torch.Tensor]) -> torch.Tensor:
    return _metrics.count_errors(detected, tx, rows)


def eval_counters(detector: Callable, tx: torch.Tensor, rx: torch.Tensor, snr: float, gamma: float,
                  rows: Optional[torch.Tensor] = None, counter: Callable = _gpu_counter,
                  group=None, reduce: bool = True, n_symbols: int = 0, rs_decoder: Callable = rs_decode) -> torch.Tensor:
    """One Monte-Carlo point on THIS rank's rows: detect -> [RS decode] -> count -> (optionally) all-reduce.
    `tx`/`rx` are the rank-local shards; `rows` are local row indices counted (None = all).
    n_symbols > 0 = the reference's use_ecc path (trainer.py:234-236): detected words are RS-decoded (on the
    device) before they are compared with the transmitted message bits.
    Returns int64[4] counters (global sums when reduce=True and a process group is up)."""
    if n_symbols > 0:
        detected = detector(rx, "val", snr, gamma)
        counters = counter(rs_decoder(detected, n_symbols)[:, : tx.shape[1]], tx, rows)
    elif counter is _gpu_counter and getattr(detector, "n_states", None) == 16 and hasattr(detector, "val_count"):
        counters = detector.val_count(rx, tx, rows)  # decode + count in one launch, decisions never stored
    else:
        detected = detector(rx, "val", snr, gamma)
        counters = counter(detected[:, : tx.shape[1]], tx, rows)
    if reduce and dist.is_available() and dist.is_initialized():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def single_eval_at_point(detector: Callable, tx: torch.Tensor, rx: torch.Tensor, snr: float, gamma: float,
                         rows: Optional[torch.Tensor] = None, counter: Callable = _gpu_counter,
                         group=None, n_symbols: int = 0) -> Tuple[float, float, torch.Tensor]:
    """trainer.py:222-241 without the data draw: detect, optional RS decode (n_symbols > 0 = use_ecc), error
    rates over the data rows.  Returns (ser, fer, counters); the reference returns ser only (:238-241)."""
    counters = eval_counters(detector, tx, rx, snr, gamma, rows, counter, group, n_symbols=n_symbols)
    ser, fer = _metrics.rates_from_counters(counters)
    return ser, fer, counters


def sharded_eval(detector: Callable, tx: torch.Tensor, rx: torch.Tensor, snr: float, gamma: float,
                 rows: Optional[torch.Tensor] = None, counter: Callable = _gpu_counter, group=None,
                 rank: Optional[int] = None, world: Optional[int] = None, n_symbols: int = 0,
                 rs_decoder: Callable = rs_decode) -> Tuple[float, float, torch.Tensor]:
    """Same point, but given the FULL (tx, rx) on every rank: each rank takes its contiguous row shard
    (shard_range), maps the global `rows` filter into it, and the counters are all-reduced."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(tx.shape[0], rank, world)
    local_rows = None
    if rows is not None:
        r = rows.to("cpu")
        r = r[(r >= lo) & (r < hi)] - lo
        local_rows = r.to(tx.device)
    if hi > lo:
        counters = eval_counters(detector, tx[lo:hi], rx[lo:hi], snr, gamma, local_rows, counter, group, reduce=False,
                                 n_symbols=n_symbols, rs_decoder=rs_decoder)
    else:
        counters = torch.zeros(4, dtype=torch.int64, device=tx.device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    ser, fer = _metrics.rates_from_counters(counters)
    return ser, fer, counters


def replica_eval(trial: Callable, n_trials: int, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, device=None, batched: bool = False) -> np.ndarray:
    """Replica mode for the evaluations that do NOT shard within a trial: with online training
    between blocks (eval_by_word with self_supervised / online_meta, trainer.py:292-347) block k's weights depend on the
    blocks before it, so one trial stays on one GPU and the parallel axis is the (SNR x seed x method) grid the reference
    walks serially (plotters/plotter_main.py:117-149).  Rank r runs trials r, r + world, ...; `trial(i)` returns that
    trial's float ser_by_word vector (all trials the same length); ONE all_gather of the float32 vectors (1.2 KB per
    trial) ends the run.  Returns [n_trials, N] (row i = trial i) on every rank.
    batched=True: `trial` takes the LIST of this rank's trial numbers and returns their vectors as rows [len(list), N] --
    the rank's trials then advance together on its GPU (trials.eval_by_word_batched) instead of one after the other.
    `device`: where the gathered tensor lives (a CUDA device under backend 'nccl' = RCCL, 'cpu' under gloo)."""
    up = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank(group) if up else 0
    if world is None:
        world = dist.get_world_size(group) if up else 1
    ids = list(range(rank, n_trials, world))
    if batched:
        rows = np.asarray(trial(ids), dtype=np.float32).reshape(len(ids), -1) if ids else np.zeros((0, 0), np.float32)
        mine = [rows[k] for k in range(len(ids))]
    else:
        mine = [np.asarray(trial(i), dtype=np.float32).reshape(-1) for i in ids]
    per_rank = (n_trials + world - 1) // world
    n = mine[0].shape[0] if mine else 0
    if up and world > 1:  # ranks without a trial still need the vector length for the collective
        nt = torch.tensor([n], dtype=torch.int64, device=device if device is not None else "cpu")
        dist.all_reduce(nt, op=dist.ReduceOp.MAX, group=group)
        n = int(nt.item())
    local = torch.full((per_rank, n), float("nan"), dtype=torch.float32)
    for k, v in enumerate(mine):
        local[k] = torch.from_numpy(v)
    if not (up and world > 1):
        return local[:n_trials].numpy()
    local = local.to(device if device is not None else "cpu")
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    out = np.empty((n_trials, n), np.float32)
    for r, g in enumerate(gathered):
        g = g.cpu().numpy()
        for k, i in enumerate(range(r, n_trials, world)):
            out[i] = g[k]
    return out


def detect_by_word(detector: Callable, rx: torch.Tensor, snr: float, gamma: float, batched: bool = True,
                   pass_count: bool = False) -> torch.Tensor:
    """The detector calls of eval_by_word (trainer.py:292-295) when no online update runs between blocks
    ("joint" variant): block k's decision does not depend on blocks < k, so the 300 B=1 calls collapse
    into one batched call (batched=True) -- or are issued one by one exactly like the reference."""
    if batched and not pass_count:
        return detector(rx, "val", snr, gamma)
    outs = []
    for count in range(rx.shape[0]):
        word = rx[count].reshape(1, -1)
        outs.append(detector(word, "val", snr, gamma, count) if pass_count else detector(word, "val", snr, gamma))
    return torch.cat(outs, dim=0)


def eval_by_word(detector, tx: torch.Tensor, rx: torch.Tensor, snr: float, gamma: float, n_symbols: int,
                 subframes_in_frame: int, self_supervised: bool = False, online_trainer=None,
                 self_supervised_iterations: int = 200, ser_thresh: float = 0.02, pass_count: bool = False,
                 verbose: bool = False, online_meta: bool = False, meta_detector=None, meta_lr: float = 0.1,
                 MAML: bool = True, window_size: int = 1, meta_train_iterations: int = 20, meta_j_num: int = 10,
                 meta_subframes: int = 5, meta_style_online_training: bool = False,
                 graphed_meta: bool = True, hip_meta: bool = True, initial_buffer=None, weights_init: str = "last_frame",
                 meta_training_weights=None, draws=None, fused_step: bool = True, observer=None) -> np.ndarray:
    """Sequential per-block online evaluation: counterpart of Trainer.eval_by_word (trainer.py:267-354).  Everything but the
    control flow stays on the GPU:
        for every block k:  detect (B=1)  ->  data block: RS decode, ser, RS re-encode | pilot: encode the known word
            ->  if ser <= ser_thresh: buffer (rx, label), label = detected word if ser > 0 else the re-encoded word
            ->  if online_meta and k % meta_subframes == 0 (k >= meta_subframes, buffer > 2): restart from the saved
                weights, meta_train_iterations x {meta_j_num random (support, query) pairs of consecutive buffered words ->
                meta.meta_train_loop}, save the weights (:331-343)
            ->  if self_supervised (and ser <= ser_thresh): online_trainer.online_training(last buffered pair) (:345-347);
                meta_style_online_training=True first restores the saved weights and trains on the whole word
                (metavnet_trainer.py:52-64), otherwise a 32-sample minibatch per iteration (vnet_trainer.py:49-60).
    initial_buffer = (tx_codewords [W, T], rx_words [W, T]): the reference's buffer_empty=False (:278-286) -- the buffer
    starts with W words drawn from the training channel and stays W long (every qualifying block pushes the oldest out,
    :325-328); None = buffer_empty=True.  weights_init (meta_weights_init, :356-366): what a meta update restarts from --
    'last_frame' the weights saved after the previous update, 'random' freshly initialised weights and a fresh optimizer,
    'meta_training' the weights in `meta_training_weights` (the reference loads its meta-trained checkpoint).
    tx [N, K] message bits, rx [N, K + 8*n_symbols] received words; block k is a pilot when k % subframes_in_frame == 0
    (trainer.py:100-102).  Returns ser_by_word [N] (0 for pilots), like the reference.
    One host sync per block (the ser decides what happens next), as in the reference (trainer.py:305).
    draws: a trials.TrialDraws -- the minibatches and j_hat values come from this trial's own streams instead of the global
    generators (the reference's are unseeded), which is what makes a run replayable inside trials.eval_by_word_batched.
    fused_step: a 16-state ViterbiNet detector with nsym <= 8 takes ONE launch per block (mvn_vnet_byword_step_f32:
    detect, RS decode, error count, re-encode); False keeps the four separate launches (the cross-check in the tests).
    observer: called at the end of every block of the update branches with a dict {stage: 'end', count, ser, pushed,
    buffer_rx, buffer_tx, meta: (support_idx [n, W], query_idx [n]) or None, trained, batch_idx, detector, saved_detector},
    and with stage 'meta' right after a meta-learning update launched by the HIP kernel: what a test needs to replay the
    block's updates on another implementation (tests/test_gpu_replay.py)."""
    import copy

    from .meta import GraphedMetaStep, copy_model, meta_train_loop

    N = tx.shape[0]
    ser_by_word = np.zeros(N)
    K = tx.shape[1]
    fused = fused_step and _fused_step_applies(detector, rx, n_symbols, pass_count)
    if not (self_supervised or online_meta or verbose):
        # No update runs between blocks, so nothing on the host depends on a block's ser: the reference's 300 B=1 detector
        # calls are issued one by one like it issues them, but the per-block error counts stay on the device and ONE
        # transfer ends the run (the reference synchronises after every block, trainer.py:305).  Pilot blocks are skipped:
        # their ser is 0 and their detection only feeds the buffer of the update branches.
        if fused:  # one launch per data block: detect + RS decode + error count (+ the re-encoding nobody reads here)
            nerr = torch.zeros(N, dtype=torch.int32, device=rx.device)
            for count in range(N):
                if count % subframes_in_frame != 0:
                    _byword_step(detector, rx[count:count + 1], tx[count:count + 1], n_symbols, False, nerr[count:count + 1],
                                 outputs=False, gamma=gamma, count=count if pass_count else None)
            e = nerr.cpu().numpy()
            data = np.arange(N) % subframes_in_frame != 0
            ser_by_word[data] = _metrics.ser_from_errors(e[data], K)  # the reference's value bit for bit (metrics.py:13-16)
            return ser_by_word
        counters = torch.zeros((N, 4), dtype=torch.int64, device=rx.device)  # row k: {bit errors, bits, ...} of block k
        for count in range(N):
            if count % subframes_in_frame == 0:
                continue
            received_word = rx[count:count + 1]
            detected_word = detector(received_word, "val", snr, gamma, count) if pass_count else detector(received_word, "val", snr, gamma)
            _metrics.count_errors(rs_decode(detected_word, n_symbols), tx[count:count + 1], None, counters[count])
        c = counters.cpu().numpy()
        data = c[:, 1] > 0
        ser_by_word[data] = _metrics.ser_from_errors(c[data, 0], K)  # the reference's value bit for bit (metrics.py:13-16)
        return ser_by_word
    if (self_supervised or online_meta) and online_trainer is None:
        raise ValueError("self_supervised / online_meta need an OnlineTrainer (it owns the Adam state)")
    if weights_init not in ("last_frame", "random", "meta_training"):
        raise ValueError("No such weights init!!!")
    if weights_init == "meta_training" and meta_training_weights is None:
        raise ValueError("weights_init='meta_training' needs meta_training_weights (six arrays in parameters() order)")
    if online_meta and meta_detector is None:
        raise ValueError("online_meta needs a META_VNETDetector")
    saved_detector = copy.deepcopy(detector) if (online_meta or meta_style_online_training) else None  # :275
    buffer_empty = initial_buffer is None
    if buffer_empty:
        buffer_rx = torch.empty([0, rx.shape[1]], device=rx.device)
        buffer_tx = torch.empty([0, rx.shape[1]], device=rx.device)
    else:
        buffer_tx, buffer_rx = [t.to(device=rx.device, dtype=torch.float32) for t in initial_buffer]
    meta_step = None  # meta.GraphedMetaStep, built at the first meta update (graphed_meta and a CUDA detector)
    graphed_meta = graphed_meta and rx.is_cuda and (online_trainer is None or online_trainer.optimizer_type == "Adam")  # captured Adam update
    # mvn_vnet_maml_train_f32: the LDS holds 4 parameter vectors (n_states <= 32); the kernel's optimizer is Adam
    hip_meta = (hip_meta and rx.is_cuda and detector.n_states <= 32 and online_trainer is not None
                and online_trainer.optimizer_type == "Adam" and online_trainer.use_kernel)
    support_idx = torch.arange(-window_size - 1, -1, device=rx.device).long()  # :288
    query_idx = -1 * torch.ones(1, device=rx.device).long()

    def draw_j_hat(high):  # trainer.py:337: the global generator, or this trial's own stream
        if draws is None:
            return torch.unique(torch.randint(low=0, high=high, size=[meta_j_num])).to(rx.device)
        return torch.as_tensor(draws.j_hat(high, meta_j_num), device=rx.device).long()

    from .detectors import VNETDetector

    # The fused ViterbiNet step also picks the word the reference buffers and computes its trellis states (what the training
    # kernels take as labels), and the block's error count travels to the host together with the trainer's status word: one
    # launch and one 8-byte copy per block before the host decides.
    step_labels = fused and isinstance(detector, VNETDetector) and online_trainer is not None and online_trainer.use_kernel
    sync_words = online_trainer.sync_words if (step_labels and getattr(online_trainer, "sync_words", None) is not None) else None
    nerr1 = (sync_words[:1] if sync_words is not None else torch.zeros(1, dtype=torch.int32, device=rx.device)) if fused else None
    buffer_lab = None
    if step_labels:
        buffer_lab = (torch.empty([0, rx.shape[1]], dtype=torch.int32, device=rx.device) if buffer_empty else
                      calculate_states(online_trainer.memory_length, buffer_tx).reshape(buffer_tx.shape).to(torch.int32))
    for count in range(N):
        transmitted_word, received_word = tx[count].reshape(1, -1), rx[count].reshape(1, -1)
        pilot = count % subframes_in_frame == 0
        seen = {"count": count, "meta": None, "trained": False, "batch_idx": None} if observer is not None else None
        status_word = None
        if step_labels:  # ONE launch: detect, RS decode, error count, re-encode, the word to buffer and its states
            label_word, label_states = _byword_step(detector, received_word, transmitted_word, n_symbols, pilot, nerr1, labels=True)
            if sync_words is not None:
                n_err, status_word = sync_words.tolist()
            else:
                n_err = int(nerr1.item())
            ser = 0.0 if pilot else float(_metrics.ser_from_errors(n_err, K))  # calculate_error_rates (:301)
            if not pilot:
                ser_by_word[count] = ser
        elif fused:  # ONE launch: detect, RS decode, error count, re-encode (pilot: encode the known word)
            detected_word, encoded_word = _byword_step(detector, received_word, transmitted_word, n_symbols, pilot, nerr1,
                                                       gamma=gamma, count=count if pass_count else None)
            ser = 0.0 if pilot else float(_metrics.ser_from_errors(int(nerr1.item()), K))  # calculate_error_rates (:301)
            if not pilot:
                ser_by_word[count] = ser
        else:
            detected_word = detector(received_word, "val", snr, gamma, count) if pass_count else detector(received_word, "val", snr, gamma)
            if not pilot:
                decoded_word = rs_decode(detected_word, n_symbols)
                ser = float(_metrics.ser_from_errors(int((decoded_word != transmitted_word).sum().item()), K))  # calculate_error_rates (:301)
                encoded_word = rs_encode(decoded_word, n_symbols)  # :304
                ser_by_word[count] = ser
            else:
                encoded_word = rs_encode(transmitted_word, n_symbols)  # pilot: the word is known (:314-316)
                ser = 0.0
        if online_trainer is not None:
            online_trainer.check_status(status_word)  # a training launch that gave up its barrier (NaN weights) raises here
        if verbose:
            print(f"current: {count, ser}")
        if ser <= ser_thresh:  # :319-329 (buffer_empty=True: the buffer only grows)
            buffer_rx = torch.cat([buffer_rx, received_word])
            if step_labels:
                buffer_tx = torch.cat([buffer_tx, label_word], dim=0)
                buffer_lab = torch.cat([buffer_lab, label_states], dim=0)
            else:
                buffer_tx = torch.cat([buffer_tx, detected_word.reshape(1, -1) if ser > 0 else encoded_word.reshape(1, -1)], dim=0)
            if not buffer_empty:  # fixed-length window: the oldest word leaves (:325-328)
                buffer_rx, buffer_tx = buffer_rx[1:], buffer_tx[1:]
                if step_labels:
                    buffer_lab = buffer_lab[1:].contiguous()
        if online_meta and count % meta_subframes == 0 and count >= meta_subframes and buffer_rx.shape[0] > 2:  # :331-343
            if weights_init == "last_frame":  # meta_weights_init (:356-366)
                copy_model(source_model=saved_detector, dest_model=detector)
            elif weights_init == "random":
                if draws is not None:  # this trial's own stream (what lets trials.eval_by_word_batched replay the run)
                    with torch.no_grad():
                        for p_, w_ in zip(detector.parameters(), draws.init_weights(detector.n_states)):
                            p_.copy_(w_)
                else:
                    for m in detector.net:
                        if hasattr(m, "reset_parameters"):
                            m.reset_parameters()
                online_trainer.reset_state()
            else:
                with torch.no_grad():
                    for p_, w_ in zip(detector.parameters(), meta_training_weights):
                        p_.copy_(torch.as_tensor(w_, dtype=p_.dtype))
            if hip_meta:  # every MAML step of this update in ONE launch of the meta-learning kernel
                if draws is not None:  # the update's j_hat values in one draw (this trial's own stream)
                    j_all = torch.as_tensor(draws.j_hat_update(buffer_rx.shape[0] - 2, meta_train_iterations, meta_j_num),
                                            device=rx.device).long()
                else:
                    j_all = torch.cat([draw_j_hat(buffer_rx.shape[0] - 2) for _ in range(meta_train_iterations)])
                sup_all = j_all.reshape(-1, 1) + support_idx.reshape(1, -1) + 1
                qry_all = j_all + query_idx + 1
                online_trainer.maml_training(buffer_rx, buffer_tx, sup_all, qry_all, meta_lr, MAML, labels=buffer_lab)
                if seen is not None:
                    seen["meta"] = (sup_all, qry_all)
            else:
                if graphed_meta and meta_step is None:  # captured once: one hipGraph replay per MAML step from here on
                    meta_step = GraphedMetaStep(detector, meta_detector, online_trainer, window_size, rx.shape[1], meta_lr,
                                                MAML)
                j_seen = []
                for _ in range(meta_train_iterations):
                    j_hat_values = draw_j_hat(buffer_rx.shape[0] - 2)
                    j_seen.append(j_hat_values)
                    for j_hat in j_hat_values:
                        if meta_step is not None:
                            meta_step(buffer_rx, buffer_tx, j_hat + support_idx + 1, j_hat + query_idx + 1)
                        else:
                            meta_train_loop(detector, meta_detector, online_trainer, buffer_rx, buffer_tx,
                                            j_hat + support_idx + 1, j_hat + query_idx + 1, meta_lr, MAML)
                if seen is not None:
                    j_all = torch.cat(j_seen)
                    seen["meta"] = (j_all.reshape(-1, 1) + support_idx.reshape(1, -1) + 1, j_all + query_idx + 1)
            copy_model(source_model=detector, dest_model=saved_detector)
            if seen is not None and seen["meta"] is not None:
                observer(dict(seen, stage="meta", detector=detector, saved_detector=saved_detector, buffer_rx=buffer_rx,
                              buffer_tx=buffer_tx))
        if self_supervised and ser <= ser_thresh:  # :345-347
            if meta_style_online_training:
                copy_model(source_model=saved_detector, dest_model=detector)  # metavnet_trainer.py:59
            batch_idx = None
            if draws is not None and not meta_style_online_training:
                batch_idx = draws.batches(count, N, rx.shape[1], self_supervised_iterations, online_trainer.train_minibatch_size)
            if seen is not None and batch_idx is None and not meta_style_online_training:
                batch_idx = online_trainer.select_batches(rx.shape[1], self_supervised_iterations)  # drawn here to be shown
            online_trainer.online_training(buffer_tx[-1].reshape(1, -1), buffer_rx[-1].reshape(1, -1),
                                           iterations=self_supervised_iterations, batch_idx=batch_idx,
                                           full_word=meta_style_online_training,
                                           labels=None if buffer_lab is None else buffer_lab[-1])
            if seen is not None:
                seen.update(trained=True, batch_idx=batch_idx)
        if seen is not None:
            seen.update(stage="end", ser=ser, pushed=ser <= ser_thresh, buffer_rx=buffer_rx, buffer_tx=buffer_tx, detector=detector,
                        saved_detector=saved_detector)
            observer(seen)
    if online_trainer is not None:
        online_trainer.check_status()
    return ser_by_word


def _fused_step_applies(detector, rx: torch.Tensor, n_symbols: int, pass_count: bool) -> bool:
    """One launch per block step (mvn_vnet_byword_step_f32 / mvn_va_byword_step_f32) serves a 16-state VNETDetector or
    VADetector whose 'val' length is the word length (Q5), words of whole bytes up to 1024 symbols and nsym <= 8; everything
    else takes the separate detect / RS / count launches.  (pass_count: the reference passes the block number to the detector,
    which is how VADetector picks the word's channel; ViterbiNet ignores it.)"""
    from .detectors import VADetector, VNETDetector

    T = rx.shape[1]
    if not (rx.is_cuda and getattr(detector, "n_states", None) == 16 and T % 8 == 0 and 8 <= T <= 1024 and 1 <= n_symbols <= 8 and T // 8 > n_symbols):
        return False
    if isinstance(detector, VNETDetector):
        return not pass_count and detector.transmission_lengths["val"] == T
    # VADetector: the fused step takes ONE row of state priors -- the word's own (pass_count) or the table's only row.  A
    # multi-row table without the block number takes the separate launches, where the detector itself raises what the
    # reference raises (the [1, S] / [W, S] broadcast of va_detector.py:64-68).
    return isinstance(detector, VADetector) and detector.transmission_length == T and (pass_count or detector.val_words == 1)


def _byword_step(detector, received_word: torch.Tensor, transmitted_word: torch.Tensor, n_symbols: int, pilot: bool,
                 nerr: torch.Tensor, outputs: bool = True, gamma: float = None, count: int = None, labels: bool = False):
    """One block of eval_by_word in one launch (trainer.py:292-316): returns (detected_word, encoded_word) [1, T]; the
    block's bit-error count goes to nerr[0] (device int32).  On a pilot the detection is skipped (never used) and
    detected_word is None.  outputs=False: the error count only (no words are stored, the re-encoding is skipped).
    gamma / count: what VADetector.forward takes to find the word's channel (count None: the detector's single table row).
    labels=True (ViterbiNet): returns (label_word [1, T], states int32 [1, T]) instead -- the word the reference pushes into its
    buffer (:322-324: the detected word if it had bit errors, else the re-encoded one) and its trellis states, both chosen and
    computed by the kernel."""
    from . import _lib
    from .detectors import VADetector

    rxw, txw = _lib.f32c(received_word), _lib.f32c(transmitted_word)
    T, K = rxw.shape[1], txw.shape[1]
    dev = rxw.device
    det = None if (pilot or not outputs) else torch.empty((1, T), dtype=torch.float32, device=dev)
    enc = torch.empty((1, T), dtype=torch.float32, device=dev) if outputs else None
    if isinstance(detector, VADetector):
        pri = detector._priors_table(rxw, gamma, "val", count)  # [W, 16] (one row when count is given)
        assert pri.shape[0] == 1  # (_fused_step_applies sends every other table to the separate launches)
        with _lib.on_device(dev):
            rc = _lib.load().mvn_va_byword_step_f32(_lib.ptr(rxw), T, _lib.ptr(txw), K, _lib.ptr(pri), 1, _lib.ptr(det), T, None, K,
                                                    _lib.ptr(enc), T, None, T, None, T, _lib.ptr(nerr), 1, T, n_symbols,
                                                    1 if pilot else 0, 16, _lib.current_stream(dev))
        _lib.check(rc, "mvn_va_byword_step_f32")
        return det, enc
    w = detector._params()
    if labels:
        word, states = torch.empty((1, T), dtype=torch.float32, device=dev), torch.empty((1, T), dtype=torch.int32, device=dev)
        with _lib.on_device(dev):
            rc = _lib.load().mvn_vnet_byword_step_f32(_lib.ptr(rxw), T, _lib.ptr(txw), K, *[_lib.ptr(_lib.f32c(p)) for p in w], None,
                                                      None, T, None, K, None, T, _lib.ptr(word), T, _lib.ptr(states), T,
                                                      _lib.ptr(nerr), 1, T, n_symbols, 1 if pilot else 0, 16,
                                                      _lib.current_stream(dev))
        _lib.check(rc, "mvn_vnet_byword_step_f32")
        return word, states
    with _lib.on_device(dev):
        rc = _lib.load().mvn_vnet_byword_step_f32(_lib.ptr(rxw), T, _lib.ptr(txw), K, *[_lib.ptr(_lib.f32c(p)) for p in w], None,
                                                  _lib.ptr(det), T, None, K, _lib.ptr(enc), T, None, T, None, T,
                                                  _lib.ptr(nerr), 1, T, n_symbols, 1 if pilot else 0, 16,
                                                  _lib.current_stream(dev))
    _lib.check(rc, "mvn_vnet_byword_step_f32")
    return det, enc
