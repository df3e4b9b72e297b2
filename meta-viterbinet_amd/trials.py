"""R independent by-word evaluations advancing in lock-step on ONE GPU.

Trainer.eval_by_word with online training between blocks (python_code/trainers/trainer.py:267-354) is sequential WITHIN a
trial -- block k's weights depend on the blocks before it -- so it does not shard; the parallel axis is
the (SNR x seed x method) grid the reference walks serially (plotters/plotter_main.py:117-149).  One trial occupies 1-9 of an
MI355X's 256 CUs; here R trials step through their blocks together:

    per block step:  ONE mvn_vnet_byword_step_f32 launch (detect + RS decode + error count + re-encode + labels for all R
                     words, each with its own weights)  ->  ONE device-to-host copy of the R error counts (the only sync:
                     every trial's next move depends on its coded ser, trainer.py:305,319)  ->  per-trial decisions on the
                     host  ->  at most ONE mvn_vnet_maml_train_trials_f32 and ONE mvn_vnet_online_train_trials_f32 launch
                     sequence for the trials that train (gridDim.y = trial, per-trial barrier counters).

All per-trial state lives in stacked device tensors (TrialBank); the reference's buffer of (received word, label word) pairs
is a list of block numbers per trial (the words themselves stay where the step kernel wrote them).  Per trial, the results --
ser_by_word, final weights, Adam moments -- are bit-identical to harness.eval_by_word run alone with the same draws
(tests/test_gpu_trials.py::test_batched_trials_equal_sequential_runs).
"""
import ctypes
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .detectors import HIDDEN1_SIZE, HIDDEN2_SIZE
from .metrics import ser_from_errors

# include/mvn.h: mvn_train_trial_t
TRIAL_DTYPE = np.dtype([("y", "u8"), ("labels", "u8"), ("idx", "u8"), ("query_idx", "u8"), ("w_in", "u8", (6,)),
                        ("w_out", "u8", (6,)), ("w_out2", "u8", (6,)), ("adam_m", "u8"), ("adam_v", "u8"), ("loss_out", "u8"),
                        ("status", "u8"), ("b1pow", "f8"), ("b2pow", "f8"), ("n", "i4"), ("reserved", "i4")])


def param_offsets(n_states: int) -> np.ndarray:
    """Offsets (floats) of W1, b1, W2, b2, W3, b3 in a flat parameter vector in parameters() order, and its length."""
    sizes = [HIDDEN1_SIZE, HIDDEN1_SIZE, HIDDEN2_SIZE * HIDDEN1_SIZE, HIDDEN2_SIZE, n_states * HIDDEN2_SIZE, n_states]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def beta_power(beta: float, step: int) -> float:
    """(double)(float)beta ** step, exactly what the single-trial C entry points compute from (beta, step0)."""
    return float(np.float32(beta)) ** int(step)


def beta_powers(beta: float, steps) -> np.ndarray:
    """beta_power for an array of step counts, one libm pow() each (a vectorised pow may round differently: the batched run
    must hand its kernels the very doubles the single-trial entry points compute)."""
    b = float(np.float32(beta))
    return np.array([b ** int(s) for s in steps], dtype=np.float64)


class TrialDraws:
    """The random draws of ONE trial's online training, a pure function of (seed, shapes): the minibatches of select_batch
    (trainer.py:534-544: torch.multinomial with weights arange(T), without replacement) for every (block, iteration), drawn
    in one call on the device when first needed, and the j_hat draws of the online meta-learning step (trainer.py:337:
    torch.randint(0, len(buffer) - 2, [meta_j_num]), then unique) from a host generator.  The reference draws from the global
    unseeded generators; giving every trial its own stream is what lets R trials run interleaved and still be replayed one
    by one (harness.eval_by_word(draws=...)) with identical results."""

    def __init__(self, seed: int, device):
        self.seed = int(seed)
        self.device = torch.device(device)
        self.rng = np.random.RandomState(self.seed % (2 ** 32))
        self._table = None
        self._shape = None
        self._init_gen = None  # host generator of init_weights, created at its first use

    def batches(self, count: int, n_blocks: int, T: int, iterations: int, M: int) -> torch.Tensor:
        """int32 [iterations, M]: the minibatch indices of block `count` (a view into the trial's table)."""
        shape = (n_blocks, T, iterations, M)
        if self._table is None:
            gen = torch.Generator(device=self.device).manual_seed(self.seed)
            w = torch.arange(T, dtype=torch.float32, device=self.device).expand(n_blocks * iterations, T)
            self._table = torch.multinomial(w, M, generator=gen).to(torch.int32).reshape(n_blocks, iterations, M)
            self._shape = shape
        elif self._shape != shape:
            raise ValueError(f"TrialDraws was drawn for {self._shape}, asked for {shape}")
        return self._table[count]

    def init_weights(self, n_states: int) -> List[torch.Tensor]:
        """Freshly initialised ViterbiNet weights for meta_weights_init('random') (trainer.py:356-359 -> initialize_detector):
        nn.Linear's default reset_parameters (kaiming_uniform(a = sqrt 5) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and
        bias), layer by layer in parameters() order, from this trial's own stream (a host generator seeded by the trial's seed;
        successive calls continue the stream).  Six CPU tensors."""
        if self._init_gen is None:
            self._init_gen = torch.Generator().manual_seed((self.seed * 2654435761 + 12345) % (2 ** 63))
        out = []
        for fan_out, fan_in in ((HIDDEN1_SIZE, 1), (HIDDEN2_SIZE, HIDDEN1_SIZE), (n_states, HIDDEN2_SIZE)):
            bound = 1.0 / np.sqrt(fan_in)
            out.append(torch.empty(fan_out, fan_in).uniform_(-bound, bound, generator=self._init_gen))
            out.append(torch.empty(fan_out).uniform_(-bound, bound, generator=self._init_gen))
        return out

    def j_hat(self, high: int, size: int) -> np.ndarray:
        """np.unique(randint(0, high, size)): sorted distinct buffer positions, like torch.unique(torch.randint(...))."""
        return self.j_hat_update(high, 1, size)

    def j_hat_update(self, high: int, iterations: int, size: int) -> np.ndarray:
        """The j_hat values of ONE meta-learning update, concatenated in step order: `iterations` times (trainer.py:335-338)
        the sorted distinct values of `size` draws from [0, high).  One generator call and one sort for the whole update."""
        x = np.sort(self.rng.randint(0, high, size=(iterations, size)), axis=1)
        keep = np.ones(x.shape, dtype=bool)
        keep[:, 1:] = x[:, 1:] != x[:, :-1]
        return x[keep]  # row-major: iteration after iteration, ascending within each


class TrialBank:
    """Weights, saved weights (the reference's saved_detector, trainer.py:275) and Adam state of R ViterbiNet detectors in
    stacked device tensors; row r is trial r, a row holds the six arrays flat in parameters() order."""

    def __init__(self, weights: Sequence[Sequence], n_states: int, memory_length: int, device, lr: float = 0.001,
                 betas=(0.9, 0.999), eps: float = 1e-8, optimizer_type: str = "Adam"):
        if optimizer_type not in ("Adam", "RMSprop", "SGD"):  # deep_learning_setup (trainer.py:163-175)
            raise NotImplementedError("No such optimizer implemented!!!")
        self.optimizer_type = optimizer_type
        self.R = len(weights)
        self.n_states, self.memory_length = n_states, memory_length
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = torch.device(device)
        self.off = param_offsets(n_states)
        self.P = int(self.off[-1])
        rows = []
        for w in weights:
            if len(w) != 6:
                raise ValueError("a trial's weights are six arrays in parameters() order")
            rows.append(torch.cat([torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a.detach().cpu(), dtype=torch.float32).reshape(-1)
                                   for a in w]))
            if rows[-1].numel() != self.P:
                raise ValueError(f"ViterbiNet parameter count {rows[-1].numel()} != {self.P}")
        self.theta = torch.stack(rows).to(self.device).contiguous()
        self.saved = self.theta.clone()
        self.exp_avg = torch.zeros_like(self.theta)
        self.exp_avg_sq = torch.zeros_like(self.theta)
        self.step = np.zeros(self.R, dtype=np.int64)

    def weights(self, r: int, saved: bool = False) -> List[torch.Tensor]:
        """Trial r's six arrays as views with the shapes of VNETDetector.parameters()."""
        row = (self.saved if saved else self.theta)[r]
        shapes = [(HIDDEN1_SIZE, 1), (HIDDEN1_SIZE,), (HIDDEN2_SIZE, HIDDEN1_SIZE), (HIDDEN2_SIZE,), (self.n_states, HIDDEN2_SIZE),
                  (self.n_states,)]
        return [row[int(self.off[a]):int(self.off[a + 1])].reshape(s) for a, s in enumerate(shapes)]

    def pointers(self, t: torch.Tensor) -> np.ndarray:
        """uint64 [R, 6]: device addresses of the six arrays of every row of `t` (theta or saved)."""
        base = np.uint64(t.data_ptr()) + np.arange(self.R, dtype=np.uint64)[:, None] * np.uint64(4 * self.P)
        return base + (self.off[:6].astype(np.uint64) * np.uint64(4))[None, :]


class _Staging:
    """Pinned host image + device copy of per-step launch inputs (trial descriptors, index lists): filled on the host,
    sent with one asynchronous copy.  The host image is only rewritten after the step's synchronisation, which follows
    the copy on the same stream."""

    def __init__(self, nbytes: int, device):
        self.host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.np = self.host.numpy()

    def send(self, nbytes: int):
        self.dev[:nbytes].copy_(self.host[:nbytes], non_blocking=True)


class _BankRows:
    """Rows [lo, hi) of a TrialBank: the same attributes, as views."""

    def __init__(self, bank: TrialBank, lo: int, hi: int):
        self.R = hi - lo
        self.n_states, self.memory_length, self.lr, self.betas, self.eps = bank.n_states, bank.memory_length, bank.lr, bank.betas, bank.eps
        self.optimizer_type = bank.optimizer_type
        self.off, self.P = bank.off, bank.P
        self.theta, self.saved = bank.theta[lo:hi], bank.saved[lo:hi]
        self.exp_avg, self.exp_avg_sq = bank.exp_avg[lo:hi], bank.exp_avg_sq[lo:hi]
        self.step = bank.step[lo:hi]

    pointers = TrialBank.pointers


def eval_by_word_batched(bank: TrialBank, tx: torch.Tensor, rx: torch.Tensor, n_symbols: int, subframes_in_frame: int,
                         draws: Sequence[TrialDraws], self_supervised: bool = False, self_supervised_iterations: int = 200,
                         ser_thresh: float = 0.02, online_meta: bool = False, meta_lr: float = 0.1, MAML: bool = True,
                         window_size: int = 1, meta_train_iterations: int = 20, meta_j_num: int = 10, meta_subframes: int = 5,
                         meta_style_online_training: bool = False, train_minibatch_size: int = 32,
                         weights_init: str = "last_frame", meta_training_weights=None, record: Optional[dict] = None,
                         cohorts: int = 1, initial_buffer=None) -> np.ndarray:
    """R trials of harness.eval_by_word (= Trainer.eval_by_word, trainer.py:267-354) at once, with the reference's switches:
    buffer_empty True / False, weights_init last_frame / random / meta_training, Adam / RMSprop / SGD.
    tx [R, N, K] message bits, rx [R, N, K + 8 n_symbols] received words (trial r = row r, its own SNR / channel / seed);
    bank: the trials' weights and optimizer state (updated in place); draws[r]: trial r's TrialDraws.
    Returns ser_by_word [R, N] (0 for pilots), row r equal to eval_by_word(..., draws=draws[r]) run alone.
    record: optional dict that receives 'nerr' [R, N], 'trained' [R, N] bool and 'meta' [R, N] bool.
    initial_buffer = (tx_codewords, rx_words), each [W0, T] (shared) or [R, W0, T]: the reference's buffer_empty=False
    (trainer.py:278-286) -- every trial's buffer starts with these W0 words and stays W0 long, each qualifying block pushing the
    oldest out (:325-328).  weights_init='random' (meta_weights_init, :356-359): before every meta-learning update the trial's
    weights are re-initialised from ITS OWN stream (TrialDraws.init_weights) and its optimizer state reset.
    bank.optimizer_type 'RMSprop' / 'SGD' (deep_learning_setup, :163-175): the online-training kernel implements them next to
    Adam, the reference's default (round 5); the meta-learning kernel implements Adam, so a run with online_meta and another
    optimizer goes trial after trial through harness.eval_by_word (its meta-learning updates on stock autograd) -- same results as
    calling it yourself, no batching.  So do detectors with another state count than 16 (the one-launch block step and its
    R-word form serve 16 states): trial after trial through harness.eval_by_word, there on the run-time-n_states training kernels.
    cohorts > 1: the trials are split into that many groups that step ALTERNATELY on the same stream: while the GPU works
    through one group's training launches the host takes the decisions and fills the descriptors of the next (the host
    work of a step can only start after the step's sync).  Same launches per trial, same results."""
    R, N = rx.shape[0], rx.shape[1]
    if R != bank.R or len(draws) != R or tx.shape[0] != R:
        raise ValueError("tx [R, N, K], rx [R, N, K + 8 n_symbols], one TrialDraws and one bank row per trial")
    ser_by_word = np.zeros((R, N))
    if record is not None:
        record.update(nerr=np.zeros((R, N), np.int64), trained=np.zeros((R, N), bool), meta=np.zeros((R, N), bool))
    if weights_init not in ("last_frame", "random", "meta_training"):
        raise ValueError("No such weights init!!!")
    if initial_buffer is not None:  # (tx codewords, rx words), shared by the trials or one set per trial
        ib = [t.to(device=rx.device, dtype=torch.float32) for t in initial_buffer]
        ib = [t.unsqueeze(0).expand(R, -1, -1) if t.dim() == 2 else t for t in ib]
        if ib[0].shape != ib[1].shape or ib[0].shape[0] != R or ib[0].shape[2] != rx.shape[2]:
            raise ValueError("initial_buffer = (tx_codewords, rx_words), each [W0, T] or [R, W0, T]")
        initial_buffer = ib
    if (bank.optimizer_type != "Adam" and online_meta) or bank.n_states != 16:
        return _one_trial_at_a_time(bank, tx, rx, n_symbols, subframes_in_frame, draws, ser_by_word, record, initial_buffer,
                                    dict(self_supervised=self_supervised, self_supervised_iterations=self_supervised_iterations,
                                         ser_thresh=ser_thresh, online_meta=online_meta, meta_lr=meta_lr, MAML=MAML, window_size=window_size,
                                         meta_train_iterations=meta_train_iterations, meta_j_num=meta_j_num, meta_subframes=meta_subframes,
                                         meta_style_online_training=meta_style_online_training, weights_init=weights_init,
                                         meta_training_weights=meta_training_weights), train_minibatch_size)
    cohorts = max(1, min(int(cohorts), R))
    bounds = [(c * R) // cohorts for c in range(cohorts + 1)]
    gens = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        rec = None if record is None else {k: record[k][lo:hi] for k in ("nerr", "trained", "meta")}
        gens.append(_cohort_steps(_BankRows(bank, lo, hi), tx[lo:hi], rx[lo:hi], n_symbols, subframes_in_frame, draws[lo:hi],
                                  ser_by_word[lo:hi], rec, self_supervised, self_supervised_iterations, ser_thresh, online_meta,
                                  meta_lr, MAML, window_size, meta_train_iterations, meta_j_num, meta_subframes,
                                  meta_style_online_training, train_minibatch_size, weights_init, meta_training_weights,
                                  None if initial_buffer is None else [t[lo:hi] for t in initial_buffer]))
    with _lib.on_device(rx.device):
        waiting = [next(g) for g in gens]  # every cohort has enqueued its first step and says which event ends it
        while gens:
            for i in range(len(gens)):  # round-robin: wait for cohort i's step, decide and enqueue, move on
                waiting[i].synchronize()
                try:
                    waiting[i] = next(gens[i])
                except StopIteration:
                    gens[i] = None
            waiting = [w for w, g in zip(waiting, gens) if g is not None]
            gens = [g for g in gens if g is not None]
    return ser_by_word


def _one_trial_at_a_time(bank, tx, rx, n_symbols, subframes_in_frame, draws, ser_by_word, record, initial_buffer, kw,
                         train_minibatch_size):
    """The trials of a bank the lock-step engine does not serve -- meta-learning with an optimizer its kernel does not implement
    (RMSprop, SGD) or another state count than 16 (no R-word block step) -- one after the other through harness.eval_by_word;
    weights, saved weights, optimizer state and step counts go back into the bank."""
    from .detectors import META_VNETDetector, VNETDetector
    from .harness import eval_by_word
    from .online import OnlineTrainer

    T = rx.shape[2]
    for r in range(bank.R):
        det = VNETDetector(bank.n_states, {"train": T, "val": T}).to(bank.device)
        with torch.no_grad():
            for p_, a in zip(det.parameters(), bank.weights(r)):
                p_.copy_(a)
        tr = OnlineTrainer(det, bank.memory_length, lr=bank.lr, betas=bank.betas, eps=bank.eps, train_minibatch_size=train_minibatch_size,
                           optimizer_type=bank.optimizer_type)
        tr.exp_avg.copy_(bank.exp_avg[r])
        tr.exp_avg_sq.copy_(bank.exp_avg_sq[r])
        tr.step = int(bank.step[r])
        last = {}

        def observer(seen, last=last):
            last.update(seen)
            if record is not None and seen["stage"] == "end":
                record["trained"][r, seen["count"]] = seen["trained"]
                record["meta"][r, seen["count"]] = seen["meta"] is not None

        ser_by_word[r] = eval_by_word(det, tx[r], rx[r], 0.0, 0.0, n_symbols, subframes_in_frame, online_trainer=tr,
                                      meta_detector=META_VNETDetector(bank.n_states, {"train": T, "val": T}), draws=draws[r],
                                      initial_buffer=None if initial_buffer is None else (initial_buffer[0][r], initial_buffer[1][r]),
                                      observer=observer, **kw)
        with torch.no_grad():
            bank.theta[r].copy_(torch.cat([p_.detach().reshape(-1) for p_ in det.parameters()]))
            saved = last.get("saved_detector")
            bank.saved[r].copy_(torch.cat([p_.detach().reshape(-1) for p_ in (saved if saved is not None else det).parameters()]))
            bank.exp_avg[r].copy_(tr.exp_avg)
            bank.exp_avg_sq[r].copy_(tr.exp_avg_sq)
        bank.step[r] = tr.step
    return ser_by_word


def _cohort_steps(bank, tx, rx, n_symbols, subframes_in_frame, draws, ser_by_word, record, self_supervised,
                  self_supervised_iterations, ser_thresh, online_meta, meta_lr, MAML, window_size, meta_train_iterations, meta_j_num,
                  meta_subframes, meta_style_online_training, train_minibatch_size, weights_init, meta_training_weights,
                  initial_buffer=None):
    """One group of trials stepping through its blocks: a generator that enqueues a step's GPU work and yields the event the
    host has to wait for before it can decide what the trials do next (eval_by_word_batched drives one or more of these)."""
    if bank.n_states != 16:
        raise NotImplementedError("the batched evaluation runs the 16-state kernels (mvn_vnet_byword_step_f32)")
    _lib.require_gpu_tensor(rx, "rx")
    lib = _lib.load()
    dev = rx.device
    R, N, T = rx.shape
    K = tx.shape[2]
    if R != bank.R or len(draws) != R or tx.shape[0] != R or tx.shape[1] != N or K != T - 8 * n_symbols:
        raise ValueError("tx [R, N, K], rx [R, N, K + 8 n_symbols], one TrialDraws and one bank row per trial")
    rx = _lib.f32c(rx)
    tx = _lib.f32c(tx).to(dev)
    S, W = bank.n_states, window_size
    b1, b2 = bank.betas
    eps_k = bank.eps
    if bank.optimizer_type == "RMSprop":  # the online-training kernel's encoding (include/mvn.h: MVN_BETA1_RMSPROP, alpha = beta2)
        b1, b2, eps_k = -1.0, 0.99, 1e-8
    elif bank.optimizer_type == "SGD":
        b1, b2, eps_k = -2.0, 0.0, 0.0
    full_word = meta_style_online_training
    M = 0 if full_word else train_minibatch_size

    # A trial's words: the W0 words its buffer starts with (buffer_empty=False), then its N blocks; word number = position here
    W0 = 0 if initial_buffer is None else int(initial_buffer[0].shape[1])
    NA = W0 + N
    labels = torch.zeros((R, NA, T), dtype=torch.int32, device=dev)  # calculate_states of every word's label word
    if W0:
        from .trellis import calculate_states

        rx = torch.cat([initial_buffer[1].to(dev), rx], dim=1).contiguous()
        labels[:, :W0] = calculate_states(bank.memory_length, initial_buffer[0].to(dev).reshape(R * W0, T)).reshape(R, W0, T).to(torch.int32)
    sync_dev = torch.zeros(2 * R, dtype=torch.int32, device=dev)     # [0:R] bit errors of the step, [R:2R] training status
    sync_host = torch.zeros(2 * R, dtype=torch.int32).pin_memory()
    nerr_np, status_np = sync_host.numpy()[:R], sync_host.numpy()[R:]
    ws_bytes = int(lib.mvn_vnet_train_trials_workspace_bytes(S, T, W, R))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    max_steps = meta_train_iterations * meta_j_num
    desc = _Staging(2 * R * TRIAL_DTYPE.itemsize, dev)               # [0:R] meta-learning descriptors, [R:2R] online training
    idx = _Staging(R * max_steps * (W + 1) * 4, dev) if online_meta else None
    d_meta = desc.np[:R * TRIAL_DTYPE.itemsize].view(TRIAL_DTYPE)
    d_onl = desc.np[R * TRIAL_DTYPE.itemsize:].view(TRIAL_DTYPE)
    desc_meta_ptr = ctypes.c_void_p(desc.dev.data_ptr())
    desc_onl_ptr = ctypes.c_void_p(desc.dev.data_ptr() + R * TRIAL_DTYPE.itemsize)
    theta_p, saved_p = bank.pointers(bank.theta), bank.pointers(bank.saved)
    m_p = np.uint64(bank.exp_avg.data_ptr()) + np.arange(R, dtype=np.uint64) * np.uint64(4 * bank.P)
    v_p = np.uint64(bank.exp_avg_sq.data_ptr()) + np.arange(R, dtype=np.uint64) * np.uint64(4 * bank.P)
    status_p = np.uint64(sync_dev.data_ptr()) + np.uint64(4) * (np.uint64(R) + np.arange(R, dtype=np.uint64))
    rx_p = np.uint64(rx.data_ptr()) + np.arange(R, dtype=np.uint64) * np.uint64(4 * NA * T)
    lab_p = np.uint64(labels.data_ptr()) + np.arange(R, dtype=np.uint64) * np.uint64(4 * NA * T)
    if weights_init == "meta_training":
        if meta_training_weights is None:
            raise ValueError("weights_init='meta_training' needs meta_training_weights (six arrays in parameters() order)")
        init_bank = TrialBank([meta_training_weights], S, bank.memory_length, dev)
        init_p = np.repeat(init_bank.pointers(init_bank.theta), R, axis=0)
    elif weights_init == "random":  # a row of fresh weights per trial, refilled from the trial's stream before every update
        init_host = torch.empty((R, bank.P), dtype=torch.float32).pin_memory()
        init_dev = torch.empty((R, bank.P), dtype=torch.float32, device=dev)
        init_p = (np.uint64(init_dev.data_ptr()) + np.arange(R, dtype=np.uint64)[:, None] * np.uint64(4 * bank.P)
                  + (bank.off[:6].astype(np.uint64) * np.uint64(4))[None, :])
    sup_off = np.arange(-W, 0)
    table_p = None
    w_stride = (ctypes.c_int64 * 6)(*([bank.P] * 6))
    wp = [ctypes.c_void_p(bank.theta.data_ptr() + 4 * int(bank.off[a])) for a in range(6)]
    stream = _lib.current_stream(dev)
    ts = torch.cuda.current_stream(dev)

    buffers: List[List[int]] = [list(range(W0)) for _ in range(R)]  # trial r's buffer: the word numbers it holds, oldest first
    tables = None
    done = torch.cuda.Event()  # (the driver holds the device guard)
    for count in range(N):
        pilot = 1 if count % subframes_in_frame == 0 else 0
        rc = lib.mvn_vnet_byword_step_f32(ctypes.c_void_p(rx.data_ptr() + 4 * (W0 + count) * T), NA * T,
                                          ctypes.c_void_p(tx.data_ptr() + 4 * count * K), N * K, *wp, w_stride,
                                          None, T, None, K, None, T, None, T,
                                          ctypes.c_void_p(labels.data_ptr() + 4 * (W0 + count) * T), NA * T,
                                          ctypes.c_void_p(sync_dev.data_ptr()), R, T, n_symbols, pilot, S, stream)
        _lib.check(rc, "mvn_vnet_byword_step_f32")
        sync_host.copy_(sync_dev, non_blocking=True)
        done.record(ts)
        yield done  # the one host sync of the step (the reference has one per trial and block, trainer.py:305)
        if status_np.any():
            raise _lib.MvnError(f"trials {np.flatnonzero(status_np).tolist()}: {lib.mvn_strerror(-7).decode()}")
        ser = ser_from_errors(nerr_np, K)  # the reference's value bit for bit (metrics.py:13-16)
        if not pilot:
            ser_by_word[:, count] = ser
        push = ser <= ser_thresh  # trainer.py:319-324
        for r in np.flatnonzero(push):
            buffers[r].append(W0 + count)
            if W0:  # buffer_empty=False: a window of fixed length, the oldest word leaves (:325-328)
                del buffers[r][0]
        if record is not None:
            record["nerr"][:, count] = nerr_np
        # ---- online meta-learning (trainer.py:331-343): restart from the saved weights, all steps in one launch
        if online_meta and count % meta_subframes == 0 and count >= meta_subframes:
            act = [r for r in range(R) if len(buffers[r]) > 2]
            if act:
                words = idx.np.view(np.int32)
                pos = 0
                offs, ns = np.empty(len(act), np.int64), np.empty(len(act), np.int32)
                for k, r in enumerate(act):
                    buf = np.asarray(buffers[r], dtype=np.int32)
                    j_hat = draws[r].j_hat_update(len(buf) - 2, meta_train_iterations, meta_j_num)
                    n = j_hat.shape[0]
                    # support j_hat + [-W .. -1], query j_hat: positions in the buffer, negative = from its end
                    words[pos:pos + n * W] = buf[(j_hat[:, None] + sup_off[None, :]) % len(buf)].reshape(-1)
                    words[pos + n * W:pos + n * (W + 1)] = buf[j_hat]
                    offs[k], ns[k] = pos, n
                    pos += n * (W + 1)
                    if record is not None:
                        record["meta"][r, count] = True
                a = np.asarray(act)
                if weights_init == "random":  # meta_weights_init('random'): fresh weights, fresh optimizer (trainer.py:356-359)
                    for r in act:
                        init_host[r].copy_(torch.cat([t.reshape(-1) for t in draws[r].init_weights(S)]))
                    init_dev.copy_(init_host, non_blocking=True)  # (rewritten only after this step's sync, which follows the copy)
                    a_dev = torch.as_tensor(a, device=dev)
                    bank.exp_avg.index_fill_(0, a_dev, 0.0)
                    bank.exp_avg_sq.index_fill_(0, a_dev, 0.0)
                    bank.step[a] = 0
                d = d_meta[:len(act)]
                d["y"], d["labels"] = rx_p[a], lab_p[a]
                d["idx"] = np.uint64(idx.dev.data_ptr()) + (4 * offs).astype(np.uint64)
                d["query_idx"] = np.uint64(idx.dev.data_ptr()) + (4 * (offs + ns.astype(np.int64) * W)).astype(np.uint64)
                d["w_in"] = saved_p[a] if weights_init == "last_frame" else init_p[a]
                d["w_out"], d["w_out2"] = theta_p[a], saved_p[a]
                d["adam_m"], d["adam_v"], d["loss_out"], d["status"] = m_p[a], v_p[a], 0, status_p[a]
                d["b1pow"], d["b2pow"] = beta_powers(b1, bank.step[a]), beta_powers(b2, bank.step[a])
                d["n"], d["reserved"] = ns, 0
                bank.step[a] += ns
                idx.send(4 * pos)
                desc.send(len(act) * TRIAL_DTYPE.itemsize)
                rc = lib.mvn_vnet_maml_train_trials_f32(desc_meta_ptr, len(act), T, W, meta_lr, 1 if MAML else 0, bank.lr,
                                                        b1, b2, bank.eps, S, _lib.ptr(ws), ws_bytes, stream)
                _lib.check(rc, "mvn_vnet_maml_train_trials_f32")
        # ---- self-supervised training on the word just buffered (trainer.py:345-347)
        if self_supervised and push.any():
            act = np.flatnonzero(push)
            if M and tables is None:
                for r in range(R):
                    draws[r].batches(0, N, T, self_supervised_iterations, M)  # draws the trial's table
                tables = [draws[r]._table for r in range(R)]
                table_p = np.array([t.data_ptr() for t in tables], dtype=np.uint64)
            d = d_onl[:len(act)]
            d["y"] = rx_p[act] + np.uint64(4 * (W0 + count) * T)
            d["labels"] = lab_p[act] + np.uint64(4 * (W0 + count) * T)
            d["idx"] = table_p[act] + np.uint64(4 * count * self_supervised_iterations * M) if M else 0
            d["query_idx"] = 0
            d["w_in"] = saved_p[act] if meta_style_online_training else theta_p[act]  # metavnet_trainer.py:59
            d["w_out"], d["w_out2"] = theta_p[act], 0
            d["adam_m"], d["adam_v"], d["loss_out"], d["status"] = m_p[act], v_p[act], 0, status_p[act]
            d["b1pow"], d["b2pow"] = beta_powers(b1, bank.step[act]), beta_powers(b2, bank.step[act])
            d["n"], d["reserved"] = self_supervised_iterations, 0
            bank.step[act] += self_supervised_iterations
            if record is not None:
                record["trained"][act, count] = True
            off = R * TRIAL_DTYPE.itemsize
            desc.dev[off:off + len(act) * TRIAL_DTYPE.itemsize].copy_(desc.host[off:off + len(act) * TRIAL_DTYPE.itemsize],
                                                                      non_blocking=True)
            rc = lib.mvn_vnet_online_train_trials_f32(desc_onl_ptr, len(act), T, M, bank.lr, b1, b2, eps_k, S,
                                                      _lib.ptr(ws), ws_bytes, stream)
            _lib.check(rc, "mvn_vnet_online_train_trials_f32")
    sync_host.copy_(sync_dev, non_blocking=True)
    done.record(ts)
    yield done
    if status_np.any():
        raise _lib.MvnError(f"trials {np.flatnonzero(status_np).tolist()}: {lib.mvn_strerror(-7).decode()}")
