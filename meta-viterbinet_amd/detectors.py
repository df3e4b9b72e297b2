"""Drop-in detectors: same constructors, forward() signatures, parameter names and exceptions as
python_code/detectors/{VA/va_detector.py, VNET/vnet_detector.py, META_VNET/meta_vnet_detector.py};
phase 'val' runs on the MI355X through libmvn_hip.so, phase 'train' is plain torch (autograd)."""
import math
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from . import _lib
from .channel import BPSKModulator, estimate_channel
from .trellis import create_transition_table

HIDDEN1_SIZE = 100  # vnet_detector.py:7
HIDDEN2_SIZE = 50  # vnet_detector.py:8
_WORKSPACE_CAP = 2 << 30  # bytes of logits scratch per call on the non-fused path


def _default_device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


_as_f32 = _lib.f32c


def _new_decisions(y: torch.Tensor, T: int) -> torch.Tensor:
    """decoded_word = torch.zeros(y.shape) (va_detector.py:90); the zero fill is skipped when the kernel writes every
    column (T == y.shape[1]) and there is at least one row."""
    if T == y.shape[1] and y.shape[0] > 0 and T > 0:
        return torch.empty(y.shape, dtype=torch.float32, device=y.device)
    return torch.zeros(y.shape, dtype=torch.float32, device=y.device)


def _check_T(T: int, y: torch.Tensor):
    if T > y.shape[1]:  # the reference indexes priors[:, i] for i < transmission_length (Q5)
        raise IndexError(f"index {y.shape[1]} is out of bounds for dimension 1 with size {y.shape[1]}")


class VADetector(nn.Module):
    """Classic full-CSI Viterbi detector (va_detector.py:13-100)."""

    def __init__(self, n_states: int, memory_length: int, transmission_length: int, val_words: int,
                 channel_type: str, noisy_est_var: float, fading: bool, fading_taps_type: int,
                 channel_coefficients):
        super().__init__()
        self.memory_length = memory_length
        self.transmission_length = transmission_length
        self.val_words = val_words
        self.n_states = n_states
        self.channel_type = channel_type
        self.noisy_est_var = noisy_est_var
        self.fading = fading
        self.fading_taps_type = fading_taps_type
        self.channel_coefficients = channel_coefficients  # dict {'train':..., 'val':...} (trainer.py:195)
        self.transition_table_array = create_transition_table(n_states)
        self.transition_table = torch.Tensor(self.transition_table_array).to(_default_device())
        self._priors_cache = {}

    def _estimate_all(self, gamma: float, phase: str) -> np.ndarray:
        """[val_words, L] float64 taps, one estimate_channel call per word (va_detector.py:54-58)."""
        return np.concatenate(
            [estimate_channel(self.memory_length, gamma, noisy_est_var=self.noisy_est_var, fading=self.fading,
                              index=index, fading_taps_type=self.fading_taps_type,
                              channel_coefficients=self.channel_coefficients[phase])
             for index in range(self.val_words)], axis=0)

    def compute_state_priors(self, h: np.ndarray) -> torch.Tensor:
        """Noise-free channel output per state, [n_states, W] fp32 (va_detector.py:42-50)."""
        all_states_decimal = np.arange(self.n_states).astype(np.uint8).reshape(-1, 1)
        all_states_binary = np.unpackbits(all_states_decimal, axis=1).astype(int)
        if self.channel_type == "ISI_AWGN":
            all_states_symbols = BPSKModulator.modulate(all_states_binary[:, -self.memory_length:])
        else:
            raise Exception("No such channel defined!!!")
        state_priors = np.dot(all_states_symbols, h.T)
        return torch.Tensor(state_priors).to(_default_device())

    def compute_likelihood_priors(self, y: torch.Tensor, snr: float, gamma: float, phase: str, count: int = None):
        """Materialised branch costs [B,T,S] (va_detector.py:52-71).  forward() does NOT use this (the
        kernel computes the same four fp32 ops in registers); kept for API parity / inspection."""
        pri = self._priors_table(y, gamma, phase, count)  # [W,S]
        priors = y.unsqueeze(dim=2) - pri.repeat(repeats=[y.shape[0] // pri.shape[0], 1]).unsqueeze(dim=1)
        return priors ** 2 / 2 - math.log(math.sqrt(2 * math.pi))

    def _priors_table(self, y, gamma, phase, count):
        """[W,S] state priors on y's device ([1,S] for one word when `count` is given).  The reference re-derives the
        channel of every word on the host at every forward (va_detector.py:54-58); with noisy_est_var == 0 that table is a
        pure function of (gamma, phase) and the detector's configuration, so it is derived once and kept on the device --
        a by-word evaluation then costs one row slice per call instead of val_words x estimate_channel + a blocking
        upload.  noisy_est_var > 0 draws from the global RNG at every call (channel_estimation.py:36): never cached."""
        if self.channel_type != "ISI_AWGN":
            raise Exception("No such channel defined!!!")
        key = None
        if not self.noisy_est_var > 0:
            key = (float(gamma), phase, self.channel_coefficients[phase], self.val_words, self.memory_length, self.n_states,
                   bool(self.fading), self.fading_taps_type, str(y.device))
        table = self._priors_cache.get(key) if key is not None else None
        if table is None:
            table = self.compute_state_priors(self._estimate_all(gamma, phase)).to(y.device).T.contiguous()  # [W,S]
            if key is not None:
                self._priors_cache = {key: table}  # one entry: the configuration an evaluation is running with
        return table if count is None else table[count].reshape(1, -1)

    def forward(self, y: torch.Tensor, phase: str, snr: float = None, gamma: float = None,
                count: int = None) -> torch.Tensor:
        """Detected words, same shape as y, fp32 {0.,1.} (va_detector.py:73-100)."""
        if phase != "val":
            raise NotImplementedError("No implemented training for this decoder!!!")
        _lib.require_gpu_tensor(y, "y")
        yc = _as_f32(y)
        B, Ty = yc.shape
        T = self.transmission_length
        pri = self._priors_table(yc, gamma, phase, count)
        W = pri.shape[0]
        if B % W != 0:
            raise RuntimeError(f"The size of tensor a ({B}) must match the size of tensor b ({W * (B // W)}) "
                               "at non-singleton dimension 0")
        _check_T(T, yc)
        decoded_word = _new_decisions(yc, T)
        with _lib.on_device(yc.device):
            rc = _lib.load().mvn_va_decode_f32(_lib.ptr(yc), Ty, _lib.ptr(pri), W, _lib.ptr(decoded_word), Ty,
                                               None, B, T, self.n_states, _lib.current_stream(yc.device))
        _lib.check(rc, "mvn_va_decode_f32")
        return decoded_word


def _va_viterbi_path(self, y: torch.Tensor, snr: float = None, gamma: float = None, count: int = None, return_all: bool = False):
    """Textbook Viterbi detection with survivor-path traceback -- NOT what the reference's VADetector.forward returns (it decides
    every symbol from a running argmin and drops acs_block's survivor indices, va_detector.py:93-95): the same
    sweep with the survivors kept (mvn_va_decode_surv_f32), then the traceback from the best final metric.  Returns the
    maximum-likelihood bits [B, y.shape[1]] (columns >= transmission_length zero), or with return_all
    (bits, running decisions = forward(y,'val'), final metrics, survivors)."""
    from .trellis import traceback

    _lib.require_gpu_tensor(y, "y")
    yc = _as_f32(y)
    B, Ty = yc.shape
    T = self.transmission_length
    pri = self._priors_table(yc, gamma, "val", count)
    W = pri.shape[0]
    if B % W != 0:
        raise RuntimeError(f"The size of tensor a ({B}) must match the size of tensor b ({W * (B // W)}) at non-singleton dimension 0")
    _check_T(T, yc)
    dec = _new_decisions(yc, T)
    S = self.n_states
    fm = torch.empty((B, S), dtype=torch.float32, device=yc.device)
    surv = torch.empty((B, T, max(1, S // 8)), dtype=torch.uint8, device=yc.device)
    with _lib.on_device(yc.device):
        rc = _lib.load().mvn_va_decode_surv_f32(_lib.ptr(yc), Ty, _lib.ptr(pri), W, _lib.ptr(dec), Ty, _lib.ptr(fm), _lib.ptr(surv),
                                                B, T, S, _lib.current_stream(yc.device))
    _lib.check(rc, "mvn_va_decode_surv_f32")
    bits = torch.zeros_like(yc)
    bits[:, :T] = traceback(surv, fm)
    return (bits, dec, fm, surv) if return_all else bits


VADetector.viterbi_path = _va_viterbi_path


def _weights_on(params, device, n_states):
    w = [_as_f32(p) if p.device == device else _as_f32(p).to(device) for p in params]
    if (len(w) != 6 or w[0].shape != (HIDDEN1_SIZE, 1) or w[1].shape != (HIDDEN1_SIZE,) or w[2].shape != (HIDDEN2_SIZE, HIDDEN1_SIZE)
            or w[3].shape != (HIDDEN2_SIZE,) or w[4].shape != (n_states, HIDDEN2_SIZE) or w[5].shape != (n_states,)):
        shapes = [(HIDDEN1_SIZE, 1), (HIDDEN1_SIZE,), (HIDDEN2_SIZE, HIDDEN1_SIZE), (HIDDEN2_SIZE,), (n_states, HIDDEN2_SIZE), (n_states,)]
        raise ValueError(f"ViterbiNet parameter shapes {[tuple(t.shape) for t in w]} != {shapes}")
    return w


def _vnet_val(y: torch.Tensor, params, n_states: int, T: int, return_logits: bool = False):
    """Shared 'val' path of VNETDetector / META_VNETDetector -> mvn_vnet_decode_f32."""
    _lib.require_gpu_tensor(y, "y")
    yc = _as_f32(y)
    B, Ty = yc.shape
    _check_T(T, yc)
    w = _weights_on(params, yc.device, n_states)  # read at call time, never cached (python_utils.py:17-27)
    lib = _lib.load()
    decoded_word = _new_decisions(yc, T)
    logits = torch.empty((B, T, n_states), dtype=torch.float32, device=yc.device) if return_logits else None
    ws, ws_bytes = None, 0
    if logits is None:
        ws_bytes = int(lib.mvn_vnet_workspace_bytes(B, T, n_states))
        if ws_bytes:  # 0: a fused kernel that needs no scratch
            if n_states != 16:  # the two-kernel route's logits, in slices of at most _WORKSPACE_CAP (16 states: <= 100 KB of hand-off lines)
                ws_bytes = max(min(ws_bytes, _WORKSPACE_CAP), T * n_states * 4)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=yc.device)
    with _lib.on_device(yc.device):
        rc = lib.mvn_vnet_decode_f32(_lib.ptr(yc), Ty, *[_lib.ptr(t) for t in w], _lib.ptr(decoded_word), Ty,
                                     _lib.ptr(logits), None, _lib.ptr(ws), ws_bytes, B, T, n_states,
                                     _lib.current_stream(yc.device))
    _lib.check(rc, "mvn_vnet_decode_f32")
    return (decoded_word, logits) if return_logits else decoded_word


def _vnet_val_count(y, params, n_states, T, tx, rows=None, counters=None, return_decisions=False):
    """One Monte-Carlo step in one launch (mvn_vnet_decode_count_f32, 16 states): decode + on-device error
    counting over the first tx.shape[1] columns of the rows in `rows` (None = all).  Decisions are only stored
    when asked for.  Returns the int64[4] counters tensor (and the decisions)."""
    _lib.require_gpu_tensor(y, "y")
    yc = _as_f32(y)
    B, Ty = yc.shape
    _check_T(T, yc)
    w = _weights_on(params, yc.device, n_states)
    txc = _as_f32(tx).to(yc.device)
    if txc.shape[0] != B or txc.shape[1] > T:
        raise ValueError("tx must be [B, K<=T]")
    if counters is None:
        counters = torch.zeros(4, dtype=torch.int64, device=yc.device)
    mask = None
    if rows is not None:
        mask = torch.zeros(B, dtype=torch.uint8, device=yc.device)
        mask[rows.to(yc.device)] = 1
    dec = torch.zeros(yc.shape, dtype=torch.float32, device=yc.device) if return_decisions else None
    ws_bytes = int(_lib.load().mvn_vnet_workspace_bytes(B, T, n_states)) if n_states == 16 else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=yc.device) if ws_bytes else None  # the dealt kernel's hand-off lines
    with _lib.on_device(yc.device):
        rc = _lib.load().mvn_vnet_decode_count_f32(_lib.ptr(yc), Ty, *[_lib.ptr(t) for t in w], _lib.ptr(txc),
                                                   txc.stride(0), txc.shape[1], _lib.ptr(mask), _lib.ptr(counters),
                                                   _lib.ptr(dec), Ty, _lib.ptr(ws), ws_bytes, B, T, n_states,
                                                   _lib.current_stream(yc.device))
    _lib.check(rc, "mvn_vnet_decode_count_f32")
    return (counters, dec) if return_decisions else counters


class VNETDetector(nn.Module):
    """ViterbiNet: the VA sweep with branch metrics from a per-symbol MLP (vnet_detector.py:11-63).
    Parameter names net.{0,2,4}.{weight,bias} match the reference so checkpoints interchange."""

    def __init__(self, n_states: int, transmission_lengths: Dict[str, int]):
        super().__init__()
        self.transmission_lengths = transmission_lengths
        self.n_states = n_states
        self.transition_table_array = create_transition_table(n_states)
        self.transition_table = torch.Tensor(self.transition_table_array).to(_default_device())
        self.initialize_dnn()

    def initialize_dnn(self):
        layers = [nn.Linear(1, HIDDEN1_SIZE), nn.Sigmoid(), nn.Linear(HIDDEN1_SIZE, HIDDEN2_SIZE), nn.ReLU(),
                  nn.Linear(HIDDEN2_SIZE, self.n_states)]
        self.net = nn.Sequential(*layers).to(_default_device())

    def _params(self):
        """The six Parameter OBJECTS in parameters() order, looked up once per `net` (walking the module tree costs more
        than a B=1 launch); their storage is still read at call time -- copy_model / load_state_dict / optimizers write
        through the same objects."""
        c = self.__dict__.get("_param_cache")
        if c is None or c[0] is not self.net:
            c = (self.net, list(self.net.parameters()))
            self.__dict__["_param_cache"] = c
        return c[1]

    def forward(self, y: torch.Tensor, phase: str, snr: float = None, gamma: float = None,
                count: int = None) -> torch.Tensor:
        """'val' -> detected words [B,T]; otherwise the logits [B,T,S] with autograd (vnet_detector.py:35-63)."""
        if phase == "val":
            return _vnet_val(y, self._params(), self.n_states, self.transmission_lengths["val"])
        return self.net(y.reshape(-1, 1)).reshape(y.shape[0], y.shape[1], self.n_states)

    def val_count(self, y, tx, rows=None, counters=None, return_decisions=False):
        """forward(y,'val') + calculate_error_rates fused in one launch (16 states); see _vnet_val_count."""
        return _vnet_val_count(y, self._params(), self.n_states, self.transmission_lengths["val"], tx, rows,
                               counters, return_decisions)

    @torch.no_grad()
    def logits(self, y: torch.Tensor) -> torch.Tensor:
        """Inference-only logits [*, S] from the HIP MLP kernel (bit-identical to the 'val' path's)."""
        _lib.require_gpu_tensor(y, "y")
        yc = _as_f32(y).reshape(-1)
        w = [_as_f32(p).to(yc.device) for p in self.net.parameters()]
        out = torch.empty((yc.numel(), self.n_states), dtype=torch.float32, device=yc.device)
        with _lib.on_device(yc.device):
            rc = _lib.load().mvn_vnet_logits_f32(_lib.ptr(yc), *[_lib.ptr(t) for t in w], _lib.ptr(out),
                                                 yc.numel(), self.n_states, _lib.current_stream(yc.device))
        _lib.check(rc, "mvn_vnet_logits_f32")
        return out.reshape(tuple(y.shape) + (self.n_states,))


def _vnet_viterbi_path(self, y: torch.Tensor, return_all: bool = False, var=None):
    """ViterbiNet detection with survivor-path traceback: the maximum-likelihood path through the learned branch metrics (cost =
    -logit, vnet_detector.py:57) -- the textbook ViterbiNet decision, NOT what the reference's forward(y, 'val') returns (it decides
    every symbol from a running argmin and drops acs_block's survivor indices, vnet_detector.py:55-57).  The same
    logits and the same sweep with the survivors kept (mvn_vnet_decode_surv_f32), then mvn_traceback_f32 from the best final
    metric.  Returns bits [B, y.shape[1]] (columns >= transmission_lengths['val'] zero), or with return_all (bits, running
    decisions = forward(y,'val'), final metrics, survivors).  var: the six weight arrays (META_VNETDetector's calling form)."""
    from .trellis import traceback

    _lib.require_gpu_tensor(y, "y")
    yc = _as_f32(y)
    B, Ty = yc.shape
    T, S = self.transmission_lengths["val"], self.n_states
    _check_T(T, yc)
    w = _weights_on(self._params() if var is None else var, yc.device, S)
    dec = _new_decisions(yc, T)
    fm = torch.empty((B, S), dtype=torch.float32, device=yc.device)
    surv = torch.empty((B, T, max(1, S // 8)), dtype=torch.uint8, device=yc.device)
    # hand-off lines of the fused detector (16 states) or the logits' scratch of the two-kernel route
    ws = torch.empty(max(int(_lib.load().mvn_vnet_surv_workspace_bytes(B, T, S)), 16), dtype=torch.uint8, device=yc.device)
    with _lib.on_device(yc.device):
        rc = _lib.load().mvn_vnet_decode_surv_f32(_lib.ptr(yc), Ty, *[_lib.ptr(t) for t in w], _lib.ptr(dec), Ty, _lib.ptr(fm), _lib.ptr(surv),
                                                  _lib.ptr(ws), ws.numel(), B, T, S, _lib.current_stream(yc.device))
    _lib.check(rc, "mvn_vnet_decode_surv_f32")
    bits = torch.zeros_like(yc)
    bits[:, :T] = traceback(surv, fm)
    return (bits, dec, fm, surv) if return_all else bits


VNETDetector.viterbi_path = _vnet_viterbi_path


class META_VNETDetector(nn.Module):
    """Functional ViterbiNet: weights arrive as var=[W1,b1,W2,b2,W3,b3] so MAML can differentiate
    through an inner step (meta_vnet_detector.py:11-47).  Owns no parameters."""

    def __init__(self, n_states: int, transmission_lengths: Dict[str, int]):
        super().__init__()
        self.transmission_lengths = transmission_lengths
        self.n_states = n_states
        self.transition_table_array = create_transition_table(n_states)
        self.transition_table = torch.Tensor(self.transition_table_array).to(_default_device())

    def forward(self, y: torch.Tensor, phase: str, var: list) -> torch.Tensor:
        if phase == "val":
            return _vnet_val(y, list(var), self.n_states, self.transmission_lengths["val"])
        x = y.reshape(-1, 1)
        x = F.linear(x, var[0], var[1])
        x = torch.sigmoid(x)
        x = F.linear(x, var[2], var[3])
        x = F.relu(x)
        x = F.linear(x, var[4], var[5])
        return x.reshape(y.shape[0], y.shape[1], self.n_states)
