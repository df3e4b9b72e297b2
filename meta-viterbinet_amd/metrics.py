"""Error metrics: counterpart of python_code/utils/metrics.py:7-17, counted on the device as integers."""
from typing import Tuple

import torch

from . import _lib


def count_errors(prediction: torch.Tensor, target: torch.Tensor, rows: torch.Tensor = None,
                 counters: torch.Tensor = None) -> torch.Tensor:
    """Accumulates int64[4] = {bit_errors, bits, frame_errors, frames} over `rows` (default: all rows)
    on the device; no host sync.  These are the counters the multi-GPU all-reduce carries."""
    _lib.require_gpu_tensor(prediction, "prediction")
    p = prediction if prediction.dtype is torch.float32 else prediction.detach().to(torch.float32)
    t = target if (target.dtype is torch.float32 and target.device == p.device) else target.detach().to(device=p.device, dtype=torch.float32)
    if p.stride(-1) != 1:
        p = p.contiguous()
    if t.stride(-1) != 1:
        t = t.contiguous()
    if p.shape[1] != t.shape[1]:
        raise RuntimeError(f"The size of tensor a ({p.shape[1]}) must match the size of tensor b ({t.shape[1]}) "
                           "at non-singleton dimension 1")
    if counters is None:
        counters = torch.zeros(4, dtype=torch.int64, device=p.device)
    r = None if rows is None else rows.to(device=p.device, dtype=torch.int64).contiguous()
    n = p.shape[0] if r is None else r.numel()
    with _lib.on_device(p.device):
        rc = _lib.load().mvn_count_errors(_lib.ptr(p), p.stride(0), _lib.ptr(t), t.stride(0), _lib.ptr(r), n,
                                          p.shape[1], _lib.ptr(counters), _lib.current_stream(p.device))
    _lib.check(rc, "mvn_count_errors")
    return counters


def ser_from_errors(n_errors, n_bits: int):
    """The reference's per-block ser from an error count, bit for bit: calculate_error_rates (metrics.py:11-16) takes the
    fp32 mean of the `equal` mask -- (n_bits - n_errors) / n_bits rounded to fp32 -- and returns max(1 - that, 0) in Python
    floats.  n_errors: int or integer array; returns float64 of the same shape (exactly the values the reference's
    eval_by_word stores in ser_by_word, golden G7 / G9)."""
    import numpy as np

    acc = (np.float32(n_bits) - np.asarray(n_errors).astype(np.float32)) / np.float32(n_bits)
    return np.maximum(1.0 - acc.astype(np.float64), 0.0)


def rates_from_counters(counters) -> Tuple[float, float]:
    """(ser, fer) = (bit_errors/bits, frame_errors/frames) in float64.  The reference takes fp32 means
    (metrics.py:13,15): the two agree to ~1e-7 relative."""
    be, bits, fe, frames = [int(v) for v in counters.tolist()]
    ser = max(be / bits, 0.0) if bits else float("nan")
    fer = max(fe / frames, 0.0) if frames else float("nan")
    return ser, fer


def calculate_error_rates(prediction: torch.Tensor, target: torch.Tensor) -> Tuple[float, float, torch.Tensor]:
    """Returns the ber, fer and indices of errored rows (metrics.py:7-17)."""
    counters = count_errors(prediction, target)
    ser, fer = rates_from_counters(counters)
    wrong = (prediction.long() != target.to(prediction.device).long()).any(dim=1)
    return ser, fer, torch.nonzero(wrong, as_tuple=False).reshape(-1)
