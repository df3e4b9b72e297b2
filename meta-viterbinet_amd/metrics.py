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
