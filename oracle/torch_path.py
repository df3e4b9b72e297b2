"""Op-for-op PyTorch-CPU restatement of the reference's 'val' path (CPU oracle, TEST INFRASTRUCTURE ONLY).

This is the stand-in for "the reference's CPU/PyTorch path" in timings on machines where the reference itself is not
present (BASELINE.md section 3, item alpha): the same ATen op sequence per trellis stage as
python_code/utils/trellis_utils.py:16-30 inside the loops of python_code/detectors/VA/va_detector.py:89-97 and
python_code/detectors/VNET/vnet_detector.py:49-59 -- index tensors rebuilt every stage, advanced-index gather,
torch.min over the last dim, argmin % 2 written column by column.  Outputs are checked bit-for-bit against the golden
vectors captured from the reference (tests/test_oracle_golden.py)."""
import math

import torch


def _stage(metrics: torch.Tensor, branch: torch.Tensor, table: torch.Tensor, n_states: int) -> torch.Tensor:
    flat_prev = table.reshape(-1).repeat(metrics.size(0)).long()
    flat_rows = torch.arange(metrics.size(0)).repeat_interleave(2 * n_states)
    candidates = (metrics + branch)[flat_rows, flat_prev].reshape(-1, n_states, 2)
    return torch.min(candidates, dim=2)[0]


def _table(n_states: int) -> torch.Tensor:
    idx = torch.arange(n_states)
    return torch.cat([idx, idx]).reshape(n_states, 2).float()


def _sweep(costs: torch.Tensor, steps: int) -> torch.Tensor:
    n_rows, _, n_states = costs.shape
    table = _table(n_states)
    metrics = torch.zeros([n_rows, n_states])
    out = torch.zeros([n_rows, costs.shape[1]])
    for i in range(steps):
        out[:, i] = torch.argmin(metrics, dim=1) % 2
        metrics = _stage(metrics, costs[:, i], table, n_states)
    return out


@torch.no_grad()
def va_val(y: torch.Tensor, state_priors_ws: torch.Tensor, steps: int = None) -> torch.Tensor:
    """VADetector.forward(y,'val') given the [W,S] prior table (va_detector.py:64-68,89-97)."""
    pri = state_priors_ws.repeat(repeats=[y.shape[0] // state_priors_ws.shape[0], 1]).unsqueeze(dim=1)
    costs = (y.unsqueeze(dim=2) - pri) ** 2 / 2 - math.log(math.sqrt(2 * math.pi))
    return _sweep(costs, y.shape[1] if steps is None else steps)


@torch.no_grad()
def vnet_val(y: torch.Tensor, weights, steps: int = None) -> torch.Tensor:
    """VNETDetector.forward(y,'val') with weights [W1,b1,W2,b2,W3,b3] (vnet_detector.py:49-59)."""
    W1, b1, W2, b2, W3, b3 = [torch.as_tensor(w) for w in weights]
    x = torch.nn.functional.linear(y.reshape(-1, 1), W1, b1)
    x = torch.sigmoid(x)
    x = torch.relu(torch.nn.functional.linear(x, W2, b2))
    logits = torch.nn.functional.linear(x, W3, b3).reshape(y.shape[0], y.shape[1], W3.shape[0])
    return _sweep(-logits, y.shape[1] if steps is None else steps)
