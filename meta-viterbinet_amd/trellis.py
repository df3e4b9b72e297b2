"""Trellis primitives: counterpart of python_code/utils/trellis_utils.py."""
import numpy as np
import torch

from . import _lib


def create_transition_table(n_states: int) -> np.ndarray:
    """[n_states,2] table; row s = the two predecessor states [(2s)%n, (2s+1)%n]
    (trellis_utils.py:7-13).  The kernels bake this map in; the array is kept for API parity."""
    return np.concatenate([np.arange(n_states), np.arange(n_states)]).reshape(n_states, 2)


def acs_block(in_prob: torch.Tensor, llrs: torch.Tensor, transition_table: torch.Tensor = None,
              n_states: int = None):
    """One Viterbi ACS stage on the GPU (trellis_utils.py:16-30):
    out[b,s] = min_j (in_prob+llrs)[b, (2s+j)%S]; returns (values [B,S] f32, argmin_j [B,S] i64)
    like torch.min(dim=2).  `transition_table` is accepted for signature parity and ignored."""
    _lib.require_gpu_tensor(in_prob, "in_prob")
    ip = in_prob.detach().to(torch.float32).contiguous()
    c = llrs.detach().to(device=ip.device, dtype=torch.float32).expand_as(ip).contiguous()
    B, S = ip.shape
    if n_states is not None and n_states != S:
        raise ValueError("n_states does not match in_prob.shape[1]")
    out = torch.empty_like(ip)
    idx = torch.empty((B, S), dtype=torch.int64, device=ip.device)
    with torch.cuda.device(ip.device):
        rc = _lib.load().mvn_acs_block_f32(_lib.ptr(ip), _lib.ptr(c), _lib.ptr(out), _lib.ptr(idx), B, S,
                                           _lib.current_stream(ip.device))
    _lib.check(rc, "mvn_acs_block_f32")
    return out, idx


def acs_sweep(cost: torch.Tensor, return_final: bool = False):
    """The T-step decision/ACS loop (va_detector.py:89-97) over materialised costs [B,T,S]."""
    _lib.require_gpu_tensor(cost, "cost")
    c = cost.detach().to(torch.float32).contiguous()
    B, T, S = c.shape
    dec = torch.zeros((B, T), dtype=torch.float32, device=c.device)
    fm = torch.empty((B, S), dtype=torch.float32, device=c.device) if return_final else None
    with torch.cuda.device(c.device):
        rc = _lib.load().mvn_acs_sweep_f32(_lib.ptr(c), _lib.ptr(dec), T, _lib.ptr(fm), B, T, S,
                                           _lib.current_stream(c.device))
    _lib.check(rc, "mvn_acs_sweep_f32")
    return (dec, fm) if return_final else dec


def acs_sweep_survivors(cost: torch.Tensor):
    """acs_sweep that also keeps what the reference's acs_block returns and drops (trellis_utils.py:30: torch.min's indices):
    returns (dec [B,T], final metrics [B,S], surv uint8 [B,T,max(1,S/8)]), bit (s & 7) of surv[b,t,s >> 3] = j with the surviving
    predecessor of state s at stage t being (2 s + j) % S.  dec and the metrics are those of acs_sweep."""
    _lib.require_gpu_tensor(cost, "cost")
    c = cost.detach().to(torch.float32).contiguous()
    B, T, S = c.shape
    dec = torch.zeros((B, T), dtype=torch.float32, device=c.device)
    fm = torch.empty((B, S), dtype=torch.float32, device=c.device)
    surv = torch.empty((B, T, max(1, S // 8)), dtype=torch.uint8, device=c.device)
    with torch.cuda.device(c.device):
        rc = _lib.load().mvn_acs_sweep_surv_f32(_lib.ptr(c), _lib.ptr(dec), T, _lib.ptr(fm), _lib.ptr(surv), B, T, S,
                                                _lib.current_stream(c.device))
    _lib.check(rc, "mvn_acs_sweep_surv_f32")
    return dec, fm, surv


def traceback(surv: torch.Tensor, final_metric: torch.Tensor, return_states: bool = False):
    """The textbook Viterbi traceback over the survivors of acs_sweep_survivors / VADetector.viterbi_path: from
    torch.argmin(final_metric[b]) back to stage 0.  Returns bits [B,T] fp32 {0.,1.} -- bits[b,t] = the least-significant bit of the
    maximum-likelihood path's state before stage t, i.e. symbol t's bit (trellis_utils.py:33-46) -- and, on request, those states
    (int32 [B,T])."""
    _lib.require_gpu_tensor(surv, "surv")
    sv = surv.contiguous()
    fm = final_metric.detach().to(torch.float32).contiguous()
    B, T = sv.shape[:2]
    S = fm.shape[1]
    if sv.dtype != torch.uint8 or sv.shape[2] != max(1, S // 8) or fm.shape[0] != B:
        raise ValueError("surv: uint8 [B, T, max(1, S/8)]; final_metric: [B, S]")
    bits = torch.zeros((B, T), dtype=torch.float32, device=sv.device)
    states = torch.empty((B, T), dtype=torch.int32, device=sv.device) if return_states else None
    with torch.cuda.device(sv.device):
        rc = _lib.load().mvn_traceback_f32(_lib.ptr(sv), _lib.ptr(fm), _lib.ptr(bits), T, _lib.ptr(states), B, T, S,
                                           _lib.current_stream(sv.device))
    _lib.check(rc, "mvn_traceback_f32")
    return (bits, states) if return_states else bits


def calculate_states(memory_length: int, transmitted_words: torch.Tensor) -> torch.Tensor:
    """Ground-truth state labels state[t] = sum_i 2^i b[t+i] (trellis_utils.py:33-46); training-side
    helper (vnet_trainer.py:44), plain torch on whatever device the words live on."""
    w = transmitted_words
    padded = torch.cat([w, torch.zeros([w.shape[0], memory_length], device=w.device, dtype=w.dtype)], dim=1)
    T = w.shape[1]
    weights = (2 ** torch.arange(memory_length, device=w.device)).to(torch.float32)
    windows = torch.stack([padded[:, i:i + T] for i in range(memory_length)], dim=2).to(torch.float32)
    return torch.sum(windows * weights, dim=2).reshape(-1).long()
