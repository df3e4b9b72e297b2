"""Reed-Solomon outer code on the GPU: counterpart of python_code/ecc/rs_main.py (encode :9-18, decode :21-37),
batched over words so the detect -> RS-decode -> count loop of trainer.py:232-239 / :295-305 never leaves the device."""
import numpy as np
import torch

from . import _lib


def _bits(t: torch.Tensor) -> torch.Tensor:
    _lib.require_gpu_tensor(t, "bits")
    if t.dtype is not torch.float32:
        t = t.detach().to(torch.float32)
    if t.dim() == 1:
        t = t.reshape(1, -1)
    return t if t.stride(-1) == 1 else t.contiguous()


def rs_decode(detected_words: torch.Tensor, n_symbols: int, return_status: bool = False):
    """[B, K + 8*n_symbols] detected bits -> [B, K] decoded message bits (fp32 {0.,1.}), like
    `[decode(w, n_symbols) for w in detected_words]` in trainer.py:235."""
    rx = _bits(detected_words)
    B, N = rx.shape
    if N % 8 or N // 8 <= n_symbols:
        raise ValueError("word length must be a multiple of 8 bits and longer than the parity")
    if N // 8 > 255:
        raise ValueError("Message is too long (%i when max is 255)" % (N // 8))
    out = torch.empty((B, N - 8 * n_symbols), dtype=torch.float32, device=rx.device)
    status = torch.empty(B, dtype=torch.int32, device=rx.device) if return_status else None
    with _lib.on_device(rx.device):
        rc = _lib.load().mvn_rs_decode_bits_f32(_lib.ptr(rx), rx.stride(0), _lib.ptr(out), out.stride(0), _lib.ptr(status), B,
                                                N, n_symbols, _lib.current_stream(rx.device))
    _lib.check(rc, "mvn_rs_decode_bits_f32")
    return (out, status) if return_status else out


def rs_encode(words: torch.Tensor, n_symbols: int) -> torch.Tensor:
    """[B, K] message bits -> [B, K + 8*n_symbols] systematic codewords (rs_main.py:9-18, trainer.py:304)."""
    msg = _bits(words)
    B, K = msg.shape
    if K % 8:
        raise ValueError("word length must be a multiple of 8 bits")
    if K // 8 + n_symbols > 255:
        raise ValueError("Message is too long (%i when max is 255)" % (K // 8 + n_symbols))
    out = torch.empty((B, K + 8 * n_symbols), dtype=torch.float32, device=msg.device)
    with _lib.on_device(msg.device):
        rc = _lib.load().mvn_rs_encode_bits_f32(_lib.ptr(msg), msg.stride(0), _lib.ptr(out), out.stride(0), B, K, n_symbols,
                                                _lib.current_stream(msg.device))
    _lib.check(rc, "mvn_rs_encode_bits_f32")
    return out


def encode(binary_word: np.ndarray, nsym: int) -> np.ndarray:
    """Single-word NumPy signature of the reference (rs_main.py:9)."""
    dev = torch.device("cuda")
    return rs_encode(torch.as_tensor(np.asarray(binary_word), dtype=torch.float32, device=dev), nsym)[0].cpu().numpy().astype(int)


def decode(binary_rx: np.ndarray, nsym: int) -> np.ndarray:
    """Single-word NumPy signature of the reference (rs_main.py:21)."""
    dev = torch.device("cuda")
    return rs_decode(torch.as_tensor(np.asarray(binary_rx), dtype=torch.float32, device=dev), nsym)[0].cpu().numpy().astype(int)
