"""Host-side channel-tap model needed by the full-CSI VA detector: counterpart of
python_code/channel/channel_estimation.py:11-49 and python_code/channel/modulator.py:12.
Tiny float64 NumPy work, evaluated once per forward() on the host exactly like the reference."""
import os

import numpy as np

COST_LENGTH = 300  # channel_estimation.py:8
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cost2100_taps.npy")
COST2100_DIR = None  # optional directory holding combined_h_{i}.mat (the reference's loader layout)
_cost_cache = {}


def _cost2100_table(memory_length: int) -> np.ndarray:
    key = (COST2100_DIR, memory_length)
    if key not in _cost_cache:
        if COST2100_DIR is not None:
            import scipy.io

            total_h = np.empty([COST_LENGTH, memory_length])
            for i in range(memory_length):
                total_h[:, i] = scipy.io.loadmat(os.path.join(COST2100_DIR, f"combined_h_{i}"))[
                    "h_channel_response_mag"].reshape(-1)
        else:
            total_h = np.load(_DATA)  # [300,4] float64 capture of resources/cost2100_channel/h_{0..3}.mat
            if memory_length != total_h.shape[1]:
                raise ValueError("cost2100 taps are recorded for memory_length 4 only")
        _cost_cache[key] = total_h
    return _cost_cache[key]


def estimate_channel(memory_length: int, gamma: float, channel_coefficients: str, noisy_est_var: float = 0,
                     fading: bool = False, index: int = 0, fading_taps_type: int = 1) -> np.ndarray:
    """[1,memory_length] float64 taps (channel_estimation.py:11-49)."""
    if channel_coefficients == "time_decay":
        h = np.reshape(np.exp(-gamma * np.arange(memory_length)), [1, memory_length])
    elif channel_coefficients == "cost2100":
        h = np.reshape(_cost2100_table(memory_length)[index], [1, memory_length]).copy()
    else:
        raise ValueError("No such channel_coefficients value!!!")
    if noisy_est_var > 0:  # global unseeded RNG, as in the reference (:36)
        h[:, 1:] += np.random.normal(0, noisy_est_var ** 0.5, [1, memory_length - 1])
    if fading and channel_coefficients == "time_decay":
        if fading_taps_type == 1:
            fading_taps = np.array([51, 39, 33, 21])
            h *= (0.8 + 0.2 * np.cos(2 * np.pi * index / fading_taps)).reshape(1, memory_length)
        elif fading_taps_type == 2:
            fading_taps = 5 * np.array([51, 39, 33, 21])
            fading_taps = np.maximum(fading_taps - 1.5 * index, 10 * np.ones(4)) - 1e-5
            h *= (0.8 + 0.2 * np.cos(np.pi * index / fading_taps)).reshape(1, memory_length)
        else:
            raise ValueError("No such fading tap type!!!")
    return h


class BPSKModulator:
    @staticmethod
    def modulate(c: np.ndarray) -> np.ndarray:
        """0 -> +1, 1 -> -1 (modulator.py:12)"""
        return 1 - 2 * c


def transmit(bits, h: np.ndarray, snr: float, memory_length: int, noise=None):
    """ISI-AWGN channel on the GPU (channel_dataset.py:71,87-95 + channel.py:12-35): `bits` [B,K] fp32 {0,1} codeword
    bits on the device, `h` [Bh,L] float64 taps (row b % Bh is used for word b), `noise` [B,K] standard-normal draws
    (torch float64/float32 tensor on the device) or None.  float64 arithmetic, fp32 result [B,K]."""
    import torch

    from . import _lib

    _lib.require_gpu_tensor(bits, "bits")
    c = bits.detach().to(torch.float32)
    if c.stride(-1) != 1:
        c = c.contiguous()
    B, K = c.shape
    hd = torch.as_tensor(np.ascontiguousarray(h, dtype=np.float64).reshape(-1, memory_length), device=c.device)
    sigma = (10 ** (snr / 10)) ** (-0.5)  # channel.py:23,31
    nz, is64 = None, 1
    if noise is not None:
        nz = noise.to(c.device).contiguous()
        if nz.dtype not in (torch.float64, torch.float32) or tuple(nz.shape) != (B, K):
            raise ValueError("noise must be a [B,K] float64/float32 tensor")
        is64 = 1 if nz.dtype == torch.float64 else 0
    y = torch.empty((B, K), dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        rc = _lib.load().mvn_isi_awgn_transmit(_lib.ptr(c), c.stride(0), K, _lib.ptr(nz), is64, _lib.ptr(hd), hd.shape[0],
                                               float(sigma), _lib.ptr(y), K, B, K, memory_length,
                                               _lib.current_stream(c.device))
    _lib.check(rc, "mvn_isi_awgn_transmit")
    return y


def generate_words(n_words: int, block_length: int, h: np.ndarray, snr: float, memory_length: int, device, seed: int,
                   want_tx: bool = True):
    """Uncoded words generated ON the device in one launch (mvn_generate_words_f32): counterpart of the inner loop of
    ChannelModelDataset.get_snr_data (channel_dataset.py:65-83) -- Bernoulli(1/2) bits, zero padding by L, BPSK, the
    anti-causal ISI channel `h` [Bh,L] (row b % Bh for word b) and white noise of std 10^(-snr/20) -- with Philox
    counter-based randomness (same distribution as the reference's RandomState streams, not the same stream).
    Returns (tx [n,K] fp32 {0,1} or None, y [n,K] fp32)."""
    import torch

    from . import _lib

    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.MvnError("generate_words runs on an MI355X (ROCm) device only")
    hd = torch.as_tensor(np.ascontiguousarray(h, dtype=np.float64).reshape(-1, memory_length), device=dev)
    y = torch.empty((n_words, block_length), dtype=torch.float32, device=dev)
    tx = torch.empty((n_words, block_length), dtype=torch.float32, device=dev) if want_tx else None
    sigma = (10 ** (snr / 10)) ** (-0.5)  # channel.py:23,31
    with torch.cuda.device(dev):
        rc = _lib.load().mvn_generate_words_f32(_lib.ptr(tx), block_length, _lib.ptr(y), block_length, _lib.ptr(hd), hd.shape[0],
                                                float(sigma), int(seed) & 0xFFFFFFFFFFFFFFFF, n_words, block_length, memory_length,
                                                _lib.current_stream(dev))
    _lib.check(rc, "mvn_generate_words_f32")
    return tx, y


class ReferenceWordStream:
    """Bit-exact twin of the reference's word source for small, reproducible runs: the two legacy NumPy
    RandomState streams of ChannelModelDataset (trainer.py:90-91: noise seed 3450002, word seed 7860002;
    channel_dataset.py:67 `randint(0, 2, (1, block_length))` per word, channel.py:31 `normal(0, 1, (1, T))` per word) drawn on
    the host with NumPy's own MT19937 / polar-Gaussian code, then encoded (optional RS, channel_dataset.py:69) and sent
    through the ISI-AWGN channel on the device by the replay kernel (mvn.transmit: float64 arithmetic, fp32 result --
    bit-identical to the reference's received words, tests/test_gpu_parity.py::test_reference_word_stream_*).  The streams
    persist across draw() calls like the dataset's RandomState members, so consecutive draws continue the reference's
    sequence.  At-scale Monte-Carlo runs use mvn.generate_words (Philox, one kernel) instead."""

    def __init__(self, block_length: int, memory_length: int, device, n_symbols: int = 0, noise_seed: int = 3450002,
                 word_seed: int = 7860002):
        self.block_length, self.memory_length, self.n_symbols = block_length, memory_length, n_symbols
        self.device = device
        self.random = np.random.RandomState(noise_seed)
        self.word_rand_gen = np.random.RandomState(word_seed)

    @property
    def transmission_length(self) -> int:
        return self.block_length + 8 * self.n_symbols  # trainer.py:196-198

    def draw(self, n_words: int, h: np.ndarray, snr: float):
        """`n_words` words through taps `h` [n_words or 1, L] (row i = estimate_channel(..., index=i)); returns
        (b [n, block_length] fp32 bits, y [n, transmission_length] fp32), both on the device."""
        import torch

        from . import ecc

        T = self.transmission_length
        bits = np.empty((n_words, self.block_length), np.float32)
        noise = np.empty((n_words, T), np.float64)
        for i in range(n_words):  # word by word: the two streams advance exactly like get_snr_data's loop
            bits[i] = self.word_rand_gen.randint(0, 2, size=(1, self.block_length))
            noise[i] = self.random.normal(0, 1, (1, T))
        b = torch.as_tensor(bits, device=self.device)
        c = ecc.rs_encode(b, self.n_symbols) if self.n_symbols else b
        y = transmit(c, h, snr, self.memory_length, torch.as_tensor(noise, device=self.device))
        return b, y
