"""ctypes/numpy front end of oracle/libmvn_oracle.so (CPU oracle, test infrastructure only).

Parity status: PINNED against tests/golden/*.npz (captured from the reference by
tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmvn_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "mvn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.mvn_oracle_expf_u10.restype = ctypes.c_float
        L.mvn_oracle_expf_u10.argtypes = [ctypes.c_float]
        L.mvn_oracle_sigmoid.restype = ctypes.c_float
        L.mvn_oracle_sigmoid.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return None if a is None else a.ctypes.data_as(t)


def _check(rc, what):
    if rc != 0:
        raise ValueError(f"{what}: oracle returned {rc}")


def max_threads() -> int:
    return int(lib().mvn_oracle_max_threads())


def create_transition_table(n_states: int) -> np.ndarray:
    """trellis_utils.py:7-13"""
    t = np.empty((n_states, 2), np.int32)
    lib().mvn_oracle_transition_table(ctypes.c_int(n_states), _p(t, _i32p))
    return t.astype(np.int64)


def acs_block(in_prob, llrs):
    """trellis_utils.py:16-30 -> (values [B,S] f32, argmin_j [B,S] i64)"""
    ip, c = _f32(in_prob), _f32(llrs)
    B, S = ip.shape
    out = np.empty((B, S), np.float32)
    idx = np.empty((B, S), np.int64)
    lib().mvn_oracle_acs_block(_p(ip), _p(c), _p(out), _p(idx, _i64p), ctypes.c_int64(B), ctypes.c_int(S))
    return out, idx


def acs_sweep(cost, want_final=True):
    """T-step decision/ACS loop over materialised costs [B,T,S] -> dec [B,T] (+ final metrics [B,S])."""
    c = _f32(cost)
    B, T, S = c.shape
    dec = np.zeros((B, T), np.float32)
    fm = np.empty((B, S), np.float32) if want_final else None
    _check(lib().mvn_oracle_acs_sweep(_p(c), _p(dec), ctypes.c_int64(T), _p(fm), ctypes.c_int64(B),
                                      ctypes.c_int(T), ctypes.c_int(S)), "acs_sweep")
    return (dec, fm) if want_final else dec


def acs_sweep_surv(cost):
    """acs_sweep + the survivors: dec [B,T], final metrics [B,S], surv uint8 [B,T,max(1,S/8)] (bit s & 7 of byte s >> 3 = torch.min's
    index j of state s; predecessor (2s + j) % S: trellis_utils.py:7-13,30)."""
    c = _f32(cost)
    B, T, S = c.shape
    dec = np.zeros((B, T), np.float32)
    fm = np.empty((B, S), np.float32)
    surv = np.zeros((B, T, max(1, S // 8)), np.uint8)
    _check(lib().mvn_oracle_acs_sweep_surv(_p(c), _p(dec), ctypes.c_int64(T), _p(fm), surv.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_int64(B), ctypes.c_int(T), ctypes.c_int(S)), "acs_sweep_surv")
    return dec, fm, surv


def traceback(surv, final_metric):
    """The textbook maximum-likelihood path from the survivors: bits [B,T] fp32 {0,1}, states int32 [B,T]."""
    sv = np.ascontiguousarray(surv, dtype=np.uint8)
    fm = _f32(final_metric)
    B, T = sv.shape[:2]
    S = fm.shape[1]
    bits = np.zeros((B, T), np.float32)
    states = np.zeros((B, T), np.int32)
    _check(lib().mvn_oracle_traceback(sv.ctypes.data_as(ctypes.c_void_p), _p(fm), _p(bits), ctypes.c_int64(T),
                                      states.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(B), ctypes.c_int(T), ctypes.c_int(S)),
           "traceback")
    return bits, states


def va_costs(y, priors, T=None):
    """va_detector.py:64-68 -> cost [B,T,S]"""
    y, pr = _f32(y), _f32(priors)
    B, Ty = y.shape
    T = Ty if T is None else T
    Bp, S = pr.shape
    cost = np.empty((B, T, S), np.float32)
    _check(lib().mvn_oracle_va_costs(_p(y), ctypes.c_int64(Ty), _p(pr), ctypes.c_int64(Bp), _p(cost),
                                     ctypes.c_int64(B), ctypes.c_int(T), ctypes.c_int(S)), "va_costs")
    return cost


def va_decode(y, priors, T=None, want_final=True):
    """VADetector.forward('val') given state priors [Bp,S] (va_detector.py:73-98)."""
    y, pr = _f32(y), _f32(priors)
    B, Ty = y.shape
    T = Ty if T is None else T
    if T > Ty:
        raise IndexError("transmission_length exceeds y.shape[1]")
    Bp, S = pr.shape
    dec = np.zeros((B, Ty), np.float32)
    fm = np.empty((B, S), np.float32) if want_final else None
    _check(lib().mvn_oracle_va_decode(_p(y), ctypes.c_int64(Ty), _p(pr), ctypes.c_int64(Bp), _p(dec),
                                      ctypes.c_int64(Ty), _p(fm), ctypes.c_int64(B), ctypes.c_int(T),
                                      ctypes.c_int(S)), "va_decode")
    return (dec, fm) if want_final else dec


def _weights(weights):
    W1, b1, W2, b2, W3, b3 = [_f32(np.asarray(w)) for w in weights]
    if W1.shape != (100, 1) or b1.shape != (100,) or W2.shape != (50, 100) or b2.shape != (50,):
        raise ValueError("bad ViterbiNet weight shapes")
    S = W3.shape[0]
    if W3.shape != (S, 50) or b3.shape != (S,):
        raise ValueError("bad ViterbiNet output-layer shapes")
    return (W1, b1, W2, b2, W3, b3), S


def vnet_logits(y, weights):
    """net(y.reshape(-1,1)) (vnet_detector.py:49) -> logits, shape y.shape + (S,)"""
    y = _f32(y)
    w, S = _weights(weights)
    out = np.empty(y.shape + (S,), np.float32)
    _check(lib().mvn_oracle_vnet_logits(_p(y), *[_p(a) for a in w], _p(out), ctypes.c_int64(y.size),
                                        ctypes.c_int(S)), "vnet_logits")
    return out


def vnet_decode(y, weights, T=None, want_logits=False, want_final=False):
    """VNETDetector.forward('val') (vnet_detector.py:35-61)."""
    y = _f32(y)
    w, S = _weights(weights)
    B, Ty = y.shape
    T = Ty if T is None else T
    if T > Ty:
        raise IndexError("transmission_length exceeds y.shape[1]")
    dec = np.zeros((B, Ty), np.float32)
    lg = np.empty((B, T, S), np.float32) if want_logits else None
    fm = np.empty((B, S), np.float32) if want_final else None
    _check(lib().mvn_oracle_vnet_decode(_p(y), ctypes.c_int64(Ty), *[_p(a) for a in w], _p(dec),
                                        ctypes.c_int64(Ty), _p(lg), _p(fm), ctypes.c_int64(B),
                                        ctypes.c_int(T), ctypes.c_int(S)), "vnet_decode")
    res = (dec,)
    if want_logits:
        res += (lg,)
    if want_final:
        res += (fm,)
    return res if len(res) > 1 else dec


def count_errors(dec, tx, rows=None):
    """metrics.py:7-17 as int64 counters {bit_errors, bits, frame_errors, frames}."""
    d, t = _f32(dec), _f32(tx)
    K = min(d.shape[1], t.shape[1])
    if d.shape[1] != t.shape[1]:
        raise ValueError("prediction/target width mismatch")
    r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
    n = d.shape[0] if r is None else r.size
    out = np.zeros(4, np.int64)
    lib().mvn_oracle_count_errors(_p(d), ctypes.c_int64(d.shape[1]), _p(t), ctypes.c_int64(t.shape[1]),
                                  _p(r, _i64p), ctypes.c_int64(n), ctypes.c_int(K), _p(out, _i64p))
    return out


def error_rates(counters):
    """(ser, fer) as metrics.py:13-17 would report them (computed in f64 from the integers)."""
    be, bits, fe, frames = [int(c) for c in counters]
    ser = max(be / bits, 0.0) if bits else float("nan")
    fer = max(fe / frames, 0.0) if frames else float("nan")
    return ser, fer


def calculate_states(memory_length: int, words) -> np.ndarray:
    """trellis_utils.py:33-46"""
    w = _f32(words)
    B, T = w.shape
    out = np.empty(B * T, np.int64)
    _check(lib().mvn_oracle_calculate_states(_p(w), ctypes.c_int64(B), ctypes.c_int(T),
                                             ctypes.c_int(memory_length), _p(out, _i64p)), "calculate_states")
    return out


def sigmoid(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    lib().mvn_oracle_sigmoid_array(_p(x), _p(out), ctypes.c_int64(x.size))
    return out


def rs_encode_bits(bits, nsym: int) -> np.ndarray:
    """ecc.rs_main.encode, batched: bits [B, K] {0,1} -> codewords [B, K + 8*nsym] (rs_main.py:9-18)."""
    b = _f32(bits)
    B, K = b.shape
    out = np.empty((B, K + 8 * nsym), np.float32)
    _check(lib().mvn_oracle_rs_encode_bits(_p(b), ctypes.c_int64(K), _p(out), ctypes.c_int64(K + 8 * nsym),
                                           ctypes.c_int64(B), ctypes.c_int(K), ctypes.c_int(nsym)), "rs_encode_bits")
    return out


def rs_decode_bits(bits, nsym: int, want_status: bool = False):
    """ecc.rs_main.decode, batched: received bits [B, N] -> message bits [B, N - 8*nsym] (rs_main.py:21-37).
    status: 0 decoded, 1 'too many errors' (uncorrected systematic part returned), 2 reference would raise."""
    b = _f32(bits)
    B, N = b.shape
    out = np.empty((B, N - 8 * nsym), np.float32)
    st = np.zeros(B, np.int32)
    _check(lib().mvn_oracle_rs_decode_bits(_p(b), ctypes.c_int64(N), _p(out), ctypes.c_int64(N - 8 * nsym),
                                           ctypes.c_int64(B), ctypes.c_int(N), ctypes.c_int(nsym),
                                           st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))), "rs_decode_bits")
    return (out, st) if want_status else out
