"""ctypes binding of libmvn_hip.so (C ABI: include/mvn.h).

There is deliberately NO CPU fallback: if the HIP library is missing or no gfx950 device is
usable, every 'val' call raises.  (The CPU oracle under oracle/ is test infrastructure and is
never imported from here.)
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVN_LIB_PATH", os.path.join(_PKG, "libmvn_hip.so"))  # override: A/B builds
ABI_VERSION = 6

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32

# name -> (restype, argtypes); mirrors include/mvn.h one to one
SIGNATURES = {
    "mvn_version": (ctypes.c_int, []),
    "mvn_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "mvn_device_info": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_char_p, ctypes.c_int]),
    "mvn_acs_block_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "mvn_acs_sweep_f32": (ctypes.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "mvn_survivor_bytes": (ctypes.c_size_t, [_i64, _i32, _i32]),
    "mvn_acs_sweep_surv_f32": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mvn_vnet_surv_workspace_bytes": (ctypes.c_size_t, [_i64, _i32, _i32]),
    "mvn_vnet_decode_surv_f32": (ctypes.c_int, [_vp, _i64] + [_vp] * 6 + [_vp, _i64, _vp, _vp, _vp, ctypes.c_size_t, _i64, _i32, _i32, _vp]),
    "mvn_va_decode_surv_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mvn_traceback_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "mvn_acs_sweep_kernel_name": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, ctypes.c_char_p, _i32]),
    "mvn_va_decode_kernel_name": (ctypes.c_int, [_i64, _i32, _i32, ctypes.c_char_p, _i32]),
    "mvn_vnet_decode_kernel_name": (ctypes.c_int, [_i64, _i32, _i32, _i32, ctypes.c_char_p, _i32]),
    "mvn_va_decode_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "mvn_vnet_logits_f32": (ctypes.c_int, [_vp] * 8 + [_i64, _i32, _vp]),
    "mvn_vnet_workspace_bytes": (ctypes.c_size_t, [_i64, _i32, _i32]),
    "mvn_vnet_decode_f32": (ctypes.c_int, [_vp, _i64] + [_vp] * 6 + [_vp, _i64, _vp, _vp, _vp, ctypes.c_size_t,
                                                                   _i64, _i32, _i32, _vp]),
    "mvn_vnet_decode_count_f32": (ctypes.c_int, [_vp, _i64] + [_vp] * 6 + [_vp, _i64, _i32, _vp, _vp, _vp, _i64,
                                                 _vp, ctypes.c_size_t, _i64, _i32, _i32, _vp]),
    "mvn_vnet_online_train_f32": (ctypes.c_int, [_vp, _vp, _i32, _vp, _i32, _i32] + [_vp] * 8 +
                                  [_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _i32, _vp]),
    "mvn_vnet_train_workspace_bytes": (ctypes.c_size_t, [_i32]),
    "mvn_vnet_online_train_ws_f32": (ctypes.c_int, [_vp, _vp, _i32, _vp, _i32, _i32] + [_vp] * 8 +
                                     [_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _i32, _vp,
                                      ctypes.c_size_t, _vp, _vp]),
    "mvn_vnet_maml_train_f32": (ctypes.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32] + [_vp] * 8 +
                                [_i64, ctypes.c_float, _i32] + [ctypes.c_float] * 4 + [_vp, _i32, _vp]),
    "mvn_vnet_maml_train_ws_f32": (ctypes.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32] + [_vp] * 8 +
                                   [_i64, ctypes.c_float, _i32] + [ctypes.c_float] * 4 + [_vp, _i32, _vp, ctypes.c_size_t, _vp, _vp]),
    "mvn_vnet_train_trials_workspace_bytes": (ctypes.c_size_t, [_i32, _i32, _i32, _i32]),
    "mvn_vnet_online_train_trials_f32": (ctypes.c_int, [_vp, _i32, _i32, _i32] + [ctypes.c_float] * 4 + [_i32, _vp, ctypes.c_size_t, _vp]),
    "mvn_vnet_maml_train_trials_f32": (ctypes.c_int, [_vp, _i32, _i32, _i32, ctypes.c_float, _i32] + [ctypes.c_float] * 4 +
                                       [_i32, _vp, ctypes.c_size_t, _vp]),
    "mvn_vnet_train_kernel_name": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, ctypes.c_size_t, ctypes.c_char_p, _i32]),
    "mvn_vnet_byword_step_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64] + [_vp] * 6 + [ctypes.POINTER(ctypes.c_int64)] +
                                 [_vp, _i64] * 4 + [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mvn_va_byword_step_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64] + [_vp, _i64] * 4 + [_vp, _i64, _vp, _i64, _i32, _i32,
                                              _i32, _i32, _vp]),
    "mvn_reload_switches": (None, []),
    "mvn_isi_awgn_transmit": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _i64, ctypes.c_double, _vp, _i64, _i64,
                                             _i32, _i32, _vp]),
    "mvn_generate_words_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, ctypes.c_double, ctypes.c_uint64, _i64, _i32, _i32, _vp]),
    "mvn_va_montecarlo_f32": (ctypes.c_int, [_vp, _i64, ctypes.c_double, ctypes.c_uint64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mvn_rs_decode_bits_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "mvn_rs_encode_bits_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "mvn_count_errors": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp]),
}

_lib = None


class MvnError(RuntimeError):
    """A libmvn_hip.so call returned non-zero."""


def load_variant(path: str):
    """Another build of the library (an A/B build, the tests' -DMVN_TEST_HOOKS build) with the same bindings; not cached."""
    if not os.path.exists(path):
        raise MvnError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the 'val' path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mvn_version() != ABI_VERSION:
        raise MvnError(f"{os.path.basename(path)} ABI {lib.mvn_version()} != expected {ABI_VERSION}")
    return lib


def load():
    """Load the HIP library; raises (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        _lib = load_variant(LIB_PATH)
    return _lib


def reload_switches():
    """The library reads its MVN_* environment switches once per process; call this after changing one in-process."""
    if _lib is not None:
        _lib.mvn_reload_switches()


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mvn_strerror(rc).decode()
        if rc in (-1, -3):
            raise ValueError(f"{what}: {msg}")
        raise MvnError(f"{what}: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a contiguous fp32/int64 torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """`with on_device(t.device):` makes t's GPU current for a launch; free when it already is (the by-word evaluation
    issues hundreds of B=1 calls, where torch.cuda.device()'s get/set pair is a measurable part of each)."""
    import torch

    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


def f32c(t):
    """t as a contiguous fp32 tensor; the tensor itself when it already is one (only its data_ptr is used: no autograd
    bookkeeping is involved, so no detach())."""
    import torch

    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return t.detach().to(torch.float32).contiguous()


def require_gpu_tensor(t, name):
    if not t.is_cuda:
        raise MvnError(
            f"{name} lives on {t.device}: the 'val' hot path only runs on an MI355X (ROCm) device; "
            "CPU execution is intentionally not provided")
