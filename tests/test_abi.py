"""CPU-only: libmvn_hip.so loads, exports every symbol include/mvn.h declares, and validates its
arguments before touching a device.  No compute calls here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build_hip()
    import meta_viterbinet_amd as mvn

    return mvn._lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mvn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    import meta_viterbinet_amd as mvn

    declared = _declared_symbols()
    assert len(declared) >= 10
    raw = ctypes.CDLL(mvn._lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/mvn.h but not exported"
    assert sorted(mvn._lib.SIGNATURES) == declared  # the Python binding covers the whole header


def test_version_and_strerror(lib):
    assert lib.mvn_version() == 3
    assert lib.mvn_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6, -99):
        assert lib.mvn_strerror(code).startswith(b"mvn:")


def test_argument_validation_needs_no_device(lib):
    # S must be a power of two in [2,256]
    for S in (0, 1, 3, 12, 512):
        assert lib.mvn_acs_sweep_f32(None, None, 8, None, 4, 8, S, None) == -2
        assert lib.mvn_va_decode_f32(None, 8, None, 1, None, 8, None, 4, 8, S, None) == -2
        assert lib.mvn_vnet_logits_f32(*([None] * 8), 4, S, None) == -2
    # negative sizes / T larger than a row stride (the reference raises IndexError there, Q5)
    assert lib.mvn_acs_sweep_f32(None, None, 8, None, -1, 8, 16, None) == -1
    assert lib.mvn_acs_sweep_f32(None, None, 7, None, 4, 8, 16, None) == -1
    assert lib.mvn_va_decode_f32(None, 7, None, 1, None, 8, None, 4, 8, 16, None) == -1
    # prior table must divide the batch (the reference's .repeat would fail to broadcast)
    assert lib.mvn_va_decode_f32(None, 8, None, 3, None, 8, None, 4, 8, 16, None) == -3
    assert lib.mvn_va_decode_f32(None, 8, None, 0, None, 8, None, 4, 8, 16, None) == -3
    # empty batches are fine and do nothing
    assert lib.mvn_acs_sweep_f32(None, None, 8, None, 0, 8, 16, None) == 0
    assert lib.mvn_vnet_decode_f32(None, 8, *([None] * 6), None, 8, None, None, None, 0, 0, 8, 16, None) == 0
    assert lib.mvn_count_errors(None, 4, None, 4, None, 0, 4, ctypes.c_void_p(8), None) == 0
    # NULL data pointers with work to do
    assert lib.mvn_acs_sweep_f32(None, None, 8, None, 4, 8, 16, None) == -4
    assert lib.mvn_count_errors(None, 4, None, 4, None, 2, 4, None, None) == -4
    assert lib.mvn_vnet_workspace_bytes(10, 100, 4) == 10 * 100 * 4 * 4
    assert lib.mvn_vnet_workspace_bytes(10, 100, 16) == 0  # fused kernel: logits never leave the chip


def test_no_cpu_fallback_in_product_path():
    """The shipped package must not import or call the oracle, and must refuse CPU tensors."""
    import torch

    import meta_viterbinet_amd as mvn

    pkg = os.path.join(ROOT, "meta-viterbinet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src and "libmvn_oracle" not in src, fn
    det = mvn.VNETDetector(16, {"train": 8, "val": 8}).to("cpu")
    with pytest.raises(mvn._lib.MvnError):
        det(torch.zeros(2, 8), "val")
    va = mvn.VADetector(16, 4, 8, 1, "ISI_AWGN", 0, False, 1, {"train": "time_decay", "val": "time_decay"})
    with pytest.raises(mvn._lib.MvnError):
        va(torch.zeros(2, 8), "val", 10, 0.2)
    with pytest.raises(mvn._lib.MvnError):
        mvn.calculate_error_rates(torch.zeros(2, 8), torch.zeros(2, 8))
