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
    assert lib.mvn_version() == 6
    assert lib.mvn_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6, -7, -99):
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
    assert lib.mvn_vnet_workspace_bytes(10, 100, 2) == 10 * 100 * 2 * 4  # two states: the two-kernel route (logits in scratch)
    assert lib.mvn_vnet_workspace_bytes(10, 100, 16) == 0  # (the cooperative kernel: no hand-off lines either)
    for S in (4, 8, 32, 64, 128):  # fused kernels: logits never leave the chip -- from the batch size where that is the faster route
        assert lib.mvn_vnet_workspace_bytes(10000, 1000, S) == 0
        assert lib.mvn_vnet_workspace_bytes(10, 100, S) == 10 * 100 * S * 4  # a few blocks: mlp_kernel + sweep over logits in scratch
    assert lib.mvn_vnet_workspace_bytes(10000, 100, 256) == 10000 * 100 * 256 * 4  # (fused on request only: MVN_FUSED_IP=1)
    # the by-word step: 16 states, whole bytes, nsym <= 8, a row stride for every output that is given
    step = lambda T, nsym, S, R=1, rx_ld=None: lib.mvn_vnet_byword_step_f32(  # noqa: E731
        None, T if rx_ld is None else rx_ld, None, T, *([None] * 6), None, None, T, None, T, None, T, None, T, None, T, None, R, T,
        nsym, 0, S, None)
    assert step(136, 2, 8) == -2 and step(136, 2, 256) == -2
    assert step(135, 2, 16) == -1 and step(136, 9, 16) == -1 and step(16, 2, 16) == -1 and step(2048, 2, 16) == -1
    assert step(136, 2, 16, rx_ld=100) == -1
    assert step(136, 2, 16, R=0) == 0 and step(136, 2, 16) == -4
    va_step = lambda T, nsym, S, Bp=1, R=1: lib.mvn_va_byword_step_f32(  # noqa: E731
        None, T, None, T, None, Bp, None, T, None, T, None, T, None, T, None, T, None, R, T, nsym, 0, S, None)
    assert va_step(136, 2, 8) == -2 and va_step(135, 2, 16) == -1 and va_step(136, 9, 16) == -1 and va_step(136, 2, 16, Bp=0) == -3
    assert va_step(136, 2, 16, R=0) == 0 and va_step(136, 2, 16) == -4
    # trial-batched training: shapes before pointers; no trials = nothing to do; workspace sizes
    assert lib.mvn_vnet_online_train_trials_f32(None, 4, 0, 0, 1e-3, 0.9, 0.999, 1e-8, 16, None, 0, None) == -1
    assert lib.mvn_vnet_online_train_trials_f32(None, 4, 136, 0, 1e-3, 0.9, 0.999, 1e-8, 256, None, 0, None) == -2  # online training: S <= 128
    assert lib.mvn_vnet_maml_train_trials_f32(None, 4, 136, 1, 0.1, 1, 1e-3, 0.9, 0.999, 1e-8, 64, None, 0, None) == -2  # meta-learning: S <= 32
    assert lib.mvn_vnet_online_train_trials_f32(None, 0, 136, 0, 1e-3, 0.9, 0.999, 1e-8, 16, None, 0, None) == 0
    assert lib.mvn_vnet_online_train_trials_f32(None, 4, 136, 0, 1e-3, 0.9, 0.999, 1e-8, 16, None, 0, None) == -4
    assert lib.mvn_vnet_maml_train_trials_f32(None, 4, 136, 0, 0.1, 1, 1e-3, 0.9, 0.999, 1e-8, 16, None, 0, None) == -1
    assert lib.mvn_vnet_maml_train_trials_f32(None, 4, 136, 1, 0.1, 1, 1e-3, 0.9, 0.999, 1e-8, 16, None, 0, None) == -4
    one = lib.mvn_vnet_train_trials_workspace_bytes(16, 136, 1, 1)
    assert one > 0 and one % 256 == 0 and lib.mvn_vnet_train_trials_workspace_bytes(16, 136, 1, 7) == 7 * one
    assert lib.mvn_vnet_train_trials_workspace_bytes(16, 32, 1, 4) == lib.mvn_vnet_train_trials_workspace_bytes(16, 32, 1, 1) * 4
    assert lib.mvn_vnet_train_trials_workspace_bytes(64, 136, 1, 4) == 0


def test_test_hooks_are_not_in_the_shipped_library(lib):
    """mvn_test_hooks (forces the training kernels' barrier to give up) exists in the tests' -DMVN_TEST_HOOKS build only."""
    import meta_viterbinet_amd as mvn

    shipped = ctypes.CDLL(os.path.join(ROOT, "meta-viterbinet_amd", "libmvn_hip.so"))
    assert not hasattr(shipped, "mvn_test_hooks")
    assert "mvn_test_hooks" not in mvn._lib.SIGNATURES and "mvn_test_hooks" not in _declared_symbols()


def test_train_kernel_name_validates(lib):
    name = ctypes.create_string_buffer(96)
    assert lib.mvn_vnet_train_kernel_name(3, 1, 136, 0, 16, 0, name, 96) == -1
    assert lib.mvn_vnet_train_kernel_name(1, 1, 136, 0, 16, 0, name, 96) == -1  # a meta-learning step has W >= 1 support words
    assert lib.mvn_vnet_train_kernel_name(0, 1, 136, 0, 256, 0, name, 96) == -2 and lib.mvn_vnet_train_kernel_name(1, 1, 136, 1, 64, 0, name, 96) == -2
    assert lib.mvn_vnet_train_kernel_name(0, 1, 136, 0, 64, 0, name, 96) == 0 and name.value == b"online_train_kernel<64, true> 1x1"
    assert lib.mvn_vnet_train_kernel_name(0, 1, 136, 0, 16, 0, None, 96) == -4
    # without a workspace every form is one workgroup per trial; minibatch iterations always are
    assert lib.mvn_vnet_train_kernel_name(0, 0, 136, 0, 16, 0, name, 96) == 0 and name.value == b"online_train_kernel<16, false> 1x1"
    assert lib.mvn_vnet_train_kernel_name(0, 7, 136, 32, 16, 1 << 30, name, 96) == 0 and name.value == b"online_train_kernel<16, true> 1x7"
    assert lib.mvn_vnet_train_kernel_name(2, 3, 136, 1, 32, 0, name, 96) == 0 and name.value == b"maml_train_kernel<0, true> 1x3"
    assert lib.mvn_vnet_train_kernel_name(2, 0, 136, 1, 32, 0, name, 96) == 0 and name.value == b"maml_train_kernel<32, false> 1x1"


def test_trial_descriptor_layout_matches_the_header():
    """trials.TRIAL_DTYPE is include/mvn.h's mvn_train_trial_t field for field (compiled here with the host compiler)."""
    import subprocess
    import tempfile

    import meta_viterbinet_amd as mvn

    fields = ["y", "labels", "idx", "query_idx", "w_in", "w_out", "w_out2", "adam_m", "adam_v", "loss_out", "status", "b1pow",
              "b2pow", "n", "reserved"]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "mvn.h"\nint main(void){printf("%zu", sizeof(mvn_train_trial_t));' + "".join(
        f'printf(" %zu", offsetof(mvn_train_trial_t, {f}));' for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = [int(v) for v in subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()]
    dt = mvn.trials.TRIAL_DTYPE
    assert out[0] == dt.itemsize
    assert out[1:] == [dt.fields[f][1] for f in fields]


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
