"""Workload for rocprofv3 (--kernel-trace / --pmc): every form of the training kernels at the sizes bench.py's trial entries
run them, each launched a few times through the C ABI's trial entry points (hand-filled descriptors, like a C caller):
    online_train_kernel<16, true>         256 trials x 200 minibatch iterations (32 samples)        grid 1x256
    online_train_kernel<16, true>         256 trials x 200 full-word iterations (136 samples)       grid 1x256
    online_train_groups_kernel<16, true>   48 trials x 200 full-word iterations, 5 workgroups each  grid 5x48, a trial's workgroups on one XCD
    maml_train_kernel<16, true>           256 trials x 40 second-order meta-learning steps          grid 1x256
    maml_train_groups_kernel<16, true>     48 trials x 40 steps, 5 workgroups each                  grid 5x48, a trial's workgroups on one XCD
Prints each launch's wall time (HIP events) and the algorithmic FLOPs, so that a --pmc pass can be turned into a roofline
entry (tools/pmc_train_summary.py).  usage: prof_train_kernels.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trial_setup import L, T, dev, mvn, w  # noqa: E402
from meta_viterbinet_amd import trials as tr_mod  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
lib = mvn._lib.load()
S, NW = 16, 12
gen = torch.Generator(device=dev).manual_seed(1)


def run(kind, R, n, M=0, second_order=1):
    """kind 'online' (M = 32: minibatch, 0: full word) or 'maml'.  Returns (ms per launch, kernel form)."""
    bank = tr_mod.TrialBank([w] * R, S, L, dev)
    rxw = torch.randn(R, NW, T, generator=gen, device=dev)
    txw = torch.randint(0, 2, (R, NW, T), generator=gen, device=dev).float()
    labels = torch.stack([mvn.calculate_states(L, txw[r]).reshape(NW, T) for r in range(R)]).to(torch.int32).contiguous()
    bidx = (torch.multinomial(torch.arange(T, dtype=torch.float32, device=dev).expand(R * n, T), 32, generator=gen)
            .to(torch.int32).reshape(R, n, 32)) if M else None
    sup = torch.randint(0, NW, (R, n, 1), generator=gen, device=dev).to(torch.int32)
    qry = torch.randint(0, NW, (R, n), generator=gen, device=dev).to(torch.int32)
    d = np.zeros(R, dtype=tr_mod.TRIAL_DTYPE)
    th = bank.pointers(bank.theta)
    status = torch.zeros(R, dtype=torch.int32, device=dev)
    for r in range(R):
        d[r]["y"] = rxw[r].data_ptr()
        d[r]["labels"] = labels[r].data_ptr()
        d[r]["idx"] = sup[r].data_ptr() if kind == "maml" else (bidx[r].data_ptr() if M else 0)
        d[r]["query_idx"] = qry[r].data_ptr() if kind == "maml" else 0
        d[r]["w_in"], d[r]["w_out"] = th[r], th[r]
        d[r]["adam_m"], d[r]["adam_v"] = bank.exp_avg[r].data_ptr(), bank.exp_avg_sq[r].data_ptr()
        d[r]["status"] = status[r].data_ptr()
        d[r]["b1pow"], d[r]["b2pow"] = 1.0, 1.0
        d[r]["n"] = n
    dd = torch.from_numpy(d.view(np.uint8)).to(dev)
    nb = int(lib.mvn_vnet_train_trials_workspace_bytes(S, T, 1, R))
    wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    st = mvn._lib.current_stream(dev)
    name = mvn._lib.ctypes.create_string_buffer(128)
    lib.mvn_vnet_train_kernel_name(0 if kind == "online" else 1 + second_order, R, T, M if kind == "online" else 1, S, nb, name, 128)

    def launch():
        if kind == "maml":
            rc = lib.mvn_vnet_maml_train_trials_f32(mvn._lib.ptr(dd), R, T, 1, 0.1, second_order, 1e-3, 0.9, 0.999, 1e-8, S, mvn._lib.ptr(wsb), nb, st)
        else:
            rc = lib.mvn_vnet_online_train_trials_f32(mvn._lib.ptr(dd), R, T, M, 1e-3, 0.9, 0.999, 1e-8, S, mvn._lib.ptr(wsb), nb, st)
        assert rc == 0

    launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        launch()
    b.record()
    b.synchronize()
    assert "--no-check" in sys.argv or (int(status.abs().sum()) == 0 and bool(torch.isfinite(bank.theta).all()))
    return a.elapsed_time(b) / reps, name.value.decode()


# algorithmic FLOPs (bench.py's training_roofline): 35 kFLOP per sample of a CE forward + backward pass; a second-order step =
# support + query gradient passes + a Hessian-vector pass of ~3 gradient passes over the support word
CASES = [("online", 256, 200, 32, 1, 35e3 * 32), ("online", 256, 200, 0, 1, 35e3 * T), ("online", 48, 200, 0, 1, 35e3 * T),
         ("maml", 256, 40, 0, 1, 35e3 * 2 * T + 105e3 * T), ("maml", 48, 40, 0, 1, 35e3 * 2 * T + 105e3 * T)]
print("kernel_form,trials,iterations,samples_per_iteration,ms_per_launch,us_per_iteration_per_trial_slot,algorithmic_TFLOPs")
for kind, R, n, M, so, flop_it in CASES:
    ms, form = run(kind, R, n, M, so)
    print(f"\"{form}\",{R},{n},{M or T},{ms:.3f},{ms * 1e3 / n:.2f},{flop_it * n * R / (ms * 1e-3) / 1e12:.2f}", flush=True)
