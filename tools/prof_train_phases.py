#!/usr/bin/env python3
"""Phase timeline of one iteration of the one-launch online-training kernel (diagnostic build -DMVN_DIAG_TRAIN: s_memtime at
the barriers of iteration 100, returned through the loss buffer).  Build the diagnostic library first (no GPU needed):
    python tools/prof_train_phases.py --build
then on the GPU box:
    python tools/prof_train_phases.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_SO = os.path.join(ROOT, "tools", "libmvn_diag_train.so")
if "--build" in sys.argv:
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIPCC_FLAGS + ["-DMVN_DIAG_TRAIN", g.HIP_SRC, "-o", DIAG_SO], check=True)
    print("built", DIAG_SO)
    sys.exit(0)

os.environ["MVN_LIB_PATH"] = DIAG_SO
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
import meta_viterbinet_amd as mvn  # noqa: E402

dev = torch.device("cuda:0")
T, L, S, N = 136, 4, 16, 420
names = ["h1 = sigmoid(W1 y + b1)", "z2 = h1 W2^T (MFMA)", "logits = relu(z2) W3^T (MFMA)", "softmax / CE / dlogits",
         "dz2, dW3, db3", "dz1 (MFMA), dW2 units 0..47 (MFMA)", "dW1, db1, db2, dW2 units 48, 49"]
os.environ["MVN_TRAIN_GROUPS"] = "0"  # the full-word run on the one-workgroup kernel too: the stamps live there
mvn._lib.reload_switches()
print("(the stamps cost: an iteration of this build is ~25 % longer than the production build's; read the table for proportions and "
      "for WHICH wave a phase waits for)")
for full in (False, True):
    torch.manual_seed(0)
    tx = torch.randint(0, 2, (1, T)).float().to(dev)
    rx = torch.randn(1, T).to(dev)
    det = mvn.VNETDetector(S, {"train": T, "val": T}).to(dev)
    tr = mvn.OnlineTrainer(det, L)
    for _ in range(3):
        loss = tr.online_training(tx, rx, iterations=N, full_word=full, return_loss=True)
    torch.cuda.synchronize()
    st = loss[102:102 + 304].cpu().numpy().view(np.uint64).astype(np.int64)
    # stamps 0..7: the LAST chunk of iteration 100 (grad_chunk start, then after each barrier); 8: iteration start; 9: after Adam
    it_total = st[9] - st[8]
    print(f"--- {'full word (136 samples = 5 chunks)' if full else 'minibatch (32 samples = 1 chunk)'}: iteration 100 = {it_total} cycles of s_memtime (2.4 GHz)")
    for k in range(7):
        print(f"  {names[k]:36s} {st[k+1]-st[k]:7d}")
    print(f"    wave 0 inside the hidden-layer phase: operands + 32 MFMAs issued {st[11]-st[10]}, results + epilogue {st[12]-st[11]}, "
          f"wait at the barrier {st[6]-st[12]}")
    print("    cycles from a phase's start to each wave's arrival at the barrier that ends it (waves 0..15):")
    for ph in range(8):
        start = st[ph] if ph < 7 else st[7]
        print(f"      {(names + ['Adam'])[ph][:28]:28s}", " ".join(f"{int(st[24 + 16 * ph + w] - start):5d}" for w in range(16)))
    print(f"    wave 0 inside the z2 phase: barrier -> tile start {st[16]-st[1]}, operand reads + 25 MFMAs issued {st[17]-st[16]}, "
          f"bias + store {st[18]-st[17]}, wait at the barrier {st[2]-st[18]}")
    print(f"  {'chunk total':36s} {st[7]-st[0]:7d}")
    print(f"  {'Adam + loss + barrier':36s} {st[9]-st[7]:7d}")
    print(f"  {'iteration start -> last chunk start':36s} {st[0]-st[8]:7d}")
