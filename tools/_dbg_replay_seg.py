"""Debug aid: growth of the HIP-vs-torch deviation over the first iterations of ONE update segment of the configs[4] flow
(tests/test_gpu_replay.py).  usage: _dbg_replay_seg.py <block count> <stage>"""
import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import meta_viterbinet_amd as mvn
from meta_viterbinet_amd.trials import TrialDraws
import test_gpu_replay as R
dev = torch.device("cuda:0")
count, stage = int(sys.argv[1]), sys.argv[2]
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w0 = [g7[f"w{i}"] for i in range(6)]
msg, rx = R._words(dev, "time_decay", 10.0, 9)
T = rx.shape[1]
kw = dict(self_supervised=True, online_meta=True, meta_train_iterations=20, meta_j_num=10, meta_subframes=5, meta_style_online_training=True)
det = R._vnet_with(w0, T, dev); tr = mvn.OnlineTrainer(det, 4); log = []
def observer(seen):
    log.append(dict(stage=seen["stage"], count=seen["count"], n_buf=seen["buffer_rx"].shape[0], meta=seen["meta"], trained=seen["trained"],
                    batch_idx=seen["batch_idx"], buffers=(seen["buffer_rx"], seen["buffer_tx"]), state=R._state(seen["detector"], seen["saved_detector"], tr)))
mvn.eval_by_word(det, msg, rx, 10.0, 0.2, 2, 25, online_trainer=tr, self_supervised_iterations=200,
                 meta_detector=mvn.META_VNETDetector(16, {"train": T, "val": T}), draws=TrialDraws(17, dev), observer=observer, meta_lr=0.1, MAML=True, **kw)
i = [k for k, r in enumerate(log) if r["count"] == count and r["stage"] == stage][0]
rec, prev = log[i], log[i - 1]["state"]
brx, btx = rec["buffers"]; brx, btx = brx[:rec["n_buf"]], btx[:rec["n_buf"]]
names = ["w1", "b1", "W2", "b2", "W3", "b3"]
for n in (1, 2, 3, 5, 8, 12, 16, 20, 25):
    det_k = R._vnet_with(w0, T, dev); tr_k = mvn.OnlineTrainer(det_k, 4)
    det_t = R._vnet_with(w0, T, dev); saved_t = R._vnet_with(w0, T, dev); tr_t = mvn.OnlineTrainer(det_t, 4, use_kernel=False)
    R._load(det_k, None, tr_k, prev); R._load(det_t, saved_t, tr_t, prev)
    with torch.no_grad():
        for p, a in zip(det_k.parameters(), prev["saved"]): p.copy_(a)
    mvn.copy_model(source_model=saved_t, dest_model=det_t)
    tr_k.online_training(btx[-1].reshape(1, -1), brx[-1].reshape(1, -1), iterations=n, full_word=True)
    tr_t._online_training_autograd(btx[-1].reshape(1, -1), brx[-1].reshape(1, -1), n, None, True, False)
    rs = []
    for nm, a, b in zip(names, det_t.parameters(), det_k.parameters()):
        q = (a.detach() - b.detach()).abs() / (2e-5 + 1e-3 * b.detach().abs())
        rs.append(f"{nm} {float(q.max()):.2f}@{int(q.argmax())}")
    qm = (tr_t.exp_avg - tr_k.exp_avg).abs() / (1e-6 + 1e-3 * tr_k.exp_avg.abs())
    qv = (tr_t.exp_avg_sq - tr_k.exp_avg_sq).abs() / (1e-6 + 1e-3 * tr_k.exp_avg_sq.abs())
    print(f"n {n:2d}: " + "  ".join(rs) + f"  m {float(qm.max()):.2f}@{int(qm.argmax())}  v {float(qv.max()):.2f}@{int(qv.argmax())}")
