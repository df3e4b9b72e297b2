import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import meta_viterbinet_amd as mvn
import test_gpu_replay as R
dev = torch.device("cuda:0")
g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_by_word.npz"))
w0 = [g7[f"w{i}"] for i in range(6)]
msg, rx = R._words(dev, "time_decay", 10.0, 9)
T = 136
# two blocks by hand: HIP vs torch, full-word 200 iterations, starting from w0 each time, Adam state carried
for use_groups in ("1", "0"):
    os.environ["MVN_TRAIN_GROUPS"] = use_groups; mvn._lib.reload_switches()
    det = R._vnet_with(w0, T, dev); tr = mvn.OnlineTrainer(det, 4)
    det_t = R._vnet_with(w0, T, dev); tr_t = mvn.OnlineTrainer(det_t, 4, use_kernel=False)
    for blk in range(3):
        cw = mvn.rs_encode(msg[blk:blk+1], 2)
        with torch.no_grad():
            for p, a in zip(det.parameters(), w0): p.copy_(torch.as_tensor(a))
            for p, a in zip(det_t.parameters(), w0): p.copy_(torch.as_tensor(a))
        # resync moments
        tr_t.exp_avg.copy_(tr.exp_avg); tr_t.exp_avg_sq.copy_(tr.exp_avg_sq); tr_t.step = tr.step
        lk = tr.online_training(cw, rx[blk:blk+1], iterations=200, full_word=True, return_loss=True)
        lt = tr_t._online_training_autograd(cw, rx[blk:blk+1], 200, None, True, True)
        torch.cuda.synchronize()
        devs = [float((a.detach()-b.detach()).abs().max()) for a, b in zip(det.parameters(), det_t.parameters())]
        rel = [float(((a.detach()-b.detach()).abs()/(2e-5+1e-3*b.detach().abs())).max()) for a, b in zip(det.parameters(), det_t.parameters())]
        print(f"groups={use_groups} block {blk}: max|dw| per tensor {['%.2e'%d for d in devs]} ratio/25it-tol {['%.1f'%r for r in rel]}  loss k {lk[0]:.5f}->{lk[-1]:.5f} t {lt[0]:.5f}->{lt[-1]:.5f}",
              " m dev %.2e v dev %.2e" % (float((tr.exp_avg-tr_t.exp_avg).abs().max()), float((tr.exp_avg_sq-tr_t.exp_avg_sq).abs().max())),
              " v rel max %.2e" % float(((tr.exp_avg_sq-tr_t.exp_avg_sq).abs()/(1e-6+1e-3*tr_t.exp_avg_sq.abs())).max()), " m rel %.2e" % float(((tr.exp_avg-tr_t.exp_avg).abs()/(1e-6+1e-3*tr_t.exp_avg.abs())).max()))
