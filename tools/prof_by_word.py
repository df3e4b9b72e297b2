import cProfile, pstats, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import meta_viterbinet_amd as mvn
dev = torch.device("cuda:0")
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g7 = np.load(os.path.join(R, "tests", "golden", "g7_by_word.npz"))
N, K, nsym, L, snr = 300, 120, 2, 4, 10.0
gen = torch.Generator(device=dev).manual_seed(5)
msg = torch.randint(0, 2, (N, K), generator=gen, device=dev).float()
cw = mvn.rs_encode(msg, nsym)
h = np.concatenate([mvn.estimate_channel(L, 0.2, "cost2100", index=i) for i in range(N)])
y = mvn.transmit(cw, h, snr, L, torch.randn(N, K + 8 * nsym, generator=gen, device=dev))
det = mvn.VNETDetector(16, {"train": 136, "val": 136}).to(dev)
with torch.no_grad():
    for p, i in zip(det.parameters(), range(6)): p.copy_(torch.tensor(g7[f"w{i}"]))
mvn.eval_by_word(det, msg, y, snr, 0.2, nsym, 25)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): mvn.eval_by_word(det, msg, y, snr, 0.2, nsym, 25)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
