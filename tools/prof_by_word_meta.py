"""cProfile of the Meta-ViterbiNet online flow (BASELINE configs[4] arguments) -- where the host time of eval_by_word goes."""
import cProfile, pstats, os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import meta_viterbinet_amd as mvn
dev = torch.device("cuda:0")
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g7 = np.load(os.path.join(R, "tests", "golden", "g7_by_word.npz"))
N, K, nsym, L, snr = 300, 120, 2, 4, 10.0
gen = torch.Generator(device=dev).manual_seed(5)
msg = torch.randint(0, 2, (N, K), generator=gen, device=dev).float()
cw = mvn.rs_encode(msg, nsym)
h = np.concatenate([mvn.estimate_channel(L, 0.2, "time_decay", index=i, fading=True) for i in range(N)])
y = mvn.transmit(cw, h, snr, L, torch.randn(N, K + 8 * nsym, generator=gen, device=dev))


def run():
    det = mvn.VNETDetector(16, {"train": 136, "val": 136}).to(dev)
    with torch.no_grad():
        for p, i in zip(det.parameters(), range(6)):
            p.copy_(torch.tensor(g7[f"w{i}"]))
    torch.manual_seed(0)
    return mvn.eval_by_word(det, msg, y, snr, 0.2, nsym, 25, self_supervised=True, online_trainer=mvn.OnlineTrainer(det, L),
                            self_supervised_iterations=200, online_meta=True, meta_detector=mvn.META_VNETDetector(16, {"train": 136, "val": 136}),
                            meta_train_iterations=20, meta_j_num=10, meta_subframes=5, meta_style_online_training=True)


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
ser = run()
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0):.1f} ms for {N} blocks, mean ser {np.mean(ser):.4f}")
pr = cProfile.Profile(); pr.enable()
run()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
