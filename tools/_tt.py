import os, sys, time
sys.argv=[sys.argv[0]]
exec(open('/root/repo/tools/time_trials.py').read().split("Rs = [int(a)")[0])
for name,(coef,kw) in FLOWS.items():
    for R in ([64,128,256] if 'configs[2]' in name else [28,56]):
        ws = [words(coef, 7.0 + (i % 6), 100 + i) for i in range(R)]
        msg, rx = torch.stack([a for a, _ in ws]), torch.stack([b for _, b in ws])
        for rep in range(2):
            bank = TrialBank([w] * R, 16, L, dev)
            draws = [TrialDraws(100 + i, dev) for i in range(R)]
            if "meta" not in name:
                for d in draws: d.batches(0, N, T, 200, 32)
            rec={"timing":True}
            torch.cuda.synchronize(); t0=time.perf_counter()
            ser = eval_by_word_batched(bank, msg, rx, nsym, sub, draws, record=rec, **kw)
            torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(name[:10], R, f"{dt*1e3:.1f} ms  {R*N/dt:.0f} blocks/s", {k: round(v*1e3,1) for k,v in rec["timing"].items()})
