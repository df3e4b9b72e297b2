#!/bin/bash
# After tools/prof_all.sh has run on the GPU box and gpurun has merged gpurun_out/r05p back: copy what is to be judged into
# profiles/ (tracked), under the round's names, and check that the three JSONs bench.py reads carry the sha of the kernel sources.
# usage (from anywhere): bash tools/collect_profiles.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r05p
P=$R/profiles
cp $O/bench.json $P/r05_bench.json
cp $O/fuzz_parity.txt $P/r05_fuzz_parity.txt
cp $O/headline_by_grid.csv $P/r05_headline_kernel_durations_by_grid.csv
cp $O/headline_kernel_stats.csv $P/r05_headline_kernel_stats.csv
cp $O/headline_pmc_10000.csv $P/r05_headline_pmc_10k.csv
cp $O/headline_pmc_40960.csv $P/r05_headline_pmc_40k.csv
cp $O/kernel_durations_by_grid.csv $P/r05_kernel_durations_by_grid.csv
cp $O/kernel_stats.csv $P/r05_kernel_stats.csv
cp $O/pmc/summary.csv $P/r05_pmc_summary.csv
cp $O/time_step.txt $P/r05_time_byword_step.txt
cp $O/time_online_training.txt $P/r05_time_online_training.txt
cp $O/time_online_states.txt $P/r05_time_online_states.txt
cp $O/time_vnet_states.txt $P/r05_time_vnet_states.txt
cp $O/time_dealt.txt $P/r05_time_dealt.txt
cp $O/time_survivors.txt $P/r05_time_survivors.txt
cp $O/time_montecarlo.txt $P/r05_time_montecarlo.txt
cp $O/time_trials.txt $P/r05_time_trials.txt
cp $O/train_kernels_time.csv $P/r05_train_kernels_time.csv
cp $O/train_pmc_table.csv $P/r05_train_pmc.csv
cp $O/va256_pmc.csv $P/r05_va256_pmc.csv
cp $O/pmc/traffic.json $O/pmc/valu_insts.json $O/train_pmc.json $P/
cd $R && python3 - <<'PY'
import json
import bench
sha = bench.csrc_sha16()
for f in ("traffic", "valu_insts", "train_pmc"):
    j = json.load(open(f"profiles/{f}.json"))
    got = j.get("csrc_sha16") or next(v.get("csrc_sha16") for v in j.values() if isinstance(v, dict))
    print(f"profiles/{f}.json: csrc {got}", "= the sources here" if got == sha else f"!= {sha}: STALE")
PY
