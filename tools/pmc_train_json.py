#!/usr/bin/env python3
"""The per-dispatch PMC table of tools/prof_train_kernels.py (tools/pmc_sq.sh + tools/pmc_dispatch_table.py) -> train_pmc.json:
per form of the training kernels, the counters PER ITERATION AND TRIAL and the fractions that say what the launch waits for.
bench.py derives the training entries' MFMA roofline from it (executed MFMA instructions x 2048 FLOP / time); stamped with the
sha of the kernel sources like profiles/traffic.json.
usage: pmc_train_json.py <table.csv> <reps used for the profiled run> <out.json>"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table, reps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
# prof_train_kernels.py's CASES, in launch order: (name, kernel, trials, iterations per launch)
CASES = [("online_minibatch", "online_train_kernel<16, true, 1024>", 256, 200), ("online_full_word", "online_train_kernel<16, true, 1024>", 256, 200),
         ("online_full_word_chunked", "online_train_groups_kernel<16, true>", 48, 200), ("maml_second_order", "maml_train_kernel<16, true>", 256, 40),
         ("maml_second_order_chunked", "maml_train_groups_kernel<16, true>", 48, 40)]


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "meta-viterbinet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


rows = {}
for r in csv.DictReader(open(table)):
    rows.setdefault(r["kernel"], []).append(r)
seen = {}
res = {"_note": "SQ counters per launch of the training kernels (tools/prof_train_kernels.py under tools/pmc_sq.sh: separate rocprofv3 --pmc "
                "passes), reduced to per-iteration-and-trial counts and to fractions of the launch: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(1024 SIMDs x launch cycles), the launch's cycles from SQ_WAVE_CYCLES x 4 / SQ_WAVES (all waves live for the whole launch); "
                "lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles); wait_* = share of the waves' cycles (SQ_WAVE_CYCLES) parked at a barrier / "
                "s_waitcnt (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY).",
       "csrc_sha16": csrc_sha16(), "cases": {}}
for name, kernel, trials, iters in CASES:
    k = seen.get(kernel, 0)
    mine = rows[kernel][k * (reps + 1) + 1:(k + 1) * (reps + 1)]  # skip the warm-up launch of the case
    seen[kernel] = k + 1
    f = lambda c: sum(float(r[c]) for r in mine) / len(mine)  # noqa: E731
    cycles = f("SQ_WAVE_CYCLES") * 4.0 / f("SQ_WAVES")
    per = trials * iters
    res["cases"][name] = {
        "kernel": kernel, "trials": trials, "iterations_per_launch": iters, "ms_profiled": f("ms_profiled"), "launch_cycles": cycles,
        "mfma_per_iteration": f("SQ_INSTS_MFMA") / per, "valu_non_mfma_per_iteration": (f("SQ_INSTS_VALU") - f("SQ_INSTS_MFMA")) / per,
        "lds_insts_per_iteration": f("SQ_INSTS_LDS") / per, "salu_per_iteration": f("SQ_INSTS_SALU") / per,
        "mfma_busy": f("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cycles), "lds_busy": f("SQ_LDS_IDX_ACTIVE") / (256.0 * cycles),
        "lds_bank_conflict_share_of_lds_cycles": f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE"),
        "wait_barrier_or_waitcnt": f("SQ_WAIT_ANY") / f("SQ_WAVE_CYCLES"), "wait_issue": f("SQ_WAIT_INST_ANY") / f("SQ_WAVE_CYCLES"),
        "issuing": f("SQ_ACTIVE_INST_ANY") / f("SQ_WAVE_CYCLES"), "wait_issue_lds": f("SQ_WAIT_INST_LDS") / f("SQ_WAVE_CYCLES"),
        "mfma_tflops_in_kernel": f("SQ_INSTS_MFMA") * 2048.0 / (f("ms_profiled") * 1e-3) / 1e12,
    }
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["cases"], indent=1))
