#!/usr/bin/env python3
"""summary.csv of tools/pmc_traffic.sh -> profiles/traffic.json (HBM bytes per launch, read by bench.py).

usage: pmc_to_traffic_json.py <summary.csv> <blocks> <block_length> <n_states> <raw-passes file name for the note>
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: gfx950's FETCH_SIZE tallies 128-B read requests at 64 B
(MI355X_MICROARCH.md, HBM); the TCC_EA0_RDREQ count * 128 B is kept next to it as the cross-check."""
import collections
import csv
import json
import sys

src, blocks, length, states, raw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(src)):
    rows[r["kernel"]][r["counter"]] = float(r["mean_per_dispatch"])
out = {
    "_note": "HBM bytes per launch from rocprofv3 PMC (separate passes, tools/pmc_traffic.sh: bench.py --skip-fused-count so "
             "every fused dispatch is decode-only) on bench.py's workload. bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: "
             "gfx950's FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM); cross-check: "
             f"TCC_EA0_RDREQ_sum*128 B == 2*FETCH_SIZE*1024. Raw passes: profiles/{raw}.",
    "workload": {"blocks": blocks, "block_length": length, "n_states": states},
}
for k in sorted(rows):
    c = rows[k]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out[k] = {"fetch_kb_raw": c["FETCH_SIZE"], "write_kb_raw": c["WRITE_SIZE"], "rdreq": c.get("TCC_EA0_RDREQ_sum"),
                  "bytes_per_launch": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024}
json.dump(out, sys.stdout, indent=1)
print()
