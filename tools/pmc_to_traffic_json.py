#!/usr/bin/env python3
"""summary.csv of tools/pmc_traffic.sh -> traffic.json (HBM bytes per launch) and valu_insts.json (VALU wave-instructions
per launch), both read by bench.py and both stamped with the sha of the kernel sources they were measured on.

usage: pmc_to_traffic_json.py <summary.csv> <output dir>
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: gfx950's FETCH_SIZE tallies 128-B read requests at 64 B
(MI355X_MICROARCH.md, HBM); the TCC_EA0_RDREQ count * 128 B is kept next to it as the cross-check."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, outdir = sys.argv[1], sys.argv[2]


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "meta-viterbinet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


rows = collections.defaultdict(dict)
for r in csv.DictReader(open(src)):
    rows[r["kernel"]][r["counter"]] = float(r["mean_per_dispatch"])
sha = csrc_sha16()
traffic = {
    "_note": "HBM bytes per launch from rocprofv3 PMC (separate passes, tools/pmc_traffic.sh: bench.py --skip-fused-count so "
             "every fused dispatch is decode-only) on bench.py's workload. bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: "
             "gfx950's FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM); cross-check: "
             "TCC_EA0_RDREQ_sum*128 B == 2*FETCH_SIZE*1024.",
    "csrc_sha16": sha,
    "workload": {"blocks": 10000, "block_length": 1000, "n_states": 16},
}
valu = {
    "_note": "SQ_INSTS_VALU / SQ_INSTS_MFMA per launch (rocprofv3 PMC pass of tools/pmc_traffic.sh) of the VALU-bound kernels of "
             "bench.py's configs entries; SQ_INSTS_VALU includes the MFMA instructions.",
    "csrc_sha16": sha,
}
shapes = {"sweep16_rows_kernel<2>": (100, 1000, 16), "va16_tile_kernel": (100, 1000, 16), "va16_split_kernel": (100, 1000, 16), "va_inplace_kernel<6>": (125000, 1000, 256),
          "va256_wave_kernel": (125000, 1000, 256)}
for k in sorted(rows):
    c = rows[k]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic[k] = {"fetch_kb_raw": c["FETCH_SIZE"], "write_kb_raw": c["WRITE_SIZE"], "rdreq": c.get("TCC_EA0_RDREQ_sum"),
                      "bytes_per_launch": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024}
    if "SQ_INSTS_VALU" in c:
        e = {"valu_insts_per_launch": c["SQ_INSTS_VALU"], "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA"),
             "salu_insts_per_launch": c.get("SQ_INSTS_SALU"), "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "waves": c.get("SQ_WAVES")}
        if k in shapes:
            e.update(dict(zip(("blocks", "block_length", "n_states"), shapes[k])))
        valu[k] = e
json.dump(traffic, open(os.path.join(outdir, "traffic.json"), "w"), indent=1)
json.dump(valu, open(os.path.join(outdir, "valu_insts.json"), "w"), indent=1)
print("wrote traffic.json, valu_insts.json for csrc", sha)
