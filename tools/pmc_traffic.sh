#!/bin/bash
# Collect HBM traffic and VALU instruction counters for the bench command in SEPARATE rocprofv3 --pmc passes (TCC slots:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots"), then write
# profiles-style summaries: <out>/summary.csv, <out>/traffic.json, <out>/valu_insts.json (copy them into profiles/).
# Usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh gpurun_out/traffic_r02
set -e
OUT=${1:-gpurun_out/traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/$tag" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-sustained --skip-fused-count > /dev/null 2>&1
done
python3 "$R/tools/pmc_summarize.py" "$R/$OUT" > "$R/$OUT/summary.csv"
python3 "$R/tools/pmc_to_traffic_json.py" "$R/$OUT/summary.csv" "$R/$OUT"
cat "$R/$OUT/summary.csv"
