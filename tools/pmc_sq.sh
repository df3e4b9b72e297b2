#!/bin/bash
# SQ counters of a workload in separate rocprofv3 --pmc passes (8 SQ slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# then one summary CSV (median per kernel over the dispatches of its most frequent grid: tools/pmc_summarize.py).
# Usage (on the GPU box, from the repo root):  bash tools/pmc_sq.sh <out dir> <python script> [args...]
set -e
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
cd /tmp
i=0
for C in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES" \
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/pass$i" -- python3 "$R/$1" "${@:2}" > "$R/$OUT/pass$i.log" 2>&1
  echo "pass $i done" >> "$R/$OUT/progress.txt"
done
python3 "$R/tools/pmc_summarize.py" "$R/$OUT" > "$R/$OUT/summary.csv"
