#!/bin/bash
# Round 5: SQ counters of the attention core alone (scripts/attn_abl.py, the bench's 32-pair call), separate --pmc
# passes, for the split-fp16 default (mode 1) and the single-probability-plane mode 3 -> profiles/r05_attn_counters.txt
set -o pipefail
OUT=gpurun_out/pmc_attn5
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NSEG=64
for mode in 1 3; do
  export SPR_ATTN_MODES=$mode
  echo "== mode $mode: $(timeout -k 10 100 python scripts/attn_abl.py 2>/dev/null | grep 'mode' | cut -c1-200)"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/m${mode}_p$i -- python scripts/attn_abl.py > $OUT/m${mode}_p$i.log 2>&1 || { tail -3 $OUT/m${mode}_p$i.log; echo "pass $i failed"; continue; }
    python scripts/pmc_summary.py $(ls $OUT/m${mode}_p$i/*/*counter_collection.csv | head -1) | grep -i "attn_s" | cut -c1-700
  done
done | tee gpurun_out/pmc_attn5.txt
