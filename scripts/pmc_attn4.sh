#!/bin/bash
# SQ counters of the attention core with three (default) and four waves per SIMD, split-fp16 and
# single-pass fp16 -> gpurun_out/pmc_attn4/*.txt (copied to profiles/r04_attn_counters.txt by hand)
set -o pipefail
OUT=gpurun_out/pmc_attn4
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "1 3" "1 4" "2 3" "2 4"; do
  set -- $cfg
  export SPR_ATTN_MODE=$1 SPR_ATTN_WPS=$2
  tag=mode$1_wps$2
  timeout -k 10 100 python scripts/attn_abl.py > $OUT/$tag.time 2>&1
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag.p$i -- python scripts/attn_abl.py > $OUT/$tag.p$i.log 2>&1 || { echo "$tag pass $i failed"; continue; }
    python scripts/pmc_summary.py $(ls $OUT/$tag.p$i/*/*counter_collection.csv | head -1) | grep -i "attn_h3" | cut -c1-600 >> $OUT/$tag.txt
  done
  echo "== $tag: $(grep us/call $OUT/$tag.time)"; cat $OUT/$tag.txt
done
