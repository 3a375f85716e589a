#!/bin/bash
# SQ counters of the fused cross-encoder chains inside the bench (separate --pmc passes) -> gpurun_out/pmc_xenc/
set -o pipefail
OUT=gpurun_out/pmc_xenc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  python scripts/pmc_summary.py $(ls -t $OUT/p$i/*/*counter_collection.csv | head -1) | grep -i "xenc_chain\|attn_s" | cut -c1-420
done
