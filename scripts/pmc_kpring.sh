#!/bin/bash
# SQ counters of the ring KPConv kernel alone (level-1 conv launches of scripts/kpconv_ring_bench.py), separate --pmc passes.
OUT=gpurun_out/pmc_kpring
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LEVELS=${LEVELS:-1} KINDS=neighbors
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_ICACHE_MISSES SQ_ICACHE_HITS SQ_WAIT_IFETCH"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 scripts/kpconv_ring_bench.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed"; continue; }
  python scripts/pmc_summary.py $(ls $OUT/p$i/*/*counter_collection.csv | head -1) | grep -i "kpconv_ring\|kpconv_mfma" | cut -c1-700
done
