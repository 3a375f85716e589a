#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun from the repo root):
#   1. bench.py (default flags, with cpu_baseline)            -> gpurun_out/final/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command    -> gpurun_out/final/stats/
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE)          -> gpurun_out/final/pmc_{fetch,write}/
# scripts/collect_round.py then turns these into the files committed under profiles/.
set -eo pipefail
OUT=gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python bench.py > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log > $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu-baseline --no-extra-legs > $OUT/stats.log 2>&1
grep '^{' $OUT/stats.log > $OUT/bench_under_rocprof.json
#   2b. the same with single-pass fp16 attention operands (mode 2): the attention roofline of BASELINE configs[4]
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_attn2 -- python bench.py --attn-mode 2 --no-cpu-baseline --no-extra-legs > $OUT/stats_attn2.log 2>&1
grep '^{' $OUT/stats_attn2.log > $OUT/bench_attn2_under_rocprof.json
#   2c. (round 5) split-fp16 scores with one probability plane (mode 3)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_attn3 -- python bench.py --attn-mode 3 --no-cpu-baseline --no-extra-legs > $OUT/stats_attn3.log 2>&1
grep '^{' $OUT/stats_attn3.log > $OUT/bench_attn3_under_rocprof.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/pmc_write.log 2>&1
cut -c1-400 $OUT/bench.json
