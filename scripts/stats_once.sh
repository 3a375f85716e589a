#!/bin/bash
# one rocprofv3 kernel-stats pass of the default bench -> gpurun_out/stats_once/
set -eo pipefail
OUT=gpurun_out/stats_once
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python bench.py --no-cpu-baseline --no-extra-legs "$@" > $OUT/a.log 2>&1
grep -h '^{' $OUT/a.log | cut -c1-160
