#!/bin/bash
# kernel-stats A/B of two environment settings of the same build (default vs $1=VALUE), one box
set -eo pipefail
OUT=gpurun_out/ab
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python bench.py --no-cpu-baseline --no-extra-legs > $OUT/a.log 2>&1
export "$1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python bench.py --no-cpu-baseline --no-extra-legs > $OUT/b.log 2>&1
grep -h '^{' $OUT/a.log $OUT/b.log | cut -c1-160
