#!/bin/bash
# usage: scripts/prof.sh <tag>   (run on the GPU box via gpurun) -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err || true
find $OUT -name "*kernel_stats*.csv" | head -3
f=$(find $OUT -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > $OUT/kernel_stats_top.csv
tail -2 $OUT/bench.json
