#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace stats of one bench configuration -> gpurun_out/pq/<tag>_kernel_stats.csv (small)
# usage: bash tools/prof_quick.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/pq; mkdir -p $OUT/$tag
python3 bench.py "$@" --no-cpu-baseline > /dev/null 2>&1    # warm-up outside the profile (MIOpen find, first import)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.log 2>&1
f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${tag}_kernel_stats.csv
rm -rf $OUT/$tag
tail -1 $OUT/$tag.log | cut -c1-200
