#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace stats + HBM-traffic PMC passes of one bench configuration (args after the tag go to bench.py).
# usage: tools/prof_extra.sh <tag> <bench.py args...>   -> gpurun_out/prof_extra/<tag>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof_extra/$TAG
rm -rf $OUT; mkdir -p $OUT
python3 bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py "$@" --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py "$@" --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py "$@" --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -path "*trace/*" -delete
find $OUT -name "*agent_info.csv" -delete
head -12 $OUT/trace/*/*kernel_stats.csv | cut -c1-150
tail -c 400 $OUT/bench.log
