#!/bin/bash
# Runs ON THE GPU BOX: per-kernel time table (rocprofv3 --kernel-trace --stats) of one bench configuration.
# usage: tools/kstats.sh <tag> <bench.py args...>   -> gpurun_out/kstats/<tag>_kernel_stats.csv (+ the bench line of the profiled run)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/kstats
mkdir -p $OUT
rm -rf /tmp/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/${TAG}_bench.json 2> $OUT/${TAG}.err
cp /tmp/ks_$TAG/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1], "total kernel ms", round(tot / 1e6, 2))
for r in rows[:18]:
    print("  %-88s %6s %9.2f ms %8.1f us %5s%%" % (r["Name"][:88], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
tail -c 300 $OUT/${TAG}_bench.json | grep -o '"ms_per_step":[0-9.]*'
