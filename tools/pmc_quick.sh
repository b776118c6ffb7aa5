#!/bin/bash
# Runs ON THE GPU BOX: one --pmc pass (counters in $1, comma-separated) over a bench.py invocation (remaining args); prints per-kernel means.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CTRS=$(echo $1 | tr ',' ' '); shift
OUT=gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 bench.py "$@" --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = max(glob.glob("gpurun_out/pmc_quick/*/*counter_collection.csv"), key=lambda p: p)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    e = agg[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:6]:
    print(k, {c: round(v[0] / v[1]) for c, v in cs.items()}, "launches", max(v[1] for v in cs.values()))
PY
rm -rf $OUT
