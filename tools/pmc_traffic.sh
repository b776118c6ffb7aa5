#!/bin/bash
# Runs ON THE GPU BOX: HBM read/write of the fused cell launches in the default bench, with and without the XCD-aware mapping
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
for x in 0 1; do
  export VPX_XCD_MAP=$x
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/x${x}_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  done
  python3 - $OUT $x <<'PY'
import csv, glob, sys
out, x = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{out}/x{x}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "EpiConvLSTM" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    tot[c] = sum(v) / max(len(v), 1)
print(f"VPX_XCD_MAP={x}: per fused launch read {2*tot['FETCH_SIZE']*1024/1e6:.1f} MB (2 x FETCH_SIZE), write {tot['WRITE_SIZE']*1024/1e6:.1f} MB")
PY
done
find $OUT -name "*.csv" -delete
