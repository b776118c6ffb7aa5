#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE / clock passes over tools/pmc_cell.py for a list of EXP values. BB = batch.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  OUT=gpurun_out/pmc_fetch_$e
  rm -rf $OUT; mkdir -p $OUT
  export EXP=$e
  i=0
  for line in "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" ${EXTRA_PASS:+"$EXTRA_PASS"}; do
    i=$((i+1))
    rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/pmc_cell.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  done
  python3 - "$OUT" "$e" <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*/")):
    agg = collections.defaultdict(lambda: [0.0, 0]); dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "cell2_kernel" not in k and "conv_gemm" not in k: continue
            a = agg[(k[:48], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "cell2_kernel" not in k and "conv_gemm" not in k: continue
            a = dur[k[:48]]; a[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; a[1] += 1
    for (k, c), (s, n) in sorted(agg.items()):
        print(f"EXP={sys.argv[2]} {c:28s} {s/n:16.1f} n={n} {k}")
    for k, (s, n) in sorted(dur.items()):
        print(f"EXP={sys.argv[2]} {'duration_us':28s} {s/n:16.1f} n={n} {k}")
PY
  find $OUT -name "*.csv" -size +200k -delete
done
