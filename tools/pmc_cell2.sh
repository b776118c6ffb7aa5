#!/bin/bash
# Runs ON THE GPU BOX: separate --pmc passes (kernel-trace only) over tools/pmc_cell.py for the cell kernel selected by
# VPX_CELL2 (2 = second generation, 0 = first); BB = batch. Output: gpurun_out/pmc_cell2_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-gen2}
OUT=gpurun_out/pmc_cell2_$TAG
rm -rf $OUT; mkdir -p $OUT
export PREC=${PREC:-bf16x3}
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/pmc_cell.py > $OUT/p$i.log 2>&1 || echo "pass $i failed: $line"
done <<'LIST'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
FETCH_SIZE
WRITE_SIZE
LIST
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*/")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_gemm_kernel" not in k and "cell2_kernel" not in k: continue
            a = agg[(k[:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_gemm_kernel" not in k and "cell2_kernel" not in k: continue
            a = dur[k[:60]]
            a[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; a[1] += 1
    for (k, c), (s, n) in sorted(agg.items()):
        print(f"{os.path.basename(d.rstrip('/')):4s} {c:32s} mean/launch {s/n:18.1f}  n={n}  {k}")
    for k, (s, n) in sorted(dur.items()):
        print(f"{os.path.basename(d.rstrip('/')):4s} {'duration_us':32s} mean/launch {s/n:18.1f}  n={n}  {k}")
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +200k -delete
