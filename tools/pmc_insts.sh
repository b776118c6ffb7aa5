#!/bin/bash
# Runs ON THE GPU BOX: instruction-mix counters of the headline cell (tools/pmc_cell.py), one --pmc pass each
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_insts; rm -rf $OUT; mkdir -p $OUT
export PREC=${PREC:-bf16x3}
i=0
for line in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/pmc_cell.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_insts/p*/")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_gemm_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for c, (s, n) in sorted(agg.items()):
        print(f"{c:32s} {s/n:16.0f}  n={n}")
PY
find $OUT -name "*.csv" -delete
