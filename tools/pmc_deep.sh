#!/bin/bash
# Runs ON THE GPU BOX: several separate --pmc passes over tools/pmc_cell.py (bf16x3 headline cell), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_deep
rm -rf $OUT; mkdir -p $OUT
export PREC=${PREC:-bf16x3}
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/pmc_cell.py > $OUT/p$i.log 2>&1 || echo "pass $i failed: $line"
done <<'LIST'
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LEVEL_WAVES
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
LIST
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/pmc_deep"
for d in sorted(glob.glob(out + "/p*/")):
    files = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_gemm_kernel" not in k: continue
            a = agg[(k[:70], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in sorted(agg.items()):
        print(f"{os.path.basename(d.rstrip('/')):4s} {c:36s} mean/launch {s/n:16.1f}  n={n}  {k}")
PY
find $OUT -name "*.csv" -size +200k -delete
