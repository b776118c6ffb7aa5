#!/bin/bash
# Runs ON THE GPU BOX: MFMA-busy / clock / instruction mix of the weight-gradient kernel in one training step
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wgrad; rm -rf $OUT; mkdir -p $OUT
i=0
for line in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --mode train --steps 1 --warmup 1 --batch ${BB:-32} --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
KERNEL = os.environ.get("WG_KERNEL", "wgrad_tg_kernel<9, 8, 6>")  # the 3x3 tap-group form; WG_KERNEL=... for another
agg = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/pmc_wgrad/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_wgrad/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
us = sum(dur) / len(dur) / 1e3
cyc = m["GRBM_GUI_ACTIVE"] / 8
print(f"{KERNEL}: {len(dur)} launches, avg {us:.0f} us, {cyc/us/1e3:.2f} GHz, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc)*100:.1f} %")
for k in sorted(m): print(f"  {k:28s} {m[k]:16.0f}")
PY
find $OUT -name "*.csv" -delete
