#!/bin/bash
# Runs ON THE GPU BOX: MFMA-busy and clock of the cell kernel under a timing ablation (VPX_DBG bits, ablation library)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export VPX_LIB=$PWD/build/libvpx_ablate.so PREC=bf16x3
OUT=gpurun_out/pmc_abl; rm -rf $OUT; mkdir -p $OUT
for d in ${BITS:-0 206}; do
  export VPX_DBG=$d
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/d$d -- python3 tools/pmc_cell.py > $OUT/d$d.log 2>&1
  python3 - $OUT/d$d $d <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm_kernel" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
us = sum(dur) / len(dur) / 1e3
cyc = m["GRBM_GUI_ACTIVE"] / 8
print(f"VPX_DBG={tag}: {us:.1f} us/launch, {cyc/1e3:.0f} k cycles -> {cyc/us/1e3:.2f} GHz, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc)*100:.1f} %")
PY
  find $OUT/d$d -name "*.csv" -delete
done
