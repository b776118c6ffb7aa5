#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace of the cell kernel under chosen ablation bits -> average kernel duration per variant
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export VPX_LIB=$PWD/build/libvpx_ablate.so PREC=${PREC:-bf16x3}
OUT=gpurun_out/abl; rm -rf $OUT; mkdir -p $OUT
for d in ${BITS:-0 15 47}; do
  export VPX_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$d -- python3 tools/pmc_cell.py > $OUT/d$d.log 2>&1
  f=$(find $OUT/d$d -name "*kernel_stats.csv" | head -1)
  echo "VPX_DBG=$d"; head -4 "$f" | cut -c1-200
  find $OUT/d$d -name "*kernel_trace.csv" -delete
done
