#!/bin/bash
# Runs ON THE GPU BOX: timing ablations of the cell kernel (library built with `make -C vp-suite_amd/csrc ablate`).
# VPX_DBG bits: 1 no MFMA/fragment reads, 2 no activation loads, 4 no weight loads, 8 no epilogue, 32 no chunk barrier
export VPX_LIB=$PWD/build/libvpx_ablate.so PREC=${PREC:-bf16x3}
for d in 0 1 2 4 8 32 6 14 15 47; do
  echo "VPX_DBG=$d"; VPX_DBG=$d python3 tools/ab_bench_cell.py 2>&1 | grep -v amdgpu | head -1
done
