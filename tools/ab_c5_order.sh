#!/bin/bash
# Runs ON THE GPU BOX (developer build): block -> (pixel tile, N tile) order of c5_kernel (VPX_C5_ORDER bits: 1 = 128-column launches,
# 2 = 64-column, 4 = 32-column take the N-tile-major order), time and L2-miss bytes of the predrnn-pp training step
export VPX_LIB=build/libvpx_ablate.so
for o in ${ORDERS:-1 3 7}; do
  export VPX_C5_ORDER=$o
  echo "== VPX_C5_ORDER=$o"
  timeout 200 python3 bench.py --model predrnn-pp --mode train --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c60-100,190-215
  timeout 400 bash tools/pmc_quick.sh FETCH_SIZE --model predrnn-pp --mode train | grep c5_kernel
done
