#!/bin/bash
# Runs ON THE GPU BOX (developer build): c3 (c5_kernel<NT, 3>) against the half tile on grids around the selection rule
export VPX_LIB=build/libvpx_ablate.so
run() { python3 bench.py "$@" --no-extras --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for cfg in "--batch 4" "--batch 4 --img 128 --channels 3 --pred 20" "--batch 8" "--cell 64,64,64,64 --batch 4" "--cell 64,64,64,64 --batch 8" "--cell 64,96,64,64 --batch 4"; do
  echo "== $cfg"
  for v in "VPX_C3_MAX=0" "VPX_C3_MAX=128 VPX_C3_NT=2" "VPX_C3_MAX=128 VPX_C3_NT=4" "VPX_C3_MAX=1024 VPX_C3_NT=2" "VPX_C3_MAX=1024 VPX_C3_NT=4"; do
    echo "   $v: $(env $v bash -c "$(declare -f run); run $cfg")"
  done
done
