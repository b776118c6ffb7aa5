#!/bin/bash
# Runs ON THE GPU BOX (developer build): the grid size from which the c5 launches run unsplit (forward: VPX_C5_MIN_TILES, backward: VPX_C5B_MIN_TILES)
export VPX_LIB=build/libvpx_ablate.so
for b in 40 48 56 64 80; do
  for m in 32 1000; do
    echo "fwd B=$b min_tiles=$m: $(VPX_C5_MIN_TILES=$m MASKS=0 BB=$b IMG=64 CH=1 PRED=10 LAYERS=4 MODE=infer timeout 200 python3 tools/ab_predrnn.py 2>&1 | tail -1 | cut -c40-70)"
  done
done
for b in 48 64 80; do
  for m in 32 1000; do
    echo "train B=$b bwd min_tiles=$m: $(VPX_C5_MIN_TILES=64 VPX_C5B_MIN_TILES=$m MASKS=0 BB=$b IMG=64 CH=1 PRED=10 LAYERS=4 MODE=train timeout 300 python3 tools/ab_predrnn.py 2>&1 | tail -1 | cut -c40-70)"
  done
done
