#!/bin/bash
# Runs ON THE GPU BOX: prices parts of the c5 K loop (developer build with -DVPX_C5_ABL linked as build/libvpx_abl5.so; results are wrong by design)
for a in 0 1 2 3 4 7 8 16 32 48 56 63; do echo "abl $a: $(VPX_LIB=build/libvpx_abl5.so VPX_C5_ABLATE=$a MODE=${MODE:-cell} BB=${BB:-4} python tools/stamp_c5.py | head -1)"; done
