for a in 0 1 2 3 4 8 16 32 48 56 63; do echo "abl $a: $(VPX_C5_ABLATE=$a MODE=cell BB=4 python tools/stamp_c5.py | head -1)"; done
