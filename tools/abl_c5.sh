#!/bin/bash
# Runs ON THE GPU BOX: prices parts of the c5 K loop by leaving them out (results are wrong by design). Needs a developer library whose
# convq.hip was compiled with -DVPX_DEV_SWITCHES -DVPX_C5_ABL and linked as build/libvpx_abl5.so (make ablate builds everything else:
#   hipcc $CXXFLAGS -DVPX_DEV_SWITCHES -DVPX_C5_ABL -c vp-suite_amd/csrc/convq.hip -o /tmp/convq_abl.o
#   hipcc -shared -fPIC --offload-arch=gfx950 -o build/libvpx_abl5.so $(ls build/obj/*.o | grep -v convq.o) /tmp/convq_abl.o).
# The mask's own branches cost the loop a third: read the differences, not the absolute numbers.
for a in 0 1 2 3 4 7 8 16 32 48 56 63; do echo "abl $a: $(VPX_LIB=build/libvpx_abl5.so VPX_C5_ABLATE=$a MODE=${MODE:-cell} BB=${BB:-4} python tools/stamp_c5.py | head -1)"; done
