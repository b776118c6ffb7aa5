#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libvpx_hip (device code un-instrumented: GPU ASan is not available on this pool) and a run of
# every CPU test that drives the library's host code — tests/test_workspace_contract.py (the dry-run sweep over every entry point: all
# kernel selection, planning, carving and bounds-check paths) and tests/test_host_logic.py. CPU only; objects and the .so go to /tmp/asan.
# Round 5: 43 passed, no sanitizer report. Round 6 (final tree, with the acstlstm / bwd_ex / c16 entry points in the sweep): 48 passed, none.
set -e
OUT=${VPX_ASAN_DIR:-/tmp/asan}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $OUT
cd "$ROOT/vp-suite_amd/csrc"
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function"
for f in *.hip; do
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $OUT/${f%.hip}.o &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o $OUT/libvpx_asan.so $OUT/*.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 VPX_LIB=$OUT/libvpx_asan.so \
  python3 -m pytest tests/test_workspace_contract.py tests/test_host_logic.py -q -s -p no:cacheprovider > $OUT/run.log 2>&1 || true
tail -2 $OUT/run.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' $OUT/run.log)"
