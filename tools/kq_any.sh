#!/bin/bash
# Runs ON THE GPU BOX: kernel stats of a bench.py configuration (args after the script go to bench.py)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kq; mkdir -p gpurun_out/kq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kq -- python3 bench.py "$@" --steps 6 --warmup 2 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/kq 8 10
rm -rf gpurun_out/kq
