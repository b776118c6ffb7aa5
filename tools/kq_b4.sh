#!/bin/bash
# Runs ON THE GPU BOX: kernel stats of the B=4 inference step (top kernels, per-step calls assume 60 cell launches per step)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kq; mkdir -p gpurun_out/kq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kq -- python3 bench.py --batch 4 --steps 10 --warmup 2 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/kq 12 14
rm -rf gpurun_out/kq
