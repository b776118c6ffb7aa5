#!/bin/bash
# Runs ON THE GPU BOX: average duration / algorithmic TFLOP/s of the weight-gradient kernel for the three
# convlstm-shi block shapes (one block forward + backward, T = 4, two iterations) from a rocprofv3 kernel trace.
# usage: BB=128 bash tools/wgrad_quick.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/wq; rm -rf $OUT; mkdir -p $OUT
for shape in 64,64,64,64,3 96,96,32,32,3 96,96,16,16,3 ${EXTRA_SHAPES}; do
  export SHAPE=$shape
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/pmc_block_train.py > /dev/null 2>&1
  f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, os, sys
Cin, Ch, H, W, K = (int(v) for v in os.environ["SHAPE"].split(","))
B, T = int(os.environ.get("BB", 32)), 4
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "wgrad" in n and "reduce" not in n:
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        fl = 2.0 * 4 * Ch * K * K * B * H * W * (Cin * T + Ch * (T - 1))  # one backward; h is absent at t = 0
        print(f"{os.environ['SHAPE']:>16s} B={B} {n[:58]:58s} calls {calls} avg {tot/calls/1e3:9.1f} us  {fl * 2 / tot / 1e3:7.1f} TF")
PY
  rm -rf $OUT/t
done
