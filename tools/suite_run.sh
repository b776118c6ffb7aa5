#!/bin/bash
# One GPU-box call: the driver's exact test command with the WHOLE log kept (round 4 kept only tails and lost a fault's HSA line),
# then optionally the same suite with NaN-poisoned allocations (VPX_CANARY=2, no -x: every finding of one pass).
# Usage (on the box): tools/suite_run.sh TAG [poison]
tag=${1:-t}
mkdir -p gpurun_out
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/${tag}_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/${tag}_suite.log
tail -5 gpurun_out/${tag}_suite.log
if [ "$2" = "poison" ]; then
  VPX_CANARY=2 python3 -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/${tag}_poison.log 2>&1
  echo "poison rc=$?" >> gpurun_out/${tag}_poison.log
  tail -40 gpurun_out/${tag}_poison.log
fi
