#!/bin/bash
# Soak of the GPU suite: N consecutive runs of the driver's exact command, each on a FRESH lease (one gpurun call per run), whole logs kept.
# usage: tools/soak.sh N   -> gpurun_out/soak_<i>_suite.log, summary lines in gpurun_out/soak_summary.txt
n=${1:-10}
for i in $(seq 1 $n); do
  tools/gpurun_retry.sh 1200 "tools/suite_run.sh soak_$i" > /tmp/soak_$i.out 2>&1
  lease=$(grep -o "status=[a-z]* rc=[0-9-]* charged=[0-9.]*s" /tmp/soak_$i.out | tail -1)
  echo "run $i: $(tail -2 gpurun_out/soak_${i}_suite.log | tr '\n' ' ') | $lease" | tee -a gpurun_out/soak_summary.txt
done
