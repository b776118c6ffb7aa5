#!/bin/bash
# round 5, run D: the stage glue's data gradients on convq (default) vs the first-generation launch (experiment bit 14), ConvLSTM training
for bit in 0 16384 0 16384; do
  for b in 128 32; do
    VPX_BENCH_EXPERIMENT=$bit python3 bench.py --mode train --batch $b --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bit', $bit, 'B', $b, 'ms', d['ms_per_step'])"
  done
done
