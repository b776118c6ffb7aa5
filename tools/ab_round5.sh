#!/bin/bash
# Runs ON THE GPU BOX: the A/B measurements of round 5, each a pair of bench.py runs in ONE session.
#   tools/ab_round5.sh fuse      sequence + reversal as one 2B batch vs two passes        (VPX_BENCH_FUSE_REVERSED)
#   tools/ab_round5.sh defer     deferred ST-LSTM weight gradients vs per-step            (VPX_BENCH_DEFER_WGRAD)
#   tools/ab_round5.sh glue      stage-glue data gradients on convq vs first generation   (VPX_BENCH_EXPERIMENT=16384)
#   tools/ab_round5.sh predrnn   the five PredRNN workloads of the extras, as they are
b() { python3 bench.py "$@" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['ms_per_step'], 'ms  frac', d['roofline']['frac'])"; }
SHARD="--model predrnn-pp --batch 2 --img 128 --channels 3 --context 10 --pred 30 --layers 4"
case "$1" in
  fuse)  for v in 1 0 1 0; do echo "fuse_reversed_pass=$v"; VPX_BENCH_FUSE_REVERSED=$v b $SHARD --mode train --steps 10 --warmup 3; VPX_BENCH_FUSE_REVERSED=$v b --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2; done ;;
  defer) for v in 1 0 1 0; do echo "defer_weight_gradients=$v"; VPX_BENCH_DEFER_WGRAD=$v b $SHARD --mode train --steps 10 --warmup 3; VPX_BENCH_DEFER_WGRAD=$v b --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2; done ;;
  glue)  for v in 0 16384 0 16384; do echo "experiment bits $v"; VPX_BENCH_EXPERIMENT=$v b --mode train --batch 128 --steps 10 --warmup 3; VPX_BENCH_EXPERIMENT=$v b --mode train --batch 32 --steps 10 --warmup 3; done ;;
  predrnn)
    echo "shard training";  b $SHARD --mode train --steps 10 --warmup 3
    echo "training B=128";  b --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2
    echo "training B=32";   b --model predrnn-pp --mode train --batch 32 --steps 10 --warmup 2
    echo "shard inference B=4"; b --model predrnn-pp --mode infer --batch 4 --img 128 --channels 3 --context 10 --pred 30 --layers 4 --steps 20 --warmup 3
    echo "inference B=128"; b --model predrnn-pp --mode infer --batch 128 --steps 10 --warmup 2 ;;
  *) echo "usage: $0 fuse|defer|glue|predrnn"; exit 2 ;;
esac
