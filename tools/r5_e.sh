#!/bin/bash
# round 5, run E: batched decoupling tail — parity tests, then the PredRNN workloads
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_stlstm.py tests/test_gpu_parity_r4.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5_e_tests.log 2>&1; tail -3 gpurun_out/r5_e_tests.log
b() { python3 bench.py "$@" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', d['ms_per_step'], d['roofline']['frac'])"; }
b --model predrnn-pp --mode train --batch 2 --img 128 --channels 3 --context 10 --pred 30 --layers 4 --steps 10 --warmup 3
b --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2
b --model predrnn-pp --mode train --batch 32 --steps 10 --warmup 2
b --model predrnn-pp --mode infer --batch 4 --img 128 --channels 3 --context 10 --pred 30 --layers 4 --steps 20 --warmup 3
b --model predrnn-pp --mode infer --batch 128 --steps 10 --warmup 2
