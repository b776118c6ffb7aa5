#!/bin/bash
# round 5, run A: the fused forward + reversed pass — its parity test, then the two PredRNN training workloads it targets
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_stlstm.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5_a_tests.log 2>&1; tail -3 gpurun_out/r5_a_tests.log
python3 bench.py --model predrnn-pp --mode train --batch 2 --img 128 --channels 3 --context 10 --pred 30 --layers 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --name c5_train_b2 > gpurun_out/r5_a_c5_train_b2.json 2> gpurun_out/r5_a_c5_train_b2.err
python3 bench.py --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --name predrnn_train_b128 > gpurun_out/r5_a_predrnn_train_b128.json 2> gpurun_out/r5_a_predrnn_train_b128.err
python3 - <<'PY'
import json
for n in ("c5_train_b2", "predrnn_train_b128"):
    try:
        d = json.loads(open(f"gpurun_out/r5_a_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e:
        print(n, "failed", e, open(f"gpurun_out/r5_a_{n}.err").read()[-800:])
PY
