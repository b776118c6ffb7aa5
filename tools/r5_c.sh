#!/bin/bash
# round 5, run C: deferred ST-LSTM weight gradients — parity tests, then the PredRNN training workloads with the switch on and off
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_stlstm.py tests/test_gpu_parity_r4.py tests/test_gpu_dp.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5_c_tests.log 2>&1; tail -3 gpurun_out/r5_c_tests.log
for defer in 1 0; do
  export VPX_BENCH_DEFER_WGRAD=$defer
  python3 bench.py --model predrnn-pp --mode train --batch 2 --img 128 --channels 3 --context 10 --pred 30 --layers 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --name c5_train_b2 > gpurun_out/r5_c_c5_train_b2_$defer.json 2> gpurun_out/r5_c_c5_train_b2_$defer.err
  python3 bench.py --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --name predrnn_train_b128 > gpurun_out/r5_c_predrnn_train_b128_$defer.json 2> gpurun_out/r5_c_predrnn_train_b128_$defer.err
done
python3 - <<'PY'
import json
for n in ("c5_train_b2", "predrnn_train_b128"):
    for defer in (1, 0):
        try:
            d = json.loads(open(f"gpurun_out/r5_c_{n}_{defer}.json").read().strip().splitlines()[-1])
            print(n, "defer", defer, d["ms_per_step"], d["roofline"]["frac"])
        except Exception as e:
            print(n, defer, "failed", e, open(f"gpurun_out/r5_c_{n}_{defer}.err").read()[-1500:])
PY
