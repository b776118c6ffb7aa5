timeout 900 python -m pytest tests/test_gpu_cell2.py tests/test_gpu_convlstm.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --mode train --steps 8 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'])"; done
timeout 500 bash tools/prof_quick.sh train --mode train --steps 6 --warmup 2 --no-extras < /dev/null
