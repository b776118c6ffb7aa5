#!/bin/bash
# round 5, run B: fused vs two-pass training_loss of PredRNN at B = 128 (same session), with per-kernel stats of both
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 1 0; do
  export VPX_BENCH_FUSE_REVERSED=$f
  python3 bench.py --model predrnn-pp --mode train --batch 128 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --name predrnn_train_b128 > gpurun_out/r5_b_fuse$f.json 2> gpurun_out/r5_b_fuse$f.err
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5_b_prof$f -o p -- python3 bench.py --model predrnn-pp --mode train --batch 128 --steps 3 --warmup 1 --prewarm 0 --no-extras --no-cpu-baseline > /dev/null 2> gpurun_out/r5_b_prof$f.err
done
python3 - <<'PY'
import json, glob, csv
for f in (1, 0):
    d = json.loads(open(f"gpurun_out/r5_b_fuse{f}.json").read().strip().splitlines()[-1])
    print("fuse", f, d["ms_per_step"], d["roofline"]["frac"])
    for p in glob.glob(f"gpurun_out/r5_b_prof{f}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(p)))
        for r in rows[:14]:
            print("   ", r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["AverageNs"][:8], r["Percentage"])
PY
