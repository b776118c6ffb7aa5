#!/bin/bash
# Runs ON THE GPU BOX: HBM bytes fetched / written by the weight-gradient kernel of one ConvLSTM block backward
# (tools/pmc_block_train.py; SHAPE / BB as there). read = 2 x FETCH_SIZE KiB-units per the gfx950 correction.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wt; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p_${c%% *} -- python3 tools/pmc_block_train.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_wt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    m = sum(v) / len(v)
    extra = f"  = {2 * m * 1024 / 1e9:.2f} GB read" if k == "FETCH_SIZE" else (f"  = {m * 1024 / 1e9:.2f} GB written" if k == "WRITE_SIZE" else "")
    print(f"{k:16s} {m:16.0f} per launch ({len(v)} launches){extra}")
PY
find $OUT -name "*.csv" -delete
