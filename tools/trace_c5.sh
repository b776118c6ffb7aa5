#!/bin/bash
# Runs ON THE GPU BOX: kernel sequence of the last cell steps of one configs[4]-shard forward (predrnn-pp, B=4, 128x128x3, 4 layers, 10->30)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --model predrnn-pp --batch 4 --img 128 --channels 3 --pred 30 --layers 4 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/kt/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vpx::", "")[:60] for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
n = len(names)
last = names[n - n // 2:]          # the timed forward (second half of the run)
c = collections.Counter(last); t = collections.Counter()
for nm, d in zip(names[n - n // 2:], dur[n - n // 2:]): t[nm] += d
print(len(last), "launches in the timed forward;", sum(t.values()) / 1e6, "ms of kernel time")
for nm, k in c.most_common(25): print(f"{k:6d}  {t[nm] / 1e3:9.1f} us total  {t[nm] / k / 1e3:7.1f} us avg  {nm}")
print("--- the last 40 launches:")
print("\n".join(names[-40:]))
PY
rm -rf gpurun_out/kt
