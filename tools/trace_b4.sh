#!/bin/bash
# Runs ON THE GPU BOX: the kernel sequence of ONE B=4 inference step (names in launch order, consecutive repeats folded)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --batch 4 --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vpx::", "")[:48] for r in rows]
# last step = the last occurrence block starting at the final conv_few_to_16
starts = [i for i, n in enumerate(names) if n.startswith("conv_few_to_16")]
seq = names[starts[-1]:]
out, prev, cnt = [], None, 0
for n in seq:
    if n == prev: cnt += 1
    else:
        if prev: out.append(f"{prev} x{cnt}")
        prev, cnt = n, 1
out.append(f"{prev} x{cnt}")
print(len(seq), "launches in the last step")
print("\n".join(out))
PY
rm -rf gpurun_out/kt
