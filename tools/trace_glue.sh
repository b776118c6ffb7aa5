#!/bin/bash
# Runs ON THE GPU BOX: full kernel trace of two bench steps -> per-launch list of the non-cell kernels (small csv)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tg; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2
out = []
for r in rows[half:]:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append(f'{d:9.1f} us  grid {r["Grid_Size_X"]:>8}x{r["Grid_Size_Y"]:>4}x{r["Grid_Size_Z"]:>3} wg {r["Workgroup_Size_X"]:>4} lds {r.get("LDS_Block_Size","?"):>7}  {n[:90]}')
open("gpurun_out/tg/launches.txt", "w").write("\n".join(out))
PY
rm -rf $OUT/t
