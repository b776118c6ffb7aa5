#!/usr/bin/env python3
"""gpurun_out/prof_extra/<tag>/ (tools/prof_extra.sh) -> profiles/<round>_extra_<name>_kernel_stats.csv, ..._under_rocprof.json and
profiles/<round>_pmc_<name>.json: per-kernel HBM bytes (read = 2 * FETCH_SIZE KiB: gfx950 wide-read correction; write = WRITE_SIZE
KiB) and MFMA-busy share, plus hbm_traffic_bytes_per_launch = bytes of the fused-cell kernels per step / the cell launches bench.py
counts per step (what `roofline.traffic` of the `extras` entry <name> reports).
usage: tools/summarize_extra.py <round> <tag> <name>"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_sha16():
    """sha256(vp-suite_amd/libvpx_hip.so)[:16] — the library these counters were collected on; bench.py reports `roofline.traffic` from a
    committed PMC file only while the library it loaded has this hash (a kernel change without a fresh PMC pass then shows null, not old bytes)"""
    import hashlib
    with open(os.path.join(ROOT, "vp-suite_amd", "libvpx_hip.so"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]

ROUND, TAG, NAME = sys.argv[1], sys.argv[2], sys.argv[3]
SRC = os.path.join(ROOT, "gpurun_out", "prof_extra", TAG)
DST = os.environ.get("VPX_PROFILES_DST") or os.path.join(ROOT, "profiles")   # (the GPU box writes under gpurun_out/: only that comes back)
CELL = ("cell2_kernel", "cell3_kernel", "EpiConvLSTM", "convlstm_pointwise_kernel", "conv_gemm_dual_kernel", "EpiSTOut", "EpiSTGate",
        "st_ln_", "st_gates", "st_out", "c5_kernel", "c1_kernel<2, 8>")   # round 4: the ST-LSTM step = c5 launches + conv_last (c1<2,8>) + K-split pointwise stages


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern))
    return max(f, key=os.path.getmtime) if f else None


def is_cell(k):
    return any(c in k for c in CELL) and "Conv2Epi" not in k


f = one("trace/*/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(DST, f"{ROUND}_extra_{NAME}_kernel_stats.csv"))
line = None
log = os.path.join(SRC, "bench.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        line = json.loads(lines[-1])
        open(os.path.join(DST, f"{ROUND}_extra_{NAME}_under_rocprof.json"), "w").write(lines[-1])
tab = collections.defaultdict(dict)
steps_in_pmc = 3 + 1   # tools/prof_extra.sh: --steps 3 --warmup 1
for d, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", "SQ_VALU_MFMA_BUSY_CYCLES"), ("pmc_sq", "GRBM_GUI_ACTIVE")):
    f = one(f"{d}/*/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"].split("(")[0][:100]
        e = tab[k].setdefault(cname, [0.0, 0])
        e[0] += float(r["Counter_Value"]); e[1] += 1
out = {"configuration": NAME, "lib_sha16": lib_sha16(), "command": f"tools/prof_extra.sh {TAG} ... (separate --pmc passes over bench.py --steps 3 --warmup 1)", "per_kernel": {}}
cell_bytes = cell_fetch_raw = cell2_fetch_raw = cell_write = 0.0
for k, cs in tab.items():
    e = {}
    if "FETCH_SIZE" in cs:
        e["launches"] = cs["FETCH_SIZE"][1]
        e["read_MB_per_launch"] = round(2.0 * cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] * 1024 / 1e6, 3)
    if "WRITE_SIZE" in cs:
        e["write_MB_per_launch"] = round(cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1] * 1024 / 1e6, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"][0] > 0:
        e["mfma_pipe_busy_frac"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (cs["GRBM_GUI_ACTIVE"][0] / 8 * 1024), 4)
    if is_cell(k):
        cell_bytes += 2.0 * cs.get("FETCH_SIZE", [0, 1])[0] * 1024 + cs.get("WRITE_SIZE", [0, 1])[0] * 1024
        cell_fetch_raw += cs.get("FETCH_SIZE", [0, 1])[0] * 1024
        if "cell2_kernel_q<vpx::Cell2Epi" in k or "cell2_kernel_x" in k:
            cell2_fetch_raw += cs.get("FETCH_SIZE", [0, 1])[0] * 1024
        cell_write += cs.get("WRITE_SIZE", [0, 1])[0] * 1024
    out["per_kernel"][k] = e
out["per_kernel"] = dict(sorted(out["per_kernel"].items(), key=lambda kv: -(kv[1].get("read_MB_per_launch", 0) + kv[1].get("write_MB_per_launch", 0)) * kv[1].get("launches", 0))[:20])
if line is not None and cell_bytes > 0:
    per_step_launches = line["roofline"]["launches"] / line["steps"]
    # Two readings of FETCH_SIZE (profiles/r06_fetch_calibration.md): the guide's x2 holds for WIDE reads (16 B per lane, whole 128-byte lines:
    # tallied at half their size); the fused cell's operand stages arrive by LDS-DMA as 64-byte segments at the pixel pitch and are tallied in
    # full (calibrated on three shapes of this very kernel). calibrated = raw FETCH + half of the wide reads (the epilogue's c_{t-1}) + WRITE;
    # x2 = the upper bound the earlier rounds reported.
    n_l = steps_in_pmc * per_step_launches
    wide = line["roofline"].get("wide_read_bytes_per_launch")
    out["hbm_traffic_bytes_per_launch"] = {"x2_upper_bound": cell_bytes / n_l, "raw_lower_bound": (cell_fetch_raw + cell_write) / n_l,
                                           "note": "HBM bytes of the fused-cell kernels per step / cell launches per step as bench.py counts them; "
                                                   "total = calibrated (FETCH x1 + wide reads / 2 + WRITE) for the ConvLSTM entries (bench.py states their wide reads), else the x2 bound"}
    # (calibrated for the fused ConvLSTM cell only — the entries whose bench line states wide reads > 0; the ST-LSTM kernels' 32-byte stage
    #  pieces are NOT calibrated: those entries keep the x2 bound as their total)
    share = cell2_fetch_raw / cell_fetch_raw if cell_fetch_raw > 0 else 0.0
    cal = bool(wide) and share >= 0.9   # ... and only where cell2_kernel_q / _x — the kernel the calibration was done on — makes >= 90 % of the cell reads
    out["hbm_traffic_bytes_per_launch"]["total"] = ((cell_fetch_raw + cell_write) / n_l + 0.5 * wide) if cal else cell_bytes / n_l
    out["hbm_traffic_bytes_per_launch"]["rule"] = "calibrated" if cal else "x2"
    out["hbm_traffic_bytes_per_launch"]["cell2_share_of_cell_reads"] = round(share, 3)
    out["algorithmic_bytes_per_launch"] = line["roofline"].get("algorithmic_bytes_per_launch")
json.dump(out, open(os.path.join(DST, f"{ROUND}_pmc_{NAME}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
