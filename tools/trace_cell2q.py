"""Developer tool (ablate build: make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): start / loop-end / end times of EVERY
workgroup of one half-tile cell2_kernel_q launch together with the CU it ran on -> are the two workgroups of a CU in phase (both in the
main loop, then both in the epilogue) or staggered? EXP = experiment word, PREC = bf16x3 | bf16."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
L = v._lib.lib()
dev = torch.device("cuda:0")
B, T = int(os.environ.get("BB", 128)), 1
Cin, Ch, H, W = [int(t) for t in os.environ.get("SHAPE", "64,64,64,64").split(",")]
x = v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev))
Wt = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03
b = torch.zeros(4 * Ch, device=dev)
pw = [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)]
h0 = torch.randn(B, Ch, H, W, device=dev) * 0.5
L.vpx_set_option(v._lib.OPT_EXPERIMENT, int(os.environ.get("EXP", "0"), 0))
with torch.no_grad():
    for _ in range(5):
        v.ops.convlstm_seq(x, h0, h0, Wt, b, *pw, seq_len=T, in_channels=Cin, precision=os.environ.get("PREC", "bf16x3"))
torch.cuda.synchronize()
n = 8192
buf = (ctypes.c_ulonglong * (n * 4))()
L.vpx_dbg_cell2_trace.argtypes = [ctypes.c_void_p]
assert L.vpx_dbg_cell2_trace(buf) == 0
rows = [(i, buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n) if buf[4 * i + 2]]
t0 = min(r[1] for r in rows)
by_cu = collections.defaultdict(list)
for i, a, m, e, hw in rows:
    key = (hw >> 32) & 0xf, (hw >> 13) & 7, (hw >> 8) & 0xf      # xcc, se, cu
    by_cu[key].append((a - t0, m - t0, e - t0, hw & 0xf, (hw >> 4) & 3, i))
print(f"{len(rows)} workgroups on {len(by_cu)} CUs; kernel span {max(r[3] for r in rows) - t0} cycles")
both_loop = one_loop = none_loop = 0
for key in sorted(by_cu)[:int(os.environ.get('SHOW', 0))]:
    print("CU", key)
    for a, m, e, slot, simd, i in sorted(by_cu[key]):
        print(f"   block {i:5d} slot {slot} simd {simd}: start {a:8d} loop end {m:8d} end {e:8d}   (loop {m - a}, epilogue {e - m})")
# per CU: time with two / one / no workgroup in its main loop (between the first start and the last end)
for key, ws in by_cu.items():
    ev = []
    for a, m, e, *_ in ws:
        ev += [(a, 1), (m, -1)]
    ev.sort()
    cur, last = 0, ev[0][0]
    for t, d in ev:
        if cur >= 2: both_loop += t - last
        elif cur == 1: one_loop += t - last
        else: none_loop += t - last
        cur += d; last = t
tot = both_loop + one_loop + none_loop
import statistics
late = [r for r in rows if r[0] >= 1024]
print(f"blocks >= 1024: median loop {statistics.median(m - a for _, a, m, e, _h in late)}, median epilogue {statistics.median(e - m for _, a, m, e, _h in late)}")
print(f"share of CU time with two workgroups in the main loop {both_loop / tot:.3f}, one {one_loop / tot:.3f}, none {none_loop / tot:.3f}")
