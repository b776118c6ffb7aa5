"""Developer tool (ablate build: make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): start / first MFMA / loop end / end of EVERY
workgroup of one launch of the fused cell step, on the eight-wave half tile (cell2_kernel_x, EXP with bit 15 set) or the four-wave one
(cell2_kernel_q, bit 15 clear), with the CU it ran on: phase lengths and how the two workgroups of a CU overlap.
EXP = experiment word, PREC = bf16x3 | bf16, BB = batch, SHAPE = Cin,Ch,H,W."""
import ctypes, os, sys, collections, statistics
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
exp = int(os.environ.get("EXP", "32768"), 0)
L.vpx_set_option(v._lib.OPT_EXPERIMENT, exp)
with torch.no_grad():
    for _ in range(5):
        v.ops.convlstm_seq(x, h0, h0, Wt, b, *pw, seq_len=T, in_channels=Cin, precision=os.environ.get("PREC", "bf16x3"))
torch.cuda.synchronize()
n = 8192
if exp & 32768:
    buf = (ctypes.c_ulonglong * (n * 8))()
    L.vpx_dbg_cell2x_trace.argtypes = [ctypes.c_void_p]
    assert L.vpx_dbg_cell2x_trace(buf) == 0
    rows = [(i, buf[8 * i], buf[8 * i + 1], buf[8 * i + 2], buf[8 * i + 3], buf[8 * i + 4]) for i in range(n) if buf[8 * i + 3]]
else:
    buf = (ctypes.c_ulonglong * (n * 4))()
    L.vpx_dbg_cell2_trace.argtypes = [ctypes.c_void_p]
    assert L.vpx_dbg_cell2_trace(buf) == 0
    rows = [(i, buf[4 * i], buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n) if buf[4 * i + 2]]
t0 = min(r[1] for r in rows)
by_cu = collections.defaultdict(list)
for i, a, f, m, e, hw in rows:
    key = (hw >> 32) & 0xf, (hw >> 13) & 7, (hw >> 8) & 0xf      # xcc, se, cu
    by_cu[key].append((a - t0, f - t0, m - t0, e - t0, i))
print(f"EXP {exp} PREC {os.environ.get('PREC', 'bf16x3')}: {len(rows)} workgroups on {len(by_cu)} CUs; kernel span {max(r[4] for r in rows) - t0} cycles")
for key in sorted(by_cu)[:int(os.environ.get('SHOW', 1))]:
    print("CU", key)
    for a, f, m, e, i in sorted(by_cu[key]):
        print(f"   block {i:5d}: start {a:8d} first MFMA {f:8d} loop end {m:8d} end {e:8d}   (prologue {f - a}, loop {m - f}, epilogue {e - m})")
both = one = none = 0
for key, ws in by_cu.items():
    ev = []
    for a, f, m, e, i in ws:
        ev += [(f, 1), (m, -1)]
    ev.sort()
    cur, last = 0, ev[0][0]
    for t, d in ev:
        if cur >= 2: both += t - last
        elif cur == 1: one += t - last
        else: none += t - last
        cur += d; last = t
tot = both + one + none
late = [r for r in rows if r[0] >= 1024]
med = lambda f: statistics.median(f(r) for r in late)
print(f"blocks >= 1024: median prologue {med(lambda r: r[2] - r[1])}, loop {med(lambda r: r[3] - r[2])}, epilogue {med(lambda r: r[4] - r[3])}, tile {med(lambda r: r[4] - r[1])}")
print(f"share of CU time with two workgroups in the main loop {both / tot:.3f}, one {one / tot:.3f}, none {none / tot:.3f}")
