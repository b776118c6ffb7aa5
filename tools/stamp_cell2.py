"""Developer tool (ablate build only: make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): per-wave s_memtime stamps
of one cell2 workgroup on the headline cell -> where a tile's cycles go (prologue / sync waits / barriers / epilogue)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
L = v._lib.lib()
dev = torch.device("cuda:0")
B, T = int(os.environ.get("BB", 128)), 3
Cin, Ch, H, W = 64, 64, 64, 64
x = v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev))
Wt = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03
b = torch.zeros(4 * Ch, device=dev)
pw = [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)]
h0 = torch.randn(B, Ch, H, W, device=dev) * 0.5
L.vpx_set_option(v._lib.OPT_CELL2, 2)
with torch.no_grad():
    for _ in range(3):
        v.ops.convlstm_seq(x, h0, h0, Wt, b, *pw, seq_len=T, in_channels=Cin, precision="bf16x3")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
L.vpx_dbg_cell2_stamps.argtypes = [ctypes.c_void_p]
assert L.vpx_dbg_cell2_stamps(buf) == 0
names = {0: "start", 1: "prologue issued", 2: "prologue landed+barrier", 40: "loop end", 42: "epilogue done"}
for w in range(8):
    st = [buf[w * 64 + i] for i in range(64)]
    t0 = st[0]
    line = [f"wave {w}:"]
    for i in (1, 2):
        line.append(f"{names[i]} +{st[i] - t0}")
    prev = st[2]
    for c in range(9):
        a, bb, cc = st[3 + 3 * c], st[4 + 3 * c], st[5 + 3 * c]
        line.append(f"c{c}: run {a - prev} vm {bb - a} bar {cc - bb}")
        prev = cc
    line.append(f"loop end +{st[40] - t0}; epilogue {st[42] - st[40]}; total {st[42] - t0}")
    e = [st[43 + i] for i in range(5)]
    line.append(f"epi: loads0+barrier {e[1] - e[0]} put0+loads1 {e[2] - e[1]} math0 {e[3] - e[2]} put1+math1 {e[4] - e[3]}")
    print(" | ".join(line))
