"""Developer tool (ablate build: make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): per-wave s_memtime stamps of one
half-tile cell2_kernel_q workgroup on the headline cell -> prologue / main loop / epilogue / store-drain cycles of a tile.
VPX_C2_STAMP_BLOCK = block id to stamp (default: a late one, so that its CU is in steady state)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VPX_C2_STAMP_BLOCK", "3000")
import torch
import vp_suite_amd as v
L = v._lib.lib()
dev = torch.device("cuda:0")
B, T = int(os.environ.get("BB", 128)), 4
Cin, Ch, H, W = [int(t) for t in os.environ.get("SHAPE", "64,64,64,64").split(",")]
x = v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev))
Wt = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03
b = torch.zeros(4 * Ch, device=dev)
pw = [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)]
h0 = torch.randn(B, Ch, H, W, device=dev) * 0.5
L.vpx_set_option(v._lib.OPT_EXPERIMENT, int(os.environ.get("EXP", "0"), 0))
with torch.no_grad():
    for _ in range(20):
        v.ops.convlstm_seq(x, h0, h0, Wt, b, *pw, seq_len=T, in_channels=Cin, precision=os.environ.get("PREC", "bf16x3"))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
L.vpx_dbg_cell2_stamps.argtypes = [ctypes.c_void_p]
assert L.vpx_dbg_cell2_stamps(buf) == 0
for w in range(4):
    st = [buf[w * 64 + i] for i in range(64)]
    t0 = st[0]
    print(f"wave {w}: prologue issued+landed {st[1] - t0}, barrier {st[2] - st[1]}, main loop {st[40] - st[2]}, "
          f"epilogue (issue) {st[41] - st[40]}, store drain {st[42] - st[41]}, total {st[42] - t0}  [s_memtime ticks, 100 MHz]")
    e = [st[43 + i] for i in range(8)]
    names = ["issue loads0", "barrier", "put0", "issue loads1", "math0+stores0", "put1", "math1+stores1"]
    print("        epilogue: " + ", ".join(f"{n} {e[i + 1] - e[i]}" for i, n in enumerate(names)))
