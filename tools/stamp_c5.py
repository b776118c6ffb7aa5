#!/usr/bin/env python3
"""Phase timing of ONE c5_kernel workgroup (developer build: make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): shader-clock
stamps at kernel start / after the prologue / after the K loop / after the epilogue, per wave. MODE=cell (ConvLSTM block, c3) | st (ST-LSTM step).
BB, BLOCK (hardware block index to stamp)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VPX_LIB", os.path.join(ROOT, "build", "libvpx_ablate.so"))
sys.path.insert(0, ROOT)
import torch
import vp_suite_amd as v
L = v._lib.lib()
B = int(os.environ.get("BB", 4)); MODE = os.environ.get("MODE", "cell"); BLOCK = int(os.environ.get("BLOCK", 100))
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
L.vpx_dbg_c5_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.vpx_dbg_c5_stamps(buf.data_ptr(), BLOCK)
torch.manual_seed(0)
if MODE == "cell":
    Cin, Ch, H, W, T = 64, 64, 64, 64, 6
    Wt = torch.randn(4 * Ch, Cin + Ch, 3, 3, device="cuda") * 0.04; b = torch.zeros(4 * Ch, device="cuda")
    pw = [torch.randn(1, Ch, H, W, device="cuda") * 0.1 for _ in range(3)]
    x = torch.rand(B, T, Cin, H, W, device="cuda")
    run = lambda: v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=Cin, precision="bf16x3")
else:
    Cin = Ch = 128; H = W = 16; k = 5
    a = [torch.randn(B, Ch, H, W, device="cuda") * 0.5 for _ in range(4)]
    shapes = [(7 * Ch, Cin, k, k), (4 * Ch, Ch, k, k), (3 * Ch, Ch, k, k), (Ch, 2 * Ch, k, k), (Ch, 2 * Ch, 1, 1)]
    w = [torch.randn(s, device="cuda") / (s[1] * s[2] * s[3]) ** 0.5 for s in shapes]
    run = lambda: v.ops.stlstm_step(*a, *w, precision="bf16x3")
with torch.no_grad():
    for _ in range(5):
        run()
    torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(8, 8)
for wv in range(4):
    t = s[wv]
    print(f"wave {wv}: prologue {t[1] - t[0]:7d}  loop {t[2] - t[1]:7d}  epilogue {t[3] - t[2] if t[3] else 0:7d}  (shader cycles, last launch of the kernel that stamped)")
