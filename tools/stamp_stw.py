#!/usr/bin/env python3
"""Phase timing of ONE stw_kernel workgroup (developer build: make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): shader cycles
spent per item requesting the next item's copies / multiplying / at the item's sync point, per wave. BB, BLOCKS (hardware block indices)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VPX_LIB", os.path.join(ROOT, "build", "libvpx_ablate.so"))
sys.path.insert(0, ROOT)
import torch
import vp_suite_amd as v
L = v._lib.lib()
B = int(os.environ.get("BB", 128)); BLOCKS = [int(x) for x in os.environ.get("BLOCKS", "0,100,300,500").split(",")]
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
L.vpx_dbg_stw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
torch.manual_seed(0)
Cin = Ch = 128; H = W = 16; k = 5
a = [(torch.randn(B, Ch, H, W, device="cuda") * 0.5).requires_grad_() for _ in range(4)]
shapes = [(7 * Ch, Cin, k, k), (4 * Ch, Ch, k, k), (3 * Ch, Ch, k, k), (Ch, 2 * Ch, k, k), (Ch, 2 * Ch, 1, 1)]
w = [(torch.randn(s, device="cuda") / (s[1] * s[2] * s[3]) ** 0.5).requires_grad_() for s in shapes]
def run():
    out = v.ops.stlstm_step(*a, *w, precision="bf16x3")
    loss = sum((o * o).sum() for o in out if torch.is_tensor(o) and o.requires_grad)
    loss.backward()
for blk in BLOCKS:
    buf.zero_()
    L.vpx_dbg_stw_stamps(buf.data_ptr(), blk)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(8, 8)
    n = max(int(s[0][3]), 1)
    print(f"block {blk}: pass {s[0][4]} pair {s[0][5]} items {n}")
    for wv in (0, 3, 4, 7):
        t = s[wv]
        print(f"   wave {wv}: per item  copies {t[0] / n:8.0f}  multiply {t[1] / n:8.0f}  sync {t[2] / n:8.0f}   (shader cycles)")
