#!/usr/bin/env python3
"""Developer tool (make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): every workgroup of the wgrad2_kernel launch of one ConvLSTM
block's backward (SHAPE = Cin,Ch,H,W; BB; T) — CU, item-loop cycles, full tile or half-empty tail tile -> per XCD: how long each CU was busy."""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VPX_LIB", os.path.join(ROOT, "build", "libvpx_ablate.so"))
sys.path.insert(0, ROOT)
import torch
import vp_suite_amd as v
L = v._lib.lib()
B, T = int(os.environ.get("BB", 128)), int(os.environ.get("T", 10))
Cin, Ch, H, W = [int(t) for t in os.environ.get("SHAPE", "64,64,64,64").split(",")]
dev = torch.device("cuda:0")
x = v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev))
Wt = (torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03).requires_grad_()
b = torch.zeros(4 * Ch, device=dev, requires_grad=True)
pw = [(torch.randn(1, Ch, H, W, device=dev) * 0.1).requires_grad_() for _ in range(3)]
for _ in range(2):
    out, hT, cT = v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=Cin, precision="bf16x3")
    (out * out).sum().backward()
torch.cuda.synchronize()
n = 8192
buf = (ctypes.c_ulonglong * (n * 4))()
L.vpx_dbg_stw_trace.argtypes = [ctypes.c_void_p]
assert L.vpx_dbg_stw_trace(buf) == 0
rows = [(i, buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n) if buf[4 * i + 1] > buf[4 * i] > 0 and (buf[4 * i + 3] & 4)]
print(f"{len(rows)} workgroups")
dur = collections.defaultdict(list); by_xcd = collections.defaultdict(list)
for i, a, e, hw, info in rows:
    kind = "half-empty tail tile" if info & 8 else "full tile"
    dur[kind].append(e - a)
    by_xcd[(hw >> 32) & 0xf].append((a, e, kind, (hw >> 8) & 0xf, (hw >> 13) & 7))
for k in sorted(dur):
    dd = sorted(dur[k]); print(f"{k}: {len(dd)} workgroups, item loop median {dd[len(dd) // 2]} cycles (min {dd[0]}, max {dd[-1]})")
for x_ in sorted(by_xcd):
    cus = collections.defaultdict(int); cnt = collections.Counter()
    for a, e, kind, cu, se in by_xcd[x_]:
        cus[(se, cu)] += e - a; cnt[kind] += 1
    vals = sorted(cus.values())
    print(f"XCD {x_}: {len(by_xcd[x_])} workgroups {dict(cnt)} on {len(cus)} CUs; per CU, sum of its item loops: min {vals[0]} median {vals[len(vals) // 2]} max {vals[-1]}")
