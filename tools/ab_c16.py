"""Timing ablations of c16_kernel (csrc/conv16.hip) in the developer build (make ablate; VPX_LIB=build/libvpx_ablate.so): the forecaster's
last glue layer (64 -> 16, 3x3, 1280 frames of 64x64) with parts of the kernel switched off by VPX_OPT_EXPERIMENT bits 20-22
(1: no fragment reads / MFMAs, 2: no tile copies after the first, 4: no stores). Results are wrong by construction; only the times count."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
dev = torch.device("cuda:0")
N, Ci, Co, H, W = int(os.environ.get("NN", 1280)), int(os.environ.get("CI", 64)), 16, 64, 64
x = v.ops.to_channels_last(torch.rand(N, Ci, H, W, device=dev) - 0.3)
if os.environ.get("ZERO"):
    x.zero_()
w = torch.randn(Ci, Co, 3, 3, device=dev) * 0.05
b = torch.randn(Co, device=dev) * 0.1
xbuf, _ = v.ops.split_convert(x)
L = v._lib.lib()
res = {}
with torch.no_grad():
    for rnd in range(5):
        for bits in (0, 1, 2, 4, 3, 5, 6, 7):
            L.vpx_set_option(v._lib.OPT_EXPERIMENT, bits << 20)
            run = lambda: v.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, 1, 1, True, 0.2, "bf16x3")
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(bits, []).append(e0.elapsed_time(e1) / 10)
L.vpx_set_option(v._lib.OPT_EXPERIMENT, 0)
names = {0: "whole kernel", 1: "no MFMAs / fragment reads", 2: "no tile copies", 4: "no stores", 3: "neither copies nor MFMAs", 5: "copies only",
         6: "MFMAs only", 7: "tile loop only"}
for bits, r in res.items():
    r.sort()
    print(f"{names[bits]:32s} {r[len(r) // 2] * 1e3:8.1f} us", flush=True)
