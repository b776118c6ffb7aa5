"""Developer script (GPU box): where does the convq data gradient of a glue layer differ from torch? Error by border / position,
with the first-generation launch (VPX_OPT_EXPERIMENT bit 14) next to it."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import vp_suite_amd as vpx

torch.manual_seed(0)
L = vpx._lib.lib()
CASES = [(False, 64, 64, 3, 2, 1, 64, 64, 24), (False, 64, 64, 3, 2, 1, 32, 32, 24), (True, 96, 96, 4, 2, 1, 16, 16, 24), (False, 64, 96, 3, 2, 1, 33, 47, 24)]
for bit in (0, 16384, 0):
    L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, bit)
    print("==== experiment bits", bit)
    for (tr, Ci, Co, k, s, p, H, W, n) in CASES:
        x = torch.randn(n, Ci, H, W, device="cuda", requires_grad=True)
        w = (torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), device="cuda") / np.sqrt(Ci * k * k)).requires_grad_(True)
        b = (0.1 * torch.randn(Co, device="cuda")).requires_grad_(True)
        y = vpx.ops.conv2d_ex(x, w, b, s, p, tr, 0.2, "bf16x3")
        gy = torch.randn_like(y)
        (y * gy).sum().backward()
        x2 = x.detach().clone().requires_grad_(True)
        w2 = w.detach().clone().requires_grad_(True)
        r = F.conv_transpose2d(x2, w2, b.detach(), stride=s, padding=p) if tr else F.conv2d(x2, w2, b.detach(), stride=s, padding=p)
        (r * torch.where(y.detach() > 0, 1.0, 0.2) * gy).sum().backward()   # LeakyReLU' at the sign of OUR output (kink: see tests/test_gpu_more.py)
        err = (x.grad - x2.grad).abs()
        scale = float(x2.grad.abs().max())
        e = err.amax(dim=(0, 1)).cpu().numpy() / scale            # [H, W]
        en = err.amax(dim=(1, 2, 3)).cpu().numpy() / scale        # per image
        print("case", (tr, Ci, Co, k, s, p, H, W), "relmax dx", float(err.max()) / scale, "dw", float((w.grad - w2.grad).abs().max() / w2.grad.abs().max()),
              "| rows", np.nonzero(e.max(axis=1) > 1e-4)[0][:12], "cols", np.nonzero(e.max(axis=0) > 1e-4)[0][:12], "images", np.nonzero(en > 1e-4)[0][:24])
