"""One ConvLSTM block forward + backward (headline shape) for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
dev = torch.device("cuda:0")
B, T = int(os.environ.get("BB", 32)), 4
PREC = os.environ.get("PREC", "bf16x3")
Cin, Ch, H, W, K = (int(v_) for v_ in os.environ.get("SHAPE", "64,64,64,64,3").split(","))
x = v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev)).requires_grad_(True)
Wt = (torch.randn(4 * Ch, Cin + Ch, K, K, device=dev) * 0.03).requires_grad_(True)
b = torch.zeros(4 * Ch, device=dev, requires_grad=True)
pw = [(torch.randn(1, Ch, H, W, device=dev) * 0.1).requires_grad_(True) for _ in range(3)]
for _ in range(2):
    out, hT, cT = v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=Cin, precision=PREC)
    out.sum().backward()
torch.cuda.synchronize()
print("done")
