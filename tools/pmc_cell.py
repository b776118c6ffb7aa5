"""Runs a few launches of the fused ConvLSTM cell at the headline shape for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
dev = torch.device("cuda:0")
if os.environ.get("EXP"):
    v._lib.lib().vpx_set_option(v._lib.OPT_EXPERIMENT, int(os.environ["EXP"]))
B, T = int(os.environ.get("BB", 32)), 4
Cin, Ch, H, W = [int(t) for t in os.environ.get('SHAPE', '64,64,64,64').split(',')]
x = v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev))
Wt = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03
b = torch.zeros(4 * Ch, device=dev)
pw = [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)]
h0 = torch.randn(B, Ch, H, W, device=dev) * 0.5
with torch.no_grad():
    for _ in range(3):
        v.ops.convlstm_seq(x, h0, h0, Wt, b, *pw, seq_len=T, in_channels=Cin, precision=os.environ.get("PREC", "bf16x3"))
torch.cuda.synchronize()
print("done")
