"""Quick kernel-level timing of the fused ConvLSTM cell through the product API (development helper)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v

dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 64), (16, 64, 64, 64), (64, 96, 32, 32), (96, 96, 16, 16), (96, 96, 32, 32), (96, 64, 64, 64)]
for B in (4, 32):
    for (Cin, Ch, H, W) in shapes:
        T = 10
        x = torch.rand(B, T, Cin, H, W, device=dev)
        x = v.ops.to_channels_last(x)
        Wt = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03
        b = torch.zeros(4 * Ch, device=dev)
        pw = [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)]
        with torch.no_grad():
            for _ in range(2):
                v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=Cin)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=Cin)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        flops = 2.0 * 4 * Ch * (Cin + Ch) * 9 * H * W * B * T
        print(f"B={B:3d} ({Cin:3d},{Ch:3d},{H}x{W}) seq of {T}: {dt*1e3:8.3f} ms  {flops/dt/1e12:7.2f} TFLOP/s "
              f"per-step {dt/T*1e6:8.1f} us", flush=True)
