"""A/B timing of the two fused-cell kernels inside ONE process (vpx_set_option), interleaved rounds, per block shape.
BB = per-GPU batch (default 128)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
L = v._lib.lib()
dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 64), (16, 64, 64, 64), (64, 96, 32, 32), (96, 96, 32, 32), (96, 64, 64, 64)]
B, T = int(os.environ.get("BB", 128)), 6
res, data = {}, {}
for s in shapes:
    Cin, Ch, H, W = s
    data[s] = (v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev)),
               torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03, torch.zeros(4 * Ch, device=dev),
               [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)])
with torch.no_grad():
    for rnd in range(4):
        for s in shapes:
            for mode in (0, 2):
                L.vpx_set_option(v._lib.OPT_CELL2, mode)
                x, Wt, b, pw = data[s]
                v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision="bf16x3")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision="bf16x3")
                torch.cuda.synchronize()
                res.setdefault((s, mode), []).append((time.perf_counter() - t0) / 3)
for s in shapes:
    Cin, Ch, H, W = s
    fl = 2.0 * 4 * Ch * (Cin + Ch * (T - 1) / T) * 9 * H * W * B * T
    line = f"B={B} {s}:"
    for mode in (0, 2):
        r = sorted(res[(s, mode)])
        line += f"  gen{1 if mode == 0 else 2}: best {fl / r[0] / 1e12:6.1f} TF median {fl / r[len(r) // 2] / 1e12:6.1f} TF ({r[len(r)//2] / T * 1e6:7.1f} us/step)"
    print(line, flush=True)
