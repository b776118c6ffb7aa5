"""Timing of the fused ConvLSTM cell for kernel variants selected by env (VPX_VARIANT / VPX_CS). Boxes differ by ~10 %,
so compare variants only within ONE gpurun command (same box), several rounds each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 64), (96, 96, 32, 32), (96, 96, 16, 16)]
B, T = int(os.environ.get("BB", 32)), 10
PREC = os.environ.get("PREC", "f32")
res = {}
data = {}
for s in shapes:
    Cin, Ch, H, W = s
    data[s] = (v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev)),
               torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03, torch.zeros(4 * Ch, device=dev),
               [torch.randn(1, Ch, H, W, device=dev) * 0.1 for _ in range(3)])
with torch.no_grad():
    for rnd in range(4):
        for s in shapes:
            x, Wt, b, pw = data[s]
            v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision=PREC)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision=PREC)
            torch.cuda.synchronize()
            res.setdefault(s, []).append((time.perf_counter() - t0) / 5)
for s in shapes:
    Cin, Ch, H, W = s
    fl = 2.0 * 4 * Ch * (Cin + Ch) * 9 * H * W * B * T
    best = min(res[s]); med = sorted(res[s])[len(res[s]) // 2]
    print(f"PREC={PREC} CS={os.environ.get('VPX_CS','auto')} B={B} {s}: best {fl/best/1e12:6.1f} TF  median {fl/med/1e12:6.1f} TF")
