import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import vp_suite_amd as v
dev = torch.device("cuda:0")
N = 1280
for (Ci, Co, H, W, k, s, p, tr) in [(96, 96, 32, 32, 3, 1, 1, False), (96, 128, 32, 32, 3, 1, 1, False), (64, 64, 64, 64, 3, 1, 1, False), (64, 32, 64, 64, 3, 1, 1, False)]:
    x = v.ops.to_channels_last(torch.rand(N if H == 32 else N // 4, Ci, H, W, device=dev) - 0.3)
    n = x.shape[0]
    w = torch.randn((Co, Ci, k, k), device=dev) * 0.05
    b = torch.randn(Co, device=dev) * 0.1
    xbuf, _ = v.ops.split_convert(x)
    with torch.no_grad():
        for mode in (0, 1):
            def run():
                if mode == 0: v.ops.conv2d_ex(x, w, b, s, p, tr, 0.0, "bf16x3")
                else: v.ops.conv2d_ex_from_split(xbuf, (n, Ci, H, W), w, b, s, p, tr, 0.0, "bf16x3")
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): run()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 10
            fl = 2.0 * Ci * Co * k * k * H * W * n
            print(f"{Ci}->{Co} {H}x{W} n={n} mode {mode}: {t*1e3:.3f} ms {fl/t/1e12:.1f} TF", flush=True)
