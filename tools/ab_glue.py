"""A/B timing of the EF stage-glue layers (convlstm-shi, B*T = 1280 frames): first-generation path (vpx_conv2d_ex_fwd on fp32 input)
against the schedule-driven K = 32 kernel on split input (vpx_conv2d_ex_fwd_from_split), one process, interleaved rounds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
dev = torch.device("cuda:0")
N = int(os.environ.get("NN", 1280))
layers = {  # name: (Ci, Co, H, W, k, s, p, transposed, slope)
    "conv2 64->64 s2": (64, 64, 64, 64, 3, 2, 1, False, 0.2),
    "conv3 96->96 s2": (96, 96, 32, 32, 3, 2, 1, False, 0.2),
    "deconv1 96->96 t2k4 16^2": (96, 96, 16, 16, 4, 2, 1, True, 0.2),
    "deconv2 96->96 t2k4 32^2": (96, 96, 32, 32, 4, 2, 1, True, 0.2),
    "deconv3 64->16 t1k3": (64, 16, 64, 64, 3, 1, 1, True, 0.2),
}
res = {}
data = {}
for name, (Ci, Co, H, W, k, s, p, tr, slope) in layers.items():
    x = v.ops.to_channels_last(torch.rand(N, Ci, H, W, device=dev) - 0.3)
    w = torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), device=dev) * 0.05
    b = torch.randn(Co, device=dev) * 0.1
    xbuf, _ = v.ops.split_convert(x)
    data[name] = (x, w, b, xbuf)
with torch.no_grad():
    for rnd in range(4):
        for name, (Ci, Co, H, W, k, s, p, tr, slope) in layers.items():
            x, w, b, xbuf = data[name]
            for mode in (0, 1, 2):
                def run():
                    if mode == 0:
                        v.ops.conv2d_ex(x, w, b, s, p, tr, slope, "bf16x3")
                    else:
                        v.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3", out_split=(mode == 2 and Co % 8 == 0),
                                                   out_fp32=(mode == 1 or Co % 8 != 0))
                run(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                res.setdefault((name, mode), []).append((time.perf_counter() - t0) / 5)
for name, (Ci, Co, H, W, k, s, p, tr, slope) in layers.items():
    Ho, Wo = (H * s, W * s) if tr else (H // s, W // s)
    fl = 2.0 * Ci * Co * k * k * (H * W if tr else Ho * Wo) * N
    line = f"{name:28s}"
    for mode, lab in ((0, "gen1 fp32-in"), (1, "convq fp32-out"), (2, "convq split-out")):
        r = sorted(res[(name, mode)]); m = r[len(r) // 2]
        line += f"  {lab}: {m * 1e3:6.3f} ms {fl / m / 1e12:6.1f} TF"
    print(line, flush=True)
