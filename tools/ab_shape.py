"""A/B timing of the two MFMA shapes of the second-generation main loop inside ONE process (vpx_set_option(VPX_OPT_MFMA_SHAPE)),
interleaved rounds, random data, per block shape: forward steps, and MODE=train a forward + backward of the block.
BB = per-GPU batch (default 128)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
L = v._lib.lib()
dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 64), (16, 64, 64, 64), (64, 96, 32, 32), (96, 96, 32, 32), (96, 64, 64, 64)]
B, T = int(os.environ.get("BB", 128)), 6
train = os.environ.get("MODE", "infer") == "train"
res, data = {}, {}
for s in shapes:
    Cin, Ch, H, W = s
    data[s] = (v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev)).requires_grad_(train),
               (torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03).requires_grad_(train), torch.zeros(4 * Ch, device=dev).requires_grad_(train),
               [(torch.randn(1, Ch, H, W, device=dev) * 0.1).requires_grad_(train) for _ in range(3)])


def once(s):
    x, Wt, b, pw = data[s]
    if train:
        out, hT, cT = v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision="bf16x3")
        out.backward(out.detach())
    else:
        with torch.no_grad():
            v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision="bf16x3")


for rnd in range(5):
    for s in shapes:
        for mode in (0, 1):
            L.vpx_set_option(v._lib.OPT_MFMA_SHAPE, mode)
            once(s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                once(s)
            torch.cuda.synchronize()
            res.setdefault((s, mode), []).append((time.perf_counter() - t0) / 3)
for s in shapes:
    Cin, Ch, H, W = s
    fl = 2.0 * 4 * Ch * (Cin + Ch * (T - 1) / T) * 9 * H * W * B * T * (3 if train else 1)
    line = f"B={B} {'train' if train else 'infer'} {s}:"
    for mode in (0, 1):
        r = sorted(res[(s, mode)])
        line += f"  {'32x32x16' if mode == 0 else '16x16x32'}: best {fl / r[0] / 1e12:6.1f} TF median {fl / r[len(r) // 2] / 1e12:6.1f} TF ({r[len(r)//2] / T * 1e6:7.1f} us/step)"
    r0, r1 = sorted(res[(s, 0)]), sorted(res[(s, 1)])
    line += f"  ratio(median) {r0[len(r0)//2] / r1[len(r1)//2]:.3f}"
    print(line, flush=True)
