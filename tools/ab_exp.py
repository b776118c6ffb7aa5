"""A/B timing of kernel experiment bits (vpx_set_option(VPX_OPT_EXPERIMENT), MODES env, default "0,1,2,3") — derived from ab_shape.py: the two MFMA shapes of the second-generation main loop inside ONE process (vpx_set_option(VPX_OPT_MFMA_SHAPE)),
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
MODES = [int(x) for x in os.environ.get("MODES", "0,1,2,3").split(",")]
PREC = os.environ.get("PREC", "bf16x3")   # bf16: inference only (the plain form of the fused cell)
ZERO = os.environ.get("ZERO", "0") == "1"  # all-zero operands: the clock the chip holds without data toggling (MI355X_MICROARCH.md, DVFS give-back 1)
res, data = {}, {}
for s in shapes:
    Cin, Ch, H, W = s
    z = 0.0 if ZERO else 1.0
    data[s] = (v.ops.to_channels_last(torch.rand(B, T, Cin, H, W, device=dev) * z).requires_grad_(train),
               (torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.03 * z).requires_grad_(train), torch.zeros(4 * Ch, device=dev).requires_grad_(train),
               [(torch.randn(1, Ch, H, W, device=dev) * 0.1 * z).requires_grad_(train) for _ in range(3)])


def once(s):
    x, Wt, b, pw = data[s]
    if train:
        out, hT, cT = v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision=PREC)
        out.backward(out.detach())
    else:
        with torch.no_grad():
            v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=T, in_channels=s[0], precision=PREC)


for rnd in range(5):
    for s in shapes:
        for mode in MODES:
            L.vpx_set_option(v._lib.OPT_EXPERIMENT, mode)
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
    line = f"B={B} {PREC}{' ZERO-DATA' if ZERO else ''} {'train' if train else 'infer'} {s}:"
    base = sorted(res[(s, MODES[0])])
    for mode in MODES:
        r = sorted(res[(s, mode)])
        line += f"  exp={mode}: {fl / r[len(r) // 2] / 1e12:6.1f} TF ({r[len(r)//2] / T * 1e6:7.1f} us/step, x{base[len(base)//2] / r[len(r)//2]:.3f})"
    print(line, flush=True)
