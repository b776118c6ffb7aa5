"""Developer repro: the x form against the four-wave half tile on one case of tests/test_gpu_cell2.py, several runs; where do outputs differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import vp_suite_amd as vpx
import test_gpu_cell2 as t
tag = os.environ.get("TAG", "plain_s7_many_tiles")
prec = os.environ.get("PREC", "bf16x3")
bits = int(os.environ.get("EXP", "32768"))
L = vpx._lib.lib()
L.vpx_set_option(vpx._lib.OPT_CELL2, 2)
L.vpx_set_option(vpx._lib.OPT_MFMA_SHAPE, 1)
with torch.no_grad():
    L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 0)
    o1, h1, c1, _ = t._run(vpx, tag, grads=False, precision=prec)
    o1b, _, c1b, _ = t._run(vpx, tag, grads=False, precision=prec)
    print("q vs q equal:", torch.equal(o1, o1b), torch.equal(c1, c1b))
    for rep in range(6):
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, bits)
        o2, h2, c2, _ = t._run(vpx, tag, grads=False, precision=prec)
        d = (o2 != o1)
        print(f"run {rep}: out mismatches {int(d.sum())} of {d.numel()}; c mismatches {int((c2 != c1).sum())}; max abs {float((o2 - o1).abs().max()):.3e}")
        if d.any():
            idx = d.nonzero()
            print("   first mismatches [b, t, ch, y, x]:", idx[:8].tolist())
            print("   by t:", [int(d[:, tt].sum()) for tt in range(d.shape[1])], " by y%16:", sorted(set((idx[:, 3] % 16).tolist()))[:16], " by x%16:", sorted(set((idx[:, 4] % 16).tolist()))[:16],
                  " ch%32:", sorted(set((idx[:, 2] % 32).tolist()))[:32])
