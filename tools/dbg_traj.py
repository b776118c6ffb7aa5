"""Developer script (GPU box): per-tensor gradient errors of random TrajGRU blocks (tests/test_gpu_fuzz.py's cases) against the oracle."""
import random, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import vp_suite_amd as vpx
from vp_suite_amd import traj_ops
from oracle import torch_ref as tr
from golden_util import name_seed, seeded_rand, seeded_randn
names = ("i2h", "i2f_conv1", "h2f_conv1", "flows_conv", "ret")
for chunk in (0, 3):
    rng = random.Random(4000 + chunk)
    for i in range(8):
        B, T, Cin, C = rng.choice([1, 2, 3]), rng.choice([1, 2, 3]), rng.choice([1, 3, 4, 8, 12]), rng.choice([4, 8, 12, 16, 32])
        H, W, nl = rng.choice([5, 8, 12, 16, 17, 24]), rng.choice([7, 8, 12, 16, 20]), rng.choice([1, 3, 5])
        has_x, has_h0 = rng.random() < 0.85, rng.random() < 0.6
        if not has_x and not has_h0: has_x = True
        prec = rng.choice(["f32", "bf16x3"])
        tag = f"fuzz.traj.{chunk}.{i}"
        shapes = {"i2h": (3 * C, Cin, 3, 3), "i2f_conv1": (32, Cin, 5, 5), "h2f_conv1": (32, C, 5, 5), "flows_conv": (2 * nl, 32, 5, 5), "ret": (3 * C, nl * C, 1, 1)}
        P = {}
        for n in names:
            s_ = shapes[n]
            P[n + ".weight"] = seeded_randn(s_, name_seed(f"{tag}.{n}.w"), 1.0 / np.sqrt(s_[1] * s_[2] * s_[3]))
            P[n + ".bias"] = seeded_randn((s_[0],), name_seed(f"{tag}.{n}.b"), 0.1)
        x = seeded_rand((B, T, Cin, H, W), name_seed(tag + ".x")) if has_x else None
        h0 = seeded_randn((B, C, H, W), name_seed(tag + ".h"), 0.5) if has_h0 else None
        g_out = seeded_randn((B, T, C, H, W), name_seed(tag + ".g"))
        res = {}
        for pr in ("f32", "bf16x3"):
            dev = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
            dx = None if x is None else x.cuda().requires_grad_(True)
            dh = None if h0 is None else h0.cuda().requires_grad_(True)
            params = [dev[f"{n}.{kind}"] for n in names for kind in ("weight", "bias")]
            out, hT = traj_ops.trajgru_seq(dx, dh, params, seq_len=T, L=nl, slope=0.2, state_hw=(H, W), precision=pr)
            ((out * g_out.cuda()).sum() + 0.5 * (hT * hT).sum()).backward()
            res[pr] = (out.detach().cpu(), {**{k: v.grad.cpu() for k, v in dev.items() if v.grad is not None}, **({"x": dx.grad.cpu()} if dx is not None else {}), **({"h0": dh.grad.cpu()} if dh is not None else {})})
        ref = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        rx = None if x is None else x.clone().requires_grad_(True)
        rh = None if h0 is None else h0.clone().requires_grad_(True)
        ro, rhT = tr.trajgru_seq(rx, rh, T, ref, nl, 0.2)
        ((ro * g_out).sum() + 0.5 * (rhT * rhT).sum()).backward()
        rg = {**{k: v.grad for k, v in ref.items() if v.grad is not None}, **({"x": rx.grad} if rx is not None else {}), **({"h0": rh.grad} if rh is not None else {})}
        worst = []
        for pr in ("f32", "bf16x3"):
            for k in rg:
                if k not in res[pr][1]: continue
                a, r = res[pr][1][k].double(), rg[k].double()
                l2 = float(((a - r) ** 2).sum().sqrt() / (r ** 2).sum().sqrt().clamp_min(1e-30))
                mx = float((a - r).abs().max() / r.abs().max().clamp_min(1e-30))
                if l2 > 5e-4: worst.append((pr, k, round(l2, 5), round(mx, 5), int(((a - r).abs() > 1e-3 * r.abs().max()).sum()), a.numel()))
        print((chunk, i), (B, T, Cin, C, H, W, nl, has_x, has_h0, prec), "out err f32", float((res["f32"][0] - ro).abs().max() / ro.abs().max()), worst)
