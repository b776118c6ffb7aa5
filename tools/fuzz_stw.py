"""Developer tool (GPU): random PredRNN-V2 training passes — deferred weight gradients (stw_kernel over a whole pass: 1..32 K slices, both block
decodes) against the first-generation per-step weight gradients (VPX_OPT_EXPERIMENT bit 6), every parameter gradient. usage: fuzz_stw.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import vp_suite_amd as vpx
from vp_suite_amd.measure import PredictionLossProvider
from vp_suite_amd.models import MODEL_CLASSES
from golden_util import fill_state_dict_, name_seed, seeded_rand

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
L = vpx._lib.lib()
lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
worst_all = 0.0
for case in range(n_cases):
    ch = int(rng.choice([16, 32, 64, 96, 128]))
    layers = int(rng.integers(1, 3))
    patch = int(rng.choice([2, 4]))
    hw = int(rng.choice([16, 32, 64])) if patch == 4 else int(rng.choice([16, 32]))
    c = int(rng.choice([1, 2]))
    B = int(rng.integers(1, 25 if ch >= 96 else 49))
    Ttot = int(rng.integers(4, 11)); P = int(rng.integers(1, Ttot - 1))
    kw = dict(img_shape=(c, hw, hw), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=layers, num_hidden=[ch] * layers, patch_size=patch,
              cell_precision="bf16x3")
    frames = seeded_rand((B, Ttot, c, hw, hw), name_seed(f"fuzz_stw.{case}")).cuda()
    res = {}
    for first_generation in (False, True):
        prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 64 if first_generation else 0)
        try:
            m = MODEL_CLASSES["predrnn-pp"]("cuda", **kw)
            fill_state_dict_(m, name_seed(f"fuzz_stw.model.{case}"))
            m = m.to("cuda")
            m.sampling_eta = 0.5
            torch.manual_seed(1000 + case)
            loss = m.training_loss(frames, frames[:, Ttot - P:], P, lp)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
        named = dict(m.named_parameters())
        res[first_generation] = (float(loss.detach()), {k: named[k].grad.detach().cpu().numpy() for k in sorted(named)})
        del m, loss
    a, b = res[False], res[True]
    # adapter.weight: the decoupling term is a mean of |cos| — where a (sample, channel) cosine sits within rounding of zero the two runs'
    # forward passes (second- vs first-generation kernels) put it on different sides of the kink and that pair's contribution flips sign
    # (tests/test_gpu_parity_r4.py holds this gradient to 5e-3 for the same reason); it does not pass through stw_kernel at all
    rel = {k: float(np.abs(a[1][k] - b[1][k]).max() / (np.abs(b[1][k]).max() + 1e-30)) for k in a[1]}
    kink = rel.pop("adapter.weight", 0.0)
    worst = max((v, k) for k, v in rel.items())
    if kink > 2e-2: worst = (kink, "adapter.weight (beyond the kink allowance)")
    maps = hw // patch
    items = 2 * B * (Ttot - 1) * ((maps + 15) // 16) * ((maps + 3) // 4)
    ok = worst[0] < (5e-5 if not worst[1].startswith('adapter') else 0.0) and abs(a[0] - b[0]) < 2e-6 * abs(b[0])
    worst_all = max(worst_all, worst[0])
    print(f"case {case:3d}: Ch={ch:3d} L={layers} maps={maps:2d} B={B:2d} T={Ttot:2d} items={items:6d}  loss diff {abs(a[0] - b[0]) / abs(b[0]):.1e}  worst grad {worst[0]:.2e} ({worst[1]}), adapter {kink:.1e}  {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
print(f"{n_cases} cases ok; worst relative gradient difference {worst_all:.2e}")
