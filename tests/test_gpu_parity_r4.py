"""Parity hardening at the sizes BASELINE.json's configs run: TRAINING gradients of the full models against
the oracle's autograd at real size — PredRNN-V2's training iteration (forward + time-reversed forward, a non-trivial sampling mask)
at configs[2] size, convlstm-shi at configs[3]'s full 10 -> 20 horizon, one deep-PredRNN step at configs[4]'s frame size. Every
figure is logged to gpurun_out/parity_r04.json (metric: max|got-ref| / max|ref|)."""
import numpy as np
import pytest
import torch

from golden_util import fill_state_dict_, name_seed, seeded_rand

pytestmark = pytest.mark.gpu


def _model(name, seed_tag, **kw):
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES[name]("cuda", action_size=0, tensor_value_range=[0.0, 1.0], **kw)
    fill_state_dict_(m, name_seed(seed_tag))
    return m.cuda().train()


def _predrnn_training_parity(vpx, parity_log, tag, img_shape, B, ctx, P, layers, w_bound):
    """model.training_loss (predrnn_v2.py:319-365: forward + reversed forward averaged, MSE + 100 x decoupling loss) with a FIXED
    Bernoulli(0.5) sampling mask fed to both sides, gradients of every parameter against the oracle's autograd."""
    from oracle import torch_ref as tr
    from vp_suite_amd.measure import PredictionLossProvider
    kw = dict(img_shape=img_shape, cell_precision="bf16x3")
    if layers:
        kw["num_layers"] = layers
    m = _model("predrnn-pp", f"parity.{tag}", **kw)
    L = m.num_layers
    c, H, W = img_shape
    frames = seeded_rand((B, ctx + P, c, H, W), name_seed(f"parity.{tag}.x"))
    gen = torch.Generator().manual_seed(name_seed(f"parity.{tag}.mask") % (2 ** 31))
    masks = [(torch.rand(B, P - 1, 1, 1, 1, generator=gen) < 0.5).float().expand(B, P - 1, m.patch_c, m.patch_h, m.patch_w).contiguous()
             for _ in range(2)]
    calls = {"n": 0}

    def fixed_mask(batch_size, context_frames, pred_frames, train):
        k = calls["n"]
        calls["n"] += 1
        return masks[k].cuda()
    m._scheduled_sampling = fixed_mask
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    x = frames.cuda()
    loss = m.training_loss(x, x[:, ctx:], P, lp)
    loss.backward()
    assert calls["n"] == 2

    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    # the decoupling loss is a mean of |cos|: a term whose cosine sits within rounding of zero takes sign +1 on one side and -1 on
    # the other, and flips a whole row of the adapter's gradient (weight 100 / (layer-steps * B * Ch) each). Count them in the oracle.
    near_kink = {"n": 0, "min": 1.0}
    plain_decouple = tr.decouple_term

    def counting_decouple(dc, dm, aw):
        with torch.no_grad():
            Bc, Chc = dc.shape[:2]
            a = torch.nn.functional.normalize(torch.nn.functional.conv2d(dc, aw).view(Bc, Chc, -1), dim=2)
            b = torch.nn.functional.normalize(torch.nn.functional.conv2d(dm, aw).view(Bc, Chc, -1), dim=2)
            cs = torch.cosine_similarity(a, b, dim=2).abs()
            near_kink["n"] += int((cs < 2e-5).sum())
            near_kink["min"] = min(near_kink["min"], float(cs.min()))
        return plain_decouple(dc, dm, aw)
    tr.decouple_term = counting_decouple
    total = 0.0
    for k, fr in enumerate((frames, torch.flip(frames, dims=[1]))):
        pred, dec = tr.predrnn_v2_forward(sd, fr, P, patch_size=m.patch_size, num_layers=L, mask_true=masks[k])
        total = total + tr.mse_measure(pred, fr[:, ctx:]) + dec
    tr.decouple_term = plain_decouple
    total = total / 2
    total.backward()
    rel = abs(float(loss) - float(total)) / abs(float(total))
    parity_log(f"{tag}.training_loss", loss.detach().reshape(1), total.detach().reshape(1), 1e-4)
    assert rel < 1e-4, (float(loss), float(total))
    bad = {}
    for k, p in m.named_parameters():
        e = parity_log(f"{tag}.grad.{k}", p.grad, sd[k].grad, w_bound)
        if k == "adapter.weight" and near_kink["n"] > 0:
            # |cos| kinks present (c3 run of round 4: max-norm 1.2e-2 from such terms, every other tensor <= 5.3e-5): hold the tensor
            # to a relative L2 error instead — a wrong kernel is off by O(1), a handful of flipped signs by < 5e-3
            g, r = p.grad.detach().cpu().numpy(), sd[k].grad.numpy()
            l2 = float(np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum()))
            if l2 > 5e-3:
                bad[k] = ("l2", l2, near_kink)
        elif e >= w_bound:
            bad[k] = e
    parity_log(f"{tag}.decouple_terms_within_2e-5_of_the_abs_kink", torch.tensor([float(near_kink["n"])]), torch.tensor([1.0]), None)
    assert not bad, (bad, near_kink)


def test_predrnn_c3_training_gradients_vs_oracle(vpx, parity_log):
    """BASELINE configs[2]: predrnn-pp, 1x64x64, 10 -> 10, 3 layers x 128 channels, B = 4: every weight gradient through 2 x 19 steps
    of BPTT (one-launch 5x5 weight gradients, K-split data gradients, decoupling tail) within 2e-4 of the oracle's."""
    _predrnn_training_parity(vpx, parity_log, "c3train", (1, 64, 64), 4, 10, 10, None, 2e-4)


def test_predrnn_c5_training_step_vs_oracle(vpx, parity_log):
    """BASELINE configs[4]'s frame size and depth (128x128x3, 4 layers; 32x32 maps, 48 input channels) at a per-GPU shard of B = 2,
    horizon 10 -> 10 (the full 10 -> 30 forward is test_c5_deep_predrnn_full_horizon_vs_oracle; its BPTT on the CPU oracle is minutes)."""
    _predrnn_training_parity(vpx, parity_log, "c5train", (3, 128, 128), 2, 10, 10, 4, 2e-4)


def test_c4_training_full_horizon_one_sample_vs_oracle(vpx, parity_log):
    """BASELINE configs[3]: convlstm-shi on 128x128x3 at the FULL 10 -> 20 horizon (BPTT through 30 steps of 6 blocks), one sample:
    loss and every gradient against the oracle's autograd. Weights / biases: max-norm 2e-4; peepholes (per-pixel sums, see
    test_c4_training_step_batch4_vs_oracle): relative L2 <= 5e-3."""
    from oracle import torch_ref as tr
    from vp_suite_amd.measure import PredictionLossProvider
    m = _model("convlstm-shi", "parity.c4full", img_shape=(3, 128, 128), cell_precision="bf16x3")
    frames = seeded_rand((1, 30, 3, 128, 128), name_seed("parity.c4full.x"))
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    pred, _ = m(frames[:, :10].cuda(), pred_frames=20)
    _, loss = lp.get_losses(pred, frames[:, 10:].cuda())
    loss.backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    rl = tr.mse_measure(tr.ef_convlstm_forward(sd, frames[:, :10], 20), frames[:, 10:])
    rl.backward()
    parity_log("c4full.loss", loss.detach().reshape(1), rl.detach().reshape(1), 1e-4)
    assert abs(float(loss) - float(rl)) < 1e-4 * abs(float(rl))
    bad = {}
    for k, p in m.named_parameters():
        g, r = p.grad.detach().cpu().numpy(), sd[k].grad.numpy()
        if k.split(".")[-1] in ("Wci", "Wcf", "Wco"):
            l2 = float(np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum()))
            parity_log(f"c4full.grad.{k}", p.grad, sd[k].grad, None)
            if l2 > 5e-3:
                bad[k] = ("l2", l2)
        elif parity_log(f"c4full.grad.{k}", p.grad, sd[k].grad, 2e-4) >= 2e-4:
            bad[k] = float(np.abs(g - r).max() / np.abs(r).max())
    assert not bad, bad
