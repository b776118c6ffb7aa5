"""GPU parity of the full models (drop-in VPModel surface) against reference-generated golden vectors."""
import json

import numpy as np
import pytest
import torch

import golden_cases as gc
from golden_util import checksum, fill_state_dict_, load_golden, name_seed, seeded_rand

pytestmark = pytest.mark.gpu
RTOL = 1e-4  # north_star: within 1e-4 relative (fp32)


from parity import relmax as _relmax   # max|a - b| / max|b|, recorded (tests/parity.py)


def _ef(vpx, tag, kw):
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["convlstm-shi"]("cuda", **kw)
    fill_state_dict_(m, name_seed("ef." + tag))
    return m.to("cuda")


@pytest.mark.parametrize("tag,kw,B,T,P", [("tiny", gc.EF_TINY_KW, 2, 3, 2), ("tiny3", gc.EF_TINY3_KW, 2, 2, 3)])
def test_ef_convlstm_forward_vs_golden(vpx, tag, kw, B, T, P):
    g = load_golden(f"ef_{tag}")
    m = _ef(vpx, tag, kw)
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, T + P, c, h, w), name_seed(f"ef.{tag}.frames")).cuda()
    with torch.no_grad():
        pred, ml = m(frames[:, :T], pred_frames=P)
        pred1 = m.pred_1(frames[:, :T])
    assert ml is None and pred.shape == (B, P, c, h, w) and pred1.shape == (B, c, h, w)
    assert _relmax(pred, g["pred"]) < RTOL and _relmax(pred1, g["pred1"]) < RTOL


@pytest.mark.parametrize("tag,c", [("full_c1", 1), ("full_c3", 3)])
def test_ef_convlstm_full_size_vs_golden(vpx, tag, c):
    """BASELINE configs C1/C2 shape: default convlstm-shi, 64x64, 10 -> 10."""
    g = load_golden(f"ef_{tag}")
    m = _ef(vpx, tag, dict(img_shape=(c, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0]))
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    x = seeded_rand((1, 10, c, 64, 64), name_seed(f"ef.{tag}.x"))
    assert abs(checksum(x) - float(g["chk_x"])) < 1e-9
    with torch.no_grad():
        pred, _ = m(x.cuda(), pred_frames=10)
    assert _relmax(pred[:, :, :, ::4, ::4], g["pred_slice"]) < RTOL
    assert abs(checksum(pred) - float(g["pred_chk"])) < 1e-3 * max(1.0, abs(float(g["pred_chk"])))


def test_reference_shape_contract(vpx):
    """The reference's own model test (tests/test_models.py:19-35): shapes of pred_1 and forward(pred_frames=5)."""
    from vp_suite_amd.models import MODEL_CLASSES
    b, p, (c, h, w) = 2, 5, (3, 64, 64)
    for key, cls in MODEL_CLASSES.items():
        model = cls("cuda", action_size=3, img_shape=(c, h, w), temporal_dim=3, action_conditional=False,
                    tensor_value_range=[0.0, 1.0]).to("cuda")
        t = p + 3 if cls.NEEDS_COMPLETE_INPUT else 3
        x = torch.randn(b, t, c, h, w, device="cuda")
        with torch.no_grad():
            assert model.pred_1(x).shape == (b, c, h, w)
            assert model(x, pred_frames=p)[0].shape == (b, p, c, h, w)


@pytest.mark.parametrize("tag,kw,B,T,P", [("tiny", gc.EF_TINY_KW, 2, 3, 2), ("tiny3", gc.EF_TINY3_KW, 2, 2, 3)])
def test_ef_convlstm_training_vs_golden(vpx, tag, kw, B, T, P):
    """Loss, per-parameter gradients, and parameters after Adam steps through the model's own train_iter / eval_iter
    (the harness semantics of base_model.py:148-216) against the reference-generated pins."""
    from vp_suite_amd.measure import PredictionLossProvider
    g = load_golden(f"ef_{tag}")
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, T + P, c, h, w), name_seed(f"ef.{tag}.frames")).cuda()
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    m = _ef(vpx, tag, kw)
    pred, _ = m(frames[:, :T], pred_frames=P)
    _, loss = lp.get_losses(pred, frames[:, T:])
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    flat = np.concatenate([named[k].grad.detach().cpu().numpy().reshape(-1) for k in sorted(named)])
    assert _relmax(flat, g["grads_flat"]) < RTOL

    m = _ef(vpx, tag, kw)
    cfg = {"device": "cuda", "context_frames": T, "pred_frames": P, "val_rec_criterion": "mse"}
    data = {"frames": frames, "actions": torch.zeros(B, T + P - 1, 0)}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for step in (1, 2, 3):
        m.train_iter(cfg, [data], opt, lp, epoch=0)
        if step in (1, 3):
            named = dict(m.named_parameters())
            pflat = np.concatenate([named[k].detach().cpu().numpy().reshape(-1) for k in sorted(named)])
            # Adam's first steps move every weight by ~lr regardless of gradient scale; compare absolutely
            assert np.abs(pflat[::3] - g[f"params_after{step}_s3"]).max() < 2e-5, step
    means, indicator = m.eval_iter(cfg, [data], lp)
    assert abs(means["mse"] - float(g["eval_mse_after3"])) < 1e-4 * abs(float(g["eval_mse_after3"]))
    assert m.training


def test_dp_trainer_on_gpu_single_rank(vpx):
    from vp_suite_amd.train import DataParallelTrainer
    m = _ef(vpx, "tiny", gc.EF_TINY_KW)
    tr = DataParallelTrainer(m, lr=1e-3, world_size=1)
    frames = seeded_rand((2, 5, 1, 16, 16), name_seed("ef.tiny.frames")).cuda()
    l0 = float(tr.step(frames[:, :3], frames[:, 3:], 2))
    for _ in range(5):
        l1 = float(tr.step(frames[:, :3], frames[:, 3:], 2))
    assert l1 < l0


def test_ef_convlstm_full_size_bf16x3_vs_golden(vpx):
    """The bench's default operand mode (split bf16, fp32 accumulate) on the BASELINE C1/C2 shape: still within the
    north-star tolerance of 1e-4 relative to the reference's fp32 CPU output."""
    g = load_golden("ef_full_c1")
    m = _ef(vpx, "full_c1", dict(img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0],
                                 cell_precision="bf16x3"))
    x = seeded_rand((1, 10, 1, 64, 64), name_seed("ef.full_c1.x"))
    with torch.no_grad():
        pred, _ = m(x.cuda(), pred_frames=10)
    err = _relmax(pred[:, :, :, ::4, ::4], g["pred_slice"])
    assert err < 1e-4, err


def test_ef_trajgru_model_vs_golden(vpx):
    """EF_TrajGRU ("trajgru", ef_traj_gru.py): forward, MSE loss and every parameter gradient of the tiny model against
    the reference-generated fixture; shape contract of the default 64x64 model (L = 13, ret with 13*96 input channels)."""
    from vp_suite_amd.models import MODEL_CLASSES
    kw = gc.EF_TRAJGRU_TINY_KW
    g = load_golden("ef_trajgru_tiny")
    m = MODEL_CLASSES["trajgru"]("cuda", **kw)
    assert list(m.state_dict().keys()) == list(json.loads(str(g["sd_shapes"])).keys())
    fill_state_dict_(m, name_seed("ef_trajgru.tiny"))
    m = m.cuda()
    c, h, w = kw["img_shape"]
    frames = seeded_rand((2, 5, c, h, w), name_seed("ef_trajgru.tiny.frames")).cuda()
    pred, ml = m(frames[:, :3], pred_frames=2)
    assert ml is None and _relmax(pred, g["pred"]) < RTOL
    from vp_suite_amd.measure import PredictionLossProvider
    _, loss = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}}).get_losses(pred, frames[:, 3:])
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    for k, p in m.named_parameters():
        if "grad." + k in g:
            assert _relmax(p.grad, g["grad." + k]) < 2e-4, k
        else:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "bf16"])
def test_ef_trajgru_default_size_forward(vpx, precision):
    """Shape contract of the default 64x64 EF-TrajGRU (L = 13, `ret` with 13*96 input channels; ef_traj_gru.py:31-75) in every operand
    mode, seeded. This is the forward that aborted GPUTEST_r04: its 5x5 flow-generator layers (h2f_conv1 64|96 -> 32, flows_conv
    32 -> 26) packed 4-12 KB more weights than vpx_conv2d_workspace_bytes sized (tests/test_workspace_contract.py pins the sizing
    rule on the CPU; the guard bands of tests/canary.py check the memory around every tensor of this call)."""
    from vp_suite_amd.models import MODEL_CLASSES
    big = MODEL_CLASSES["trajgru"]("cuda", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision=precision)
    fill_state_dict_(big, name_seed("ef_trajgru.default"))
    big = big.cuda()
    x = seeded_rand((2, 4, 1, 64, 64), name_seed("ef_trajgru.default.x")).cuda()
    with torch.no_grad():
        out, _ = big(x, pred_frames=3)
        out2, _ = big(x, pred_frames=3)
    assert out.shape == (2, 3, 1, 64, 64) and bool(torch.isfinite(out).all())
    torch.use_deterministic_algorithms(True)
    try:
        with torch.no_grad():
            d1, _ = big(x, pred_frames=3)
            d2, _ = big(x, pred_frames=3)
    finally:
        torch.use_deterministic_algorithms(False)
    assert torch.equal(d1, d2)                       # no K-split atomics in deterministic mode: bit-reproducible
    # (plain bf16 rounds every activation to 8 bits of mantissa: a last-bit difference in summation order can flip a rounding)
    bound = 1e-2 if precision == "bf16" else 1e-4
    assert _relmax(out, d1) < bound and _relmax(out2, d1) < bound


@pytest.mark.parametrize("tag", list(gc.PRED_ACTION_CASES))
def test_predrnn_action_conditional_vs_golden(vpx, tag):
    """Action-conditional PredRNN-V2 (predrnn_v2.py:62-121, 143-149, 178-221): `action_conditional=True` x
    {residual_on_action_conv, layer_norm} — prediction, decoupling loss, total loss and all gradients against the
    reference fixtures; the reference's argument checks."""
    from vp_suite_amd.measure import PredictionLossProvider
    from vp_suite_amd.models import MODEL_CLASSES
    from golden_util import seeded_randn
    g = load_golden(f"predrnn_action_{tag}")
    kw = dict(gc.PRED_ACTION_KW, **gc.PRED_ACTION_CASES[tag])
    m = MODEL_CLASSES["predrnn-pp"]("cuda", **kw)
    assert m.conv_actions_on_input and m.reverse_scheduled_sampling and not hasattr(m, "conv_last")
    assert list(m.state_dict().keys()) == list(json.loads(str(g["sd_shapes"])).keys())
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    fill_state_dict_(m, name_seed("predrnn_action." + tag))
    m = m.cuda().eval()
    c, h, w = kw["img_shape"]
    frames = seeded_rand((2, 5, c, h, w), name_seed(f"predrnn_action.{tag}.frames")).cuda()
    actions = seeded_randn((2, 5, kw["action_size"]), name_seed(f"predrnn_action.{tag}.actions")).cuda()
    pred, ml = m(frames, pred_frames=2, actions=actions)
    assert _relmax(pred, g["pred"]) < RTOL
    assert abs(float(ml["ST-LSTM decouple loss"]) - float(g["decouple"])) < 1e-4 * abs(float(g["decouple"]))
    _, loss = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}}).get_losses(pred, frames[:, 3:])
    loss = loss + ml["ST-LSTM decouple loss"]
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    flat = np.concatenate([named[k].grad.detach().cpu().numpy().reshape(-1) for k in sorted(named)])
    assert _relmax(flat, g["grads_flat"]) < 2e-4
    with pytest.raises(ValueError):
        m(frames, pred_frames=2)                                  # actions are mandatory
    with pytest.raises(ValueError):
        m(frames, pred_frames=2, actions=actions[..., :2])        # wrong action size
