"""GPU parity of the ST-LSTM cell (C ABI: vpx_stlstm_step_fwd/_bwd) and the PredRNN-V2 model against
reference-generated golden vectors."""
import json

import numpy as np
import pytest
import torch

import golden_cases as gc
from golden_util import checksum, fill_state_dict_, load_golden, name_seed, seeded_rand, seeded_state_dict

pytestmark = pytest.mark.gpu
RTOL = 1e-5
GRTOL = 5e-5


from parity import relmax as _relmax   # max|a - b| / max|b|, recorded (tests/parity.py)


@pytest.mark.parametrize("tag", ["plain", "k3", "ln"])
def test_stlstm_cell_vs_golden(vpx, tag):
    from vp_suite_amd.model_blocks import SpatioTemporalLSTMCell
    Cin, Ch, H, W, k, ln, B = gc.STLSTM_CASES[tag]
    g = load_golden(f"stlstm_{tag}")
    cell = SpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, ln)
    fill_state_dict_(cell, name_seed("stlstm." + tag))
    cell = cell.cuda()
    inp = {n: v.cuda() for n, v in gc.stlstm_inputs(tag, Cin, Ch, H, W, B).items()}
    lv = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h", "c", "m")}
    outs = cell(lv["x"], lv["h"], lv["c"], lv["m"])
    for o, n in zip(outs, ("h_new", "c_new", "m_new", "delta_c", "delta_m")):
        assert _relmax(o, g[n]) < RTOL, n
    sum((o * inp[gn]).sum() for o, gn in zip(outs, ("g_h", "g_c", "g_m", "g_dc", "g_dm"))).backward()
    for n in ("x", "h", "c", "m"):
        assert _relmax(lv[n].grad, g["d" + n]) < GRTOL, n
    for key, prm in cell.named_parameters():
        assert _relmax(prm.grad, g["grad." + key]) < GRTOL, key
    # second call re-uses the packed weights held in the cell's workspace
    with torch.no_grad():
        o2 = cell(inp["x"], inp["h"], inp["c"], inp["m"])
        o3 = cell(inp["x"], inp["h"], inp["c"], inp["m"])
        assert _relmax(o3[0], g["h_new"]) < RTOL and _relmax(o2[0], o3[0]) < 1e-6
        # small maps split the contraction over workgroups (atomic partial sums): bit-reproducible only on request
        torch.use_deterministic_algorithms(True)
        try:
            d2 = cell(inp["x"], inp["h"], inp["c"], inp["m"])
            d3 = cell(inp["x"], inp["h"], inp["c"], inp["m"])
        finally:
            torch.use_deterministic_algorithms(False)
    assert _relmax(d3[0], g["h_new"]) < RTOL and torch.equal(d2[0], d3[0])


def test_stlstm_real_shape_vs_oracle(vpx):
    """PredRNN default cell shapes (BASELINE C3): layer 0 (16 -> 128) and layer 1 (128 -> 128), 16x16, k=5."""
    from oracle import torch_ref as tr
    from vp_suite_amd.model_blocks import SpatioTemporalLSTMCell
    for Cin in (16, 128):
        Ch, H, W, k, B = 128, 16, 16, 5, 2
        cell = SpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, False)
        fill_state_dict_(cell, name_seed(f"stlstm.real{Cin}"))
        sd = {kk: v.clone() for kk, v in cell.state_dict().items()}
        inp = gc.stlstm_inputs(f"real{Cin}", Cin, Ch, H, W, B)
        with torch.no_grad():
            ref = tr.stlstm_cell(inp["x"], inp["h"], inp["c"], inp["m"], sd, "", False)
            out = cell.cuda()(*(inp[n].cuda() for n in ("x", "h", "c", "m")))
        for o, r in zip(out, ref):
            assert _relmax(o, r) < RTOL


def _predrnn(tag, kw, **extra):
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["predrnn-pp"]("cuda", **kw, **extra)
    fill_state_dict_(m, name_seed("predrnn." + tag))
    return m.to("cuda")


def test_predrnn_tiny_vs_golden(vpx):
    from vp_suite_amd.measure import PredictionLossProvider
    tag, kw, B, Ttot, P = "tiny", gc.PRED_TINY_KW, 2, 5, 2
    g = load_golden(f"predrnn_{tag}")
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn.{tag}.frames")).cuda()
    m = _predrnn(tag, kw).eval()
    pred, ml = m(frames, pred_frames=P)
    assert _relmax(pred, g["eval.pred"]) < 1e-4
    assert abs(float(ml["ST-LSTM decouple loss"]) - float(g["eval.decouple"])) < 1e-4 * abs(float(g["eval.decouple"]))
    with torch.no_grad():
        assert _relmax(m.pred_1(frames[:, :Ttot - P + 1]), g["eval.pred1"]) < 1e-4
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    _, loss = lp.get_losses(pred, frames[:, Ttot - P:])
    loss = loss + ml["ST-LSTM decouple loss"]
    assert abs(float(loss) - float(g["eval.loss"])) < 1e-4 * abs(float(g["eval.loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    flat = np.concatenate([named[k].grad.detach().cpu().numpy().reshape(-1) for k in sorted(named)])
    assert _relmax(flat, g["eval.grads_flat"]) < 1e-4
    # reverse scheduled sampling, eval
    with torch.no_grad():
        pr, _ = _predrnn(tag, kw, reverse_scheduled_sampling=True).eval()(frames, pred_frames=P)
    assert _relmax(pr, g["rss_eval.pred"]) < 1e-4
    # train=True with the reference's random draws injected
    m = _predrnn(tag, kw)
    m.sampling_eta = 0.5
    flips = torch.from_numpy(g["train.random_flip"])
    real_rand = torch.rand
    try:
        torch.rand = lambda *a, **k: flips.to(k.get("device", "cpu"))
        with torch.no_grad():
            pt, _ = m(frames, pred_frames=P, train=True)
    finally:
        torch.rand = real_rand
    assert abs(m.sampling_eta - float(g["train.eta_after"])) < 1e-12
    assert _relmax(pt, g["train.pred"]) < 1e-4


def test_predrnn_train_iter_vs_golden(vpx):
    """The model's own train_iter (forward + reversed forward averaged, predrnn_v2.py:319-365) for 2 Adam steps."""
    from vp_suite_amd.measure import PredictionLossProvider
    tag, kw, B, Ttot, P = "tiny", gc.PRED_TINY_KW, 2, 5, 2
    g = load_golden(f"predrnn_{tag}")
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn.{tag}.frames")).cuda()
    m = _predrnn(tag, kw, sampling_changing_rate=2.0)
    cfg = {"device": "cuda", "context_frames": Ttot - P, "pred_frames": P, "val_rec_criterion": "mse"}
    data = {"frames": frames, "actions": torch.zeros(B, Ttot - 1, 0)}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    for step in (1, 2):
        m.train_iter(cfg, [data], opt, lp, epoch=0)
        named = dict(m.named_parameters())
        pflat = np.concatenate([named[k].detach().cpu().numpy().reshape(-1) for k in sorted(named)])
        assert np.abs(pflat[::5] - g[f"params_after{step}_s5"]).max() < 5e-5, step
    assert m.training_iteration == int(g["training_iteration_after2"])
    assert abs(m.sampling_eta - float(g["sampling_eta_after2"])) < 1e-9


@pytest.mark.parametrize("mode", ["standard", "reverse_sampling", "action"])
def test_fused_reversed_pass_equals_two_passes(vpx, mode):
    """training_loss with the sequence and its time-reversal as ONE batch of 2B samples (fuse_reversed_pass, the default) against the
    reference's two passes one after the other (predrnn_v2.py:326-352): same sampling masks from the same RNG stream (real coin flips:
    sampling_eta ~ 0.5), same schedule state afterwards, loss and every gradient equal up to fp32 summation order. Crossed with
    defer_weight_gradients (ops.STWeightBank: a cell's weight gradients once per pass over all its steps; the action-conditional model
    has no such path and must simply ignore the switch)."""
    from golden_util import seeded_randn
    from vp_suite_amd.measure import PredictionLossProvider
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    if mode == "action":
        kw, extra, tag = dict(gc.PRED_ACTION_KW, **gc.PRED_ACTION_CASES["plain"]), dict(sampling_eta=0.5), "action"
        from vp_suite_amd.models import MODEL_CLASSES
        def make():
            m = MODEL_CLASSES["predrnn-pp"]("cuda", **kw)
            fill_state_dict_(m, name_seed("predrnn_action.plain"))
            return m.cuda()
    else:
        kw, tag = dict(img_shape=(1, 32, 32), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[16, 16], cell_precision="bf16x3"), "fuse"
        extra = dict(reverse_scheduled_sampling=True) if mode == "reverse_sampling" else {}
        def make():
            return _predrnn(tag, dict(kw, **extra))
    c, h, w = kw["img_shape"]
    B, Ttot, P = 3, 7, 3
    frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn.{tag}.frames")).cuda()
    actions = seeded_randn((B, Ttot, kw["action_size"]), name_seed(f"predrnn.{tag}.actions")).cuda() if mode == "action" else None
    res = {}
    for fused, defer in ((True, True), (True, False), (False, True), (False, False)):
        m = make()
        m.fuse_reversed_pass, m.defer_weight_gradients = fused, defer
        m.sampling_eta, m.training_iteration = 0.5, 30000   # both schedules in their stochastic range
        torch.manual_seed(1234)
        kwargs = {"actions": actions} if actions is not None else {}
        loss = m.training_loss(frames, frames[:, Ttot - P:], P, lp, **kwargs)
        loss.backward()
        named = dict(m.named_parameters())
        res[(fused, defer)] = (float(loss), np.concatenate([named[k].grad.detach().cpu().numpy().reshape(-1) for k in sorted(named)]),
                               m.sampling_eta, m.training_iteration, float(torch.rand(1, device="cuda")))
    b = res[(False, False)]   # the reference's schedule: two passes, every step's weight gradients on their own
    for key in ((True, True), (True, False), (False, True)):
        a = res[key]
        assert abs(a[0] - b[0]) < 2e-6 * abs(b[0]), key
        assert _relmax(a[1], b[1]) < 2e-5, key
        assert a[2] == b[2] and a[3] == b[3] and a[4] == b[4], key   # schedule state and the RNG stream's position


@pytest.mark.parametrize("frozen", ["cell0", "all"])
def test_training_forward_with_frozen_cells(vpx, frozen):
    """with cell 0's weights frozen its step at t = 0 has nothing that requires a gradient (frames and zero states) and saves
    nothing for the backward — the deferred-weight-gradient banks must not be built then (the reference and the per-step path run this
    case). Loss equal to the unfrozen model's; the trainable cells' gradients equal to the per-step path's; a frozen cell has none; and a
    cell driven directly afterwards still runs with its own defaults (the model no longer flips use_shadows on the modules)."""
    from vp_suite_amd.measure import PredictionLossProvider
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    kw = dict(img_shape=(1, 32, 32), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[16, 16], cell_precision="bf16x3")
    B, Ttot, P = 3, 6, 3
    frames = seeded_rand((B, Ttot, 1, 32, 32), name_seed("predrnn.frozen.frames")).cuda()
    res = {}
    for defer in (True, False):
        m = _predrnn("fuse", kw)
        m.defer_weight_gradients = defer
        if frozen == "cell0":
            m.cell_list[0].requires_grad_(False)
        else:
            m.requires_grad_(False)
        m.sampling_eta = 0.5
        torch.manual_seed(99)
        loss = m.training_loss(frames, frames[:, Ttot - P:], P, lp)
        if frozen == "cell0":
            loss.backward()
        named = dict(m.named_parameters())
        assert all(p.grad is None for k, p in named.items() if k.startswith("cell_list.0.") or frozen == "all")
        res[defer] = (float(loss.detach()), {k: p.grad.detach().cpu().numpy() for k, p in named.items() if p.grad is not None})
        assert all(c.use_shadows is False for c in m.cell_list)
    assert abs(res[True][0] - res[False][0]) < 2e-6 * abs(res[False][0])
    for k, g in res[False][1].items():
        assert _relmax(res[True][1][k], g) < 2e-5, k
    m = _predrnn("fuse", kw)
    m.sampling_eta = 0.5
    torch.manual_seed(99)
    ref = float(m.training_loss(frames, frames[:, Ttot - P:], P, lp).detach())
    assert abs(res[False][0] - ref) < 2e-6 * abs(ref)


@pytest.mark.parametrize("B,Ttot,slices", [(24, 12, 8), (48, 12, 16), (128, 17, 32)])
def test_deferred_weight_gradients_on_whole_slices_per_xcd(vpx, B, Ttot, slices):
    """The one-launch ST-LSTM weight gradient (stw_kernel) over a whole pass: 8 / 16 / 32 K slices, whole slices per XCD (round 5: its
    block decode for slice counts in 8s; `stw_slices`) — against the FIRST-GENERATION per-step weight gradients (VPX_OPT_EXPERIMENT bit 6:
    other kernels, other summation order) on the same model, same frames, same sampling masks. 8x8 maps of two items each: 2B x (Ttot - 1)
    images -> 1 056 / 2 112 / 8 192 items."""
    from vp_suite_amd.measure import PredictionLossProvider
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    kw = dict(img_shape=(1, 32, 32), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[16, 16], cell_precision="bf16x3")
    P = 4
    frames = seeded_rand((B, Ttot, 1, 32, 32), name_seed(f"predrnn.slices{slices}.frames")).cuda()
    L = vpx._lib.lib()
    res = {}
    for first_generation in (False, True):
        prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 64 if first_generation else 0)
        try:
            m = _predrnn("slices", kw)
            m.sampling_eta = 0.5
            torch.manual_seed(4321)
            loss = m.training_loss(frames, frames[:, Ttot - P:], P, lp)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
        named = dict(m.named_parameters())
        res[first_generation] = (float(loss.detach()), {k: named[k].grad.detach().cpu().numpy() for k in sorted(named)})
    a, b = res[False], res[True]
    assert abs(a[0] - b[0]) < 2e-6 * abs(b[0])
    worst = max((_relmax(a[1][k], b[1][k]), k) for k in a[1])
    assert worst[0] < 5e-5, worst


@pytest.mark.parametrize("B,slices", [(32, 16), (96, 32)])
def test_deferred_weight_gradients_at_model_width_on_whole_slices(vpx, B, slices):
    """The same comparison at the reference's default width (128 hidden channels: the pair table of the bench's launches — eight 128-row
    tiles of dG8, 36 k x k pairs + the 1x1 tensor's), 64x64 frames -> 16x16 maps of four items: 2B x 11 images = 2 816 / 8 448 items ->
    16 / 32 K slices, two / four whole slices per XCD. One training pass each way; every parameter gradient."""
    from vp_suite_amd.measure import PredictionLossProvider
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    kw = dict(img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[128, 128], cell_precision="bf16x3")
    Ttot, P = 12, 4
    frames = seeded_rand((B, Ttot, 1, 64, 64), name_seed(f"predrnn.wide{slices}.frames")).cuda()
    L = vpx._lib.lib()
    res = {}
    for first_generation in (False, True):
        prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 64 if first_generation else 0)
        try:
            m = _predrnn("wide", kw)
            m.sampling_eta = 0.5
            torch.manual_seed(777)
            loss = m.training_loss(frames, frames[:, Ttot - P:], P, lp)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
        named = dict(m.named_parameters())
        res[first_generation] = (float(loss.detach()), {k: named[k].grad.detach().cpu().numpy() for k in sorted(named)})
        del m, loss
        torch.cuda.empty_cache()
    a, b = res[False], res[True]
    assert abs(a[0] - b[0]) < 2e-6 * abs(b[0])
    worst = max((_relmax(a[1][k], b[1][k]), k) for k in a[1])
    assert worst[0] < 5e-5, worst


@pytest.mark.parametrize("variant", ["plain", "layer_norm"])
def test_batched_decoupling_tail_equals_per_step_tails(vpx, variant):
    """batch_decoupling_tail (the default): every layer-step writes its delta_c / delta_m into one slab and ONE vpx_decouple_fwd/_bwd pair
    runs over all of them, against one tail per layer-step as the reference's loop does (predrnn_v2.py:197-211, 229). Training loss, the
    decoupling value itself in eval mode, and every gradient; plain cells (second-generation kernels, deferred weight gradients) and the
    LayerNorm variant (first-generation path)."""
    from vp_suite_amd.measure import PredictionLossProvider
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    kw = dict(img_shape=(1, 32, 32), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[16, 16], cell_precision="bf16x3",
              layer_norm=(variant == "layer_norm"))
    B, Ttot, P = 3, 6, 3
    frames = seeded_rand((B, Ttot, 1, 32, 32), name_seed("predrnn.btail.frames")).cuda()
    res = {}
    for batched in (True, False):
        m = _predrnn("btail", kw)
        m.batch_decoupling_tail = batched
        m.sampling_eta = 0.5
        torch.manual_seed(99)
        loss = m.training_loss(frames, frames[:, Ttot - P:], P, lp)
        loss.backward()
        named = dict(m.named_parameters())
        flat = np.concatenate([named[k].grad.detach().cpu().numpy().reshape(-1) for k in sorted(named)])
        with torch.no_grad():
            _, ml = m.eval()(frames, pred_frames=P)
        res[batched] = (float(loss), flat, float(ml["ST-LSTM decouple loss"]))
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) < 2e-6 * abs(b[0]) and abs(a[2] - b[2]) < 2e-6 * abs(b[2])
    assert _relmax(a[1], b[1]) < 2e-5


def test_predrnn_full_size_vs_golden(vpx):
    """BASELINE config C3: default predrnn-pp, 64x64, 10 -> 10."""
    g = load_golden("predrnn_full_c1")
    m = _predrnn("full_c1", dict(img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0])).eval()
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    frames = seeded_rand((1, 20, 1, 64, 64), name_seed("predrnn.full_c1.frames"))
    with torch.no_grad():
        pred, ml = m(frames.cuda(), pred_frames=10)
    assert _relmax(pred[:, :, :, ::4, ::4], g["pred_slice"]) < 1e-4
    assert abs(float(ml["ST-LSTM decouple loss"]) - float(g["decouple"])) < 1e-3 * abs(float(g["decouple"]))


@pytest.mark.parametrize("B,Cin,Ch,H,W", [(3, 8, 24, 12, 20), (2, 16, 128, 16, 16), (5, 48, 40, 9, 33), (9, 48, 128, 32, 32)])
def test_stlstm_second_generation_backward_matches_first_generation(vpx, B, Cin, Ch, H, W):
    """Round-4 kernels of the ST-LSTM step's backward against the first-generation launches — same operands, same bf16x3 split, fp32
    summation order only:
      * stw (wgrad2.hip): the four 5x5 weight gradients + conv_last's in ONE launch on split operands (pair table, three tap
        passes, K split of tap row 4 / of the centre tap, half-empty column tiles, ragged maps and row tiles) — VPX_OPT_EXPERIMENT
        bit 6 switches it (and everything built on its split dG8) off;
      * c5 (convq.hip): the 5x5 data gradients (conv_o's adjoint; dx | dh | dm as the jobs of one launch) on 16x16-pixel tiles with
        8-channel stages — bit 7 switches it off alone;
      * c5 forward (channels in 32s): both gate groups as the jobs of one launch (gate-interleaved N tiles, fused gate math, c_new /
        m_new also written in the split format), conv_o + output gate as another — bit 8 switches it off;
      * c1 (conv1.hip): conv_last and its adjoint as a streaming kernel with register-resident weights (Ch = 128 only) — bit 9;
      * on grids below 48 (forward) / 96 (backward) pixel tiles (every shape here) the c5 launches run K-SPLIT (chunks of K as separate jobs writing partial sums,
        st_pointwise.hip adds them): the default; bit 10 forces the unsplit forms, bit 11 the first generation."""
    from golden_util import seeded_randn
    k = 5
    tag = f"stw.{B}.{Cin}.{Ch}.{H}.{W}"
    names = ("x", "h", "c", "m")
    inp = {n: seeded_randn((B, Cin if n == "x" else Ch, H, W), name_seed(f"{tag}.{n}"), 0.5).cuda() for n in names}
    shapes = {"Wx": (7 * Ch, Cin, k, k), "Wh": (4 * Ch, Ch, k, k), "Wm": (3 * Ch, Ch, k, k), "Wo": (Ch, 2 * Ch, k, k), "Wlast": (Ch, 2 * Ch, 1, 1)}
    Ws = {n: seeded_randn(s, name_seed(f"{tag}.{n}"), 1.0 / np.sqrt(s[1] * s[2] * s[3])).cuda() for n, s in shapes.items()}
    gout = [seeded_randn((B, Ch, H, W), name_seed(f"{tag}.g{i}")).cuda() for i in range(5)]

    def run():
        a = [inp[n].clone().requires_grad_(True) for n in names]
        w = [Ws[n].clone().requires_grad_(True) for n in shapes]
        outs = vpx.ops.stlstm_step(*a, *w, precision="bf16x3")
        sum((o * g).sum() for o, g in zip(outs, gout)).backward()
        return [o.detach() for o in outs] + [t.grad for t in a + w]
    L = vpx._lib.lib()
    labels = ["h_new", "c_new", "m_new", "delta_c", "delta_m"] + list(names) + list(shapes)
    # deterministic mode for every run: the first generation's K-split data gradients otherwise add their partial sums with
    # atomics, and two runs then see dG differing in the last bits (the comparisons below would be flaky at their 5e-6 bar)
    torch.use_deterministic_algorithms(True)
    FORCE = 1024   # the c5 launches also below their grid bar (these test shapes are small)
    L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, FORCE)
    try:
        new = run()
        res = {}
        # 0 = the product's choice on these small grids: the K-split job forms of c5 (partial sums + pointwise stages); 2048 = the
        # first generation there
        # (round 6) 1 << 27: conv_last reads the fp32 c_new / m_new and converts in the kernel, instead of the split copies the gate stage leaves
        for bits in (FORCE | 64, FORCE | 128, FORCE | 256, FORCE | 64 | 256, FORCE | 512, FORCE | (1 << 27), 0, 1 << 27, 2048, 2048 | 64):
            prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, bits)
            try:
                res[bits] = run()
            finally:
                L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
        again = run()
    finally:
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 0)
        torch.use_deterministic_algorithms(False)
    for bits, old in res.items():
        for n, a, b in zip(labels, new, old):
            assert _relmax(a, b) < 5e-6, (bits, n, _relmax(a, b))
    for a, b in zip(new, again):
        assert torch.equal(a, b)   # no atomics in either kernel or in the slice reduction: bit-reproducible
    # the 1x1 layer on split sources multiplies the very same (hi, lo) pairs the in-kernel conversion makes: bit-identical h_new
    assert torch.equal(new[0], res[FORCE | (1 << 27)][0]) and torch.equal(res[0][0], res[1 << 27][0])


@pytest.mark.parametrize("B", [3, 100])
def test_stlstm_split_shadows_change_nothing(vpx, B):
    """h_new / c_new / m_new of a step carry a shadow in the split operand format (`_vpx_sp`); the next step takes the shadows of
    its x, h, m instead of converting them, the backward those of all five sources. Same bits either way: two chained steps with the
    outputs passed on as they are (shadows used) against the same steps on clones (no shadow: the library converts), forward and
    every gradient — small grid (K-split forms) and 100 pixel tiles (unsplit forms). A shadow whose tensor was written in place is dropped."""
    from golden_util import seeded_randn
    Cin = Ch = 128
    H = W = 16
    k = 5
    tag = f"shadow.{B}"
    shapes = {"Wx": (7 * Ch, Cin, k, k), "Wh": (4 * Ch, Ch, k, k), "Wm": (3 * Ch, Ch, k, k), "Wo": (Ch, 2 * Ch, k, k), "Wlast": (Ch, 2 * Ch, 1, 1)}
    Ws = {n: seeded_randn(s_, name_seed(f"{tag}.{n}"), 1.0 / np.sqrt(s_[1] * s_[2] * s_[3])).cuda() for n, s_ in shapes.items()}
    st = [seeded_randn((B, Ch, H, W), name_seed(f"{tag}.s{i}"), 0.5).cuda() for i in range(4)]
    gout = [seeded_randn((B, Ch, H, W), name_seed(f"{tag}.g{i}")).cuda() for i in range(5)]

    def run(pass_on):
        w = [Ws[n].clone().requires_grad_(True) for n in shapes]
        a = [t.clone().requires_grad_(True) for t in st]
        o1 = vpx.ops.stlstm_step(*a, *w, precision="bf16x3", use_shadows=True)
        assert all(hasattr(t, "_vpx_sp") for t in o1[:3])
        x2, h2, c2, m2 = o1[0], o1[0], o1[1], o1[2]          # next layer: x = h_new; same cell next step: h, c; zig-zag memory: m_new
        if not pass_on:
            x2, h2, c2, m2 = (t * 1.0 for t in (x2, h2, c2, m2))   # new tensors: no shadow
            assert not hasattr(x2, "_vpx_sp")
        o2 = vpx.ops.stlstm_step(x2, h2, c2, m2, *w, precision="bf16x3", use_shadows=True)
        sum((o * g).sum() for o, g in zip(o2, gout)).backward()
        return [o.detach() for o in o2] + [t.grad for t in a + w]
    torch.use_deterministic_algorithms(True)
    try:
        with_sh, without = run(True), run(False)
    finally:
        torch.use_deterministic_algorithms(False)
    for a, b in zip(with_sh, without):
        assert torch.equal(a, b)
    # what ends a shadow's life: an in-place write (version counter), a consumer with another channel count, a new epoch (the next
    # model forward), an explicit invalidate — the answer to writes the version counter cannot see (`.data`, raw-pointer kernels)
    Wl = [Ws[n] for n in shapes]
    with torch.no_grad():
        o = vpx.ops.stlstm_step(*st, *Wl, precision="bf16x3", use_shadows=True)
        assert vpx.ops._shadow_of(o[0], o[0], Ch) is not None
        assert vpx.ops._shadow_of(o[0], o[0], Ch // 2) is None
        o[0].mul_(2.0)
        assert vpx.ops._shadow_of(o[0], o[0], Ch) is None
        assert vpx.ops._shadow_of(o[2], o[2], Ch) is not None
        vpx.ops.new_shadow_epoch()
        assert vpx.ops._shadow_of(o[2], o[2], Ch) is None
        # a write through .data: invisible to the version counter — after invalidate_shadow the step reads the tensor itself
        o = vpx.ops.stlstm_step(*st, *Wl, precision="bf16x3", use_shadows=True)
        o[0].data.mul_(0.5)
        vpx.ops.invalidate_shadow(o[0])
        got = vpx.ops.stlstm_step(o[0], o[0], o[1], o[2], *Wl, precision="bf16x3", use_shadows=True)
        want = vpx.ops.stlstm_step(*(t.clone() for t in (o[0], o[0], o[1], o[2])), *Wl, precision="bf16x3")
        for a, b in zip(got, want):
            assert _relmax(a, b) < 1e-6
        # without the opt-in (a cell driven by user code) no shadow is ever attached or read
        o = vpx.ops.stlstm_step(*st, *Wl, precision="bf16x3")
        assert not any(hasattr(t, "_vpx_sp") for t in o)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_single_tile_weight_gradients_with_many_k_slices(vpx, prec):
    """Layers whose whole dW is one output tile (the decoupling adapter, a 1x1 / 3x3 head with <= 64 channels) cut their pixel sum into up to
    256 K slices (round 5: 32 left most of the chip idle). At sizes where more than 32 slices are launched: the adapter gradient against the
    oracle's restatement of predrnn_v2.py:197-206 under autograd, conv2d's against torch."""
    from golden_util import seeded_randn
    from oracle import torch_ref as tr
    B, Ch, H, W = 48, 64, 16, 32     # 192 work items -> 192 slices
    A = seeded_randn((Ch, Ch, 1, 1), name_seed("slices.adapter"), 1.0 / np.sqrt(Ch))
    dc = seeded_randn((B, Ch, H, W), name_seed("slices.dc"))
    dm = seeded_randn((B, Ch, H, W), name_seed("slices.dm"))
    ref = [t.clone().requires_grad_(True) for t in (dc, dm, A)]
    rv = tr.decouple_term(*ref)
    rv.backward()
    mine = [t.cuda().requires_grad_(True) for t in (dc, dm, A)]
    v = vpx.ops.decouple_term(*mine, prec) if prec != "f32" else vpx.ops.decouple_term(*mine)
    v.backward()
    assert abs(float(v) - float(rv)) < 1e-5 * abs(float(rv))
    tol = 2e-5 if prec == "f32" else 1e-4
    for a, r, n in zip(mine, ref, ("d_delta_c", "d_delta_m", "d_adapter")):
        assert _relmax(a.grad, r.grad) < tol, (n, _relmax(a.grad, r.grad))
    for (Ci, Co, k, N, Hh, Ww) in [(64, 16, 1, 40, 16, 16), (32, 48, 3, 25, 24, 16)]:   # 80 / 75 items
        x = seeded_randn((N, Ci, Hh, Ww), name_seed(f"slices.x{Ci}{k}"))
        w = seeded_randn((Co, Ci, k, k), name_seed(f"slices.w{Ci}{k}"), 1.0 / np.sqrt(Ci * k * k))
        b = seeded_randn((Co,), name_seed(f"slices.b{Ci}{k}"), 0.1)
        gy = seeded_randn((N, Co, Hh, Ww), name_seed(f"slices.g{Ci}{k}"))
        ref = [t.clone().requires_grad_(True) for t in (x, w, b)]
        (torch.nn.functional.conv2d(ref[0], ref[1], ref[2], padding=k // 2) * gy).sum().backward()
        mine = [t.cuda().requires_grad_(True) for t in (x, w, b)]
        (vpx.ops.conv2d_same(mine[0], mine[1], mine[2], precision=prec) * gy.cuda()).sum().backward()
        for a, r in zip(mine, ref):
            assert _relmax(a.grad, r.grad) < tol, (prec, Ci, k, _relmax(a.grad, r.grad))


def test_decouple_and_conv2d_vs_golden_and_autograd(vpx):
    """K4 (vpx_decouple_fwd/_bwd) against the reference-generated pin; K5-style conv2d fwd/bwd against torch autograd."""
    from golden_util import seeded_randn
    g = load_golden("decouple_tiny")
    B, Ch, H, W = [int(v) for v in g["shape"]]
    values = {}
    for prec in ("f32", "bf16x3"):   # the tail runs in the arithmetic the caller names (the model's operand mode), f32 by default
        A = seeded_randn((Ch, Ch, 1, 1), name_seed("decouple.adapter"), 1.0 / np.sqrt(Ch)).cuda().requires_grad_(True)
        dc = seeded_randn((B, Ch, H, W), name_seed("decouple.dc")).cuda().requires_grad_(True)
        dm = seeded_randn((B, Ch, H, W), name_seed("decouple.dm")).cuda().requires_grad_(True)
        v = vpx.ops.decouple_term(dc, dm, A, prec) if prec != "f32" else vpx.ops.decouple_term(dc, dm, A)
        values[prec] = float(v)
        assert abs(float(v) - float(g["value"])) < 1e-6
        v.backward()
        assert _relmax(dc.grad, g["d_dc"]) < GRTOL and _relmax(dm.grad, g["d_dm"]) < GRTOL
        assert _relmax(A.grad, g["d_adapter"]) < GRTOL
    with pytest.raises(KeyError):
        vpx.ops.decouple_term(dc, dm, A, "fp64")
    for prec, tol in (("f32", 2e-5), ("bf16x3", 1e-4)):
        for (Ci, Co, k, Hh, Ww) in [(128, 16, 1, 16, 16), (24, 40, 3, 9, 21)]:
            x = seeded_randn((3, Ci, Hh, Ww), name_seed(f"c2g.x{Ci}{k}"))
            w = seeded_randn((Co, Ci, k, k), name_seed(f"c2g.w{Ci}{k}"), 1.0 / np.sqrt(Ci * k * k))
            b = seeded_randn((Co,), name_seed(f"c2g.b{Ci}{k}"), 0.1)
            gy = seeded_randn((3, Co, Hh, Ww), name_seed(f"c2g.g{Ci}{k}"))
            ref = [t.clone().requires_grad_(True) for t in (x, w, b)]
            (torch.nn.functional.conv2d(ref[0], ref[1], ref[2], padding=k // 2) * gy).sum().backward()
            mine = [t.cuda().requires_grad_(True) for t in (x, w, b)]
            y = vpx.ops.conv2d_same(mine[0], mine[1], mine[2], precision=prec)
            (y * gy.cuda()).sum().backward()
            for a, r in zip(mine, ref):
                assert _relmax(a.grad, r.grad) < tol, (prec, Ci, k)


def test_predrnn_layernorm_tiny_vs_golden(vpx):
    """PredRNN-V2 with layer_norm=True (the upstream default, tests/test_impl_match/_predrnn_v2.py:41): forward, loss and
    all gradients (incl. the LayerNorm parameters) against the reference-generated pins."""
    from vp_suite_amd.measure import PredictionLossProvider
    tag, kw, B, Ttot, P = "tiny_ln", gc.PRED_TINY_LN_KW, 2, 6, 3
    g = load_golden(f"predrnn_{tag}")
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn.{tag}.frames")).cuda()
    m = _predrnn(tag, kw).eval()
    pred, ml = m(frames, pred_frames=P)
    assert _relmax(pred, g["eval.pred"]) < 1e-4
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    _, loss = lp.get_losses(pred, frames[:, Ttot - P:])
    loss = loss + ml["ST-LSTM decouple loss"]
    assert abs(float(loss) - float(g["eval.loss"])) < 1e-4 * abs(float(g["eval.loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    flat = np.concatenate([named[k].grad.detach().cpu().numpy().reshape(-1) for k in sorted(named)])
    assert _relmax(flat, g["eval.grads_flat"]) < 2e-4


@pytest.mark.gpu
def test_flat_adam_invalidates_packed_weight_caches(vpx):
    """The ST-LSTM cell re-uses its packed weights while (data_ptr, _version) of the parameters are unchanged; the
    flat-bucket Adam kernel updates parameters outside autograd and must bump the versions: two training steps of a tiny
    PredRNN through FlatAdam equal two steps through torch.optim.Adam."""
    import copy
    from vp_suite_amd.train import FlatAdam
    from vp_suite_amd.models import MODEL_CLASSES
    torch.manual_seed(3)
    kw = dict(img_shape=(1, 16, 16), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[8, 8],
              patch_size=2, filter_size=3)
    m1 = MODEL_CLASSES["predrnn-pp"]("cuda", **kw).to("cuda")
    m2 = copy.deepcopy(m1)
    frames = torch.rand(2, 6, 1, 16, 16, device="cuda")
    o1, o2 = torch.optim.Adam(m1.parameters(), lr=1e-2), FlatAdam.from_module(m2, lr=1e-2)
    for _ in range(2):
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            pred, losses = m(frames, pred_frames=3, train=False)
            (((pred - frames[:, 3:]) ** 2).mean() + sum(losses.values())).backward()
            o.step()
    with torch.no_grad():
        p1, _ = m1(frames, pred_frames=3)
        p2, _ = m2(frames, pred_frames=3)
    assert (p1 - p2).abs().max() < 1e-5


@pytest.mark.gpu
def test_backward_weight_pack_reuse_matches_repacking(vpx, monkeypatch):
    """The ST-LSTM backward keeps its five transposed weight packs in a per-cell workspace while the weights are unchanged
    (VPX_FLAG_WEIGHTS_PACKED); ops.BWD_WEIGHT_PACK_REUSE = False repacks on every call. Two optimizer steps (the weights change in
    between, the packs must follow) give the same model either way."""
    import copy
    from vp_suite_amd.models import MODEL_CLASSES
    torch.manual_seed(5)
    kw = dict(img_shape=(1, 16, 16), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=2, num_hidden=[8, 8],
              patch_size=2, filter_size=3)
    m1 = MODEL_CLASSES["predrnn-pp"]("cuda", **kw).to("cuda")
    m2 = copy.deepcopy(m1)
    frames = torch.rand(2, 6, 1, 16, 16, device="cuda")
    # deterministic mode (no K-split atomics): both models then see bit-identical gradients — Adam turns a last-bit difference
    # of a near-zero gradient into a full lr-sized step (the test was flaky without it, 1 run in 6)
    torch.use_deterministic_algorithms(True)
    try:
        for m, reuse in ((m1, True), (m2, False)):
            monkeypatch.setattr(vpx.ops, "BWD_WEIGHT_PACK_REUSE", reuse)
            o = torch.optim.Adam(m.parameters(), lr=1e-2)
            for _ in range(2):
                o.zero_grad()
                pred, losses = m(frames, pred_frames=3, train=False)
                (((pred - frames[:, 3:]) ** 2).mean() + sum(losses.values())).backward()
                o.step()
    finally:
        torch.use_deterministic_algorithms(False)
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1 - p2).abs().max() < 1e-6, n1


@pytest.mark.gpu
def test_layernorm_cell_pack_reuse_follows_parameter_updates(vpx):
    """The LayerNorm ST-LSTM keeps its five weight packs and the transposed LayerNorm parameters in the cell's workspace while
    (data_ptr, _version) of all of them are unchanged (VPX_FLAG_WEIGHTS_PACKED): an in-place update of a LayerNorm weight or
    of a convolution weight must be seen by the next call."""
    import copy
    from vp_suite_amd.model_blocks import SpatioTemporalLSTMCell
    Cin, Ch, H, W, k, ln, B = gc.STLSTM_CASES["ln"]
    cell = SpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, ln)
    fill_state_dict_(cell, name_seed("stlstm.ln"))
    cell = cell.cuda()
    inp = {n: v.cuda() for n, v in gc.stlstm_inputs("ln", Cin, Ch, H, W, B).items()}
    args = (inp["x"], inp["h"], inp["c"], inp["m"])
    with torch.no_grad():
        o1 = cell(*args)
        o1b = cell(*args)                      # packs and transposed parameters re-used
        assert _relmax(o1b[0], o1[0]) < 1e-6
        cell.conv_h[1].weight.mul_(1.5)        # LayerNorm gamma of conv_h, in place
        cell.conv_x[0].weight.mul_(0.5)        # a convolution weight, in place
        o2 = cell(*args)
        fresh = copy.deepcopy(cell)            # pickled state drops the workspace: packs from scratch
        o3 = fresh(*args)
    assert _relmax(o2[0], o3[0]) < 1e-6 and _relmax(o2[1], o3[1]) < 1e-6
    assert _relmax(o2[0], o1[0]) > 1e-3
