"""Further GPU parity cases: the ndrplz multi-layer sequence block (a4), the ST-LSTM C ABI on reference-layout (NCHW)
buffers, the BASELINE C4 / C5 shapes (128x128x3), the plain-bf16 operand mode with its own stated tolerance."""
import ctypes

import numpy as np
import pytest
import torch

import golden_cases as gc
from golden_util import fill_state_dict_, load_golden, name_seed, seeded_rand, seeded_randn

pytestmark = pytest.mark.gpu


from parity import relmax as _relmax   # max|a - b| / max|b|, recorded (tests/parity.py)


@pytest.mark.parametrize("tag", list(gc.NDRPLZ_SEQ_CASES))
def test_ndrplz_sequence_block_vs_golden(vpx, tag):
    """ConvLSTM_ndrplz: multi-layer, odd frame sizes, batch_first on/off, return_all_layers; forward + all gradients."""
    from vp_suite_amd.model_blocks import ConvLSTM_ndrplz
    Cin, hid, ks, H, W, B, T, bias, batch_first = gc.NDRPLZ_SEQ_CASES[tag]
    g = load_golden(f"ndrplz_seq_{tag}")
    blk = ConvLSTM_ndrplz(Cin, hid, ks, len(hid), batch_first=batch_first, bias=bias, return_all_layers=True)
    fill_state_dict_(blk, name_seed("ndrplz_seq." + tag))
    blk = blk.cuda()
    shape = (B, T, Cin, H, W) if batch_first else (T, B, Cin, H, W)
    x = seeded_rand(shape, name_seed(f"ndrplz_seq.{tag}.x")).cuda().requires_grad_(True)
    outs, states = blk(x)
    for i in range(len(hid)):
        assert _relmax(outs[i], g[f"out{i}"]) < 1e-5 and _relmax(states[i][1], g[f"c{i}"]) < 1e-5
        assert _relmax(states[i][0], g[f"h{i}"]) < 1e-5
    sum((o * seeded_randn(o.shape, name_seed(f"ndrplz_seq.{tag}.g{i}")).cuda()).sum() for i, o in enumerate(outs)).backward()
    assert _relmax(x.grad, g["dx"]) < 5e-5
    for key, prm in blk.named_parameters():
        assert _relmax(prm.grad, g["grad." + key]) < 5e-5, key
    with pytest.raises(NotImplementedError):
        blk(x, hidden_state=[None])


@pytest.mark.parametrize("tag", ["plain", "ln"])
def test_stlstm_c_abi_nchw(vpx, tag):
    """vpx_stlstm_step_fwd/_bwd called directly on reference-layout (NCHW) buffers; "ln" = the LayerNorm variant
    (predrnn.py:24-40), whose NCHW backward form transposes through the workspace."""
    from golden_util import seeded_state_dict
    L = vpx._lib.lib()
    Cin, Ch, H, W, k, ln, B = gc.STLSTM_CASES[tag]
    g = load_golden(f"stlstm_{tag}")
    sd = {kk: v.cuda().contiguous() for kk, v in seeded_state_dict(g, name_seed("stlstm." + tag)).items()}
    inp = {n: v.cuda().contiguous() for n, v in gc.stlstm_inputs(tag, Cin, Ch, H, W, B).items()}
    d = vpx._lib.STLSTMDesc(B, Cin, Ch, H, W, k, int(ln), vpx._lib.LAYOUT_NCHW, vpx._lib.PREC_F32, vpx._lib.FLAG_SAVE_FOR_BWD)
    ln_names = [f"conv_{n}.1.{wb}" for n in "xhmo" for wb in ("weight", "bias")]
    ln_arr = (ctypes.c_void_p * 8)(*[sd[n].data_ptr() for n in ln_names]) if ln else None
    dln = [torch.empty_like(sd[n]) for n in ln_names] if ln else []
    dln_arr = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in dln]) if ln else None
    ws_bytes, rs_bytes = L.vpx_stlstm_workspace_bytes(ctypes.byref(d)), L.vpx_stlstm_reserve_bytes(ctypes.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    rs = torch.empty(rs_bytes, dtype=torch.uint8, device="cuda")
    outs = [torch.empty(B, Ch, H, W, device="cuda") for _ in range(5)]
    p = vpx._lib.ptr
    Ws = [sd["conv_x.0.weight"], sd["conv_h.0.weight"], sd["conv_m.0.weight"], sd["conv_o.0.weight"], sd["conv_last.weight"]]
    rc = L.vpx_stlstm_step_fwd(ctypes.byref(d), p(inp["x"]), p(inp["h"]), p(inp["c"]), p(inp["m"]), *[p(w) for w in Ws],
                               ln_arr, *[p(o) for o in outs], p(rs), rs_bytes, p(ws), ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    for o, n in zip(outs, ("h_new", "c_new", "m_new", "delta_c", "delta_m")):
        assert _relmax(o, g[n]) < 1e-5, n
    grads_in = [inp[n] for n in ("g_h", "g_c", "g_m", "g_dc", "g_dm")]
    dins = [torch.empty_like(inp[n]) for n in ("x", "h", "c", "m")]
    dWs = [torch.empty_like(w) for w in Ws]
    rc = L.vpx_stlstm_step_bwd(ctypes.byref(d), p(inp["x"]), p(inp["h"]), p(inp["c"]), p(inp["m"]), p(outs[1]), p(outs[2]),
                               *[p(w) for w in Ws], ln_arr, p(rs), rs_bytes, *[p(t) for t in grads_in],
                               *[p(t) for t in dins], *[p(t) for t in dWs], dln_arr, p(ws), ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    torch.cuda.synchronize()
    for t, n in zip(dins, ("dx", "dh", "dc", "dm")):
        assert _relmax(t, g[n]) < 5e-5, n
    for t, n in zip(dWs, ("conv_x.0.weight", "conv_h.0.weight", "conv_m.0.weight", "conv_o.0.weight", "conv_last.weight")):
        assert _relmax(t, g["grad." + n]) < 5e-5, n
    for t, n in zip(dln, ln_names):
        assert _relmax(t, g["grad." + n]) < 5e-5, n


def test_ef_convlstm_128x128x3_vs_oracle(vpx):
    """BASELINE config C4 shape (KTH-like 128x128x3, 10 -> 20 shortened to 3 -> 2): product vs the pinned torch restatement."""
    from oracle import torch_ref as tr
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["convlstm-shi"]("cuda", img_shape=(3, 128, 128), action_size=0, tensor_value_range=[0.0, 1.0],
                                      cell_precision="bf16x3")
    fill_state_dict_(m, name_seed("ef.c4"))
    m = m.cuda()
    x = seeded_rand((1, 3, 3, 128, 128), name_seed("ef.c4.x"))
    with torch.no_grad():
        ref = tr.ef_convlstm_forward({k: v.cpu() for k, v in m.state_dict().items()}, x, 2)
        pred, _ = m(x.cuda(), pred_frames=2)
    assert pred.shape == (1, 2, 3, 128, 128) and _relmax(pred, ref) < 1e-4


def test_predrnn_deep_128x128x3_vs_oracle(vpx):
    """BASELINE config C5 shape: 4 ST-LSTM layers, 128x128x3 (patch 4 -> 48 channels on 32x32 maps)."""
    from oracle import torch_ref as tr
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["predrnn-pp"]("cuda", img_shape=(3, 128, 128), action_size=0, tensor_value_range=[0.0, 1.0],
                                    num_layers=4, cell_precision="bf16x3")
    fill_state_dict_(m, name_seed("predrnn.c5"))
    m = m.cuda().eval()
    frames = seeded_rand((1, 5, 3, 128, 128), name_seed("predrnn.c5.x"))
    with torch.no_grad():
        ref, rdec = tr.predrnn_v2_forward({k: v.cpu() for k, v in m.state_dict().items()}, frames, 2, patch_size=4,
                                          num_layers=4)
        pred, ml = m(frames.cuda(), pred_frames=2)
    assert _relmax(pred, ref) < 1e-4
    assert abs(float(ml["ST-LSTM decouple loss"]) - float(rdec)) < 1e-3 * abs(float(rdec))


def test_plain_bf16_mode_has_its_own_tolerance(vpx):
    """VPX_PREC_BF16 (bf16 operands, fp32 accumulate and state): NOT a 1e-4 path. Measured error of bf16 autocast vs the
    fp32 reference is ~2e-3 of the output range; held to 1e-2 here, forward and gradients."""
    Cin, Ch, H, W, k, B, T, _ = gc.HZZONE_CASES["mid"]
    inp = {n: v.cuda() for n, v in gc.hzzone_inputs("mid", Cin, Ch, H, W, k, B, T).items()}
    g = load_golden("hzzone_mid_states")
    out, hT, cT = vpx.ops.convlstm_seq(inp["x"], inp["h0"], inp["c0"], inp["W"], inp["b"], inp["Wci"], inp["Wcf"], inp["Wco"],
                                       seq_len=T, in_channels=Cin, precision="bf16")
    e = _relmax(out, g["out"])
    assert 1e-5 < e < 1e-2, e   # clearly not fp32-exact, clearly within the bf16 budget
    g2 = load_golden("hzzone_tiny_states")
    Cin, Ch, H, W, k, B, T, _ = gc.HZZONE_CASES["tiny"]
    inp = {n: v.cuda() for n, v in gc.hzzone_inputs("tiny", Cin, Ch, H, W, k, B, T).items()}
    lv = {n: inp[n].clone().requires_grad_(True) for n in ("x", "W")}
    out, hT, cT = vpx.ops.convlstm_seq(lv["x"], inp["h0"], inp["c0"], lv["W"], inp["b"], inp["Wci"], inp["Wcf"], inp["Wco"],
                                       seq_len=T, in_channels=Cin, precision="bf16")
    ((out * inp["g_out"]).sum() + (hT * inp["g_hT"]).sum() + (cT * inp["g_cT"]).sum()).backward()
    assert _relmax(lv["W"].grad, g2["dW"]) < 2e-2 and _relmax(lv["x"].grad, g2["dx"]) < 2e-2


def test_conv2d_ex_vs_torch(vpx):
    """vpx_conv2d_ex_fwd / _bwd (strided conv, transposed conv incl. the 4-phase stride-2 form, fused bias + LeakyReLU)
    against torch fp64 / autograd on the CPU."""
    import torch.nn.functional as F
    cases = [  # (transposed, Ci, Co, k, stride, pad, H, W)
        (False, 1, 16, 3, 1, 1, 20, 24), (False, 64, 64, 3, 2, 1, 32, 32), (False, 12, 20, 3, 2, 1, 17, 23),
        (False, 16, 1, 1, 1, 0, 16, 16), (True, 96, 96, 4, 2, 1, 8, 8), (True, 10, 14, 4, 2, 1, 7, 9),
        (True, 64, 16, 3, 1, 1, 16, 16), (True, 6, 5, 5, 2, 2, 6, 7), (False, 8, 8, 4, 2, 1, 12, 12),
        (False, 3, 16, 3, 1, 1, 19, 21), (False, 16, 3, 1, 1, 0, 18, 20),   # few-channel layers: wgrad_small_kernel (with 1->16, 16->1 above)
    ]
    _conv2d_ex_cases(vpx, cases, F)   # backward: vpx_conv2d_ex_bwd (adjoint layer + strided MFMA weight gradient)
    # a layer outside the library's glue backward (kernel smaller than its stride) raises instead of falling back to ATen — in the
    # FORWARD of a call that will need gradients (not after a whole forward pass has been spent); inference runs
    x = torch.rand(1, 4, 8, 8, device="cuda", requires_grad=True)
    w = torch.rand(4, 4, 1, 1, device="cuda", requires_grad=True)
    with pytest.raises(vpx.ops.VpxError, match="unsupported"):
        vpx.ops.conv2d_ex(x, w, None, 2, 0, False, 0.2, "f32")
    with torch.no_grad():
        y = vpx.ops.conv2d_ex(x, w, None, 2, 0, False, 0.2, "f32")
    ref = F.leaky_relu(F.conv2d(x.detach(), w.detach(), None, 2, 0), 0.2)
    assert float((y - ref).abs().max()) < 1e-5


def test_glue_data_gradient_on_convq_vs_torch_and_first_generation(vpx):
    """Round 5: the data gradient of a stage-glue layer whose ADJOINT is a layer the schedule-driven kernel takes (bf16x3, >= 64 output
    channels, a full grid) runs on convq, fed by the LeakyReLU' pass's split-format copy of dy. The EF model's heavy layers
    (ef_conv_lstm.py:36-65) at 24 frames: against torch autograd and against the first-generation launch (VPX_OPT_EXPERIMENT bit 14)."""
    import torch.nn.functional as F
    cases = [  # (transposed, Ci, Co, k, stride, pad, H, W)
        (False, 64, 64, 3, 2, 1, 64, 64), (False, 96, 96, 3, 2, 1, 32, 32), (True, 96, 96, 4, 2, 1, 16, 16), (True, 96, 96, 4, 2, 1, 32, 32),
        (True, 64, 16, 3, 1, 1, 64, 64), (False, 64, 96, 3, 2, 1, 33, 47),
    ]
    L = vpx._lib.lib()
    grads = {}
    for bit in (0, 16384):
        prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, bit)
        try:
            grads[bit] = _conv2d_ex_cases(vpx, cases, F, n=24, precs=(("bf16x3", 5e-5),))
        finally:
            L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
    for a, b in zip(grads[0], grads[16384]):
        assert _relmax(a, b) < 2e-5   # same products, same operand split: fp32 summation order only


def test_glue_weight_gradient_on_split_operands_vs_torch_and_tap_group_kernel(vpx):
    """Round 6: the weight gradient of a stage-glue layer (bf16x3, channel counts in groups of 8) runs per stride residue on wgrad2_kernel's
    glue form — both operands (x and the LeakyReLU'-scaled dy) once more in the split format, staged by LDS-DMA, sub-image addressing in the
    copy's source address. The EF model's layers (ef_conv_lstm.py:36-65) and ragged / narrow ones at 24 frames: against torch autograd and
    against the tap-group kernel on fp32 operands (VPX_OPT_EXPERIMENT bit 29)."""
    import torch.nn.functional as F
    cases = [  # (transposed, Ci, Co, k, stride, pad, H, W)
        (False, 64, 64, 3, 2, 1, 64, 64), (False, 96, 96, 3, 2, 1, 32, 32), (True, 96, 96, 4, 2, 1, 16, 16), (True, 96, 96, 4, 2, 1, 32, 32),
        (True, 64, 16, 3, 1, 1, 64, 64), (False, 64, 96, 3, 2, 1, 33, 47), (False, 24, 40, 3, 1, 1, 19, 21), (True, 8, 136, 4, 2, 1, 9, 7),
        (False, 16, 8, 2, 2, 0, 12, 20),
    ]
    L = vpx._lib.lib()
    grads = {}
    for bit in (0, 1 << 29):
        prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, bit)
        try:
            grads[bit] = []
            _conv2d_ex_cases(vpx, cases, F, n=24, precs=(("bf16x3", 5e-5),), dws=grads[bit])
        finally:
            L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
    for a, b in zip(grads[0], grads[1 << 29]):
        assert _relmax(a, b) < 2e-5   # same products, same operand split: fp32 summation order only


def _conv2d_ex_cases(vpx, cases, F, n=2, precs=(("f32", 2e-5), ("bf16x3", 5e-5)), dws=None):
    dxs = []
    for prec, tol in precs:
        for tr, Ci, Co, k, s, p, H, W in cases:
            tag = f"cex.{tr}.{Ci}.{Co}.{k}.{s}"
            x = seeded_randn((n, Ci, H, W), name_seed(tag + "x"))
            wshape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
            w = seeded_randn(wshape, name_seed(tag + "w"), 1.0 / np.sqrt(Ci * k * k))
            b = seeded_randn((Co,), name_seed(tag + "b"), 0.1)
            if tr:
                ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=s, padding=p)
            else:
                ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)
            ref = F.leaky_relu(ref, 0.2).float()
            lv = [t.cuda().requires_grad_(True) for t in (x, w, b)]
            y = vpx.ops.conv2d_ex(lv[0], lv[1], lv[2], s, p, tr, 0.2, prec)
            assert y.shape == ref.shape, (tag, y.shape, ref.shape)
            assert _relmax(y, ref) < tol, (prec, tag, _relmax(y, ref))
            if True:  # gradients in both operand modes (bf16x3 keeps fp32-level accuracy)
                gy = seeded_randn(ref.shape, name_seed(tag + "g"))
                rl = [t.clone().requires_grad_(True) for t in (x, w, b)]
                rr = F.conv_transpose2d(rl[0], rl[1], rl[2], stride=s, padding=p) if tr else \
                    F.conv2d(rl[0], rl[1], rl[2], stride=s, padding=p)
                # LeakyReLU' taken at the sign of OUR forward output, as the library does: where a pre-activation is within rounding of
                # zero the two forwards may disagree about its sign, and autograd of F.leaky_relu on the reference's own output would
                # then differ by 0.8 * gy at that element — a property of the activation's kink, not of the gradient kernels
                slope_map = torch.where(y.detach().cpu() > 0, 1.0, 0.2)
                (rr * slope_map * gy).sum().backward()
                (y * gy.cuda()).sum().backward()
                for a, r in zip(lv, rl):
                    assert _relmax(a.grad, r.grad) < 1e-4, (prec, tag)
                dxs.append(lv[0].grad.detach().clone())
                if dws is not None:
                    dws.append(lv[1].grad.detach().clone())
    return dxs


def test_edge_shapes_vs_oracle(vpx):
    """Edge cases: single sample / single step, maps smaller than one tile, 1 input channel, channel counts that are not
    multiples of anything, 7x7 and rectangular kernels, inputs longer than seq_len, non-contiguous (NCHW) inputs."""
    from oracle import torch_ref as tr
    cases = [  # (Cin, Ch, H, W, kh, kw, B, T)
        (1, 5, 3, 5, 3, 3, 1, 1), (7, 9, 5, 37, 3, 3, 2, 2), (2, 33, 19, 6, 7, 7, 1, 2), (3, 4, 9, 9, 1, 1, 2, 3),
        (5, 70, 8, 8, 3, 5, 1, 2), (130, 8, 6, 6, 3, 3, 1, 1),
    ]
    for prec, tol in (("f32", 1e-5), ("bf16x3", 3e-5)):
        for (Cin, Ch, H, W, kh, kw, B, T) in cases:
            tag = f"edge.{Cin}.{Ch}.{H}.{kh}"
            Wt = seeded_randn((4 * Ch, Cin + Ch, kh, kw), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * kh * kw))
            b = seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1)
            x = seeded_rand((B, T + 1, Cin, H, W), name_seed(tag + "x"))      # one frame more than seq_len
            h0 = seeded_randn((B, Ch, H, W), name_seed(tag + "h"), 0.5)
            c0 = seeded_randn((B, Ch, H, W), name_seed(tag + "c"), 0.5)
            with torch.no_grad():
                hr, cr = h0, c0
                ref = []
                for t in range(T):
                    hr, cr = tr.convlstm_ndrplz_cell(x[:, t], hr, cr, Wt, b)
                    ref.append(hr)
                ref = torch.stack(ref, 1)
                out, hT, cT = vpx.ops.convlstm_seq(x.cuda(), h0.cuda(), c0.cuda(), Wt.cuda(), b.cuda(), seq_len=T,
                                                   in_channels=Cin, gate_order=vpx._lib.GATE_IFOG, precision=prec)
            assert out.shape == (B, T, Ch, H, W)
            assert _relmax(out, ref) < tol and _relmax(cT, cr) < tol, (prec, tag)
    with pytest.raises(ValueError):   # empty sequence / batch are rejected like any other bad dimension
        vpx.ops.convlstm_seq(torch.zeros(1, 0, 3, 4, 4, device="cuda"), None, None, torch.zeros(16, 7, 3, 3, device="cuda"),
                             None, seq_len=0, in_channels=3)


@pytest.mark.gpu
def test_mse_loss_kernel_vs_oracle(vpx):
    """vpx_mse_loss (value + gradient in one pass) against the restated measure (base_measure.py:55-57), incl. a
    channels-last prediction and an element count that is not a multiple of the vector width."""
    from oracle.torch_ref import mse_measure as mse_ref
    from vp_suite_amd import ops
    for shape, cl in (((3, 4, 1, 16, 16), False), ((2, 3, 3, 9, 7), True), ((4, 10, 1, 64, 64), False)):
        pred = seeded_rand(shape, name_seed("mse.p" + str(shape))).cuda()
        tgt = seeded_rand(shape, name_seed("mse.t" + str(shape))).cuda()
        if cl:
            pred = ops.to_channels_last(pred.flatten(0, 1)).unflatten(0, shape[:2])
        pred.requires_grad_(True)
        loss = ops.mse_loss(pred, tgt, 1.0)
        (loss * 3.0).backward()
        pr = pred.detach().cpu().clone().requires_grad_(True)
        ref = mse_ref(pr, tgt.cpu())
        (ref * 3.0).backward()
        assert abs(float(loss) - float(ref)) < 1e-6 * abs(float(ref)), shape
        assert _relmax(pred.grad, pr.grad) < 1e-6, shape


@pytest.mark.gpu
def test_flat_adam_kernel_vs_oracle_and_torch(vpx):
    """vpx_adam_step over flat buckets against the numpy restatement and torch.optim.Adam (vpsuite.py:353), 3 steps,
    odd length (tail path), grad_scale folding the data-parallel mean."""
    from oracle.torch_ref import adam_step_ref
    from vp_suite_amd import ops
    n = 100003
    p0 = seeded_randn((n,), name_seed("adam.p")).numpy()
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    dp, dm, dv = torch.from_numpy(p0.copy()).cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in (1, 2, 3):
        g = (seeded_randn((n,), name_seed(f"adam.g{step}")).numpy() * (10.0 ** (step - 2))).astype(np.float32)
        p, m, v = adam_step_ref(p, g, m, v, step, 1e-3, grad_scale=0.5)
        ops.adam_step(dp, torch.from_numpy(g).cuda(), dm, dv, step, 1e-3, grad_scale=0.5)
        assert np.abs(dp.cpu().numpy() - p).max() < 3e-7, step
        assert _relmax(dm, m) < 1e-6 and _relmax(dv, v) < 1e-6


@pytest.mark.gpu
def test_flat_adam_drives_train_iter_like_torch_adam(vpx):
    """FlatAdam.from_module as the drop-in for torch.optim.Adam(model.parameters()) inside the reference harness
    (train_iter: zero_grad -> backward -> step, base_model.py:174-176): same parameters after 3 iterations, and the
    golden pins of the reference run (params_after3)."""
    from vp_suite_amd.measure import PredictionLossProvider
    from vp_suite_amd.train import FlatAdam
    import golden_cases as gc
    from test_gpu_models import _ef
    kw, B, T, P = gc.EF_TINY_KW, 2, 3, 2
    g = load_golden("ef_tiny")
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, T + P, c, h, w), name_seed("ef.tiny.frames")).cuda()
    lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
    cfg = {"device": "cuda", "context_frames": T, "pred_frames": P, "val_rec_criterion": "mse"}
    data = {"frames": frames, "actions": torch.zeros(B, T + P - 1, 0)}
    m = _ef(vpx, "tiny", kw)
    opt = FlatAdam.from_module(m, lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=5, factor=0.2, min_lr=1e-6)  # vpsuite.py:354
    for _ in range(3):
        m.train_iter(cfg, [data], opt, lp, epoch=0)
    sched.step(1.0)
    named = dict(m.named_parameters())
    pflat = np.concatenate([named[k].detach().cpu().numpy().reshape(-1) for k in sorted(named)])
    assert np.abs(pflat[::3] - g["params_after3_s3"]).max() < 2e-5
    sd = m.state_dict()  # parameters stay ordinary, individually addressable tensors
    assert all(sd[k].shape == v.shape for k, v in named.items())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "action"])
def test_phydnet_single_step_convlstm_vs_golden(vpx, tag):
    """PhyDNet's SingleStepConvLSTM (phydnet.py:117-175) on the fused cell: per-frame calls with persistent (H, C),
    first_timestep reset, action inflation; outputs, final states and all gradients against the reference fixture."""
    from vp_suite_amd.model_blocks import SingleStepConvLSTM
    isz, idim, hdims, nl, ks, ac, asz, B, steps = gc.PHY_SSC_CASES[tag]
    g = load_golden(f"phy_ssc_{tag}")
    blk = SingleStepConvLSTM(isz, idim, hdims, nl, ks, ac, asz, "cuda")
    fill_state_dict_(blk, name_seed("phy_ssc." + tag))
    blk = blk.cuda()
    frames = seeded_rand((B, steps, idim, *isz), name_seed(f"phy_ssc.{tag}.frames")).cuda().requires_grad_(True)
    actions = seeded_randn((B, steps, max(asz, 1)), name_seed(f"phy_ssc.{tag}.actions"))[:, :, :asz].cuda()
    loss = 0.0
    for t in range(steps):
        (H, C), out = blk(frames[:, t], actions[:, t], first_timestep=(t == 0))
        assert _relmax(out[-1], g[f"out{t}"]) < 1e-5, t
        loss = loss + (out[-1] * seeded_randn(out[-1].shape, name_seed(f"phy_ssc.{tag}.g{t}")).cuda()).sum()
    for j in range(nl):
        assert _relmax(H[j], g[f"H{j}"]) < 1e-5 and _relmax(C[j], g[f"C{j}"]) < 1e-5
    loss.backward()
    assert _relmax(frames.grad, g["dframes"]) < 5e-5
    for key, prm in blk.named_parameters():
        assert _relmax(prm.grad, g["grad." + key]) < 5e-5, key


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "ln"])
def test_action_conditional_stlstm_cell_vs_golden(vpx, tag):
    """ActionConditionalSpatioTemporalLSTMCell (predrnn.py:86-169): six biased convolutions on the library kernel,
    conv_h(h) * conv_a(a) gating; one step forward + all gradients against the reference fixture."""
    from vp_suite_amd.model_blocks import ActionConditionalSpatioTemporalLSTMCell
    Cin, Ch, H, W, k, ln, B = gc.ACSTLSTM_CASES[tag]
    g = load_golden(f"acstlstm_{tag}")
    cell = ActionConditionalSpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, ln)
    fill_state_dict_(cell, name_seed("acstlstm." + tag))
    cell = cell.cuda()
    inp = {n: v.cuda() for n, v in gc.acstlstm_inputs(tag, Cin, Ch, H, W, B).items()}
    lv = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h", "c", "m", "a")}
    outs = cell(lv["x"], lv["h"], lv["c"], lv["m"], lv["a"])
    for o, n in zip(outs, ("h_new", "c_new", "m_new", "delta_c", "delta_m")):
        assert _relmax(o, g[n]) < 1e-5, n
    sum((o * inp[gn]).sum() for o, gn in zip(outs, ("g_h", "g_c", "g_m", "g_dc", "g_dm"))).backward()
    for n in lv:
        assert _relmax(lv[n].grad, g["d" + n]) < 5e-5, n
    for key, prm in cell.named_parameters():
        assert _relmax(prm.grad, g["grad." + key]) < 5e-5, key


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "ln"])
def test_action_conditional_cell_inference_and_frozen_parameters(vpx, tag):
    """vpx_acstlstm_step_fwd without VPX_FLAG_SAVE_FOR_BWD (no reserve: the stage buffers live in the workspace) gives the training-mode
    outputs bit for bit; with some parameters frozen (NULL dparams / dln entries, unwanted data gradients) the remaining parameter
    gradients are unchanged, and bf16x3 stays within the ST-LSTM tolerance of the fixture."""
    from vp_suite_amd.model_blocks import ActionConditionalSpatioTemporalLSTMCell
    Cin, Ch, H, W, k, ln, B = gc.ACSTLSTM_CASES[tag]
    g = load_golden(f"acstlstm_{tag}")
    cell = ActionConditionalSpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, ln)
    fill_state_dict_(cell, name_seed("acstlstm." + tag))
    cell = cell.cuda()
    inp = {n: v.cuda() for n, v in gc.acstlstm_inputs(tag, Cin, Ch, H, W, B).items()}
    args = [inp[n] for n in ("x", "h", "c", "m", "a")]
    with torch.no_grad():
        inf = cell(*args)
    lv = [t.clone().requires_grad_(True) for t in args]
    outs = cell(*lv)
    for a, b in zip(inf, outs):
        assert torch.equal(a, b)
    loss = lambda o: sum((t * inp[gn]).sum() for t, gn in zip(o, ("g_h", "g_c", "g_m", "g_dc", "g_dm")))
    loss(outs).backward()
    full = {key: prm.grad.clone() for key, prm in cell.named_parameters()}
    frozen = [key for i, key in enumerate(full) if i % 3 == 0]
    for key, prm in cell.named_parameters():
        prm.grad = None
        prm.requires_grad_(key not in frozen)
    lv2 = [t.clone().requires_grad_(i in (1, 2, 4)) for i, t in enumerate(args)]
    loss(cell(*lv2)).backward()
    for key, prm in cell.named_parameters():
        assert (prm.grad is None) == (key in frozen), key
        if prm.grad is not None:
            assert torch.equal(prm.grad, full[key]), key
    for i in (1, 2, 4):   # (data gradients of small maps: K-split partial sums meet in float atomics — equal to rounding, not bit for bit)
        assert _relmax(lv2[i].grad, lv[i].grad) < 1e-6
    assert lv2[0].grad is None and lv2[3].grad is None
    cell.precision = "bf16x3"
    with torch.no_grad():
        o3 = cell(*args)
    for o, n in zip(o3, ("h_new", "c_new", "m_new", "delta_c", "delta_m")):
        assert _relmax(o, g[n]) < 1e-4, n


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["full", "noinput"])
def test_trajgru_block_vs_golden(vpx, tag):
    """TrajGRU (traj_gru.py:164-214) as one library-backed sequence op (flow convolutions, bilinear warps, 1x1 ret, GRU
    gates: all HIP, explicit BPTT); encoder form (inputs, zero state) and forecaster form (inputs=None) against the
    reference fixture, incl. gradients; then both operand modes against each other and the block's restrictions."""
    from vp_suite_amd.model_blocks import TrajGRU
    in_c, enc_c, H, W, L, B, T, mode = gc.TRAJGRU_CASES[tag]
    g = load_golden(f"trajgru_{tag}")
    blk = TrajGRU("cuda", in_c, enc_c, H, W, L=L)
    fill_state_dict_(blk, name_seed("trajgru." + tag))
    blk = blk.cuda()
    x = seeded_rand((B, T, in_c, H, W), name_seed(f"trajgru.{tag}.x")).cuda().requires_grad_(True)
    h0 = seeded_randn((B, enc_c, H, W), name_seed(f"trajgru.{tag}.h0"), 0.5).cuda().requires_grad_(True)
    out, hT = blk(x, None, T) if mode == "full" else blk(None, h0, T)
    assert _relmax(out, g["out"]) < 2e-5 and _relmax(hT, g["hT"]) < 2e-5
    (out * seeded_randn(out.shape, name_seed(f"trajgru.{tag}.g")).cuda()).sum().backward()
    if mode == "full":
        assert _relmax(x.grad, g["dx"]) < 1e-4
    else:
        assert _relmax(h0.grad, g["dh0"]) < 1e-4
    for key, prm in blk.named_parameters():
        if "grad." + key in g:
            assert _relmax(prm.grad, g["grad." + key]) < 1e-4, key
    with pytest.raises(ValueError):
        blk(None, None, 1)
    blk.precision = "bf16x3"
    with torch.no_grad():
        o3, _ = blk(x.detach(), None, T) if mode == "full" else blk(None, h0.detach(), T)
    assert _relmax(o3, g["out"]) < 5e-5
    from vp_suite_amd.model_blocks.traj_gru import Activation
    with pytest.raises(NotImplementedError):
        TrajGRU("cuda", in_c, enc_c, H, W, L=L, act_type=Activation("relu"))
    with pytest.raises(NotImplementedError):
        TrajGRU("cuda", in_c, enc_c, H, W, L=L, zoneout=0.1)


@pytest.mark.gpu
def test_trajgru_backward_is_bit_reproducible_on_request(vpx):
    """torch.use_deterministic_algorithms(True): the warp backward scatters 2^40-scaled integers (vpx_trajgru_warp_bwd_det) — two
    runs give bit-identical gradients, equal to the float-atomic path within its summation noise."""
    from vp_suite_amd.model_blocks import TrajGRU
    in_c, enc_c, H, W, L, B, T, mode = gc.TRAJGRU_CASES["full"]
    blk = TrajGRU("cuda", in_c, enc_c, H, W, L=L)
    fill_state_dict_(blk, name_seed("trajgru.full"))
    blk = blk.cuda()
    x0 = seeded_rand((B, T, in_c, H, W), name_seed("trajgru.full.x")).cuda()
    gout = seeded_randn((B, T, enc_c, H, W), name_seed("trajgru.full.g")).cuda()

    def run():
        for p_ in blk.parameters():
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        out, _ = blk(x, None, T)
        (out * gout).sum().backward()
        return [x.grad.clone()] + [p_.grad.clone() for p_ in blk.parameters()]

    ref = run()
    torch.use_deterministic_algorithms(True)
    try:
        a, b = run(), run()
    finally:
        torch.use_deterministic_algorithms(False)
    for u, v_, r in zip(a, b, ref):
        assert torch.equal(u, v_)
        assert _relmax(u, r) < 2e-5


@pytest.mark.gpu
def test_small_grids_fused_and_split_paths_agree(vpx):
    """On nearly-empty grids the ConvLSTM step runs as a K-split convolution + pointwise gates (atomics); with
    torch.use_deterministic_algorithms(True) the same shapes take the fused launch. Both must meet the oracle, the
    deterministic one bit-reproducibly, forward and backward."""
    from oracle import torch_ref as tr
    Cin, Ch, H, W, B, T = 24, 40, 16, 16, 4, 3
    Wt = seeded_randn((4 * Ch, Cin + Ch, 3, 3), name_seed("sg.W"), 1.0 / np.sqrt((Cin + Ch) * 9))
    b = seeded_randn((4 * Ch,), name_seed("sg.b"), 0.1)
    x = seeded_rand((B, T, Cin, H, W), name_seed("sg.x"))
    pw = [seeded_randn((1, Ch, H, W), name_seed("sg.p%d" % i), 0.1) for i in range(3)]
    xr = x.clone().requires_grad_(True)
    Wr = Wt.clone().requires_grad_(True)
    ref, _ = tr.convlstm_hzzone_seq(xr, None, T, Wr, b, *pw)
    (ref ** 2).sum().backward()
    br = b.clone().requires_grad_(True)
    ref2, _ = tr.convlstm_hzzone_seq(x, None, T, Wt, br, *pw)
    (ref2 ** 2).sum().backward()
    runs = {}
    for det in (False, True, True):
        torch.use_deterministic_algorithms(det)
        try:
            xg = x.cuda().requires_grad_(True)
            Wg = Wt.cuda().requires_grad_(True)
            bg = b.cuda().requires_grad_(True)
            pg = [p.cuda().requires_grad_(True) for p in pw]
            out, hT, cT = vpx.ops.convlstm_seq(xg, None, None, Wg, bg, *pg, seq_len=T, in_channels=Cin)
            (out ** 2).sum().backward()
        finally:
            torch.use_deterministic_algorithms(False)
        assert _relmax(out, ref) < 1e-5 and _relmax(xg.grad, xr.grad) < 5e-5 and _relmax(Wg.grad, Wr.grad) < 5e-5, det
        assert _relmax(bg.grad, br.grad) < 5e-5, det
        runs.setdefault(det, []).append([t.detach().clone() for t in (out, xg.grad, Wg.grad, bg.grad, *[p.grad for p in pg])])
    # deterministic mode: bit-identical forward AND every gradient (dx, dW, db, peepholes) — ADVICE r1: the bias gradient
    # used float atomics regardless of the mode
    for a, c in zip(runs[True][0], runs[True][1]):
        assert torch.equal(a, c)


def test_bias_gradients_are_bit_reproducible(vpx):
    """db of the ConvLSTM block and of the glue layers sums in a fixed order in EVERY mode (no atomics): two runs agree
    bit for bit on a shape with several thousand partial rows, and match a float64 column sum."""
    Cin, Ch, H, W, B, T = 16, 32, 32, 32, 6, 3
    Wt = seeded_randn((4 * Ch, Cin + Ch, 3, 3), name_seed("db.W"), 1.0 / np.sqrt((Cin + Ch) * 9)).cuda()
    x = seeded_rand((B, T, Cin, H, W), name_seed("db.x")).cuda()
    got = []
    for _ in range(2):
        bg = seeded_randn((4 * Ch,), name_seed("db.b"), 0.1).cuda().requires_grad_(True)
        out, _, _ = vpx.ops.convlstm_seq(x, None, None, Wt, bg, seq_len=T, in_channels=Cin, gate_order=vpx._lib.GATE_IFOG)
        (out ** 2).sum().backward()
        got.append(bg.grad.clone())
    assert torch.equal(got[0], got[1])
    w = seeded_randn((24, 16, 3, 3), name_seed("db.gw"), 0.1).cuda()
    xs = seeded_randn((10, 16, 40, 40), name_seed("db.gx")).cuda()
    gy = seeded_randn((10, 24, 20, 20), name_seed("db.gy")).cuda()
    res = []
    for _ in range(2):
        bb = torch.zeros(24, device="cuda", requires_grad=True)
        y = vpx.ops.conv2d_ex(xs, w, bb, 2, 1, False, 0.2, "f32")
        (y * gy).sum().backward()
        res.append(bb.grad.clone())
    assert torch.equal(res[0], res[1])
    want = (gy.double() * torch.where(y > 0, 1.0, 0.2).double()).sum(dim=(0, 2, 3))
    assert _relmax(res[0], want.float()) < 1e-5


def test_real_launch_after_dry_run():
    """a dry run (VPX_OPT_DRY_RUN) must leave no state behind that a later real launch depends on — in a FRESH process (no function
    attribute set yet) the same calls run dry first, then for real, on a shape of each fused-cell kernel family (the small-grid kernel
    used to record its LDS attribute as set during the dry run), and the real results match a process that never ran dry."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import sys, torch
sys.path.insert(0, %r)
import vp_suite_amd as v
L = v._lib.lib()
dry_first = sys.argv[1] == "1"
outs = []
for (Cin, Ch, H, W, B) in ((96, 96, 16, 16, 3), (64, 64, 64, 64, 8), (16, 64, 32, 32, 2)):
    g = torch.Generator().manual_seed(Cin + Ch + H)
    x = v.ops.to_channels_last(torch.rand(B, 3, Cin, H, W, generator=g).cuda())
    Wt = (torch.randn(4 * Ch, Cin + Ch, 3, 3, generator=g) * 0.03).cuda()
    b = torch.zeros(4 * Ch).cuda()
    pw = [(torch.randn(1, Ch, H, W, generator=g) * 0.1).cuda() for _ in range(3)]
    for dry in ((1, 0) if dry_first else (0,)):
        L.vpx_set_option(v._lib.OPT_DRY_RUN, dry)
        with torch.no_grad():
            out, hT, cT = v.ops.convlstm_seq(x, None, None, Wt, b, *pw, seq_len=3, in_channels=Cin, precision="bf16x3")
    torch.cuda.synchronize()
    outs.append(float(out.double().abs().sum()) + float(cT.double().abs().sum()))
print("RESULT", " ".join(repr(o) for o in outs))
''' % root
    res = []
    for dry_first in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", script, dry_first], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1])
    assert res[0] == res[1], res
