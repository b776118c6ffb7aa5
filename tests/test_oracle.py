"""Pins the oracle (oracle/vpx_oracle.c and oracle/torch_ref.py) against the golden vectors generated from the real
reference (tools/gen_golden.py). CPU only. Tolerance: the reference's own convention is atol=1e-4, rtol=0
(tests/test_impl_match/_convlstm_hzzone.py:91 etc.); the oracle is held to 2e-6 absolute on O(1) outputs and 1e-5
relative (to the tensor's max) on gradients."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from golden_util import checksum, load_golden, name_seed, seeded_rand, seeded_randn, seeded_state_dict
from oracle import oracle as orc
from oracle import torch_ref as tr

torch.set_num_threads(4)

ATOL = 2e-6
GREL = 1e-5


def _relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


@pytest.mark.parametrize("tag", list(gc.HZZONE_CASES))
@pytest.mark.parametrize("mode", ["full", "states", "noinput"])
def test_c_oracle_hzzone(tag, mode):
    Cin, Ch, H, W, k, B, T, with_grads = gc.HZZONE_CASES[tag]
    inp = {n: v.numpy() for n, v in gc.hzzone_inputs(tag, Cin, Ch, H, W, k, B, T).items()}
    g = load_golden(f"hzzone_{tag}_{mode}")
    assert abs(checksum(inp["W"]) - float(g["chk_W"])) < 1e-9 and abs(checksum(inp["x"]) - float(g["chk_x"])) < 1e-9
    x = None if mode == "noinput" else inp["x"]
    h0 = None if mode == "full" else inp["h0"]
    c0 = None if mode == "full" else inp["c0"]
    dims = dict(B=B, T=T, Cin=Cin, Ch=Ch, H=H, Wd=W, kh=k, kw=k)
    f = orc.convlstm_seq_fwd(x, h0, c0, inp["W"], inp["b"], inp["Wci"], inp["Wcf"], inp["Wco"], save=True, **dims)
    for n in ("out", "hT", "cT"):
        assert np.abs(f[n] - g[n]).max() < ATOL, n
    if with_grads:
        r = orc.convlstm_seq_bwd(x, h0, c0, inp["W"], inp["Wci"], inp["Wcf"], inp["Wco"], f, inp["g_out"],
                                 inp["g_hT"], inp["g_cT"], **dims)
        for n in ("dx", "dh0", "dc0", "dW", "db", "dWci", "dWcf", "dWco"):
            if n in g:
                assert _relmax(r[n], g[n]) < GREL, n


@pytest.mark.parametrize("tag", list(gc.HZZONE_CASES))
def test_torch_ref_hzzone(tag):
    Cin, Ch, H, W, k, B, T, with_grads = gc.HZZONE_CASES[tag]
    inp = gc.hzzone_inputs(tag, Cin, Ch, H, W, k, B, T)
    g = load_golden(f"hzzone_{tag}_states")
    out, (hT, cT) = tr.convlstm_hzzone_seq(inp["x"], (inp["h0"], inp["c0"]), T, inp["W"], inp["b"], inp["Wci"],
                                           inp["Wcf"], inp["Wco"], padding=k // 2)
    assert np.abs(out.numpy() - g["out"]).max() < ATOL
    assert np.abs(cT.numpy() - g["cT"]).max() < ATOL


@pytest.mark.parametrize("tag", list(gc.NDRPLZ_CELL_CASES))
def test_oracles_ndrplz_cell(tag):
    Cin, Ch, H, W, kh, kw, bias, B = gc.NDRPLZ_CELL_CASES[tag]
    inp = gc.ndrplz_cell_inputs(tag, Cin, Ch, H, W, kh, kw, bias, B)
    g = load_golden(f"ndrplz_cell_{tag}")
    b = inp["b"] if bias else None
    # C oracle: one step of the sequence function with T=1, gate order (i,f,o,g), no peephole
    f = orc.convlstm_seq_fwd(inp["x"].numpy()[:, None], inp["h"].numpy(), inp["c"].numpy(), inp["W"].numpy(),
                             None if b is None else b.numpy(), B=B, T=1, Cin=Cin, Ch=Ch, H=H, Wd=W, kh=kh, kw=kw,
                             gate_order=orc.GATE_IFOG, save=True)
    assert np.abs(f["hT"] - g["h_next"]).max() < ATOL and np.abs(f["cT"] - g["c_next"]).max() < ATOL
    r = orc.convlstm_seq_bwd(inp["x"].numpy()[:, None], inp["h"].numpy(), inp["c"].numpy(), inp["W"].numpy(), None,
                             None, None, f, None, inp["g_h"].numpy(), inp["g_c"].numpy(), B=B, T=1, Cin=Cin, Ch=Ch,
                             H=H, Wd=W, kh=kh, kw=kw, gate_order=orc.GATE_IFOG)
    assert _relmax(r["dx"][:, 0], g["dx"]) < GREL and _relmax(r["dh0"], g["dh"]) < GREL
    assert _relmax(r["dc0"], g["dc"]) < GREL and _relmax(r["dW"], g["dW"]) < GREL
    if bias:
        assert _relmax(r["db"], g["db"]) < GREL
    # torch restatement
    hn, cn = tr.convlstm_ndrplz_cell(inp["x"], inp["h"], inp["c"], inp["W"], b)
    assert np.abs(hn.numpy() - g["h_next"]).max() < ATOL and np.abs(cn.numpy() - g["c_next"]).max() < ATOL


@pytest.mark.parametrize("tag", list(gc.NDRPLZ_SEQ_CASES))
def test_torch_ref_ndrplz_seq(tag):
    Cin, hid, ks, H, W, B, T, bias, batch_first = gc.NDRPLZ_SEQ_CASES[tag]
    g = load_golden(f"ndrplz_seq_{tag}")
    sd = seeded_state_dict(g, name_seed("ndrplz_seq." + tag))
    shape = (B, T, Cin, H, W) if batch_first else (T, B, Cin, H, W)
    x = seeded_rand(shape, name_seed(f"ndrplz_seq.{tag}.x"))
    params = [(sd[f"cell_list.{i}.conv.weight"], sd.get(f"cell_list.{i}.conv.bias")) for i in range(len(hid))]
    outs, states = tr.convlstm_ndrplz_seq(x, params, batch_first=batch_first)
    for i in range(len(hid)):
        assert np.abs(outs[i].numpy() - g[f"out{i}"]).max() < ATOL
        assert np.abs(states[i][1].numpy() - g[f"c{i}"]).max() < ATOL


@pytest.mark.parametrize("tag", list(gc.STLSTM_CASES))
def test_oracles_stlstm(tag):
    Cin, Ch, H, W, k, ln, B = gc.STLSTM_CASES[tag]
    g = load_golden(f"stlstm_{tag}")
    sd = seeded_state_dict(g, name_seed("stlstm." + tag))
    assert abs(checksum(sd["conv_x.0.weight"]) - float(g["chk_wx"])) < 1e-9
    inp = gc.stlstm_inputs(tag, Cin, Ch, H, W, B)
    names = ("h_new", "c_new", "m_new", "delta_c", "delta_m")
    # torch restatement incl. gradients
    leaves = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h", "c", "m")}
    psd = {kk: v.clone().requires_grad_(True) for kk, v in sd.items()}
    outs = tr.stlstm_cell(leaves["x"], leaves["h"], leaves["c"], leaves["m"], psd, "", ln)
    for o, n in zip(outs, names):
        assert np.abs(o.detach().numpy() - g[n]).max() < ATOL, n
    sum((o * inp[gn]).sum() for o, gn in zip(outs, ("g_h", "g_c", "g_m", "g_dc", "g_dm"))).backward()
    for n in ("x", "h", "c", "m"):
        assert _relmax(leaves[n].grad.numpy(), g["d" + n]) < GREL, n
    for kk in sd:
        assert _relmax(psd[kk].grad.numpy(), g["grad." + kk]) < GREL, kk
    # C oracle forward
    lnd = None
    if ln:
        lnd = {f"{a}_{b}": sd[f"conv_{a}.1.{'weight' if b == 'g' else 'bias'}"].numpy() for a in "xhmo" for b in "gb"}
    co = orc.stlstm_step_fwd(inp["x"].numpy(), inp["h"].numpy(), inp["c"].numpy(), inp["m"].numpy(),
                             sd["conv_x.0.weight"].numpy(), sd["conv_h.0.weight"].numpy(),
                             sd["conv_m.0.weight"].numpy(), sd["conv_o.0.weight"].numpy(),
                             sd["conv_last.weight"].numpy(), lnd, B=B, Cin=Cin, Ch=Ch, H=H, Wd=W, k=k)
    for o, n in zip(co, names):
        assert np.abs(o - g[n]).max() < ATOL, n


def test_oracles_decouple():
    g = load_golden("decouple_tiny")
    B, Ch, H, W = [int(v) for v in g["shape"]]
    A = seeded_randn((Ch, Ch, 1, 1), name_seed("decouple.adapter"), 1.0 / np.sqrt(Ch)).requires_grad_(True)
    dc = seeded_randn((B, Ch, H, W), name_seed("decouple.dc")).requires_grad_(True)
    dm = seeded_randn((B, Ch, H, W), name_seed("decouple.dm")).requires_grad_(True)
    v = tr.decouple_term(dc, dm, A)
    assert abs(float(v) - float(g["value"])) < 1e-6
    v.backward()
    assert _relmax(dc.grad.numpy(), g["d_dc"]) < GREL and _relmax(A.grad.numpy(), g["d_adapter"]) < GREL
    vc = orc.decouple_fwd(dc.detach().numpy(), dm.detach().numpy(), A.detach().numpy(), B=B, Ch=Ch, HW=H * W)
    assert abs(vc - float(g["value"])) < 1e-6


@pytest.mark.parametrize("tag,kw,B,T,P", [("tiny", gc.EF_TINY_KW, 2, 3, 2), ("tiny3", gc.EF_TINY3_KW, 2, 2, 3)])
def test_torch_ref_ef(tag, kw, B, T, P):
    g = load_golden(f"ef_{tag}")
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(g, name_seed("ef." + tag)).items()}
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, T + P, c, h, w), name_seed(f"ef.{tag}.frames"))
    assert abs(checksum(frames) - float(g["chk_frames"])) < 1e-9
    pred = tr.ef_convlstm_forward(sd, frames[:, :T], P)
    assert np.abs(pred.detach().numpy() - g["pred"]).max() < ATOL
    loss = tr.mse_measure(pred, frames[:, T:])
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    flat = np.concatenate([sd[k].grad.numpy().reshape(-1) for k in sorted(sd)])
    assert _relmax(flat, g["grads_flat"]) < GREL


@pytest.mark.parametrize("tag", ["full_c1"])
def test_torch_ref_ef_full(tag):
    g = load_golden(f"ef_{tag}")
    sd = seeded_state_dict(g, name_seed("ef." + tag))
    x = seeded_rand((1, 10, 1, 64, 64), name_seed(f"ef.{tag}.x"))
    with torch.no_grad():
        pred = tr.ef_convlstm_forward(sd, x, 10)
    assert np.abs(pred[:, :, :, ::4, ::4].numpy() - g["pred_slice"]).max() < 1e-5


@pytest.mark.parametrize("tag,kw,B,Ttot,P", [("tiny", gc.PRED_TINY_KW, 2, 5, 2), ("tiny_ln", gc.PRED_TINY_LN_KW, 2, 6, 3)])
def test_torch_ref_predrnn(tag, kw, B, Ttot, P):
    g = load_golden(f"predrnn_{tag}")
    sd = {k: v.requires_grad_(True) for k, v in seeded_state_dict(g, name_seed("predrnn." + tag)).items()}
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn.{tag}.frames"))
    common = dict(patch_size=kw["patch_size"], num_layers=kw["num_layers"], layer_norm=kw.get("layer_norm", False))
    pred, dec = tr.predrnn_v2_forward(sd, frames, P, **common)
    assert np.abs(pred.detach().numpy() - g["eval.pred"]).max() < 1e-5
    assert abs(float(dec) - float(g["eval.decouple"])) < 1e-4 * abs(float(g["eval.decouple"]))
    loss = tr.mse_measure(pred, frames[:, Ttot - P:]) + dec
    assert abs(float(loss) - float(g["eval.loss"])) < 1e-5 * abs(float(g["eval.loss"]))
    loss.backward()
    flat = np.concatenate([sd[k].grad.numpy().reshape(-1) for k in sorted(sd)])
    assert _relmax(flat, g["eval.grads_flat"]) < 5e-5
    with torch.no_grad():
        pred_r, dec_r = tr.predrnn_v2_forward(sd, frames, P, reverse_scheduled_sampling=True, **common)
        assert np.abs(pred_r.numpy() - g["rss_eval.pred"]).max() < 1e-5
        # train=True: mask[b, j] = 1 where random_flip < eta (predrnn_v2.py:294-297)
        eta = float(g["train.eta_after"])
        ps = kw["patch_size"]
        mask = torch.zeros(B, P - 1, ps * ps * c, h // ps, w // ps)
        mask[torch.from_numpy(g["train.random_flip"]) < eta] = 1
        pred_t, _ = tr.predrnn_v2_forward(sd, frames, P, mask_true=mask, **common)
        assert np.abs(pred_t.numpy() - g["train.pred"]).max() < 1e-5


def test_adam_restatement_matches_torch_adam():
    """Pins oracle.torch_ref.adam_step_ref to PyTorch's own Adam (the optimizer the reference constructs, vpsuite.py:353)."""
    import numpy as np, torch
    from oracle.torch_ref import adam_step_ref
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal(1000).astype(np.float32)
    grads = [rng.standard_normal(1000).astype(np.float32) * s for s in (1.0, 1e-3, 30.0)]
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-3)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step, g in enumerate(grads, 1):
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v = adam_step_ref(p, g, m, v, step, 1e-3)
        assert np.abs(p - tp.detach().numpy()).max() < 2e-7, step


@pytest.mark.parametrize("tag", ["plain", "action"])
def test_torch_ref_phydnet_single_step_convlstm(tag):
    """oracle restatement of PhyDNet's SingleStepConvLSTM against the reference-generated fixture."""
    import torch
    import golden_cases as gc
    from golden_util import load_golden, name_seed, seeded_rand, seeded_randn, seeded_state_dict
    from oracle.torch_ref import phydnet_single_step_convlstm
    isz, idim, hdims, nl, ks, ac, asz, B, steps = gc.PHY_SSC_CASES[tag]
    g = load_golden(f"phy_ssc_{tag}")
    sd = seeded_state_dict(g, name_seed("phy_ssc." + tag))
    frames = seeded_rand((B, steps, idim, *isz), name_seed(f"phy_ssc.{tag}.frames"))
    actions = seeded_randn((B, steps, max(asz, 1)), name_seed(f"phy_ssc.{tag}.actions"))[:, :, :asz]
    outs, H, C = phydnet_single_step_convlstm(sd, frames, actions, hdims, ac)
    for t in range(steps):
        assert (outs[t] - torch.from_numpy(g[f"out{t}"])).abs().max() < 1e-6
    for j in range(nl):
        assert (C[j] - torch.from_numpy(g[f"C{j}"])).abs().max() < 1e-6


@pytest.mark.parametrize("tag", ["plain", "ln"])
def test_torch_ref_action_conditional_stlstm(tag):
    """oracle restatement of the action-conditional ST-LSTM cell against the reference-generated fixture."""
    import torch
    import golden_cases as gc
    from golden_util import load_golden, name_seed, seeded_state_dict
    from oracle.torch_ref import acstlstm_cell
    Cin, Ch, H, W, k, ln, B = gc.ACSTLSTM_CASES[tag]
    g = load_golden(f"acstlstm_{tag}")
    sd = seeded_state_dict(g, name_seed("acstlstm." + tag))
    inp = gc.acstlstm_inputs(tag, Cin, Ch, H, W, B)
    outs = acstlstm_cell(inp["x"], inp["h"], inp["c"], inp["m"], inp["a"], sd, layer_norm=ln)
    for o, n in zip(outs, ("h_new", "c_new", "m_new", "delta_c", "delta_m")):
        assert (o - torch.from_numpy(g[n])).abs().max() < 2e-6, n


@pytest.mark.parametrize("tag", ["full", "noinput"])
def test_torch_ref_trajgru(tag):
    """oracle restatement of the TrajGRU block against the reference-generated fixture."""
    import torch
    import golden_cases as gc
    from golden_util import load_golden, name_seed, seeded_rand, seeded_randn, seeded_state_dict
    from oracle.torch_ref import trajgru_seq
    in_c, enc_c, H, W, L, B, T, mode = gc.TRAJGRU_CASES[tag]
    g = load_golden(f"trajgru_{tag}")
    sd = seeded_state_dict(g, name_seed("trajgru." + tag))
    x = seeded_rand((B, T, in_c, H, W), name_seed(f"trajgru.{tag}.x"))
    h0 = seeded_randn((B, enc_c, H, W), name_seed(f"trajgru.{tag}.h0"), 0.5)
    out, hT = trajgru_seq(x, None, T, sd, L) if mode == "full" else trajgru_seq(None, h0, T, sd, L)
    assert (out - torch.from_numpy(g["out"])).abs().max() < 2e-6
    assert (hT - torch.from_numpy(g["hT"])).abs().max() < 2e-6


def test_torch_ref_ef_trajgru_model():
    """oracle/torch_ref.ef_trajgru_forward (EF skeleton + trajgru_seq) pinned to the reference's EF_TrajGRU fixture: prediction,
    loss and every parameter gradient."""
    import golden_cases as gc
    from oracle import torch_ref as tr
    g = load_golden("ef_trajgru_tiny")
    sd = {k: v.clone().requires_grad_(True) for k, v in seeded_state_dict(g, name_seed("ef_trajgru.tiny")).items()}
    c, h, w = gc.EF_TRAJGRU_TINY_KW["img_shape"]
    frames = seeded_rand((2, 5, c, h, w), name_seed("ef_trajgru.tiny.frames"))
    pred = tr.ef_trajgru_forward(sd, frames[:, :3], 2, L=3)
    assert np.abs(pred.detach().numpy() - g["pred"]).max() < 2e-6
    loss = tr.mse_measure(pred, frames[:, 3:])
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    for k, v in sd.items():
        if "grad." + k in g:
            ref = g["grad." + k]
            assert np.abs(v.grad.numpy() - ref).max() < 1e-5 * (np.abs(ref).max() + 1e-30) + 1e-9, k


@pytest.mark.parametrize("tag", ["residual", "plain", "ln"])
def test_torch_ref_predrnn_action_model(tag):
    """oracle/torch_ref.predrnn_v2_action_forward pinned to the reference's action-conditional PredRNN-V2 fixtures."""
    import golden_cases as gc
    from oracle import torch_ref as tr
    g = load_golden(f"predrnn_action_{tag}")
    kw, extra = gc.PRED_ACTION_KW, gc.PRED_ACTION_CASES[tag]
    sd = seeded_state_dict(g, name_seed("predrnn_action." + tag))
    c, h, w = kw["img_shape"]
    frames = seeded_rand((2, 5, c, h, w), name_seed(f"predrnn_action.{tag}.frames"))
    actions = seeded_randn((2, 5, kw["action_size"]), name_seed(f"predrnn_action.{tag}.actions"))
    with torch.no_grad():
        pred, dec = tr.predrnn_v2_action_forward(sd, frames, actions, 2, patch_size=kw["patch_size"], num_layers=kw["num_layers"],
                                                 layer_norm=extra.get("layer_norm", False), residual=extra["residual_on_action_conv"])
    assert np.abs(pred.numpy() - g["pred"]).max() < 5e-6
    assert abs(float(dec) - float(g["decouple"])) < 1e-5 * abs(float(g["decouple"]))
