#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference via tools/ref_shim.py).

Runs only in the build container (the reference never travels). The fixtures are data: shapes/seeds + the
reference's outputs. Inputs / parameters / cotangents are regenerated from seeds (tests/golden_util.py).

Reference entry points exercised:
  a2  model_blocks.ConvLSTM.forward            vp_suite/model_blocks/conv_lstm_hzzone.py:38-70
  a3  ConvLSTMCell.forward                     vp_suite/model_blocks/conv_lstm_ndrplz.py:28-43
  a4  ConvLSTM_ndrplz.forward                  vp_suite/model_blocks/conv_lstm_ndrplz.py:92-131
  a5  SpatioTemporalLSTMCell.forward           vp_suite/model_blocks/predrnn.py:57-83
  K4  decoupling tail                          vp_suite/models/predrnn_v2.py:197-211
  a7  EF_ConvLSTM forward / train_iter         vp_suite/models/precipitation_nowcasting/ef_blocks.py:184-187
  a8  PredRNN_V2 forward / train_iter          vp_suite/models/predrnn_v2.py:131-230, 319-365
  a9  VPModel.train_iter / eval_iter + MSE     vp_suite/base/base_model.py:148-216, base_measure.py:57

usage: python tools/gen_golden.py [--only PREFIX]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_shim  # noqa: E402
from golden_cases import *  # noqa: E402,F401,F403  (case tables + seeded input builders)
from golden_util import (GOLDEN_DIR, checksum, fill_state_dict_, name_seed, seeded_rand,  # noqa: E402
                         seeded_randn)

torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _save(name, **arrays):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def _sd_meta(module):
    """key/shape table of a module's state_dict, so tests can rebuild seeded parameters without the reference."""
    import json
    sd = module.state_dict()
    return dict(sd_keys=np.array(sorted(sd.keys())), sd_shapes=np.array(json.dumps({k: list(v.shape) for k, v in sd.items()})))


# --------------------------------------------------------------------------------------------------------------
# a2: hzzone ConvLSTM block
# --------------------------------------------------------------------------------------------------------------
def gen_hzzone():
    from vp_suite.model_blocks import ConvLSTM
    for tag, (Cin, Ch, H, W, k, B, T, with_grads) in HZZONE_CASES.items():
        inp = hzzone_inputs(tag, Cin, Ch, H, W, k, B, T)
        for mode in ("full", "states", "noinput"):
            blk = ConvLSTM("cpu", Cin, Ch, H, W, k, 1, k // 2)
            with torch.no_grad():
                blk._conv.weight.copy_(inp["W"])
                blk._conv.bias.copy_(inp["b"])
                blk.Wci.copy_(inp["Wci"])
                blk.Wcf.copy_(inp["Wcf"])
                blk.Wco.copy_(inp["Wco"])
            x = inp["x"].clone().requires_grad_(True)
            h0 = inp["h0"].clone().requires_grad_(True)
            c0 = inp["c0"].clone().requires_grad_(True)
            if mode == "full":
                out, (hT, cT) = blk(x, None, T)
            elif mode == "states":
                out, (hT, cT) = blk(x, (h0, c0), T)
            else:
                out, (hT, cT) = blk(None, (h0, c0), T)
            arrays = dict(shape=np.array([Cin, Ch, H, W, k, B, T]), out=_np(out), hT=_np(hT), cT=_np(cT),
                          chk_W=checksum(inp["W"]), chk_x=checksum(inp["x"]), chk_c0=checksum(inp["c0"]))
            if with_grads:
                loss = (out * inp["g_out"]).sum() + (hT * inp["g_hT"]).sum() + (cT * inp["g_cT"]).sum()
                loss.backward()
                arrays.update(dW=_np(blk._conv.weight.grad), db=_np(blk._conv.bias.grad),
                              dWci=_np(blk.Wci.grad), dWcf=_np(blk.Wcf.grad), dWco=_np(blk.Wco.grad))
                if mode != "noinput":
                    arrays["dx"] = _np(x.grad)
                if mode != "full":
                    arrays["dh0"] = _np(h0.grad)
                    arrays["dc0"] = _np(c0.grad)
            _save(f"hzzone_{tag}_{mode}", **arrays)


# --------------------------------------------------------------------------------------------------------------
# a3/a4: ndrplz ConvLSTM cell + multi-layer block
# --------------------------------------------------------------------------------------------------------------
def gen_ndrplz():
    from vp_suite.model_blocks.conv_lstm_ndrplz import ConvLSTM as ConvLSTM_ndrplz
    from vp_suite.model_blocks.conv_lstm_ndrplz import ConvLSTMCell
    for tag, (Cin, Ch, H, W, kh, kw, bias, B) in NDRPLZ_CELL_CASES.items():
        inp = ndrplz_cell_inputs(tag, Cin, Ch, H, W, kh, kw, bias, B)
        cell = ConvLSTMCell(Cin, Ch, (kh, kw), bias)
        with torch.no_grad():
            cell.conv.weight.copy_(inp["W"])
            if bias:
                cell.conv.bias.copy_(inp["b"])
        x = inp["x"].clone().requires_grad_(True)
        h = inp["h"].clone().requires_grad_(True)
        c = inp["c"].clone().requires_grad_(True)
        hn, cn = cell(x, (h, c))
        ((hn * inp["g_h"]).sum() + (cn * inp["g_c"]).sum()).backward()
        arrays = dict(shape=np.array([Cin, Ch, H, W, kh, kw, int(bias), B]), h_next=_np(hn), c_next=_np(cn),
                      dx=_np(x.grad), dh=_np(h.grad), dc=_np(c.grad), dW=_np(cell.conv.weight.grad),
                      chk_W=checksum(inp["W"]), chk_x=checksum(inp["x"]))
        if bias:
            arrays["db"] = _np(cell.conv.bias.grad)
        _save(f"ndrplz_cell_{tag}", **arrays)

    for tag, (Cin, hid, ks, H, W, B, T, bias, batch_first) in NDRPLZ_SEQ_CASES.items():
        blk = ConvLSTM_ndrplz(Cin, hid, ks, len(hid), batch_first=batch_first, bias=bias, return_all_layers=True)
        fill_state_dict_(blk, name_seed("ndrplz_seq." + tag))
        shape = (B, T, Cin, H, W) if batch_first else (T, B, Cin, H, W)
        x = seeded_rand(shape, name_seed(f"ndrplz_seq.{tag}.x")).requires_grad_(True)
        outs, states = blk(x)
        loss = sum((o * seeded_randn(o.shape, name_seed(f"ndrplz_seq.{tag}.g{i}"))).sum() for i, o in enumerate(outs))
        loss.backward()
        arrays = dict(dx=_np(x.grad), chk_x=checksum(x))
        for i, o in enumerate(outs):
            arrays[f"out{i}"] = _np(o)
            arrays[f"h{i}"] = _np(states[i][0])
            arrays[f"c{i}"] = _np(states[i][1])
        for key, prm in blk.named_parameters():
            arrays["grad." + key] = _np(prm.grad)
        arrays.update(_sd_meta(blk))
        _save(f"ndrplz_seq_{tag}", **arrays)


# --------------------------------------------------------------------------------------------------------------
# a5: ST-LSTM cell step; K4: decoupling tail
# --------------------------------------------------------------------------------------------------------------
def gen_phydnet_ssc():
    """rank-4 widening: PhyDNet's single-step ConvLSTM stack (vp_suite/model_blocks/phydnet.py:117-175), rolled out for
    a few frames incl. the first_timestep reset and the action-inflation concat; gradients w.r.t. frames and weights."""
    from vp_suite.model_blocks.phydnet import SingleStepConvLSTM
    for tag, (isz, idim, hdims, nl, ks, ac, asz, B, steps) in PHY_SSC_CASES.items():
        blk = SingleStepConvLSTM(isz, idim, hdims, nl, ks, ac, asz, "cpu")
        fill_state_dict_(blk, name_seed("phy_ssc." + tag))
        frames = seeded_rand((B, steps, idim, *isz), name_seed(f"phy_ssc.{tag}.frames")).requires_grad_(True)
        actions = seeded_randn((B, steps, max(asz, 1)), name_seed(f"phy_ssc.{tag}.actions"))[:, :, :asz]
        loss, arrays = 0.0, {}
        for t in range(steps):
            (H, C), out = blk(frames[:, t], actions[:, t], first_timestep=(t == 0))
            gt = seeded_randn(out[-1].shape, name_seed(f"phy_ssc.{tag}.g{t}"))
            loss = loss + (out[-1] * gt).sum()
            arrays[f"out{t}"] = _np(out[-1])
        for j in range(nl):
            arrays[f"H{j}"] = _np(H[j]); arrays[f"C{j}"] = _np(C[j])
        loss.backward()
        arrays["dframes"] = _np(frames.grad)
        for key, prm in blk.named_parameters():
            arrays["grad." + key] = _np(prm.grad)
        arrays.update(_sd_meta(blk))
        _save(f"phy_ssc_{tag}", **arrays)


def gen_stlstm():
    from vp_suite.model_blocks import SpatioTemporalLSTMCell
    for tag, (Cin, Ch, H, W, k, ln, B) in STLSTM_CASES.items():
        cell = SpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, ln)
        fill_state_dict_(cell, name_seed("stlstm." + tag))
        inp = stlstm_inputs(tag, Cin, Ch, H, W, B)
        x, h, c, m = (inp[n].clone().requires_grad_(True) for n in ("x", "h", "c", "m"))
        outs = cell(x, h, c, m)
        loss = sum((o * inp[g]).sum() for o, g in zip(outs, ("g_h", "g_c", "g_m", "g_dc", "g_dm")))
        loss.backward()
        arrays = dict(shape=np.array([Cin, Ch, H, W, k, int(ln), B]),
                      h_new=_np(outs[0]), c_new=_np(outs[1]), m_new=_np(outs[2]), delta_c=_np(outs[3]),
                      delta_m=_np(outs[4]), dx=_np(x.grad), dh=_np(h.grad), dc=_np(c.grad), dm=_np(m.grad),
                      chk_x=checksum(inp["x"]), chk_wx=checksum(cell.conv_x[0].weight))
        for key, prm in cell.named_parameters():
            arrays["grad." + key] = _np(prm.grad)
        arrays.update(_sd_meta(cell))
        _save(f"stlstm_{tag}", **arrays)


def gen_acstlstm():
    """rank-3 widening: the action-conditional ST-LSTM cell (vp_suite/model_blocks/predrnn.py:86-169), one step + grads."""
    from vp_suite.model_blocks.predrnn import ActionConditionalSpatioTemporalLSTMCell
    for tag, (Cin, Ch, H, W, k, ln, B) in ACSTLSTM_CASES.items():
        cell = ActionConditionalSpatioTemporalLSTMCell(Cin, Ch, H, W, k, 1, ln)
        fill_state_dict_(cell, name_seed("acstlstm." + tag))
        inp = acstlstm_inputs(tag, Cin, Ch, H, W, B)
        lv = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h", "c", "m", "a")}
        outs = cell(lv["x"], lv["h"], lv["c"], lv["m"], lv["a"])
        sum((o * inp[g]).sum() for o, g in zip(outs, ("g_h", "g_c", "g_m", "g_dc", "g_dm"))).backward()
        arrays = dict(h_new=_np(outs[0]), c_new=_np(outs[1]), m_new=_np(outs[2]), delta_c=_np(outs[3]), delta_m=_np(outs[4]))
        for n in lv:
            arrays["d" + n] = _np(lv[n].grad)
        for key, prm in cell.named_parameters():
            arrays["grad." + key] = _np(prm.grad)
        arrays.update(_sd_meta(cell))
        _save(f"acstlstm_{tag}", **arrays)


def gen_trajgru():
    """rank-4 widening: TrajGRU block (vp_suite/model_blocks/traj_gru.py:164-214), encoder form (inputs + zero state)
    and forecaster form (inputs=None, given state); outputs + gradients."""
    from vp_suite.model_blocks.traj_gru import TrajGRU
    for tag, (in_c, enc_c, H, W, L, B, T, mode) in TRAJGRU_CASES.items():
        blk = TrajGRU("cpu", in_c, enc_c, H, W, L=L)
        fill_state_dict_(blk, name_seed("trajgru." + tag))
        x = seeded_rand((B, T, in_c, H, W), name_seed(f"trajgru.{tag}.x")).requires_grad_(True)
        h0 = seeded_randn((B, enc_c, H, W), name_seed(f"trajgru.{tag}.h0"), 0.5).requires_grad_(True)
        out, hT = blk(x, None, T) if mode == "full" else blk(None, h0, T)
        (out * seeded_randn(out.shape, name_seed(f"trajgru.{tag}.g"))).sum().backward()
        arrays = dict(out=_np(out), hT=_np(hT))
        if mode == "full":
            arrays["dx"] = _np(x.grad)
        else:
            arrays["dh0"] = _np(h0.grad)
        for key, prm in blk.named_parameters():
            if prm.grad is not None:
                arrays["grad." + key] = _np(prm.grad)
        arrays.update(_sd_meta(blk))
        _save(f"trajgru_{tag}", **arrays)


def gen_predrnn_action():
    """Action-conditional PredRNN-V2 (predrnn_v2.py:62-121, 143-149, 178-221 + ActionConditionalSpatioTemporalLSTMCell):
    eval forward (reverse scheduled sampling is forced by the model), decoupling loss, loss and all gradients."""
    from vp_suite.models import MODEL_CLASSES
    from golden_cases import PRED_ACTION_KW, PRED_ACTION_CASES
    PR = MODEL_CLASSES["predrnn-pp"]
    B, Ttot, P = 2, 5, 2
    c, h, w = PRED_ACTION_KW["img_shape"]
    for tag, extra in PRED_ACTION_CASES.items():
        frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn_action.{tag}.frames"))
        actions = seeded_randn((B, Ttot, PRED_ACTION_KW["action_size"]), name_seed(f"predrnn_action.{tag}.actions"))
        model = PR("cpu", **PRED_ACTION_KW, **extra)
        fill_state_dict_(model, name_seed("predrnn_action." + tag))
        model.eval()
        pred, ml = model(frames, pred_frames=P, actions=actions)
        _, loss = _loss_provider(c).get_losses(pred, frames[:, Ttot - P:])
        loss = loss + ml["ST-LSTM decouple loss"]
        model.zero_grad()
        loss.backward()
        _, gflat = _flat_sorted({k: p.grad for k, p in model.named_parameters()})
        arrays = dict(pred=_np(pred), decouple=_np(ml["ST-LSTM decouple loss"]), loss=_np(loss), grads_flat=gflat,
                      n_params=np.array(sum(p.numel() for p in model.parameters())))
        arrays.update(_sd_meta(model))
        _save(f"predrnn_action_{tag}", **arrays)


def gen_ef_trajgru():
    """EF_TrajGRU tiny model (ef_traj_gru.py + ef_blocks.py): forward, loss and every parameter gradient."""
    from vp_suite.models import MODEL_CLASSES
    from golden_cases import EF_TRAJGRU_TINY_KW as kw
    B, T, P = 2, 3, 2
    m = MODEL_CLASSES["trajgru"]("cpu", **kw)
    fill_state_dict_(m, name_seed("ef_trajgru.tiny"))
    c, h, w = kw["img_shape"]
    frames = seeded_rand((B, T + P, c, h, w), name_seed("ef_trajgru.tiny.frames"))
    pred, _ = m(frames[:, :T], pred_frames=P)
    loss = ((pred - frames[:, T:]) ** 2).sum(dim=(4, 3, 2)).mean(dim=1).mean(dim=0)
    loss.backward()
    arrays = dict(pred=_np(pred), loss=_np(loss))
    for key, prm in m.named_parameters():
        if prm.grad is not None:   # the top forecaster block gets no input: its i2h / i2f_conv1 parameters are unused
            arrays["grad." + key] = _np(prm.grad)
    arrays.update(_sd_meta(m))
    _save("ef_trajgru_tiny", **arrays)


def gen_decouple():
    import torch.nn.functional as F
    B, Ch, H, W = 2, 8, 6, 5
    adapter = torch.nn.Conv2d(Ch, Ch, 1, 1, 0, bias=False)
    with torch.no_grad():
        adapter.weight.copy_(seeded_randn((Ch, Ch, 1, 1), name_seed("decouple.adapter"), 1.0 / np.sqrt(Ch)))
    dc = seeded_randn((B, Ch, H, W), name_seed("decouple.dc")).requires_grad_(True)
    dm = seeded_randn((B, Ch, H, W), name_seed("decouple.dm")).requires_grad_(True)
    # verbatim op sequence of predrnn_v2.py:197-198 + 210-211 (restated, not copied: one layer, one step)
    a = F.normalize(adapter(dc).view(B, Ch, -1), dim=2)
    b = F.normalize(adapter(dm).view(B, Ch, -1), dim=2)
    val = torch.mean(torch.abs(torch.cosine_similarity(a, b, dim=2)))
    val.backward()
    _save("decouple_tiny", shape=np.array([B, Ch, H, W]), value=_np(val), d_dc=_np(dc.grad), d_dm=_np(dm.grad),
          d_adapter=_np(adapter.weight.grad))


# --------------------------------------------------------------------------------------------------------------
# a7/a9: EF_ConvLSTM model, harness pins
# --------------------------------------------------------------------------------------------------------------
def _flat_sorted(named):
    keys = sorted(named.keys())
    return keys, np.concatenate([_np(named[k]).reshape(-1) for k in keys])


def _loss_provider(img_c):
    from vp_suite.measure.loss_provider import PredictionLossProvider
    return PredictionLossProvider({"device": "cpu", "losses_and_scales": {"mse": 1.0}, "img_c": img_c})


def gen_ef():
    from vp_suite.models import MODEL_CLASSES
    EF = MODEL_CLASSES["convlstm-shi"]
    for tag, kw, B, T, P in (("tiny", EF_TINY_KW, 2, 3, 2), ("tiny3", EF_TINY3_KW, 2, 2, 3)):
        model = EF("cpu", **kw)
        fill_state_dict_(model, name_seed("ef." + tag))
        c, h, w = kw["img_shape"]
        frames = seeded_rand((B, T + P, c, h, w), name_seed(f"ef.{tag}.frames"))
        x, target = frames[:, :T], frames[:, T:]
        pred, ml = model(x, pred_frames=P)
        assert ml is None
        pred1 = model.pred_1(x)
        _, loss = _loss_provider(c).get_losses(pred, target)
        model.zero_grad()
        loss.backward()
        keys, gflat = _flat_sorted({k: p.grad for k, p in model.named_parameters()})
        arrays = dict(pred=_np(pred), pred1=_np(pred1), loss=_np(loss), grads_flat=gflat,
                      n_params=np.array(sum(p.numel() for p in model.parameters())),
                      chk_frames=checksum(frames), chk_w=checksum(model.encoder.rnn1._conv.weight))
        # harness pin (a9): 3 Adam steps through the reference's own train_iter, then eval_iter
        model = EF("cpu", **kw)
        fill_state_dict_(model, name_seed("ef." + tag))
        cfg = {"device": "cpu", "context_frames": T, "pred_frames": P, "val_rec_criterion": "mse"}
        data = {"frames": frames, "actions": torch.zeros(B, T + P - 1, 0)}
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        lp = _loss_provider(c)
        for step in (1, 2, 3):
            model.train_iter(cfg, [data], opt, lp, epoch=0)
            if step in (1, 3):
                _, pflat = _flat_sorted(dict(model.named_parameters()))
                arrays[f"params_after{step}_s3"] = pflat[::3]
                arrays[f"params_after{step}_chk"] = np.float64(checksum(pflat))
        all_losses, indicator = model.eval_iter(cfg, [data], lp)
        arrays["eval_mse_after3"] = np.float32(all_losses["mse"])
        arrays["eval_indicator_after3"] = _np(indicator)
        arrays.update(_sd_meta(model))
        _save(f"ef_{tag}", **arrays)

    # full-size default model (BASELINE config C1/C2 shape), weights seeded by name; slices + checksum only
    for tag, c in (("full_c1", 1), ("full_c3", 3)):
        model = EF("cpu", img_shape=(c, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0])
        fill_state_dict_(model, name_seed("ef." + tag))
        B, T, P = 1, 10, 10
        x = seeded_rand((B, T, c, 64, 64), name_seed(f"ef.{tag}.x"))
        with torch.no_grad():
            pred, _ = model(x, pred_frames=P)
        _save(f"ef_{tag}", pred_slice=_np(pred[:, :, :, ::4, ::4]), pred_chk=np.float64(checksum(pred)),
              pred_absmean=np.float64(pred.abs().mean().item()),
              n_params=np.array(sum(p.numel() for p in model.parameters())), chk_x=checksum(x),
              chk_w=checksum(model.encoder.rnn1._conv.weight), **_sd_meta(model))


# --------------------------------------------------------------------------------------------------------------
# a8: PredRNN_V2
# --------------------------------------------------------------------------------------------------------------
def gen_predrnn():
    from vp_suite.models import MODEL_CLASSES
    PR = MODEL_CLASSES["predrnn-pp"]
    for tag, kw, B, Ttot, P in (("tiny", PRED_TINY_KW, 2, 5, 2), ("tiny_ln", PRED_TINY_LN_KW, 2, 6, 3)):
        c, h, w = kw["img_shape"]
        frames = seeded_rand((B, Ttot, c, h, w), name_seed(f"predrnn.{tag}.frames"))
        arrays = dict(chk_frames=checksum(frames))
        for variant, extra in (("eval", {}), ("rss_eval", {"reverse_scheduled_sampling": True})):
            model = PR("cpu", **kw, **extra)
            fill_state_dict_(model, name_seed("predrnn." + tag))
            model.eval()
            pred, ml = model(frames, pred_frames=P)
            arrays[f"{variant}.pred"] = _np(pred)
            arrays[f"{variant}.decouple"] = _np(ml["ST-LSTM decouple loss"])
            if variant == "eval":
                arrays["eval.pred1"] = _np(model.pred_1(frames[:, :Ttot - P + 1]))
                target = frames[:, Ttot - P:]
                _, loss = _loss_provider(c).get_losses(pred, target)
                loss = loss + ml["ST-LSTM decouple loss"]
                model.zero_grad()
                loss.backward()
                _, gflat = _flat_sorted({k: p.grad for k, p in model.named_parameters()})
                arrays["eval.loss"] = _np(loss)
                arrays["eval.grads_flat"] = gflat
                arrays["n_params"] = np.array(sum(p.numel() for p in model.parameters()))
                arrays["chk_w"] = checksum(model.cell_list[0].conv_x[0].weight)

        # train=True with scheduled sampling: the mask is torch.rand(B, P-1) < eta (predrnn_v2.py:294-297); store the
        # random_flip draws so the test can inject the identical mask.
        model = PR("cpu", **kw)
        fill_state_dict_(model, name_seed("predrnn." + tag))
        model.sampling_eta = 0.5
        torch.manual_seed(1234)
        flips = torch.rand(B, P - 1)
        torch.manual_seed(1234)
        pred, ml = model(frames, pred_frames=P, train=True)
        arrays["train.random_flip"] = _np(flips)
        arrays["train.eta_after"] = np.float64(model.sampling_eta)
        arrays["train.pred"] = _np(pred)
        arrays["train.decouple"] = _np(ml["ST-LSTM decouple loss"])

        # harness pin: the reference's own PredRNN train_iter (forward + reversed forward, averaged) with
        # scheduled_sampling disabled is not runnable (predrnn_v2.py:285-287 returns a tuple) -> pin with sampling on
        # and eta forced to 0 by a large sampling_changing_rate (mask all zeros, deterministic).
        model = PR("cpu", **kw, sampling_changing_rate=2.0)
        fill_state_dict_(model, name_seed("predrnn." + tag))
        cfg = {"device": "cpu", "context_frames": Ttot - P, "pred_frames": P, "val_rec_criterion": "mse"}
        data = {"frames": frames, "actions": torch.zeros(B, Ttot - 1, 0)}
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        lp = _loss_provider(c)
        for step in (1, 2):
            model.train_iter(cfg, [data], opt, lp, epoch=0)
            if step in (1, 2):
                _, pflat = _flat_sorted(dict(model.named_parameters()))
                arrays[f"params_after{step}_s5"] = pflat[::5]
                arrays[f"params_after{step}_chk"] = np.float64(checksum(pflat))
        arrays["training_iteration_after2"] = np.array(model.training_iteration)
        arrays["sampling_eta_after2"] = np.float64(model.sampling_eta)
        arrays.update(_sd_meta(model))
        _save(f"predrnn_{tag}", **arrays)

    # full-size default model (BASELINE config C3): slices + checksum
    model = PR("cpu", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0])
    fill_state_dict_(model, name_seed("predrnn.full_c1"))
    model.eval()
    frames = seeded_rand((1, 20, 1, 64, 64), name_seed("predrnn.full_c1.frames"))
    with torch.no_grad():
        pred, ml = model(frames, pred_frames=10)
    _save("predrnn_full_c1", pred_slice=_np(pred[:, :, :, ::4, ::4]), pred_chk=np.float64(checksum(pred)),
          decouple=_np(ml["ST-LSTM decouple loss"]), chk_frames=checksum(frames),
          n_params=np.array(sum(p.numel() for p in model.parameters())),
          chk_w=checksum(model.cell_list[0].conv_x[0].weight), **_sd_meta(model))


GENERATORS = {"hzzone": gen_hzzone, "ndrplz": gen_ndrplz, "stlstm": gen_stlstm, "decouple": gen_decouple,
              "ef": gen_ef, "predrnn": gen_predrnn, "phy_ssc": gen_phydnet_ssc, "acstlstm": gen_acstlstm, "trajgru": gen_trajgru, "ef_trajgru": gen_ef_trajgru, "predrnn_action": gen_predrnn_action}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    ref_shim.load_reference()
    for name, fn in GENERATORS.items():
        if args.only and not name.startswith(args.only):
            continue
        print(f"[{name}]")
        fn()
