"""Plain-PyTorch (CPU, fp32, autograd) functional restatement of the reference hot path — the floating-point oracle
for model-level forward/backward checks and the "PyTorch CPU path" baseline timed by bench.py.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package never imports it. Parity status: PINNED against tests/golden/ (generated from the real reference by
tools/gen_golden.py; checked in tests/test_oracle.py).

Everything here is a pure function of a `state_dict`-like mapping {reference parameter name: tensor} so that it shares
no code with the product's nn.Module classes. Reference lines restated:
  convlstm_hzzone_seq      vp_suite/model_blocks/conv_lstm_hzzone.py:38-70
  convlstm_ndrplz_cell     vp_suite/model_blocks/conv_lstm_ndrplz.py:28-43
  convlstm_ndrplz_seq      vp_suite/model_blocks/conv_lstm_ndrplz.py:92-131
  stlstm_cell              vp_suite/model_blocks/predrnn.py:57-83 (LayerNorm variant :24-40)
  decouple_term            vp_suite/models/predrnn_v2.py:197-198, 209-211
  ef_convlstm_forward      vp_suite/models/precipitation_nowcasting/ef_blocks.py:67-82, 100-114, 184-187 with the
                           layer table of ef_conv_lstm.py:31-108
  predrnn_v2_forward       vp_suite/models/predrnn_v2.py:131-250
  mse_measure              vp_suite/base/base_measure.py:57 (sum c,h,w -> mean t -> mean b)
"""
import torch
import torch.nn.functional as F


def convlstm_hzzone_seq(inputs, states, seq_len, W, b, Wci, Wcf, Wco, padding=1):
    Ch = W.shape[0] // 4
    Cin = W.shape[1] - Ch
    if states is None:
        B = inputs.shape[0]
        Hs, Ws = Wci.shape[-2:]
        h = torch.zeros(B, Ch, Hs, Ws, dtype=W.dtype)
        c = torch.zeros(B, Ch, Hs, Ws, dtype=W.dtype)
    else:
        h, c = states
        B = h.shape[0]
    outs = []
    for t in range(seq_len):
        x = torch.zeros(B, Cin, *h.shape[-2:], dtype=W.dtype) if inputs is None else inputs[:, t]
        pre = F.conv2d(torch.cat([x, h], dim=1), W, b, stride=1, padding=padding)
        i, f, g, o = torch.chunk(pre, 4, dim=1)
        i = torch.sigmoid(i + Wci * c)
        f = torch.sigmoid(f + Wcf * c)
        c = f * c + i * torch.tanh(g)
        o = torch.sigmoid(o + Wco * c)
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1), (h, c)


def convlstm_ndrplz_cell(x, h, c, W, b):
    kh, kw = W.shape[-2:]
    Ch = W.shape[0] // 4
    pre = F.conv2d(torch.cat([x, h], dim=1), W, b, padding=(kh // 2, kw // 2))
    ci, cf, co, cg = torch.split(pre, Ch, dim=1)
    i, f, o, g = torch.sigmoid(ci), torch.sigmoid(cf), torch.sigmoid(co), torch.tanh(cg)
    c_next = f * c + i * g
    return o * torch.tanh(c_next), c_next


def convlstm_ndrplz_seq(x, layer_params, batch_first=True):
    """layer_params: list of (W, b|None). x [B,T,C,H,W] (or [T,B,...] if not batch_first).
    Returns (list of per-layer outputs [B,T,Ch,H,W], list of (h,c))."""
    if not batch_first:
        x = x.permute(1, 0, 2, 3, 4)
    B, T, _, H, Wd = x.shape
    cur = x
    outs, states = [], []
    for W, b in layer_params:
        Ch = W.shape[0] // 4
        h = torch.zeros(B, Ch, H, Wd, dtype=x.dtype)
        c = torch.zeros(B, Ch, H, Wd, dtype=x.dtype)
        seq = []
        for t in range(T):
            h, c = convlstm_ndrplz_cell(cur[:, t], h, c, W, b)
            seq.append(h)
        cur = torch.stack(seq, dim=1)
        outs.append(cur)
        states.append((h, c))
    return outs, states


def stlstm_cell(x, h, c, m, p, prefix="", layer_norm=False, forget_bias=1.0):
    """p: mapping with keys prefix+'conv_x.0.weight' ... 'conv_last.weight' (+ 'conv_x.1.weight/bias' LayerNorm)."""
    def conv(name, inp):
        w = p[f"{prefix}{name}.0.weight"]
        y = F.conv2d(inp, w, None, stride=1, padding=w.shape[-1] // 2)
        if layer_norm:
            g, b = p[f"{prefix}{name}.1.weight"], p[f"{prefix}{name}.1.bias"]
            y = F.layer_norm(y, list(g.shape), g, b)
        return y
    Ch = h.shape[1]
    xc, hc, mc = conv("conv_x", x), conv("conv_h", h), conv("conv_m", m)
    i_x, f_x, g_x, i_xp, f_xp, g_xp, o_x = torch.split(xc, Ch, dim=1)
    i_h, f_h, g_h, o_h = torch.split(hc, Ch, dim=1)
    i_m, f_m, g_m = torch.split(mc, Ch, dim=1)
    i_t = torch.sigmoid(i_x + i_h)
    f_t = torch.sigmoid(f_x + f_h + forget_bias)
    g_t = torch.tanh(g_x + g_h)
    delta_c = i_t * g_t
    c_new = f_t * c + delta_c
    i_p = torch.sigmoid(i_xp + i_m)
    f_p = torch.sigmoid(f_xp + f_m + forget_bias)
    g_p = torch.tanh(g_xp + g_m)
    delta_m = i_p * g_p
    m_new = f_p * m + delta_m
    mem = torch.cat((c_new, m_new), 1)
    o_t = torch.sigmoid(o_x + o_h + conv("conv_o", mem))
    h_new = o_t * torch.tanh(F.conv2d(mem, p[f"{prefix}conv_last.weight"]))
    return h_new, c_new, m_new, delta_c, delta_m


def decouple_term(delta_c, delta_m, adapter_w):
    B, Ch = delta_c.shape[:2]
    a = F.normalize(F.conv2d(delta_c, adapter_w).view(B, Ch, -1), dim=2)
    b = F.normalize(F.conv2d(delta_m, adapter_w).view(B, Ch, -1), dim=2)
    return torch.mean(torch.abs(torch.cosine_similarity(a, b, dim=2)))


def mse_measure(pred, target):
    return ((pred - target) ** 2).sum(dim=(4, 3, 2)).mean(dim=1).mean(dim=0)


# EF_ConvLSTM layer table (ef_conv_lstm.py:36-65): (kernel, stride, pad)
EF_ENC_CONV = [(3, 1, 1), (3, 2, 1), (3, 2, 1)]
EF_DEC_CONV = [(4, 2, 1), (4, 2, 1), (3, 1, 1)]


def ef_convlstm_forward(sd, x, pred_frames):
    """sd: reference-named state dict of EF_ConvLSTM (default layer table, any channel widths)."""
    def rnn(prefix, inputs, states, T):
        return convlstm_hzzone_seq(inputs, states, T, sd[prefix + "._conv.weight"], sd[prefix + "._conv.bias"],
                                   sd[prefix + ".Wci"], sd[prefix + ".Wcf"], sd[prefix + ".Wco"], padding=1)
    return _ef_forward(sd, x, pred_frames, rnn)


def ef_trajgru_forward(sd, x, pred_frames, L=13):
    """sd: reference-named state dict of EF_TrajGRU (ef_traj_gru.py; default layer table, any channel widths, one L)."""
    def rnn(prefix, inputs, states, T):
        p = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}
        return trajgru_seq(inputs, states, T, p, L)
    return _ef_forward(sd, x, pred_frames, rnn)


def _ef_forward(sd, x, pred_frames, rnn):
    """Encoder-Forecaster skeleton (ef_blocks.py:52-128): rnn(prefix, inputs, states, T) -> (outputs, state)."""
    B, T = x.shape[:2]
    cur = x
    enc_states = []
    for n in range(3):
        k, s, p = EF_ENC_CONV[n]
        name = f"encoder.stage{n + 1}.conv{n + 1}_leaky_1"
        y = F.conv2d(cur.reshape(-1, *cur.shape[2:]), sd[name + ".weight"], sd[name + ".bias"], stride=s, padding=p)
        y = F.leaky_relu(y, 0.2)
        cur = y.reshape(B, T, *y.shape[1:])
        cur, st = rnn(f"encoder.rnn{n + 1}", cur, None, T)
        enc_states.append(st)
    cur = None
    for idx, n in enumerate((3, 2, 1)):  # forecaster.rnn3 first (ef_blocks.py:109-114)
        cur, _ = rnn(f"forecaster.rnn{n}", cur, enc_states[n - 1], pred_frames)
        k, s, p = EF_DEC_CONV[idx]
        name = f"forecaster.stage{n}.deconv{idx + 1}_leaky_1"
        y = F.conv_transpose2d(cur.reshape(-1, *cur.shape[2:]), sd[name + ".weight"], sd[name + ".bias"], stride=s,
                               padding=p)
        y = F.leaky_relu(y, 0.2)
        if n == 1:  # identity, then conv3_3 (1x1)
            y = F.conv2d(y, sd["forecaster.stage1.conv3_3.weight"], sd["forecaster.stage1.conv3_3.bias"])
        cur = y.reshape(B, pred_frames, *y.shape[1:])
    return cur


def reshape_patch(x, ps):
    b, t, c, h, w = x.shape
    x = x.view(b, t, c, h // ps, ps, w // ps, ps).permute(0, 1, 4, 6, 2, 3, 5).contiguous()
    return x.view(b, t, ps * ps * c, h // ps, w // ps)


def reshape_patch_back(xp, ps):
    b, t, cpp, hp, wp = xp.shape
    c = cpp // (ps * ps)
    xp = xp.reshape(b, t, ps, ps, c, hp, wp).permute(0, 1, 4, 5, 2, 6, 3)
    return xp.reshape(b, t, c, hp * ps, wp * ps)


def predrnn_v2_forward(sd, frames, pred_frames, *, patch_size, num_layers, layer_norm=False, mask_true=None,
                       reverse_scheduled_sampling=False, decoupling_loss_scale=100.0):
    """Non-action-conditional PredRNN-V2 forward. mask_true: [B, n_mask, patch_c, h_, w_] (zeros in eval for the
    standard schedule; predrnn_v2.py:300-309)."""
    B, Ttot = frames.shape[:2]
    ctx = Ttot - pred_frames
    xp = reshape_patch(frames, patch_size)
    hs = [None] * num_layers
    Hh, Ww = xp.shape[-2:]
    nh = [sd[f"cell_list.{i}.conv_h.0.weight"].shape[1] for i in range(num_layers)]
    h_t = [torch.zeros(B, nh[i], Hh, Ww) for i in range(num_layers)]
    c_t = [torch.zeros(B, nh[i], Hh, Ww) for i in range(num_layers)]
    memory = torch.zeros(B, nh[0], Hh, Ww)
    if mask_true is None:
        n_mask = ctx + pred_frames - 2 if reverse_scheduled_sampling else pred_frames - 1
        mask_true = torch.zeros(B, n_mask, *xp.shape[2:])
        if reverse_scheduled_sampling:
            mask_true[:, :ctx - 1] = 1
    first_blend = 1 if reverse_scheduled_sampling else ctx
    x_gen = None
    frames_out, dec = [], []
    for t in range(Ttot - 1):
        if t < first_blend:
            net = xp[:, t]
        else:
            mk = mask_true[:, t - first_blend]
            net = mk * xp[:, t] + (1 - mk) * x_gen
        for i in range(num_layers):
            inp = net if i == 0 else h_t[i - 1]
            h_t[i], c_t[i], memory, dc, dm = stlstm_cell(inp, h_t[i], c_t[i], memory, sd, f"cell_list.{i}.", layer_norm)
            dec.append(decouple_term(dc, dm, sd["adapter.weight"]))
        x_gen = F.conv2d(h_t[num_layers - 1], sd["conv_last.weight"])
        frames_out.append(x_gen)
    pred = reshape_patch_back(torch.stack(frames_out[-pred_frames:], dim=1), patch_size)
    return pred, decoupling_loss_scale * torch.mean(torch.stack(dec, dim=0))


def adam_step_ref(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """numpy restatement of one torch.optim.Adam update (torch/optim/adam.py::_single_tensor_adam, amsgrad=False,
    maximize=False; pinned torch==1.10.1, requirements.txt:13 — the formula is unchanged in the torch 2.10 here), the
    optimizer vp_suite/vpsuite.py:353 builds. float32 arithmetic like PyTorch; returns (p, m, v)."""
    import numpy as np
    f32 = np.float32
    g = (g.astype(f32) * f32(grad_scale)).astype(f32)
    if weight_decay:
        g = g + f32(weight_decay) * p
    m = (f32(beta1) * m + f32(1.0 - beta1) * g).astype(f32)
    v = (f32(beta2) * v + f32(1.0 - beta2) * g * g).astype(f32)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (np.sqrt(v) / f32(np.sqrt(bc2)) + f32(eps)).astype(f32)
    p = (p - f32(lr / bc1) * (m / denom)).astype(f32)
    return p, m, v


def phydnet_single_step_convlstm(sd, frames, actions, hidden_dims, action_conditional):
    """Functional restatement of SingleStepConvLSTM (vp_suite/model_blocks/phydnet.py:117-175) rolled over the frames
    of `frames` [B,T,C,H,W] (first_timestep on t = 0): per step, optional action inflation + concat (:150-152), then
    the ConvLSTMCell stack (:154-158, cells of conv_lstm_ndrplz.py:28-43). Returns (outputs per step, H list, C list)."""
    import torch
    B, T, _, Hh, Ww = frames.shape
    n_layers = len(hidden_dims)
    H = [frames.new_zeros(B, hd, Hh, Ww) for hd in hidden_dims]
    C = [frames.new_zeros(B, hd, Hh, Ww) for hd in hidden_dims]
    outs = []
    for t in range(T):
        inp = frames[:, t]
        if action_conditional:
            inp = torch.cat([inp, actions[:, t].unsqueeze(-1).unsqueeze(-1).expand(-1, -1, Hh, Ww)], dim=-3)
        for j in range(n_layers):
            H[j], C[j] = convlstm_ndrplz_cell(inp if j == 0 else H[j - 1], H[j], C[j], sd[f"cell_list.{j}.conv.weight"],
                                              sd[f"cell_list.{j}.conv.bias"])
        outs.append(H[-1])
    return outs, H, C


def acstlstm_cell(x, h, c, m, a, p, layer_norm=False, forget_bias=1.0):
    """Functional restatement of ActionConditionalSpatioTemporalLSTMCell.forward (vp_suite/model_blocks/predrnn.py:139-169):
    six biased convolutions, conv_h(h) * conv_a(a) gating (:144), optional LayerNorm([C,H,W]) after conv_x/h/a/m/o."""
    import torch
    import torch.nn.functional as F
    nh = c.shape[1]

    def conv(name, t):
        w = p[f"{name}.0.weight"]
        y = F.conv2d(t, w, p[f"{name}.0.bias"], padding=w.shape[-1] // 2)
        if layer_norm:
            y = F.layer_norm(y, y.shape[1:], p[f"{name}.1.weight"], p[f"{name}.1.bias"])
        return y
    xc, hc, ac, mc = conv("conv_x", x), conv("conv_h", h), conv("conv_a", a), conv("conv_m", m)
    i_x, f_x, g_x, i_xp, f_xp, g_xp, o_x = torch.split(xc, nh, dim=1)
    i_h, f_h, g_h, o_h = torch.split(hc * ac, nh, dim=1)
    i_m, f_m, g_m = torch.split(mc, nh, dim=1)
    i_t, f_t, g_t = torch.sigmoid(i_x + i_h), torch.sigmoid(f_x + f_h + forget_bias), torch.tanh(g_x + g_h)
    delta_c = i_t * g_t
    c_new = f_t * c + delta_c
    i_p, f_p, g_p = torch.sigmoid(i_xp + i_m), torch.sigmoid(f_xp + f_m + forget_bias), torch.tanh(g_xp + g_m)
    delta_m = i_p * g_p
    m_new = f_p * m + delta_m
    mem = torch.cat((c_new, m_new), 1)
    o_t = torch.sigmoid(o_x + o_h + conv("conv_o", mem))
    h_new = o_t * torch.tanh(F.conv2d(mem, p["conv_last.weight"], p["conv_last.bias"]))
    return h_new, c_new, m_new, delta_c, delta_m


def predrnn_v2_action_forward(sd, frames, actions, pred_frames, *, patch_size, num_layers, layer_norm=False, residual=True,
                              decoupling_loss_scale=100.0):
    """Action-conditional PredRNN-V2 forward in eval mode (vp_suite/models/predrnn_v2.py:131-230 with action_conditional =
    conv_actions_on_input = reverse_scheduled_sampling = True, as the model forces them, :64-67): stride-2 5x5 convolutions
    of frame and action map (:178-188), action-conditional cells (:190-201), stride-2 transposed convolutions with
    `output_size` (:212-218), reverse-scheduled-sampling test mask (:300-309: ground truth for the context, own output after)."""
    import torch
    import torch.nn.functional as F
    B, Ttot = frames.shape[:2]
    ctx = Ttot - pred_frames
    xp = reshape_patch(frames, patch_size)
    ph, pw = xp.shape[-2:]
    ap = actions[..., None, None].expand(-1, -1, -1, ph, pw)
    k = sd["conv_input1.weight"].shape[-1]
    nh = [sd[f"cell_list.{i}.conv_h.0.weight"].shape[1] for i in range(num_layers)]
    rh, rw = ph // 4, pw // 4
    h_t = [torch.zeros(B, nh[i], rh, rw) for i in range(num_layers)]
    c_t = [torch.zeros(B, nh[i], rh, rw) for i in range(num_layers)]
    memory = torch.zeros(B, nh[0], rh, rw)
    x_gen, outs, dec = None, [], []

    def deconv(t, name, size):
        w = sd[name + ".weight"]
        base = [(t.shape[-2 + d] - 1) * 2 - 2 * (k // 2) + k for d in (0, 1)]
        return F.conv_transpose2d(t, w, stride=2, padding=k // 2, output_padding=(size[0] - base[0], size[1] - base[1]))
    for t in range(Ttot - 1):
        net = xp[:, t] if (t < 1 or t - 1 < ctx - 1) else x_gen   # mask = 1 for the first ctx-1 blended steps, 0 after
        s1 = net.shape[-2:]
        net = in1 = F.conv2d(net, sd["conv_input1.weight"], stride=2, padding=k // 2)
        s2 = net.shape[-2:]
        net = in2 = F.conv2d(net, sd["conv_input2.weight"], stride=2, padding=k // 2)
        act = F.conv2d(F.conv2d(ap[:, t], sd["action_conv_input1.weight"], stride=2, padding=k // 2),
                       sd["action_conv_input2.weight"], stride=2, padding=k // 2)
        for i in range(num_layers):
            inp = net if i == 0 else h_t[i - 1]
            p = {kk[len(f"cell_list.{i}."):]: v for kk, v in sd.items() if kk.startswith(f"cell_list.{i}.")}
            h_t[i], c_t[i], memory, dc, dm = acstlstm_cell(inp, h_t[i], c_t[i], memory, act, p, layer_norm)
            dec.append(decouple_term(dc, dm, sd["adapter.weight"]))
        top = h_t[num_layers - 1]
        x_gen = deconv(top + in2 if residual else top, "deconv_output1", s2)
        x_gen = deconv(x_gen + in1 if residual else x_gen, "deconv_output2", s1)
        outs.append(x_gen)
    pred = reshape_patch_back(torch.stack(outs[-pred_frames:], dim=1), patch_size)
    return pred, decoupling_loss_scale * torch.mean(torch.stack(dec, dim=0))


def trajgru_seq(inputs, states, seq_len, p, L, slope=0.2):
    """Functional restatement of TrajGRU.forward (vp_suite/model_blocks/traj_gru.py:164-214; zoneout 0): i2h over all
    frames, per step flow generation (:134-146), L bilinear warps of h (:148-162, default grid_sample alignment), 1x1 ret
    conv and the GRU gate arithmetic (:190-203)."""
    import torch
    import torch.nn.functional as F
    nf = p["ret.weight"].shape[0] // 3

    def conv(name, t, pad):
        return F.conv2d(t, p[name + ".weight"], p[name + ".bias"], padding=pad)
    ref = inputs if inputs is not None else states
    if states is None:
        states = inputs.new_zeros(inputs.shape[0], nf, *inputs.shape[-2:])
    H, W = states.shape[-2:]
    if inputs is not None:
        b, _, c, h, w = inputs.shape
        i2h = conv("i2h", inputs[:, :seq_len].reshape(-1, c, h, w), 1).reshape(b, seq_len, 3 * nf, H, W)
        i2h = torch.split(i2h, nf, dim=2)
    else:
        i2h = None
    xx = torch.arange(W).view(1, 1, 1, W).expand(1, 1, H, W)
    yy = torch.arange(H).view(1, 1, H, 1).expand(1, 1, H, W)
    grid = torch.cat((xx, yy), 1).to(ref.dtype)
    prev, outs = states, []
    for t in range(seq_len):
        f1 = conv("h2f_conv1", prev, 2)
        if inputs is not None:
            f1 = conv("i2f_conv1", inputs[:, t], 2) + f1
        flows = torch.split(conv("flows_conv", F.leaky_relu(f1, slope), 2), 2, dim=1)
        warped = []
        for flow in flows:
            vg = grid - flow
            vx = 2.0 * vg[:, 0] / max(W - 1, 1) - 1.0
            vy = 2.0 * vg[:, 1] / max(H - 1, 1) - 1.0
            warped.append(F.grid_sample(prev, torch.stack((vx, vy), -1), align_corners=False))
        h2h = torch.split(F.conv2d(torch.cat(warped, 1), p["ret.weight"], p["ret.bias"]), nf, dim=1)
        if i2h is not None:
            r = torch.sigmoid(i2h[0][:, t] + h2h[0]); u = torch.sigmoid(i2h[1][:, t] + h2h[1])
            n = F.leaky_relu(i2h[2][:, t] + r * h2h[2], slope)
        else:
            r = torch.sigmoid(h2h[0]); u = torch.sigmoid(h2h[1])
            n = F.leaky_relu(r * h2h[2], slope)
        prev = u * prev + (1 - u) * n
        outs.append(prev)
    return torch.stack(outs, 1), prev
