"""Encoder-Forecaster ConvLSTM ("convlstm-shi") — drop-in for
vp_suite/models/precipitation_nowcasting/ef_blocks.py:15-187 and ef_conv_lstm.py:8-108.

Identical hyper-parameter names/defaults, module tree and state_dict keys
(`encoder.stage{n}.conv{n}_leaky_1.*`, `encoder.rnn{n}.{_conv.weight,_conv.bias,Wci,Wcf,Wco}`,
`forecaster.rnn{n}.*`, `forecaster.stage{n}.deconv*_leaky_1.*`, `forecaster.stage1.conv3_3.*`), so reference
checkpoints load unchanged. The six recurrent blocks run as fused HIP kernels; activations stay channels-last (NHWC)
between the glue convolutions (MIOpen) and the recurrent kernels, so no layout round trips happen inside the model."""
from collections import OrderedDict

import torch
from torch import nn

from .. import ops
from ..base import VPModel
from ..model_blocks import ConvLSTM
from ..utils import conv_output_shape, convtransp_output_shape


def _stage(spec: "OrderedDict[str, list]") -> nn.Sequential:
    """Layer-name driven stage builder with the reference's naming rules (ef_blocks.py:15-49): 'identity', 'pool',
    'deconv*' (ConvTranspose2d), 'conv*' (Conv2d); a 'relu' / 'leaky' tag in the name appends the activation
    (LeakyReLU slope 0.2) under the key '<tag>_<name>'. Values: [c_in, c_out, kernel, stride, pad]."""
    layers = []
    for name, v in spec.items():
        if "identity" in name:
            layers.append((name, nn.Identity()))
            continue
        if "pool" in name:
            layers.append((name, nn.MaxPool2d(kernel_size=v[0], stride=v[1], padding=v[2])))
            continue
        if "deconv" in name:
            op = nn.ConvTranspose2d(v[0], v[1], v[2], v[3], v[4])
        elif "conv" in name:
            op = nn.Conv2d(v[0], v[1], v[2], v[3], v[4])
        else:
            raise NotImplementedError
        layers.append((name, op))
        if "relu" in name:
            layers.append(("relu_" + name, nn.ReLU(inplace=True)))
        elif "leaky" in name:
            layers.append(("leaky_" + name, nn.LeakyReLU(negative_slope=0.2, inplace=True)))
    seq = nn.Sequential(OrderedDict(layers))
    _validate_stage(seq)
    return seq


def _validate_stage(subnet: nn.Sequential):
    """Construction-time check of a stage's convolutions against what the library's glue entry points implement — there is no second
    backend, so an unsupported layer is an error when the model is BUILT (not at its first forward, and not as late as its backward)."""
    mods = list(subnet.children())
    for i, m in enumerate(mods):
        if not isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            continue
        if _conv_cfg(m) is None:
            raise ops.VpxError(f"EF stage glue: {m} is outside vpx_conv2d_ex (stride 1 or 2, kernel <= 7, square stride / padding, no "
                               f"dilation / groups / output_padding; transposed: padding <= kernel - 1 at stride 1, kernel >= 2 at "
                               f"stride 2): unsupported layer configuration")
        if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU) and mods[i + 1].negative_slope < 0:
            raise ops.VpxError(f"EF stage glue: LeakyReLU with negative slope {mods[i + 1].negative_slope} after {m}: unsupported")


def _conv_cfg(m):
    """(kh, kw, stride, pad, transposed) of a Conv2d / ConvTranspose2d the library's glue entry points can take, else None."""
    if not isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        return None
    tr = isinstance(m, nn.ConvTranspose2d)
    kh, kw = m.kernel_size
    simple = (m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1] and m.dilation == (1, 1)
              and m.groups == 1 and (not tr or m.output_padding == (0, 0)))
    if simple and ops.glue_supported(kh, kw, m.stride[0], m.padding[0], tr):
        return kh, kw, m.stride[0], m.padding[0], tr
    return None


def _glue_precision(cell_precision: str) -> str:
    """Arithmetic of the stage glue for a model whose cells run in `cell_precision`. The plain-bf16 cell mode (inference extras) keeps the
    glue on bf16x3: the schedule-driven split-operand kernels (convq, operand-format handovers) only exist there, they are FASTER than the
    first-generation bf16 forms (measured B = 128: glue 3.7 vs 1.9 ms of a 14.2 ms step) and more accurate."""
    return "bf16x3" if cell_precision == "bf16" else cell_precision


def _stage_takes_split(subnet: nn.Sequential, n, c, h, w, precision):
    """True when the stage's first layer can read its input in the split-bf16 operand format (inference): the producing
    recurrent block then writes that format only."""
    mods = list(subnet.children())
    cfg = _conv_cfg(mods[0]) if mods else None
    if cfg is None or precision != "bf16x3" or torch.is_grad_enabled():
        return False
    kh, kw, stride, pad, tr = cfg
    co = mods[0].out_channels
    return ops.conv2d_ex_takes_split(n, h, w, c, co, kh, kw, stride, pad, tr, precision)


def _run_stage(subnet: nn.Sequential, x, precision: str, split_last: bool = False):
    """Executes a stage built by _stage() on a channels-last [N,C,H,W] batch — or on (buffer, (N,C,H,W)) in the split-bf16 operand
    format, which the stage's first convolution then reads directly. Conv2d / ConvTranspose2d layers (with a directly following
    LeakyReLU fused in) run through libvpx_hip's glue entry points; a configuration they do not implement was refused when the
    stage was built (_validate_stage). Only parameter-free layers (pool, ReLU, Identity) run as stock modules. Tensors must live on
    the GPU: a CPU tensor raises VpxError naming exactly that (there is no CPU path)."""
    mods = list(subnet.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        cfg = _conv_cfg(m)
        x_split = isinstance(x, tuple)
        if not x_split and not x.is_cuda:
            raise ops.VpxError(f"EF stage glue: tensors must live on the GPU (got device '{x.device}'). The model runs only as HIP kernels "
                               f"on MI355X; there is no CPU fallback.")
        if cfg is not None:
            kh, kw, stride, pad, tr = cfg
            slope = 0.0
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                slope = float(mods[i + 1].negative_slope)
                i += 1
            co = m.out_channels
            last_to_split = split_last and i + 1 == len(mods) and co % 8 == 0   # the stage's output goes straight to a recurrent block
            if (not x_split and tr and stride == 2 and precision == "bf16x3" and not torch.is_grad_enabled() and x.shape[1] % 8 == 0
                    and ops.conv2d_ex_prefers_split(x.shape[0], x.shape[2], x.shape[3], x.shape[1], co, kh, kw, stride, pad, tr, precision)):
                # fp32 input (the producing block ran on a small-grid kernel): the four output phases in one convq launch are worth
                # one conversion pass (40 frames 96 -> 96 at 32x32: 0.15 -> 0.06 ms)
                x = (ops.split_convert(x)[0], tuple(x.shape))
                x_split = True
            if x_split:
                y, ybuf, shp = ops.conv2d_ex_from_split(x[0], x[1], m.weight, m.bias, stride, pad, tr, slope, precision,
                                                        out_split=last_to_split, out_fp32=not last_to_split)
                if last_to_split:
                    return ybuf, shp
                x = y
            elif last_to_split:
                return ops.conv2d_ex_split(x, m.weight, m.bias, stride, pad, tr, slope, precision)
            else:
                x = ops.conv2d_ex(x, m.weight, m.bias, stride, pad, tr, slope, precision)
            i += 1
            continue
        if isinstance(x, tuple):
            raise RuntimeError(f"EF stage glue: {m} cannot read a split-format input (stage was offered one by mistake)")
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            # (unreachable for stages built by _stage(); a module swapped in afterwards) no stock-torch convolution in the product path
            raise ops.VpxError(f"EF stage glue: {m} is outside vpx_conv2d_ex (ops.glue_supported): unsupported layer configuration")
        x = m(x)   # activations / Identity only
        i += 1
    return x


def _apply_framewise(subnet: nn.Module, seq, precision: str = "f32", consumer=None):
    """Runs a 2-D stage on every frame of [B,T,C,H,W] (a tensor, or an ops.SplitActivation handed over by the producing block),
    channels-last in and out (B*T is folded into the batch).
    `consumer`: the recurrent block the result feeds; when it takes split-format input (inference on the second-generation
    cell kernel) the stage's last convolution writes that format directly and an ops.SplitActivation is returned."""
    b, t = seq.shape[:2]
    want_split = consumer is not None and hasattr(consumer, "takes_split_input") and consumer.takes_split_input(b, t)
    if isinstance(seq, ops.SplitActivation):
        _, _, c, h, w = seq.shape
        y = _run_stage(subnet, (seq.buf, (b * t, c, h, w)), precision, want_split)
    else:
        flat = ops.to_channels_last(seq).reshape(b * t, *seq.shape[2:])  # a view: NHWC memory folds B,T for free
        y = _run_stage(subnet, flat.contiguous(memory_format=torch.channels_last), precision, want_split)
    if isinstance(y, tuple):
        buf, (n, c, h, w) = y
        return ops.SplitActivation(buf, (b, t, c, h, w))
    y = y.contiguous(memory_format=torch.channels_last)
    return y.view(b, t, *y.shape[1:])


SPLIT_HANDOVER = True   #: inference: recurrent blocks hand their output sequences to the stage glue in operand format only (A/B switch)


def _block_out_split(rnn, next_stage, batch, seq_len):
    """Should this recurrent block hand its output sequence out in operand format only? Yes when it can (inference, second-generation
    cell kernel) and its consumer — the next stage's first convolution, or nobody — does not need the fp32 copy."""
    if not SPLIT_HANDOVER or not hasattr(rnn, "writes_split_output") or not rnn.writes_split_output(batch, seq_len):
        return False
    if next_stage is None:
        return True
    return _stage_takes_split(next_stage, batch * seq_len, rnn.enc_c, rnn.state_h, rnn.state_w, _glue_precision(getattr(rnn, "precision", "f32")))


class Encoder(nn.Module):
    def __init__(self, subnets, rnns):
        super().__init__()
        assert len(subnets) == len(rnns)
        self.blocks = len(subnets)
        for index, (params, rnn) in enumerate(zip(subnets, rnns), 1):
            setattr(self, f"stage{index}", _stage(params))
            setattr(self, f"rnn{index}", rnn)

    def forward_by_stage(self, input, subnet, rnn, next_stage=None, last=False):
        input = _apply_framewise(subnet, input, _glue_precision(getattr(rnn, "precision", "f32")), consumer=rnn)
        b, t = input.shape[:2]
        if (last or next_stage is not None) and _block_out_split(rnn, next_stage, b, t):
            return rnn(input, None, seq_len=t, out_split=True)
        return rnn(input, None, seq_len=t)

    def forward(self, input):
        hidden_states = []
        for i in range(1, self.blocks + 1):
            nxt = getattr(self, f"stage{i + 1}") if i < self.blocks else None   # the last block's sequence has no reader (states only)
            input, state = self.forward_by_stage(input, getattr(self, f"stage{i}"), getattr(self, f"rnn{i}"), nxt, last=i == self.blocks)
            hidden_states.append(state)
        return tuple(hidden_states)


class Forecaster(nn.Module):
    def __init__(self, subnets, rnns):
        super().__init__()
        assert len(subnets) == len(rnns)
        self.blocks = len(subnets)
        for index, (params, rnn) in enumerate(zip(subnets, rnns)):
            setattr(self, f"rnn{self.blocks - index}", rnn)
            setattr(self, f"stage{self.blocks - index}", _stage(params))

    def forward_by_stage(self, input, state, pred_frames, subnet, rnn, next_rnn=None):
        b = state[0].shape[0]
        if _block_out_split(rnn, subnet, b, pred_frames):
            input, _ = rnn(input, state, pred_frames, out_split=True)
        else:
            input, _ = rnn(input, state, pred_frames)
        return _apply_framewise(subnet, input, _glue_precision(getattr(rnn, "precision", "f32")), consumer=next_rnn)

    def forward(self, hidden_states, pred_frames):
        # like the reference (ef_blocks.py:109-110) the top block is addressed as stage3/rnn3 and gets no input
        input = self.forward_by_stage(None, hidden_states[-1], pred_frames, self.stage3, self.rnn3,
                                      getattr(self, f"rnn{self.blocks - 1}", None) if self.blocks > 1 else None)
        for i in range(self.blocks - 1, 0, -1):
            input = self.forward_by_stage(input, hidden_states[i - 1], pred_frames, getattr(self, f"stage{i}"),
                                          getattr(self, f"rnn{i}"), getattr(self, f"rnn{i - 1}") if i > 1 else None)
        return input


class Encoder_Forecaster(VPModel):
    NAME = "Encoder-Forecaster Structure (Shi et al.)"

    def __init__(self, device, **model_kwargs):
        super().__init__(device, **model_kwargs)
        for name, val in [(k, v) for k, v in vars(self).items() if k.startswith(("enc_", "dec_"))]:
            want = 2 * self.num_layers if name in ("enc_c", "dec_c") else self.num_layers
            if len(val) != want:
                raise AttributeError(f"Speficied {self.num_layers} layers, but len of attribute '{name}' "
                                     f"doesn't match that ({val}).")
        hw = (self.img_h, self.img_w)
        enc_h, enc_w = [], []
        for n in range(self.num_layers):
            hw = conv_output_shape(hw, self.enc_conv_k[n], self.enc_conv_s[n], self.enc_conv_p[n])
            enc_h.append(hw[0]); enc_w.append(hw[1])
        dec_h, dec_w = [hw[0]], [hw[1]]
        for n in range(self.num_layers - 1):
            hw = convtransp_output_shape(hw, self.dec_conv_k[n], self.dec_conv_s[n], self.dec_conv_p[n])
            dec_h.append(hw[0]); dec_w.append(hw[1])
        final = convtransp_output_shape(hw, self.dec_conv_k[-1], self.dec_conv_s[-1], self.dec_conv_p[-1])
        if (self.img_h, self.img_w) != tuple(final):
            sizes = list(zip(enc_h, enc_w)) + list(zip(dec_h, dec_w))
            raise AttributeError(f"Model layer hyperparameters yield wrong output size: {tuple(final)} "
                                 f"(expected: {(self.img_h, self.img_w)}). All hidden sizes: {sizes}")
        self.enc_rnn_state_h, self.enc_rnn_state_w = enc_h, enc_w
        self.dec_rnn_state_h, self.dec_rnn_state_w = dec_h, dec_w
        enc_convs, enc_rnns, dec_convs, dec_rnns = self._build_encoder_decoder()
        self.encoder = Encoder(enc_convs, enc_rnns).to(self.device)
        self.forecaster = Forecaster(dec_convs, dec_rnns).to(self.device)
        self.NON_CONFIG_VARS.extend(["encoder", "forecaster"])

    def _build_encoder_decoder(self):
        raise NotImplementedError

    def pred_1(self, x, **kwargs):
        return self(x, pred_frames=1, **kwargs)[0].squeeze(dim=1)

    def forward(self, x, pred_frames: int = 1, **kwargs):
        return self.forecaster(self.encoder(x), pred_frames), None


class EF_ConvLSTM(Encoder_Forecaster):
    NAME = "EF-ConvLSTM (Shi et al.)"
    PAPER_REFERENCE = "https://arxiv.org/abs/1506.04214"
    CODE_REFERENCE = "https://github.com/Hzzone/Precipitation-Nowcasting"
    MATCHES_REFERENCE = "Yes"

    # hyper-parameters: names and defaults of ef_conv_lstm.py:31-65 (c=channels, k=kernel, s=stride, p=padding)
    num_layers = 3
    enc_c = [16, 64, 64, 96, 96, 96]
    dec_c = [96, 96, 96, 96, 64, 16]
    enc_conv_names = ["conv1_leaky_1", "conv2_leaky_1", "conv3_leaky_1"]
    enc_conv_k, enc_conv_s, enc_conv_p = [3, 3, 3], [1, 2, 2], [1, 1, 1]
    dec_conv_names = ["deconv1_leaky_1", "deconv2_leaky_1", "deconv3_leaky_1"]
    dec_conv_k, dec_conv_s, dec_conv_p = [4, 4, 3], [2, 2, 1], [1, 1, 1]
    enc_rnn_k, enc_rnn_s, enc_rnn_p = [3, 3, 3], [1, 1, 1], [1, 1, 1]
    dec_rnn_k, dec_rnn_s, dec_rnn_p = [3, 3, 3], [1, 1, 1], [1, 1, 1]
    final_conv_1_name, final_conv_1_c, final_conv_1_k, final_conv_1_s, final_conv_1_p = "identity", 16, 3, 1, 1
    final_conv_2_name, final_conv_2_k, final_conv_2_s, final_conv_2_p = "conv3_3", 1, 1, 0
    cell_precision = "f32"  #: arithmetic of the fused ConvLSTM kernels ("f32" | "bf16x3" | "bf16")
    train_peepholes: bool = True  #: False = the reference's behaviour on GPU devices (peepholes fixed, not in state_dict)

    def _build_encoder_decoder(self):
        enc_convs, enc_rnns, dec_convs, dec_rnns = [], [], [], []
        c_prev = self.img_c
        for n in range(self.num_layers):
            c_mid, c_out = self.enc_c[2 * n], self.enc_c[2 * n + 1]
            enc_convs.append(OrderedDict({self.enc_conv_names[n]: [c_prev, c_mid, self.enc_conv_k[n],
                                                                   self.enc_conv_s[n], self.enc_conv_p[n]]}))
            enc_rnns.append(self._rnn(c_mid, c_out, self.enc_rnn_state_h[n], self.enc_rnn_state_w[n],
                                      self.enc_rnn_k[n], self.enc_rnn_s[n], self.enc_rnn_p[n]))
            c_prev = c_out
        for n in range(self.num_layers):
            c_mid, c_out = self.dec_c[2 * n], self.dec_c[2 * n + 1]
            dec_rnns.append(self._rnn(c_prev, c_mid, self.dec_rnn_state_h[n], self.dec_rnn_state_w[n],
                                      self.dec_rnn_k[n], self.dec_rnn_s[n], self.dec_rnn_p[n]))
            spec = OrderedDict({self.dec_conv_names[n]: [c_mid, c_out, self.dec_conv_k[n], self.dec_conv_s[n],
                                                         self.dec_conv_p[n]]})
            if n == self.num_layers - 1:
                spec[self.final_conv_1_name] = [c_out, self.final_conv_1_c, self.final_conv_1_k, self.final_conv_1_s,
                                                self.final_conv_1_p]
                spec[self.final_conv_2_name] = [self.final_conv_1_c, self.img_c, self.final_conv_2_k,
                                                self.final_conv_2_s, self.final_conv_2_p]
            dec_convs.append(spec)
            c_prev = c_out
        return enc_convs, enc_rnns, dec_convs, dec_rnns

    def _rnn(self, c_in, c_state, h, w, k, s, p):
        blk = ConvLSTM(device=self.device, in_channels=c_in, enc_channels=c_state, state_h=h, state_w=w,
                       kernel_size=k, stride=s, padding=p, train_peepholes=self.train_peepholes)
        blk.precision = self.cell_precision
        return blk
