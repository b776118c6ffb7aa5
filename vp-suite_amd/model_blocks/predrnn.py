"""Spatio-Temporal LSTM cell (PredRNN-V2) — drop-in for vp_suite/model_blocks/predrnn.py:7-83.

Same constructor signature and parameter tree (`conv_x.0.weight [7Ch,Cin,k,k]`, `conv_h.0.weight [4Ch,Ch,k,k]`,
`conv_m.0.weight [3Ch,Ch,k,k]`, `conv_o.0.weight [Ch,2Ch,k,k]`, `conv_last.weight [Ch,2Ch,1,1]`, all bias-free) and the
same return tuple `(h_new, c_new, m_new, delta_c, delta_m)`. The five convolutions + gate math of :57-83 run as four
fused implicit-GEMM launches (csrc/stlstm_api.hip); with layer_norm=True the convolutions are normalised one by one
(csrc/stlstm_ln_api.hip, csrc/layernorm.hip).

`ActionConditionalSpatioTemporalLSTMCell` (:86-169) is provided as a composite: its six convolutions (with bias) run on
the library's implicit-GEMM kernel (`ops.conv2d_same`, forward and backward), the `conv_h(h) * conv_a(a)` product, the
optional LayerNorms and the gate arithmetic are ATen pointwise ops — the product of two contractions cannot be folded
into one GEMM epilogue, and this variant is not on any BASELINE configuration."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..base import VPModelBlock


class SpatioTemporalLSTMCell(VPModelBlock):
    NAME = "Spatio-Temporal LSTM Cell"
    PAPER_REFERENCE = "https://arxiv.org/abs/2103.09504"
    CODE_REFERENCE = "https://github.com/thuml/predrnn-pytorch"
    MATCHES_REFERENCE = "Yes"

    precision = "f32"

    def __init__(self, in_channel, num_hidden, height, width, filter_size, stride, layer_norm):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("ST-LSTM cell: the recurrence needs stride 1 (state-preserving convolutions)")
        self.num_hidden = num_hidden
        self.padding = filter_size // 2
        self._forget_bias = 1.0
        self.layer_norm = bool(layer_norm)

        def conv(c_in, c_out):
            mods = [nn.Conv2d(c_in, c_out, kernel_size=filter_size, stride=stride, padding=self.padding, bias=False)]
            if layer_norm:
                mods.append(nn.LayerNorm([c_out, height, width]))
            return nn.Sequential(*mods)

        self.conv_x = conv(in_channel, num_hidden * 7)
        self.conv_h = conv(num_hidden, num_hidden * 4)
        self.conv_m = conv(num_hidden, num_hidden * 3)
        self.conv_o = conv(num_hidden * 2, num_hidden)
        self.conv_last = nn.Conv2d(num_hidden * 2, num_hidden, kernel_size=1, stride=1, padding=0, bias=False)
        self._ws = ops.STWorkspace()

    def __getstate__(self):  # workspaces are never pickled with the model (vpsuite.py:394 pickles whole modules)
        state = self.__dict__.copy()
        state["_ws"] = ops.STWorkspace()
        return state

    def forward(self, x_t, h_t, c_t, m_t):
        ln = ()
        if self.layer_norm:  # LayerNorm([C,H,W]) after conv_x / conv_h / conv_m / conv_o (predrnn.py:24-40)
            ln = tuple(p for seq in (self.conv_x, self.conv_h, self.conv_m, self.conv_o) for p in (seq[1].weight, seq[1].bias))
        return ops.stlstm_step(x_t, h_t, c_t, m_t, self.conv_x[0].weight, self.conv_h[0].weight, self.conv_m[0].weight,
                               self.conv_o[0].weight, self.conv_last.weight, precision=self.precision, wsholder=self._ws,
                               ln=ln)


class ActionConditionalSpatioTemporalLSTMCell(VPModelBlock):
    NAME = "Spatio-Temporal LSTM Cell (Action-Conditional)"
    PAPER_REFERENCE = "https://arxiv.org/abs/2103.09504"
    CODE_REFERENCE = "https://github.com/thuml/predrnn-pytorch"
    MATCHES_REFERENCE = "Yes"

    precision = "f32"

    def __init__(self, in_channel, num_hidden, height, width, filter_size, stride, layer_norm):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("ST-LSTM cell: the recurrence needs stride 1 (state-preserving convolutions)")
        self.num_hidden = num_hidden
        self.padding = filter_size // 2
        self._forget_bias = 1.0
        self.layer_norm = bool(layer_norm)

        def conv(c_in, c_out):  # same module tree as predrnn.py:102-136 (Conv2d WITH bias, optional LayerNorm)
            mods = [nn.Conv2d(c_in, c_out, kernel_size=filter_size, stride=stride, padding=self.padding)]
            if layer_norm:
                mods.append(nn.LayerNorm([c_out, height, width]))
            return nn.Sequential(*mods)

        self.conv_x = conv(in_channel, num_hidden * 7)
        self.conv_h = conv(num_hidden, num_hidden * 4)
        self.conv_a = conv(num_hidden, num_hidden * 4)
        self.conv_m = conv(num_hidden, num_hidden * 3)
        self.conv_o = conv(num_hidden * 2, num_hidden)
        self.conv_last = nn.Conv2d(num_hidden * 2, num_hidden, kernel_size=1, stride=1, padding=0)

    def _conv(self, seq, t):
        y = ops.conv2d_same(t, seq[0].weight, seq[0].bias, precision=self.precision)
        if self.layer_norm:
            y = F.layer_norm(y, seq[1].normalized_shape, seq[1].weight, seq[1].bias, seq[1].eps)
        return y

    def forward(self, x_t, h_t, c_t, m_t, a_t):
        nh = self.num_hidden
        x_concat, h_concat = self._conv(self.conv_x, x_t), self._conv(self.conv_h, h_t)
        a_concat, m_concat = self._conv(self.conv_a, a_t), self._conv(self.conv_m, m_t)
        i_x, f_x, g_x, i_x_prime, f_x_prime, g_x_prime, o_x = torch.split(x_concat, nh, dim=1)
        i_h, f_h, g_h, o_h = torch.split(h_concat * a_concat, nh, dim=1)   # predrnn.py:144
        i_m, f_m, g_m = torch.split(m_concat, nh, dim=1)
        i_t = torch.sigmoid(i_x + i_h)
        f_t = torch.sigmoid(f_x + f_h + self._forget_bias)
        g_t = torch.tanh(g_x + g_h)
        delta_c = i_t * g_t
        c_new = f_t * c_t + delta_c
        i_t_prime = torch.sigmoid(i_x_prime + i_m)
        f_t_prime = torch.sigmoid(f_x_prime + f_m + self._forget_bias)
        g_t_prime = torch.tanh(g_x_prime + g_m)
        delta_m = i_t_prime * g_t_prime
        m_new = f_t_prime * m_t + delta_m
        mem = torch.cat((c_new, m_new), 1)
        o_t = torch.sigmoid(o_x + o_h + self._conv(self.conv_o, mem))
        h_new = o_t * torch.tanh(ops.conv2d_same(mem, self.conv_last.weight, self.conv_last.bias, precision=self.precision))
        return h_new, c_new, m_new, delta_c, delta_m
