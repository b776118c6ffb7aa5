"""Spatio-Temporal LSTM cell (PredRNN-V2) — drop-in for vp_suite/model_blocks/predrnn.py:7-83.

Same constructor signature and parameter tree (`conv_x.0.weight [7Ch,Cin,k,k]`, `conv_h.0.weight [4Ch,Ch,k,k]`,
`conv_m.0.weight [3Ch,Ch,k,k]`, `conv_o.0.weight [Ch,2Ch,k,k]`, `conv_last.weight [Ch,2Ch,1,1]`, all bias-free) and the
same return tuple `(h_new, c_new, m_new, delta_c, delta_m)`. The five convolutions + gate math of :57-83 run as four
fused implicit-GEMM launches (csrc/stlstm_api.hip); with layer_norm=True the convolutions are normalised one by one
(csrc/stlstm_ln_api.hip, csrc/layernorm.hip).

`ActionConditionalSpatioTemporalLSTMCell` (:86-169): the whole step is one library call each way (`ops.acstlstm_step` ->
`vpx_acstlstm_step_fwd / _bwd`, csrc/acst.hip): its six biased convolutions on the implicit-GEMM kernel, the optional LayerNorms,
the `conv_h(h) * conv_a(a)` product (:144), both gate groups, the state updates and the output gate (the product of two
contractions cannot be folded into one GEMM epilogue, so the convolutions stay separate launches inside the call)."""
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
    #: hand the states on with split-format shadows (ops.stlstm_step(use_shadows=True)). Set by a model whose time loop owns the state
    #: tensors from one step to the next (PredRNN_V2.forward); off for a cell that user code drives directly.
    use_shadows = False

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

    def forward(self, x_t, h_t, c_t, m_t, delta_out=None, use_shadows=None, precision=None):
        # use_shadows / precision: per-call overrides for a model whose time loop owns the states (PredRNN_V2.forward) — the module's own
        # attributes stay what user code that drives the cell directly set them to
        ln = ()
        if self.layer_norm:  # LayerNorm([C,H,W]) after conv_x / conv_h / conv_m / conv_o (predrnn.py:24-40)
            ln = tuple(p for seq in (self.conv_x, self.conv_h, self.conv_m, self.conv_o) for p in (seq[1].weight, seq[1].bias))
        return ops.stlstm_step(x_t, h_t, c_t, m_t, self.conv_x[0].weight, self.conv_h[0].weight, self.conv_m[0].weight,
                               self.conv_o[0].weight, self.conv_last.weight, precision=self.precision if precision is None else precision,
                               wsholder=self._ws, ln=ln, use_shadows=self.use_shadows if use_shadows is None else use_shadows,
                               delta_out=delta_out)


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

    def forward(self, x_t, h_t, c_t, m_t, a_t):
        # the whole step — six biased convolutions (+ LayerNorms), conv_h(h) * conv_a(a) (predrnn.py:144), both gate groups, the c / m
        # updates, conv_o / conv_last on mem = (c_new | m_new), the output gate — is one library call each way (vpx_acstlstm_step_fwd / _bwd)
        seqs = (self.conv_x, self.conv_h, self.conv_a, self.conv_m, self.conv_o)
        params = [p for seq in seqs for p in (seq[0].weight, seq[0].bias)] + [self.conv_last.weight, self.conv_last.bias]
        ln = ()
        if self.layer_norm:
            for seq in seqs:
                if abs(float(seq[1].eps) - 1e-5) > 1e-12:
                    raise ValueError("the library's LayerNorm kernels use eps = 1e-5 (nn.LayerNorm's default)")
            ln = [p for seq in seqs for p in (seq[1].weight, seq[1].bias)]
        return ops.acstlstm_step(x_t, h_t, c_t, m_t, a_t, params, ln, precision=self.precision, forget_bias=self._forget_bias)
