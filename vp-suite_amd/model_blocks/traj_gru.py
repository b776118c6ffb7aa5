"""TrajGRU (Shi et al. 2017) — drop-in for vp_suite/model_blocks/traj_gru.py:74-214 on the same Encoder-Forecaster
skeleton as the ConvLSTM block.

Composite: the five convolutions (i2h 3x3 over all frames at once, i2f / h2f / flows 5x5, ret 1x1) run on the library's
implicit-GEMM kernel (`ops.conv2d_same`, forward and backward); the bilinear warp (`F.grid_sample`, :148-162) and the GRU
gate arithmetic (:190-203) are ATen ops. Same constructor signature, parameter names / shapes and return convention
`(stack(h_t) [B,T,C,H,W], h_T)`. Restrictions (raise NotImplementedError): i2h stride 1 with "same" padding and no
dilation — what the reference's own encoder/forecaster configuration uses."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..base import VPModelBlock


class Activation:
    def __init__(self, act_type, negative_slope=0.2, inplace=True):
        self._act_type = act_type
        self.negative_slope = negative_slope
        self.inplace = inplace

    def __call__(self, input):
        if self._act_type == 'leaky':
            return F.leaky_relu(input, negative_slope=self.negative_slope, inplace=self.inplace)
        elif self._act_type == 'relu':
            return F.relu(input, inplace=self.inplace)
        elif self._act_type == 'sigmoid':
            return torch.sigmoid(input)
        raise NotImplementedError


class TrajGRU(VPModelBlock):
    NAME = "TrajGRU"
    PAPER_REFERENCE = "https://arxiv.org/abs/1706.03458"
    CODE_REFERENCE = "https://github.com/Hzzone/Precipitation-Nowcasting"
    MATCHES_REFERENCE = "Yes"

    precision = "f32"

    def __init__(self, device, in_c, enc_c, state_h, state_w, zoneout=0.0, L=5, i2h_kernel=(3, 3), i2h_stride=(1, 1),
                 i2h_pad=(1, 1), h2h_kernel=(5, 5), h2h_dilate=(1, 1),
                 act_type=Activation('leaky', negative_slope=0.2, inplace=True)):
        super().__init__()
        if tuple(i2h_stride) != (1, 1) or tuple(2 * p for p in i2h_pad) != tuple(k - 1 for k in i2h_kernel):
            raise NotImplementedError("TrajGRU block: i2h must be a stride-1 'same' convolution")
        self.device = device
        self._num_filter = enc_c
        self._state_height, self._state_width = state_h, state_w
        self._act_type = act_type
        self._L = L
        self._zoneout = zoneout
        self.i2h = nn.Conv2d(in_c, enc_c * 3, i2h_kernel, i2h_stride, i2h_pad)
        self.i2f_conv1 = nn.Conv2d(in_c, 32, (5, 5), 1, (2, 2))
        self.h2f_conv1 = nn.Conv2d(enc_c, 32, (5, 5), 1, (2, 2))
        self.flows_conv = nn.Conv2d(32, L * 2, (5, 5), 1, (2, 2))
        self.ret = nn.Conv2d(enc_c * L, enc_c * 3, (1, 1), 1)

    def _conv(self, mod, t):
        return ops.conv2d_same(t, mod.weight, mod.bias, precision=self.precision)

    def _flow_generator(self, inputs, states):
        f_conv1 = self._conv(self.h2f_conv1, states)
        if inputs is not None:
            f_conv1 = self._act_type(self._conv(self.i2f_conv1, inputs) + f_conv1)
        else:  # a library op's output must not be modified in place (autograd would bypass its backward): out of place
            act = self._act_type
            f_conv1 = Activation(act._act_type, act.negative_slope, False)(f_conv1) if isinstance(act, Activation) else act(f_conv1.clone())
        return torch.split(self._conv(self.flows_conv, f_conv1), 2, dim=1)

    def _warp(self, input, flow, grid):
        # traj_gru.py:148-162: pixel grid + flow, normalised to [-1, 1], bilinear grid_sample (default alignment)
        B, C, H, W = input.shape
        vgrid = grid + flow
        vx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
        vy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
        return F.grid_sample(input, torch.stack((vx, vy), dim=-1), align_corners=False)

    def forward(self, inputs, states, seq_len):
        if inputs is None and states is None:
            raise ValueError("TrajGRU received 'None' both in input and state")
        ref = inputs if inputs is not None else states
        dev = ref.device
        if states is None:
            states = torch.zeros((inputs.shape[0], self._num_filter, self._state_height, self._state_width),
                                 dtype=torch.float, device=dev)
        nf = self._num_filter
        if inputs is not None:
            b, _, c, h, w = inputs.shape
            i2h = self._conv(self.i2h, inputs[:, :seq_len].reshape(-1, c, h, w))   # all frames in one launch (:171-173)
            i2h = i2h.reshape(b, seq_len, *i2h.shape[1:])
            i2h_slice = torch.split(i2h, nf, dim=2)
        else:
            i2h_slice = None
        H, W = states.shape[-2:]
        xx = torch.arange(0, W, device=dev).view(1, 1, 1, W).expand(1, 1, H, W)
        yy = torch.arange(0, H, device=dev).view(1, 1, H, 1).expand(1, 1, H, W)
        grid = torch.cat((xx, yy), 1).float()
        prev_h, outputs, next_h = states, [], None
        for t in range(seq_len):
            flows = self._flow_generator(inputs[:, t] if inputs is not None else None, prev_h)
            warped = torch.cat([self._warp(prev_h, -flow, grid) for flow in flows], dim=1)
            h2h_slice = torch.split(self._conv(self.ret, warped), nf, dim=1)
            if i2h_slice is not None:
                reset_gate = torch.sigmoid(i2h_slice[0][:, t] + h2h_slice[0])
                update_gate = torch.sigmoid(i2h_slice[1][:, t] + h2h_slice[1])
                new_mem = self._act_type(i2h_slice[2][:, t] + reset_gate * h2h_slice[2])
            else:
                reset_gate = torch.sigmoid(h2h_slice[0])
                update_gate = torch.sigmoid(h2h_slice[1])
                new_mem = self._act_type(reset_gate * h2h_slice[2])
            next_h = update_gate * prev_h + (1 - update_gate) * new_mem
            if self._zoneout > 0.0:
                mask = F.dropout2d(torch.zeros_like(prev_h), p=self._zoneout)
                next_h = torch.where(mask.bool(), next_h, prev_h)
            outputs.append(next_h)
            prev_h = next_h
        return torch.stack(outputs, dim=1), next_h
