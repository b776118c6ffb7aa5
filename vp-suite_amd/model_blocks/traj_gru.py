"""TrajGRU (Shi et al. 2017) — drop-in for vp_suite/model_blocks/traj_gru.py:74-214 on the Encoder-Forecaster skeleton
(SURVEY.md §8f rank 4).

The block owns the reference's five convolutions under the reference's names (`i2h`, `i2f_conv1`, `h2f_conv1`, `flows_conv`,
`ret`: same shapes, same default init, same state_dict) and hands the whole sequence to ONE library-backed autograd
Function (`traj_ops.trajgru_seq`): flow generation, the L bilinear warps, the 1x1 `ret` convolution and the GRU gates run
as HIP launches forward and backward (csrc/trajgru.hip + the implicit-GEMM convolution kernel); there is no per-step
Python autograd graph and no ATen compute op. Return convention `(h_1..h_T as [B,T,C,H,W], h_T)` as in the reference.

Restrictions (NotImplementedError): i2h must be a stride-1 "same" convolution without dilation and the activation a
LeakyReLU with positive slope — what the reference's own encoder / forecaster configuration (ef_traj_gru.py) uses;
zoneout > 0 (the reference's expression `torch.where(dropout2d(zeros), ...)` raises on a current torch: float mask)."""
from torch import nn

from .. import traj_ops
from ..base import VPModelBlock


class Activation:
    """Activation selector with the reference's constructor contract `(act_type, negative_slope=0.2, inplace=True)`
    (traj_gru.py:8-28). The block reads `kind` / `negative_slope` and runs the activation inside its kernels; calling the
    object applies it to a tensor (API compatibility for user code)."""
    KINDS = ("leaky", "relu", "sigmoid")

    def __init__(self, act_type, negative_slope=0.2, inplace=True):
        self._act_type, self.negative_slope, self.inplace = act_type, negative_slope, inplace

    @property
    def kind(self):
        if self._act_type not in self.KINDS:
            raise NotImplementedError
        return self._act_type

    def __call__(self, input):
        import torch.nn.functional as F
        fn = {"leaky": lambda t: F.leaky_relu(t, self.negative_slope, self.inplace), "relu": lambda t: F.relu(t, self.inplace),
              "sigmoid": lambda t: t.sigmoid()}
        return fn[self.kind](input)


class TrajGRU(VPModelBlock):
    NAME = "TrajGRU"
    PAPER_REFERENCE = "https://arxiv.org/abs/1706.03458"
    CODE_REFERENCE = "https://github.com/Hzzone/Precipitation-Nowcasting"
    MATCHES_REFERENCE = "Yes"

    precision = "f32"  #: arithmetic of the convolution kernels: "f32" (exact), "bf16x3", "bf16"

    def __init__(self, device, in_c, enc_c, state_h, state_w, zoneout=0.0, L=5, i2h_kernel=(3, 3), i2h_stride=(1, 1),
                 i2h_pad=(1, 1), h2h_kernel=(5, 5), h2h_dilate=(1, 1),
                 act_type=Activation('leaky', negative_slope=0.2, inplace=True)):
        super().__init__()
        same = tuple(i2h_stride) == (1, 1) and all(2 * p == k - 1 for p, k in zip(i2h_pad, i2h_kernel)) and i2h_kernel[0] == i2h_kernel[1]
        if not same:
            raise NotImplementedError("TrajGRU block: i2h must be a square stride-1 'same' convolution")
        if h2h_kernel[0] % 2 != 1 or h2h_kernel[1] % 2 != 1:
            raise AssertionError("Only support odd number, get h2h_kernel= %s" % str(h2h_kernel))
        if getattr(act_type, "_act_type", None) != "leaky" or not act_type.negative_slope > 0:
            raise NotImplementedError("TrajGRU block: the activation must be LeakyReLU with a positive slope")
        if zoneout > 0.0:
            raise NotImplementedError("TrajGRU block: zoneout > 0 is not implemented (the reference's zoneout expression "
                                      "fails on current torch: torch.where needs a boolean mask)")
        self.device = device
        self._num_filter = enc_c
        self._state_height, self._state_width = state_h, state_w
        self._h2h_kernel, self._h2h_dilate = h2h_kernel, h2h_dilate   # kept like the reference keeps them: unused by TrajGRU
        self._act_type = act_type
        self._L = L
        self._zoneout = zoneout
        # reset / update / candidate projections of the input; flow generator (input and hidden branch, 32 features);
        # 2L flow channels; 1x1 mixing of the L warped states (traj_gru.py:99-132)
        self.i2h = nn.Conv2d(in_c, enc_c * 3, i2h_kernel, i2h_stride, i2h_pad)
        self.i2f_conv1 = nn.Conv2d(in_c, traj_ops.FLOW_FEATURES, (5, 5), 1, (2, 2))
        self.h2f_conv1 = nn.Conv2d(enc_c, traj_ops.FLOW_FEATURES, (5, 5), 1, (2, 2))
        self.flows_conv = nn.Conv2d(traj_ops.FLOW_FEATURES, L * 2, (5, 5), 1, (2, 2))
        self.ret = nn.Conv2d(enc_c * L, enc_c * 3, (1, 1), 1)

    def forward(self, inputs, states, seq_len):
        """inputs [B,T,Cin,H,W] or None; states h_0 [B,C,H,W] or None (zeros)."""
        params = [t for m in (self.i2h, self.i2f_conv1, self.h2f_conv1, self.flows_conv, self.ret) for t in (m.weight, m.bias)]
        return traj_ops.trajgru_seq(inputs, states, params, seq_len=seq_len, L=self._L, slope=float(self._act_type.negative_slope),
                                    state_hw=(self._state_height, self._state_width), precision=self.precision)
