"""ConvLSTM (Shi et al.) with peephole connections — drop-in for vp_suite/model_blocks/conv_lstm_hzzone.py:7-70.

Same constructor signature, parameter names/shapes/init (`_conv.weight [4Ch,Cin+Ch,k,k]`, `_conv.bias`, `Wci/Wcf/Wco
[1,Ch,H,W]` zeros) and return convention `(stack(h_t) [B,T,Ch,H,W], (h_T, c_T))`. The python time loop with its
cat/conv2d/chunk/sigmoid/tanh op sequence (:52-70) is replaced by one call into libvpx_hip.so.

Peepholes: on non-CPU devices the reference silently de-registers the peephole tensors (`nn.Parameter(...).to(device)`
yields a plain tensor, :30-32): a GPU-built reference model keeps them at zero, never trains them and has no `Wci/Wcf/Wco`
keys in its state_dict. `train_peepholes=True` (default) keeps them real parameters, as on the reference's CPU path (the
oracle); `train_peepholes=False` reproduces the reference's GPU behaviour (fixed tensors, no gradient, not in
state_dict). Either way a state_dict WITHOUT peephole keys (a GPU-trained reference checkpoint) loads under
strict=True, the peepholes keeping their current values (zeros after construction)."""
import torch
from torch import nn

from .. import _lib, ops
from ..base import VPModelBlock


class ConvLSTM(VPModelBlock):
    NAME = "ConvLSTM (Shi et al.)"
    PAPER_REFERENCE = "https://arxiv.org/abs/1506.04214"
    CODE_REFERENCE = "https://github.com/Hzzone/Precipitation-Nowcasting"
    MATCHES_REFERENCE = "Yes"

    precision = "f32"  #: arithmetic of the fused cell kernel: "f32" (exact), "bf16x3", "bf16"

    _PEEPHOLES = ("Wci", "Wcf", "Wco")

    def __init__(self, device, in_channels, enc_channels, state_h, state_w, kernel_size, stride=1, padding=1,
                 train_peepholes=True):
        super().__init__()
        if stride != 1 or 2 * padding != kernel_size - 1:
            # the recurrence feeds h (state size) back through the same conv: only stride-1 "same" convs are consistent
            raise NotImplementedError("ConvLSTM block needs stride=1 and padding=kernel_size//2 (state-preserving conv)")
        self.device = device
        self._conv = nn.Conv2d(in_channels + enc_channels, 4 * enc_channels, kernel_size, stride, padding)
        self.state_h, self.state_w = state_h, state_w
        shape = (1, enc_channels, state_h, state_w)
        self.train_peepholes = bool(train_peepholes)
        for name in self._PEEPHOLES:
            if self.train_peepholes:
                setattr(self, name, nn.Parameter(torch.zeros(shape)))
            else:  # the reference's GPU behaviour: plain tensors that follow .to(), absent from state_dict / parameters()
                self.register_buffer(name, torch.zeros(shape), persistent=False)
        self.in_c, self.enc_c = in_channels, enc_channels

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints of GPU-built reference models hold no peephole keys (see the module docstring): not an error."""
        n0 = len(missing_keys)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        peep = {prefix + n for n in self._PEEPHOLES}
        missing_keys[n0:] = [k for k in missing_keys[n0:] if k not in peep]
        if not self.train_peepholes:  # CPU-built reference checkpoints do carry them: take the values, stay untrained
            for n in self._PEEPHOLES:
                if prefix + n in state_dict:
                    with torch.no_grad():
                        getattr(self, n).copy_(state_dict[prefix + n])
                    if prefix + n in unexpected_keys:
                        unexpected_keys.remove(prefix + n)

    def takes_split_input(self, batch, seq_len):
        """True when forward(), in inference, consumes an `ops.SplitActivation` (the stage glue then writes the block's
        input directly in the kernels' operand format: no fp32 copy, no conversion pass)."""
        k = self._conv.kernel_size[0]
        return (not torch.is_grad_enabled()) and ops.convlstm_takes_split(batch, seq_len, self.in_c, self.enc_c, self.state_h,
                                                                         self.state_w, k, _lib.GATE_IFGO, self.precision)

    def writes_split_output(self, batch, seq_len):
        """True when forward(..., out_split=True), in inference, can hand its output sequence out as an `ops.SplitActivation`
        (no fp32 copy is written; for consumers that read the operand format)."""
        k = self._conv.kernel_size[0]
        return (not torch.is_grad_enabled()) and ops.convlstm_writes_split(batch, seq_len, self.in_c, self.enc_c, self.state_h,
                                                                           self.state_w, k, _lib.GATE_IFGO, self.precision)

    def forward(self, inputs, states, seq_len, out_split=False):
        """inputs [B,T,Cin,H,W] (or an ops.SplitActivation of that shape) or None (zero input every step); states (h, c) or
        None (zero states). out_split: see writes_split_output (not part of the reference's signature; default off)."""
        if inputs is None and states is None:
            raise ValueError("inputs and states should not be all none")
        if states is None:
            # zero states contribute nothing: the library skips the h-segment at t=0 instead of reading zeros
            h0 = c0 = None
        else:
            h0, c0 = states
        out, hT, cT = ops.convlstm_seq(inputs, h0, c0, self._conv.weight, self._conv.bias, self.Wci, self.Wcf, self.Wco,
                                       seq_len=seq_len, in_channels=self.in_c, gate_order=_lib.GATE_IFGO,
                                       precision=self.precision, out_split=out_split)
        return out, (hT, cT)
