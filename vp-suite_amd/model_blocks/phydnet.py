"""PhyDNet's ConvLSTM branch — drop-in for `SingleStepConvLSTM` (vp_suite/model_blocks/phydnet.py:117-175): a stack of
`ConvLSTMCell`s (conv_lstm_ndrplz.py:7-48, bias=True) that consumes ONE frame per call and keeps its (H, C) lists
between calls. The constructor signature, the public attributes (`H`, `C`, `cell_list`, ...) and the return convention
`((H, C), H)` are the reference's contract; the body is this package's:

  * a layer step is one fused cell launch of the library (`vpx_convlstm_seq_fwd`, T = 1, gate order i,f,o,g);
  * the step right after `init_hidden` passes NO state to the library (NULL = zeros): the kernel then skips the whole
    recurrent half of the contraction instead of multiplying a zero tensor, and nothing is read for c;
  * the action plane is written into the channel tail of one preallocated channels-last input buffer (no expand + cat pair).

The PhyCell / encoder-decoder parts of PhyDNet are outside the hot path."""
import torch
from torch import nn

from .. import ops
from .conv_lstm_ndrplz import ConvLSTMCell


class SingleStepConvLSTM(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dims, n_layers, kernel_size, action_conditional, action_size, device):
        super().__init__()
        self.input_size, self.input_dim, self.hidden_dims = input_size, input_dim, hidden_dims
        self.n_layers, self.kernel_size = n_layers, kernel_size
        self.action_conditional, self.action_size, self.device = action_conditional, action_size, device
        widths = [input_dim + (action_size if action_conditional else 0)] + list(hidden_dims[:n_layers])
        self.cell_list = nn.ModuleList(ConvLSTMCell(input_dim=cin, hidden_dim=ch, kernel_size=kernel_size, bias=True)
                                       for cin, ch in zip(widths[:-1], widths[1:]))
        self.H, self.C = [], []
        self._pristine = []   # the zero tensors handed out by init_hidden, while nobody has replaced or written them

    def _is_untouched_zero(self, j):
        """True while layer j's state still is the zero pair of init_hidden (same objects, never written in place)."""
        return (j < len(self._pristine) and self.H[j] is self._pristine[j][0] and self.C[j] is self._pristine[j][1]
                and self.H[j]._version == 0 and self.C[j]._version == 0)

    def _bottom_input(self, frame, action):
        if not self.action_conditional:
            return frame
        b, (hh, ww) = frame.size(0), self.input_size
        buf = ops.new_channels_last((b, self.input_dim + self.action_size, hh, ww), frame.device)
        buf[:, :self.input_dim].copy_(frame)
        buf[:, self.input_dim:].copy_(action[:, :, None, None].expand(b, self.action_size, hh, ww))
        return buf

    def forward(self, frame, action, first_timestep=False):
        if first_timestep:
            self.init_hidden(frame.size(0))
        below = self._bottom_input(frame, action)
        for j, cell in enumerate(self.cell_list):
            h, c = (None, None) if self._is_untouched_zero(j) else (self.H[j], self.C[j])
            _, self.H[j], self.C[j] = cell._run(below.unsqueeze(1), h, c, 1)
            below = self.H[j]
        self._pristine = []
        return (self.H, self.C), self.H

    def init_hidden(self, batch_size):
        hh, ww = self.input_size
        self._pristine = [(torch.zeros(batch_size, ch, hh, ww, device=self.device),
                           torch.zeros(batch_size, ch, hh, ww, device=self.device)) for ch in self.hidden_dims[:self.n_layers]]
        self.H, self.C = [p[0] for p in self._pristine], [p[1] for p in self._pristine]

    def set_hidden(self, hidden):
        self.H, self.C = hidden
        self._pristine = []
