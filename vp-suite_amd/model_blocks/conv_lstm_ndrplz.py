"""ConvLSTM (Palazzi, Abati) — drop-in for vp_suite/model_blocks/conv_lstm_ndrplz.py:7-149: a single-step
`ConvLSTMCell` (gate split order i,f,o,g; no peephole; optional bias) and the multi-layer sequence block
(registered as `ConvLSTM_ndrplz` in the reference). The per-layer python time loop (:112-121) is one library call."""
import torch
from torch import nn

from .. import _lib, ops
from ..base import VPModelBlock


class ConvLSTMCell(nn.Module):
    precision = "f32"

    def __init__(self, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        self.conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size, padding=self.padding, bias=bias)

    def _run(self, x_seq, h, c, seq_len):
        return ops.convlstm_seq(x_seq, h, c, self.conv.weight, self.conv.bias, seq_len=seq_len,
                                in_channels=self.input_dim, gate_order=_lib.GATE_IFOG, precision=self.precision)

    def forward(self, input_tensor, cur_state):
        h_cur, c_cur = cur_state
        _, h_next, c_next = self._run(input_tensor.unsqueeze(1), h_cur, c_cur, 1)
        return h_next, c_next

    def init_hidden(self, batch_size, image_size):
        height, width = image_size
        dev = self.conv.weight.device
        return (torch.zeros(batch_size, self.hidden_dim, height, width, device=dev),
                torch.zeros(batch_size, self.hidden_dim, height, width, device=dev))


class ConvLSTM(VPModelBlock):
    NAME = "ConvLSTM (Palazzi, Abati)"
    CODE_REFERENCE = "https://github.com/ndrplz/ConvLSTM_pytorch"
    MATCHES_REFERENCE = "Yes (Code Reference)"

    def __init__(self, input_dim, hidden_dim, kernel_size, num_layers, batch_first=False, bias=True,
                 return_all_layers=False):
        super().__init__()
        self._check_kernel_size_consistency(kernel_size)
        kernel_size = self._extend_for_multilayer(kernel_size, num_layers)
        hidden_dim = self._extend_for_multilayer(hidden_dim, num_layers)
        if not len(kernel_size) == len(hidden_dim) == num_layers:
            raise ValueError('Inconsistent list length.')
        self.input_dim, self.hidden_dim, self.kernel_size = input_dim, hidden_dim, kernel_size
        self.num_layers, self.batch_first, self.bias = num_layers, batch_first, bias
        self.return_all_layers = return_all_layers
        self.cell_list = nn.ModuleList([
            ConvLSTMCell(input_dim if i == 0 else hidden_dim[i - 1], hidden_dim[i], kernel_size[i], bias)
            for i in range(num_layers)])

    def forward(self, input_tensor, hidden_state=None):
        if not self.batch_first:
            input_tensor = input_tensor.permute(1, 0, 2, 3, 4)  # (t,b,c,h,w) -> (b,t,c,h,w)
        if hidden_state is not None:
            raise NotImplementedError()
        seq_len = input_tensor.size(1)
        layer_outputs, last_states = [], []
        cur = input_tensor
        for cell in self.cell_list:
            # zero initial states (init_hidden) == absent states for the library
            cur, h, c = cell._run(cur, None, None, seq_len)
            layer_outputs.append(cur)
            last_states.append([h, c])
        if not self.return_all_layers:
            layer_outputs, last_states = layer_outputs[-1:], last_states[-1:]
        return layer_outputs, last_states

    def _init_hidden(self, batch_size, image_size):
        return [cell.init_hidden(batch_size, image_size) for cell in self.cell_list]

    @staticmethod
    def _check_kernel_size_consistency(kernel_size):
        ok = isinstance(kernel_size, tuple) or (isinstance(kernel_size, list) and
                                                all(isinstance(e, tuple) for e in kernel_size))
        if not ok:
            raise ValueError('`kernel_size` must be tuple or list of tuples')

    @staticmethod
    def _extend_for_multilayer(param, num_layers):
        return param if isinstance(param, list) else [param] * num_layers
