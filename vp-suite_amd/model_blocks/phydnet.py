"""PhyDNet's ConvLSTM branch — drop-in for `SingleStepConvLSTM` (vp_suite/model_blocks/phydnet.py:117-175): a stack of
`ConvLSTMCell`s (conv_lstm_ndrplz.py:7-48, bias=True) that consumes ONE frame per call and keeps its (H, C) lists
between calls. Every layer step is one fused cell launch of the library (`vpx_convlstm_seq_fwd`, T = 1, gate order
i,f,o,g). The PhyCell / encoder-decoder parts of PhyDNet are outside the hot path."""
import torch
from torch import nn

from .conv_lstm_ndrplz import ConvLSTMCell


class SingleStepConvLSTM(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dims, n_layers, kernel_size, action_conditional, action_size, device):
        super().__init__()
        self.input_size = input_size
        self.input_dim = input_dim
        self.hidden_dims = hidden_dims
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.H, self.C = [], []
        self.action_size = action_size
        self.action_conditional = action_conditional
        self.device = device
        cells = []
        cur_input_dim = self.input_dim + (self.action_size if self.action_conditional else 0)
        for i in range(self.n_layers):
            cells.append(ConvLSTMCell(input_dim=cur_input_dim, hidden_dim=self.hidden_dims[i],
                                      kernel_size=self.kernel_size, bias=True))
            cur_input_dim = self.hidden_dims[i]
        self.cell_list = nn.ModuleList(cells)

    def forward(self, frame, action, first_timestep=False):
        batch_size = frame.size(0)
        if first_timestep:
            self.init_hidden(batch_size)  # init Hidden at each forward start (phydnet.py:146-148)
        inp = frame
        if self.action_conditional:
            inflated_action = action.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, *self.input_size)
            inp = torch.cat([inp, inflated_action], dim=-3)
        for j, cell in enumerate(self.cell_list):
            self.H[j], self.C[j] = cell(inp if j == 0 else self.H[j - 1], (self.H[j], self.C[j]))
        return (self.H, self.C), self.H  # (hidden, output)

    def init_hidden(self, batch_size):
        self.H = [torch.zeros(batch_size, hd, self.input_size[0], self.input_size[1], device=self.device)
                  for hd in self.hidden_dims[:self.n_layers]]
        self.C = [torch.zeros_like(h) for h in self.H]

    def set_hidden(self, hidden):
        H, C = hidden
        self.H, self.C = H, C
