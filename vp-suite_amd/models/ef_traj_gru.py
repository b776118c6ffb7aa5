"""Encoder-Forecaster TrajGRU ("trajgru") — drop-in for vp_suite/models/precipitation_nowcasting/ef_traj_gru.py:8-128 on
the skeleton of ef_conv_lstm.py (same Encoder / Forecaster, same stage glue in libvpx_hip). Hyper-parameter names and
defaults are the reference's (:31-75); state_dict keys `encoder.rnn{n}.{i2h,i2f_conv1,h2f_conv1,flows_conv,ret}.*` etc."""
from collections import OrderedDict

from ..model_blocks.traj_gru import Activation, TrajGRU
from .ef_conv_lstm import Encoder_Forecaster


class EF_TrajGRU(Encoder_Forecaster):
    NAME = "EF-TrajGRU (Shi et al.)"
    PAPER_REFERENCE = "https://arxiv.org/abs/1706.03458"
    CODE_REFERENCE = "https://github.com/Hzzone/Precipitation-Nowcasting"
    MATCHES_REFERENCE: str = "Yes"

    # hyper-parameters (c=channels, k=kernel, s=stride, p=padding, d=dilation, z=zoneout): ef_traj_gru.py:31-75
    activation = Activation('leaky', negative_slope=0.2, inplace=True)
    num_layers = 3
    enc_c = [16, 64, 64, 96, 96, 96]
    dec_c = [96, 96, 96, 96, 64, 16]
    enc_conv_names = ["conv1_leaky_1", "conv2_leaky_1", "conv3_leaky_1"]
    enc_conv_k, enc_conv_s, enc_conv_p = [3, 3, 3], [1, 2, 2], [1, 1, 1]
    dec_conv_names = ["deconv1_leaky_1", "deconv2_leaky_1", "deconv3_leaky_1"]
    dec_conv_k, dec_conv_s, dec_conv_p = [4, 4, 3], [2, 2, 1], [1, 1, 1]
    enc_rnn_z = [0.0, 0.0, 0.0]
    enc_rnn_L = [13, 13, 13]
    enc_rnn_i2h_k = [(3, 3), (3, 3), (3, 3)]
    enc_rnn_i2h_s = [(1, 1), (1, 1), (1, 1)]
    enc_rnn_i2h_p = [(1, 1), (1, 1), (1, 1)]
    enc_rnn_h2h_k = [(5, 5), (5, 5), (3, 3)]
    enc_rnn_h2h_d = [(1, 1), (1, 1), (1, 1)]
    dec_rnn_z = [0.0, 0.0, 0.0]
    dec_rnn_L = [13, 13, 13]
    dec_rnn_i2h_k = [(3, 3), (3, 3), (3, 3)]
    dec_rnn_i2h_s = [(1, 1), (1, 1), (1, 1)]
    dec_rnn_i2h_p = [(1, 1), (1, 1), (1, 1)]
    dec_rnn_h2h_k = [(3, 3), (5, 5), (5, 5)]
    dec_rnn_h2h_d = [(1, 1), (1, 1), (1, 1)]
    final_conv_1_name, final_conv_1_c, final_conv_1_k, final_conv_1_s, final_conv_1_p = "identity", 16, 3, 1, 1
    final_conv_2_name, final_conv_2_k, final_conv_2_s, final_conv_2_p = "conv3_3", 1, 1, 0
    cell_precision = "f32"  #: arithmetic of the convolution kernels ("f32" | "bf16x3" | "bf16")

    def __init__(self, device, **model_kwargs):
        super().__init__(device, **model_kwargs)
        self.NON_CONFIG_VARS.extend(["activation"])

    def _block(self, side, n, c_in, c_state, h, w):
        g = lambda name: getattr(self, f"{side}_rnn_{name}")[n]  # noqa: E731
        blk = TrajGRU(device=self.device, in_c=c_in, enc_c=c_state, state_h=h, state_w=w, zoneout=g("z"), L=g("L"),
                      i2h_kernel=g("i2h_k"), i2h_stride=g("i2h_s"), i2h_pad=g("i2h_p"), h2h_kernel=g("h2h_k"),
                      h2h_dilate=g("h2h_d"), act_type=self.activation)
        blk.precision = self.cell_precision
        return blk

    def _build_encoder_decoder(self):
        enc_convs, enc_rnns, dec_convs, dec_rnns = [], [], [], []
        c_prev = self.img_c
        for n in range(self.num_layers):
            c_mid, c_out = self.enc_c[2 * n], self.enc_c[2 * n + 1]
            enc_convs.append(OrderedDict({self.enc_conv_names[n]: [c_prev, c_mid, self.enc_conv_k[n], self.enc_conv_s[n],
                                                                   self.enc_conv_p[n]]}))
            enc_rnns.append(self._block("enc", n, c_mid, c_out, self.enc_rnn_state_h[n], self.enc_rnn_state_w[n]))
            c_prev = c_out
        for n in range(self.num_layers):
            c_mid, c_out = self.dec_c[2 * n], self.dec_c[2 * n + 1]
            dec_rnns.append(self._block("dec", n, c_prev, c_mid, self.dec_rnn_state_h[n], self.dec_rnn_state_w[n]))
            spec = OrderedDict({self.dec_conv_names[n]: [c_mid, c_out, self.dec_conv_k[n], self.dec_conv_s[n], self.dec_conv_p[n]]})
            if n == self.num_layers - 1:
                spec[self.final_conv_1_name] = [c_out, self.final_conv_1_c, self.final_conv_1_k, self.final_conv_1_s,
                                                self.final_conv_1_p]
                spec[self.final_conv_2_name] = [self.final_conv_1_c, self.img_c, self.final_conv_2_k, self.final_conv_2_s,
                                                self.final_conv_2_p]
            dec_convs.append(spec)
            c_prev = c_out
        return enc_convs, enc_rnns, dec_convs, dec_rnns
