"""Case tables and seeded input builders of the golden fixtures (shared by tools/gen_golden.py and the tests).
Pure data + torch CPU RNG; no reference code."""
import numpy as np

from golden_util import name_seed, seeded_rand, seeded_randn

HZZONE_CASES = {
    # tag: (Cin, Ch, H, W, k, B, T, with_grads)
    "tiny": (3, 8, 12, 10, 3, 2, 4, True),
    "k5": (4, 6, 9, 11, 5, 2, 3, True),
    "mid": (16, 32, 16, 12, 3, 2, 3, False),
}


def hzzone_inputs(tag, Cin, Ch, H, W, k, B, T):
    """Regenerable inputs/params of a hzzone block case (also used by the tests)."""
    p = f"hzzone.{tag}."
    fan = (Cin + Ch) * k * k
    d = {
        "W": seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(p + "W"), 1.0 / np.sqrt(fan)),
        "b": seeded_randn((4 * Ch,), name_seed(p + "b"), 0.1),
        "Wci": seeded_randn((1, Ch, H, W), name_seed(p + "Wci"), 0.1),
        "Wcf": seeded_randn((1, Ch, H, W), name_seed(p + "Wcf"), 0.1),
        "Wco": seeded_randn((1, Ch, H, W), name_seed(p + "Wco"), 0.1),
        "x": seeded_rand((B, T, Cin, H, W), name_seed(p + "x")),
        "h0": seeded_randn((B, Ch, H, W), name_seed(p + "h0"), 0.5),
        "c0": seeded_randn((B, Ch, H, W), name_seed(p + "c0"), 0.5),
        "g_out": seeded_randn((B, T, Ch, H, W), name_seed(p + "g_out")),
        "g_hT": seeded_randn((B, Ch, H, W), name_seed(p + "g_hT")),
        "g_cT": seeded_randn((B, Ch, H, W), name_seed(p + "g_cT")),
    }
    return d


NDRPLZ_CELL_CASES = {
    # tag: (Cin, Ch, H, W, kh, kw, bias, B)
    "k33": (3, 8, 13, 11, 3, 3, True, 2),
    "k53_nobias": (5, 6, 9, 14, 5, 3, False, 2),
}
NDRPLZ_SEQ_CASES = {
    # tag: (Cin, hidden_dims, kernel_sizes, H, W, B, T, bias, batch_first)   (odd frame sizes as in
    # tests/test_impl_match/_convlstm_ndrplz.py:45, scaled down)
    "l3": (3, [8, 8, 6], [(3, 3), (3, 3), (3, 3)], 13, 11, 2, 4, True, True),
    "l2_tfirst": (2, [6, 4], [(5, 5), (3, 3)], 10, 9, 2, 3, True, False),
}


def ndrplz_cell_inputs(tag, Cin, Ch, H, W, kh, kw, bias, B):
    p = f"ndrplz_cell.{tag}."
    fan = (Cin + Ch) * kh * kw
    return {
        "W": seeded_randn((4 * Ch, Cin + Ch, kh, kw), name_seed(p + "W"), 1.0 / np.sqrt(fan)),
        "b": seeded_randn((4 * Ch,), name_seed(p + "b"), 0.1),
        "x": seeded_rand((B, Cin, H, W), name_seed(p + "x")),
        "h": seeded_randn((B, Ch, H, W), name_seed(p + "h"), 0.5),
        "c": seeded_randn((B, Ch, H, W), name_seed(p + "c"), 0.5),
        "g_h": seeded_randn((B, Ch, H, W), name_seed(p + "g_h")),
        "g_c": seeded_randn((B, Ch, H, W), name_seed(p + "g_c")),
    }


STLSTM_CASES = {
    # tag: (Cin, Ch, H, W, k, layer_norm, B)
    "plain": (6, 8, 10, 9, 5, False, 2),
    "ln": (6, 8, 10, 9, 5, True, 2),
    "k3": (16, 16, 8, 8, 3, False, 2),
}


def stlstm_inputs(tag, Cin, Ch, H, W, B):
    p = f"stlstm.{tag}."
    d = {n: seeded_randn((B, Ch, H, W), name_seed(p + n), 0.5) for n in ("h", "c", "m")}
    d["x"] = seeded_randn((B, Cin, H, W), name_seed(p + "x"), 0.5)
    for n in ("g_h", "g_c", "g_m", "g_dc", "g_dm"):
        d[n] = seeded_randn((B, Ch, H, W), name_seed(p + n))
    return d


EF_TINY_KW = dict(img_shape=(1, 16, 16), action_size=0, tensor_value_range=[0.0, 1.0],
                  enc_c=[2, 4, 4, 6, 6, 6], dec_c=[6, 6, 6, 6, 4, 2], final_conv_1_c=2)
EF_TINY3_KW = dict(img_shape=(3, 16, 24), action_size=0, tensor_value_range=[0.0, 1.0],
                   enc_c=[2, 4, 4, 6, 6, 6], dec_c=[6, 6, 6, 6, 4, 2], final_conv_1_c=2)


PRED_TINY_KW = dict(img_shape=(1, 16, 16), action_size=0, tensor_value_range=[0.0, 1.0], patch_size=4,
                    num_layers=2, num_hidden=[8, 8], filter_size=5)
PRED_TINY_LN_KW = dict(img_shape=(2, 16, 24), action_size=0, tensor_value_range=[0.0, 1.0], patch_size=2,
                       num_layers=3, num_hidden=[8, 8, 8], filter_size=3, layer_norm=True)


# action-conditional PredRNN-V2 (predrnn_v2.py:62-121, 178-221): tag -> extra kwargs on top of PRED_ACTION_KW
PRED_ACTION_KW = dict(img_shape=(1, 32, 32), action_size=3, tensor_value_range=[0.0, 1.0], patch_size=2, num_layers=2,
                      num_hidden=[8, 8], filter_size=5, action_conditional=True)
PRED_ACTION_CASES = {"residual": dict(residual_on_action_conv=True), "plain": dict(residual_on_action_conv=False),
                     "ln": dict(residual_on_action_conv=True, layer_norm=True)}

# PhyDNet SingleStepConvLSTM (phydnet.py:117-175): tag -> (input_size, input_dim, hidden_dims, n_layers, kernel, action_conditional, action_size, B, steps)
PHY_SSC_CASES = {
    "plain": ((12, 10), 4, [8, 6], 2, (3, 3), False, 0, 2, 3),
    "action": ((9, 11), 3, [6], 1, (3, 3), True, 2, 2, 3),
}

# ActionConditionalSpatioTemporalLSTMCell (predrnn.py:86-169): tag -> (Cin, Ch, H, W, k, layer_norm, B)
ACSTLSTM_CASES = {
    "plain": (5, 8, 9, 10, 5, False, 2),
    "ln": (4, 8, 8, 7, 3, True, 2),
}


def acstlstm_inputs(tag, Cin, Ch, H, W, B):
    p = f"acstlstm.{tag}."
    d = {"x": seeded_randn((B, Cin, H, W), name_seed(p + "x"))}
    for n in ("h", "c", "m", "a", "g_h", "g_c", "g_m", "g_dc", "g_dm"):
        d[n] = seeded_randn((B, Ch, H, W), name_seed(p + n), 0.5 if n in ("h", "c", "m", "a") else 1.0)
    return d

# EF_TrajGRU (ef_traj_gru.py) tiny model: kwargs, B, context, pred
EF_TRAJGRU_TINY_KW = dict(img_shape=(2, 16, 16), action_size=0, tensor_value_range=[0.0, 1.0],
                          enc_c=[4, 8, 8, 12, 12, 12], dec_c=[12, 12, 12, 12, 8, 4], final_conv_1_c=4,
                          enc_rnn_L=[3, 3, 3], dec_rnn_L=[3, 3, 3])

# TrajGRU (traj_gru.py:74-214): tag -> (in_c, enc_c, H, W, L, B, T, mode)   mode: "full" | "noinput"
TRAJGRU_CASES = {
    "full": (4, 8, 10, 9, 3, 2, 3, "full"),
    "noinput": (4, 8, 8, 8, 5, 2, 2, "noinput"),
}
