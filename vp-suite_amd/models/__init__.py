"""Model registry with the reference's keys (vp_suite/models/__init__.py:14-26) for the models on the hot path."""
from .ef_conv_lstm import EF_ConvLSTM, Encoder_Forecaster  # noqa: F401
from .ef_traj_gru import EF_TrajGRU  # noqa: F401
from .predrnn_v2 import PredRNN_V2  # noqa: F401

MODEL_CLASSES = {
    "convlstm-shi": EF_ConvLSTM,
    "predrnn-pp": PredRNN_V2,
    "trajgru": EF_TrajGRU,
}
AVAILABLE_MODELS = MODEL_CLASSES.keys()
