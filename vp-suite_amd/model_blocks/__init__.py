"""Model blocks with the reference's registry names (vp_suite/model_blocks/__init__.py)."""
from .conv_lstm_hzzone import ConvLSTM  # noqa: F401
from .conv_lstm_ndrplz import ConvLSTM as ConvLSTM_ndrplz  # noqa: F401
from .conv_lstm_ndrplz import ConvLSTMCell  # noqa: F401
from .predrnn import ActionConditionalSpatioTemporalLSTMCell, SpatioTemporalLSTMCell  # noqa: F401
from .phydnet import SingleStepConvLSTM  # noqa: F401
from .traj_gru import TrajGRU  # noqa: F401
