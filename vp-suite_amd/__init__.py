"""vp_suite_amd — MI355X-native implementation of vp-suite's ConvLSTM / ST-LSTM recurrent hot path.

Host side mirrors the reference's Python surface (VPModel / model blocks); the per-timestep cell runs as hand-written
HIP kernels for gfx950 behind the C ABI of include/vpx.h (libvpx_hip.so). No CPU fallback exists."""
from . import _lib  # noqa: F401
from ._lib import VpxError, build as build_extension  # noqa: F401
from . import ops  # noqa: F401

__version__ = "0.1.0"
