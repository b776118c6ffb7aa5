"""ctypes binding of libvpx_hip.so (C ABI: include/vpx.h). No torch types cross the boundary: only raw device
pointers (tensor.data_ptr()), sizes and the current HIP stream handle.

There is NO CPU fallback: if the shared library is missing or fails to load, every op raises."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VPX_LIB: developer override to A/B two builds of the library inside one GPU session (tools/ab_*); never a fallback
LIB_PATH = os.environ.get("VPX_LIB") or os.path.join(_HERE, "libvpx_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

GATE_IFGO, GATE_IFOG = 0, 1
LAYOUT_NHWC, LAYOUT_NCHW = 0, 1
PREC_F32, PREC_BF16X3, PREC_BF16 = 0, 1, 2
FLAG_SAVE_FOR_BWD = 1
FLAG_WEIGHTS_PACKED = 2
FLAG_X_SPLIT = 4
FLAG_OUT_SPLIT = 8

OPT_CELL2 = 1
OPT_CELL3 = 2
OPT_EXPERIMENT = 4
OPT_DRY_RUN = 5      # host-side work only, no HIP call (tests/test_workspace_contract.py)
OPT_MFMA_SHAPE = 3   # 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_16x16x32_bf16 in the second-generation kernels' main loop

EXPORTED_SYMBOLS = [
    "vpx_version", "vpx_last_error", "vpx_set_deterministic", "vpx_set_option", "vpx_option_epoch",
    "vpx_convlstm_workspace_bytes", "vpx_convlstm_reserve_bytes", "vpx_convlstm_takes_split_input", "vpx_convlstm_writes_split_output", "vpx_convlstm_seq_fwd",
    "vpx_convlstm_seq_bwd",
    "vpx_stlstm_workspace_bytes", "vpx_stlstm_reserve_bytes", "vpx_stlstm_step_fwd", "vpx_stlstm_step_bwd",
    "vpx_stlstm_uses_split", "vpx_stlstm_step_fwd_ex", "vpx_stlstm_step_bwd_ex",
    "vpx_stlstm_defers_wgrad", "vpx_stlstm_wgrad_batch_workspace_bytes", "vpx_stlstm_wgrad_batch",
    "vpx_decouple_workspace_bytes", "vpx_decouple_fwd", "vpx_decouple_bwd",
    "vpx_conv2d_workspace_bytes", "vpx_conv2d_nhwc_fwd", "vpx_conv2d_bwd_workspace_bytes", "vpx_conv2d_nhwc_bwd",
    "vpx_conv2d_ex_out_shape", "vpx_conv2d_ex_workspace_bytes", "vpx_conv2d_ex_fwd", "vpx_conv2d_ex_fwd_split",
    "vpx_conv2d_ex_takes_split", "vpx_split_convert", "vpx_conv2d_ex_split_workspace_bytes", "vpx_conv2d_ex_fwd_from_split",
    "vpx_conv2d_ex_bwd_workspace_bytes", "vpx_conv2d_ex_bwd", "vpx_conv2d_ex_bwd_uses_split", "vpx_conv2d_ex_bwd_ex",
    "vpx_conv2d_nhwc_fwd_ex",
    "vpx_acstlstm_workspace_bytes", "vpx_acstlstm_reserve_bytes", "vpx_acstlstm_step_fwd", "vpx_acstlstm_step_bwd",
    "vpx_trajgru_workspace_bytes", "vpx_trajgru_reserve_bytes", "vpx_trajgru_seq_fwd", "vpx_trajgru_seq_bwd",
    "vpx_nchw_to_nhwc", "vpx_nhwc_to_nchw",
    "vpx_layernorm_workspace_bytes", "vpx_layernorm_fwd", "vpx_layernorm_bwd",
    "vpx_mse_loss_workspace_bytes", "vpx_mse_loss", "vpx_adam_step",
]


class ConvLSTMDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "T", "Cin", "Ch", "H", "W", "kh", "kw", "gate_order", "layout",
                                              "precision", "flags")]


class STLSTMDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "Cin", "Ch", "H", "W", "k", "layer_norm", "layout", "precision",
                                              "flags")]


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "H", "W", "Ci", "Co", "kh", "kw", "stride", "pad", "transposed")] + \
               [("leaky_slope", ctypes.c_float), ("precision", ctypes.c_int32), ("out_pad_h", ctypes.c_int32),
                ("out_pad_w", ctypes.c_int32)]


class ACSTLSTMDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "Cin", "Ch", "H", "W", "k", "layer_norm", "precision", "flags")] + [("forget_bias", ctypes.c_float)]


class TrajGRUDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "T", "Cin", "C", "H", "W", "L", "k_i2h", "precision", "flags")] + [("slope", ctypes.c_float)]


class STLSTMShadows(ctypes.Structure):
    """vpx_stlstm_shadows: split-format copies of (x, h, m, c_new, m_new) handed in, buffers for (h_new, c_new, m_new) handed out."""
    _fields_ = [("inp", ctypes.c_void_p * 5), ("out", ctypes.c_void_p * 3), ("dg8_out", ctypes.c_void_p)]


class VpxError(RuntimeError):
    pass


_lib = None


def build(force: bool = False, jobs: int = 4) -> str:
    """Compiles every HIP source under csrc/ for gfx950 into vp-suite_amd/libvpx_hip.so (hipcc cross-compiles
    without a GPU). No-op when the library is newer than all sources."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "vpx.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC_DIR, "-s", f"-j{jobs}"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VpxError(f"HIP extension not built: {LIB_PATH} is missing (run `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` or `make -C {CSRC_DIR}`). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        sz = ctypes.c_size_t
        L.vpx_version.restype = ctypes.c_int
        L.vpx_last_error.restype = ctypes.c_char_p
        L.vpx_set_deterministic.restype = ctypes.c_int
        L.vpx_set_deterministic.argtypes = [ctypes.c_int]
        L.vpx_option_epoch.restype = ctypes.c_int
        L.vpx_option_epoch.argtypes = []
        L.vpx_set_option.restype = ctypes.c_int
        L.vpx_set_option.argtypes = [ctypes.c_int, ctypes.c_int]
        for name in ("vpx_convlstm_workspace_bytes", "vpx_convlstm_reserve_bytes"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [ctypes.POINTER(ConvLSTMDesc)]
        for name in ("vpx_stlstm_workspace_bytes", "vpx_stlstm_reserve_bytes"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [ctypes.POINTER(STLSTMDesc)]
        L.vpx_convlstm_takes_split_input.restype = ctypes.c_int
        L.vpx_convlstm_takes_split_input.argtypes = [ctypes.POINTER(ConvLSTMDesc)]
        L.vpx_convlstm_writes_split_output.restype = ctypes.c_int
        L.vpx_convlstm_writes_split_output.argtypes = [ctypes.POINTER(ConvLSTMDesc)]
        L.vpx_convlstm_seq_fwd.restype = ctypes.c_int
        L.vpx_convlstm_seq_fwd.argtypes = [ctypes.POINTER(ConvLSTMDesc)] + [vp] * 11 + [vp, sz, vp, sz, vp]
        L.vpx_convlstm_seq_bwd.restype = ctypes.c_int
        L.vpx_convlstm_seq_bwd.argtypes = [ctypes.POINTER(ConvLSTMDesc)] + [vp] * 8 + [vp, sz] + [vp] * 11 + [vp, sz, vp]
        L.vpx_stlstm_uses_split.restype = ctypes.c_int
        L.vpx_stlstm_uses_split.argtypes = [ctypes.POINTER(STLSTMDesc)]
        L.vpx_stlstm_step_fwd.restype = ctypes.c_int
        L.vpx_stlstm_step_fwd.argtypes = [ctypes.POINTER(STLSTMDesc)] + [vp] * 9 + [vp] + [vp] * 5 + [vp, sz, vp, sz, vp]
        L.vpx_stlstm_step_bwd.restype = ctypes.c_int
        L.vpx_stlstm_step_bwd.argtypes = [ctypes.POINTER(STLSTMDesc)] + [vp] * 11 + [vp] + [vp, sz] + [vp] * 5 + [vp] * 9 + [vp] + [vp, sz, vp]
        L.vpx_stlstm_defers_wgrad.restype = ctypes.c_int
        L.vpx_stlstm_defers_wgrad.argtypes = [ctypes.POINTER(STLSTMDesc)]
        L.vpx_stlstm_wgrad_batch_workspace_bytes.restype = sz
        L.vpx_stlstm_wgrad_batch_workspace_bytes.argtypes = [ctypes.POINTER(STLSTMDesc)]
        L.vpx_stlstm_wgrad_batch.restype = ctypes.c_int
        L.vpx_stlstm_wgrad_batch.argtypes = [ctypes.POINTER(STLSTMDesc), vp, vp] + [vp] * 5 + [vp, sz, vp]
        L.vpx_stlstm_step_fwd_ex.restype = ctypes.c_int
        L.vpx_stlstm_step_fwd_ex.argtypes = L.vpx_stlstm_step_fwd.argtypes + [ctypes.POINTER(STLSTMShadows)]
        L.vpx_stlstm_step_bwd_ex.restype = ctypes.c_int
        L.vpx_stlstm_step_bwd_ex.argtypes = L.vpx_stlstm_step_bwd.argtypes + [ctypes.POINTER(STLSTMShadows)]
        L.vpx_decouple_workspace_bytes.restype = sz
        L.vpx_decouple_workspace_bytes.argtypes = [ctypes.c_int] * 4
        L.vpx_decouple_fwd.restype = ctypes.c_int
        L.vpx_decouple_fwd.argtypes = [vp] * 4 + [ctypes.c_int] * 5 + [vp, sz, vp]
        L.vpx_decouple_bwd.restype = ctypes.c_int
        L.vpx_decouple_bwd.argtypes = [vp] * 7 + [ctypes.c_int] * 5 + [vp, sz, vp]
        L.vpx_conv2d_workspace_bytes.restype = sz
        L.vpx_conv2d_workspace_bytes.argtypes = [ctypes.c_int] * 4
        L.vpx_conv2d_nhwc_fwd.restype = ctypes.c_int
        L.vpx_conv2d_nhwc_fwd.argtypes = [vp] * 4 + [ctypes.c_int] * 8 + [vp, sz, vp]
        L.vpx_conv2d_bwd_workspace_bytes.restype = sz
        L.vpx_conv2d_bwd_workspace_bytes.argtypes = [ctypes.c_int] * 7
        L.vpx_conv2d_nhwc_bwd.restype = ctypes.c_int
        L.vpx_conv2d_nhwc_bwd.argtypes = [vp] * 6 + [ctypes.c_int] * 8 + [vp, sz, vp]
        L.vpx_conv2d_ex_out_shape.restype = ctypes.c_int
        L.vpx_conv2d_ex_out_shape.argtypes = [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.vpx_conv2d_ex_workspace_bytes.restype = sz
        L.vpx_conv2d_ex_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
        L.vpx_conv2d_ex_fwd.restype = ctypes.c_int
        L.vpx_conv2d_ex_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 4 + [vp, sz, vp]
        L.vpx_conv2d_ex_bwd_workspace_bytes.restype = sz
        L.vpx_conv2d_ex_bwd_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
        L.vpx_conv2d_ex_bwd.restype = ctypes.c_int
        L.vpx_conv2d_ex_bwd.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 7 + [vp, sz, vp]
        L.vpx_conv2d_ex_bwd_uses_split.restype = ctypes.c_int
        L.vpx_conv2d_ex_bwd_uses_split.argtypes = [ctypes.POINTER(ConvDesc)]
        L.vpx_conv2d_ex_bwd_ex.restype = ctypes.c_int
        L.vpx_conv2d_ex_bwd_ex.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 8 + [vp, sz, vp]
        ll, fl, ci = ctypes.c_longlong, ctypes.c_float, ctypes.c_int
        L.vpx_conv2d_nhwc_fwd_ex.restype = ci
        L.vpx_conv2d_nhwc_fwd_ex.argtypes = [vp] * 4 + [ci] * 9 + [fl, vp, sz, vp]
        for name in ("vpx_acstlstm_workspace_bytes", "vpx_acstlstm_reserve_bytes"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [ctypes.POINTER(ACSTLSTMDesc)]
        L.vpx_acstlstm_step_fwd.restype = ci
        L.vpx_acstlstm_step_fwd.argtypes = [ctypes.POINTER(ACSTLSTMDesc)] + [vp] * 12 + [vp, sz, vp, sz, vp]
        L.vpx_acstlstm_step_bwd.restype = ci
        L.vpx_acstlstm_step_bwd.argtypes = [ctypes.POINTER(ACSTLSTMDesc)] + [vp] * 7 + [vp, sz] + [vp] * 12 + [vp, sz, vp]
        for name in ("vpx_trajgru_workspace_bytes", "vpx_trajgru_reserve_bytes"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [ctypes.POINTER(TrajGRUDesc)]
        L.vpx_trajgru_seq_fwd.restype = ci
        L.vpx_trajgru_seq_fwd.argtypes = [ctypes.POINTER(TrajGRUDesc), vp, vp, vp, vp, vp, sz, vp, sz, vp]
        L.vpx_trajgru_seq_bwd.restype = ci
        L.vpx_trajgru_seq_bwd.argtypes = [ctypes.POINTER(TrajGRUDesc), vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, sz, vp]
        L.vpx_conv2d_ex_fwd_split.restype = ci
        L.vpx_conv2d_ex_fwd_split.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 5 + [vp, sz, vp]
        L.vpx_conv2d_ex_takes_split.restype = ci
        L.vpx_conv2d_ex_takes_split.argtypes = [ctypes.POINTER(ConvDesc)]
        L.vpx_split_convert.restype = ci
        L.vpx_split_convert.argtypes = [vp, vp, ll, ci, vp]
        L.vpx_conv2d_ex_split_workspace_bytes.restype = sz
        L.vpx_conv2d_ex_split_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
        L.vpx_conv2d_ex_fwd_from_split.restype = ci
        L.vpx_conv2d_ex_fwd_from_split.argtypes = [ctypes.POINTER(ConvDesc), vp, ll, ll, ci, vp, vp, vp, vp, ci, vp, sz, vp]
        L.vpx_layernorm_workspace_bytes.restype = sz
        L.vpx_layernorm_workspace_bytes.argtypes = [ci]
        L.vpx_layernorm_fwd.restype = ci
        L.vpx_layernorm_fwd.argtypes = [vp] * 6 + [ci, ll, vp, sz, vp]
        L.vpx_layernorm_bwd.restype = ci
        L.vpx_layernorm_bwd.argtypes = [vp] * 7 + [ci, ci, ci, vp, sz, vp]
        L.vpx_mse_loss_workspace_bytes.restype = sz
        L.vpx_mse_loss_workspace_bytes.argtypes = []
        L.vpx_mse_loss.restype = ctypes.c_int
        L.vpx_mse_loss.argtypes = [vp, vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_float, vp, vp, vp, sz, vp]
        L.vpx_adam_step.restype = ctypes.c_int
        L.vpx_adam_step.argtypes = [vp] * 4 + [ctypes.c_longlong] + [ctypes.c_double] * 5 + [ctypes.c_int, ctypes.c_double, vp]
        for name in ("vpx_nchw_to_nhwc", "vpx_nhwc_to_nchw"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [vp, vp] + [ctypes.c_int] * 4 + [vp]
        _lib = L
    return _lib


def check(rc: int, what: str):
    """Maps library error codes to the exception types the reference raises for the same misuse
    (ValueError for shape/argument problems, e.g. predrnn_v2.py:136-137; NotImplementedError conv_lstm_ndrplz.py:100)."""
    if rc == 0:
        return
    msg = lib().vpx_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    if rc == -4:
        raise NotImplementedError(f"{what}: {msg}")
    raise VpxError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor or None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
