"""ctypes wrapper around oracle/libvpx_oracle.so (plain-C restatement, oracle/vpx_oracle.c). numpy in / numpy out.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvpx_oracle.so")
_lib = None

GATE_IFGO = 0  # hzzone (conv_lstm_hzzone.py:62)
GATE_IFOG = 1  # ndrplz (conv_lstm_ndrplz.py:34)


def build(force: bool = False):
    src = os.path.join(_HERE, "vpx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvpx_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_decouple_fwd.restype = ctypes.c_double
    return _lib


def _f(a):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def convlstm_seq_fwd(x, h0, c0, W, b, Wci=None, Wcf=None, Wco=None, *, B, T, Cin, Ch, H, Wd, kh, kw,
                     gate_order=GATE_IFGO, save=False):
    """x [B,T,Cin,H,W] or None; returns dict(out,hT,cT[,gates,cs])."""
    x, h0, c0, W, b, Wci, Wcf, Wco = map(_f, (x, h0, c0, W, b, Wci, Wcf, Wco))
    out = np.empty((B, T, Ch, H, Wd), np.float32)
    hT = np.empty((B, Ch, H, Wd), np.float32)
    cT = np.empty((B, Ch, H, Wd), np.float32)
    gates = np.empty((T, B, 4 * Ch, H, Wd), np.float32) if save else None
    cs = np.empty((T, B, Ch, H, Wd), np.float32) if save else None
    rc = lib().orc_convlstm_seq_fwd(_p(x), _p(h0), _p(c0), _p(W), _p(b), _p(Wci), _p(Wcf), _p(Wco), _p(out), _p(hT),
                                    _p(cT), _p(gates), _p(cs), B, T, Cin, Ch, H, Wd, kh, kw, gate_order)
    assert rc == 0
    r = dict(out=out, hT=hT, cT=cT)
    if save:
        r.update(gates=gates, cs=cs)
    return r


def convlstm_seq_bwd(x, h0, c0, W, Wci, Wcf, Wco, fwd, dout, dhT, dcT, *, B, T, Cin, Ch, H, Wd, kh, kw,
                     gate_order=GATE_IFGO):
    x, h0, c0, W, Wci, Wcf, Wco, dout, dhT, dcT = map(_f, (x, h0, c0, W, Wci, Wcf, Wco, dout, dhT, dcT))
    Ct = Cin + Ch
    r = dict(dx=np.zeros((B, T, Cin, H, Wd), np.float32), dh0=np.empty((B, Ch, H, Wd), np.float32),
             dc0=np.empty((B, Ch, H, Wd), np.float32), dW=np.empty((4 * Ch, Ct, kh, kw), np.float32),
             db=np.empty((4 * Ch,), np.float32), dWci=np.empty((1, Ch, H, Wd), np.float32),
             dWcf=np.empty((1, Ch, H, Wd), np.float32), dWco=np.empty((1, Ch, H, Wd), np.float32))
    rc = lib().orc_convlstm_seq_bwd(_p(x), _p(h0), _p(c0), _p(W), _p(Wci), _p(Wcf), _p(Wco), _p(fwd["out"]),
                                    _p(fwd["gates"]), _p(fwd["cs"]), _p(dout), _p(dhT), _p(dcT), _p(r["dx"]),
                                    _p(r["dh0"]), _p(r["dc0"]), _p(r["dW"]), _p(r["db"]), _p(r["dWci"]),
                                    _p(r["dWcf"]), _p(r["dWco"]), B, T, Cin, Ch, H, Wd, kh, kw, gate_order)
    assert rc == 0
    return r


def stlstm_step_fwd(x, h, c, m, Wx, Wh, Wm, Wo, Wlast, ln=None, *, B, Cin, Ch, H, Wd, k):
    """ln: None or dict with keys x_g,x_b,h_g,h_b,m_g,m_b,o_g,o_b. Returns (h_new,c_new,m_new,delta_c,delta_m)."""
    x, h, c, m, Wx, Wh, Wm, Wo, Wlast = map(_f, (x, h, c, m, Wx, Wh, Wm, Wo, Wlast))
    lnp = [None] * 8
    if ln is not None:
        lnp = [_f(ln[k_]) for k_ in ("x_g", "x_b", "h_g", "h_b", "m_g", "m_b", "o_g", "o_b")]
    outs = [np.empty((B, Ch, H, Wd), np.float32) for _ in range(5)]
    rc = lib().orc_stlstm_step_fwd(_p(x), _p(h), _p(c), _p(m), _p(Wx), _p(Wh), _p(Wm), _p(Wo), _p(Wlast),
                                   *[_p(a) for a in lnp], *[_p(o) for o in outs], B, Cin, Ch, H, Wd, k)
    assert rc == 0
    return tuple(outs)


def decouple_fwd(delta_c, delta_m, adapter, *, B, Ch, HW):
    delta_c, delta_m, adapter = map(_f, (delta_c, delta_m, adapter))
    return float(lib().orc_decouple_fwd(_p(delta_c), _p(delta_m), _p(adapter), B, Ch, HW))


def conv2d(x, w, bias, *, stride=1, ph=0, pw=0):
    x, w, bias = map(_f, (x, w, bias))
    B, Ci, H, Wd = x.shape
    Co, _, kh, kw = w.shape
    Ho = (H + 2 * ph - kh) // stride + 1
    Wo = (Wd + 2 * pw - kw) // stride + 1
    y = np.empty((B, Co, Ho, Wo), np.float32)
    lib().orc_conv2d(_p(x), _p(w), _p(bias), _p(y), B, Ci, H, Wd, Co, kh, kw, stride, ph, pw, 0)
    return y
