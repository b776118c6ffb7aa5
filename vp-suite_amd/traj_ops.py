"""TrajGRU over a sequence as ONE autograd Function on ONE library call each way (include/vpx.h: vpx_trajgru_seq_fwd / _bwd): the
reference's python time loop (vp_suite/model_blocks/traj_gru.py:164-214) and its autograd graph are an explicit forward / BPTT
schedule of launches INSIDE the library — per step two 5x5 convolutions summed into the flow features (+ LeakyReLU), the 5x5 flow
convolution, the L bilinear warps (csrc/trajgru.hip), the 1x1 `ret` convolution and the GRU gate kernel; the input projection i2h
runs once over all frames. torch only allocates (outputs, reserve, one workspace sized by the library's query) and transposes the
input to time-major once.

Memory layout: every activation is NHWC and TIME-MAJOR ([T][B][H*W][C]) so that a time slice is a dense batch of images; the
returned sequence is a strided view [B,T,C,H,W] of that slab (no torch.stack)."""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr
from .ops import PRECISIONS, _require_gpu, _stream, _sync_determinism

FLOW_FEATURES = 32  # channels of the flow generator's hidden layer (traj_gru.py:108-122, fixed by the reference)


def _nhwc(t):
    """[..., C, H, W] logical -> dense [..., H, W, C] memory (a view when the tensor already is channels-last)."""
    nd = t.dim()
    return t.permute(*range(nd - 3), nd - 2, nd - 1, nd - 3).contiguous()


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _TrajGRUSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, i2h_w, i2h_b, i2f_w, i2f_b, h2f_w, h2f_b, fl_w, fl_b, ret_w, ret_b, seq_len, L, slope, precision,
                need_grad, state_hw):
        ref = x if x is not None else h0
        _require_gpu(ref, "trajgru_seq")
        dev = ref.device
        C = ret_w.shape[0] // 3
        T = int(seq_len)
        if x is not None:
            B, Tx, Cin, H, W = x.shape
            if Tx != T:
                raise ValueError(f"trajgru_seq: input holds {Tx} frames, seq_len is {T}")
            x_tm = _nhwc(x.transpose(0, 1))                      # [T,B,H,W,Cin] dense: one transposing copy of the input
        else:
            B, _, H, W = h0.shape
            Cin, x_tm = int(i2f_w.shape[1]), None
        if (H, W) != tuple(state_hw):
            raise ValueError(f"trajgru_seq: feature map {H}x{W} does not match the block's state size {tuple(state_hw)}")
        params = [t.contiguous() for t in (i2h_w, i2h_b, i2f_w, i2f_b, h2f_w, h2f_b, fl_w, fl_b, ret_w, ret_b)]
        d = _lib.TrajGRUDesc(B, T, Cin, C, H, W, int(L), int(i2h_w.shape[-1]), precision, _lib.FLAG_SAVE_FOR_BWD if need_grad else 0, float(slope))
        Lb = _lib.lib()
        ws_bytes = Lb.vpx_trajgru_workspace_bytes(ctypes.byref(d))
        if ws_bytes == 0:
            check(-4 if b"must be" in Lb.vpx_last_error() else -1, "vpx_trajgru_workspace_bytes")
        rs_bytes = Lb.vpx_trajgru_reserve_bytes(ctypes.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        reserve = torch.empty(max(rs_bytes, 1), dtype=torch.uint8, device=dev)
        h_init = _nhwc(h0) if h0 is not None else None
        hs = torch.empty(T, B, H, W, C, device=dev, dtype=torch.float32)   # h_1 .. h_T, time-major
        check(Lb.vpx_trajgru_seq_fwd(ctypes.byref(d), ptr(x_tm), ptr(h_init), _ptr_array(params), ptr(hs), ptr(reserve), rs_bytes,
                                     ptr(ws), ws_bytes, _stream()), "vpx_trajgru_seq_fwd")
        out = hs.permute(1, 0, 4, 2, 3)                          # [B,T,C,H,W] view of the time-major slab
        hT = hs[T - 1].permute(0, 3, 1, 2)
        if need_grad:
            ctx.save_for_backward(x_tm, h_init, hs, reserve, *params)
            ctx.desc, ctx.rs_bytes = d, rs_bytes
        return out, hT

    @staticmethod
    def backward(ctx, dout, dhT):
        _sync_determinism()
        x_tm, h_init, hs, reserve, *params = ctx.saved_tensors
        d = ctx.desc
        dev = hs.device
        Lb = _lib.lib()
        B, T, Cin, C, H, W = d.B, d.T, d.Cin, d.C, d.H, d.W
        # gradient of the output sequence, time-major NHWC (a view when the consumer kept the slab's layout)
        dout_tm = None if dout is None else dout.permute(1, 0, 3, 4, 2).contiguous()
        dhT_n = None if dhT is None else _nhwc(dhT)
        have_x = x_tm is not None
        dparams = [torch.empty_like(p) if (have_x or i >= 4) else None for i, p in enumerate(params)]
        dx_tm = torch.empty(T, B, H, W, Cin, device=dev, dtype=torch.float32) if (have_x and ctx.needs_input_grad[0]) else None
        dh0_n = torch.empty(B, H, W, C, device=dev, dtype=torch.float32) if (h_init is not None and ctx.needs_input_grad[1]) else None
        ws_bytes = Lb.vpx_trajgru_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(Lb.vpx_trajgru_seq_bwd(ctypes.byref(d), ptr(x_tm), ptr(h_init), _ptr_array(params), ptr(hs), ptr(reserve), ctx.rs_bytes,
                                     ptr(dout_tm), ptr(dhT_n), ptr(dx_tm), ptr(dh0_n), _ptr_array(dparams), ptr(ws), ws_bytes, _stream()),
              "vpx_trajgru_seq_bwd")
        dx = None if dx_tm is None else dx_tm.permute(1, 0, 4, 2, 3)
        dh0 = None if dh0_n is None else dh0_n.permute(0, 3, 1, 2)
        return (dx, dh0, *dparams, None, None, None, None, None, None)


def trajgru_seq(x, h0, params, *, seq_len, L, slope, state_hw, precision="f32"):
    """x: [B,T,Cin,H,W] or None; h0: [B,C,H,W] or None (zero state; not both None). params: the block's ten tensors in the
    order (i2h, i2f_conv1, h2f_conv1, flows_conv, ret) x (weight, bias). Returns (out [B,T,C,H,W], h_T [B,C,H,W])."""
    if x is None and h0 is None:
        raise ValueError("TrajGRU received 'None' both in input and state")
    if x is not None and x.shape[1] > seq_len:
        x = x[:, :seq_len]
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, h0, *params))
    return _TrajGRUSeqFn.apply(x, h0, *params, int(seq_len), int(L), float(slope), PRECISIONS[precision], need_grad, tuple(state_hw))
