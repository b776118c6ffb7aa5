"""TrajGRU over a sequence as ONE autograd Function on the C ABI (include/vpx.h): the reference's python time loop
(vp_suite/model_blocks/traj_gru.py:164-214) and its autograd graph are replaced by an explicit forward / BPTT schedule of
library launches — per step two 5x5 convolutions summed into the flow features (+ LeakyReLU), the 5x5 flow convolution,
the L bilinear warps (csrc/trajgru.hip), the 1x1 `ret` convolution and the GRU gate kernel; the input projection i2h runs
once over all frames. No ATen compute op on the path (torch only allocates, and transposes the input to time-major once).

Memory layout: every activation is NHWC and TIME-MAJOR ([T][B][H*W][C]) so that a time slice is a dense batch of images;
the returned sequence is a strided view [B,T,C,H,W] of that slab (no torch.stack)."""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr
from .ops import PRECISIONS, _require_gpu, _stream, _sync_determinism


def _nhwc(t):
    """[..., C, H, W] logical -> dense [..., H, W, C] memory (a view when the tensor already is channels-last)."""
    nd = t.dim()
    return t.permute(*range(nd - 3), nd - 2, nd - 1, nd - 3).contiguous()


class _Lib:
    """Thin call helpers over raw NHWC buffers (all on torch's current stream)."""

    def __init__(self, dev, prec):
        self.L = _lib.lib()
        self.dev, self.prec = dev, prec
        self._ws = None

    def ws(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
        return self._ws

    def conv(self, x, w, b, y, N, H, W, Ci, Co, k, accumulate=False, leaky=0.0):
        nb = self.L.vpx_conv2d_workspace_bytes(Ci, Co, k, k)
        ws = self.ws(nb)
        check(self.L.vpx_conv2d_nhwc_fwd_ex(ptr(x), ptr(w), ptr(b), ptr(y), N, H, W, Ci, Co, k, k, self.prec, int(accumulate),
                                            float(leaky), ptr(ws), ws.numel(), _stream()), "vpx_conv2d_nhwc_fwd_ex")

    def conv_bwd(self, x, w, dy, dx, dw, db, N, H, W, Ci, Co, k):
        nb = self.L.vpx_conv2d_bwd_workspace_bytes(N, H, W, Ci, Co, k, k)
        ws = self.ws(nb)
        check(self.L.vpx_conv2d_nhwc_bwd(ptr(x), ptr(w), ptr(dy), ptr(dx), ptr(dw), ptr(db), N, H, W, Ci, Co, k, k, self.prec,
                                         ptr(ws), ws.numel(), _stream()), "vpx_conv2d_nhwc_bwd")

    def axpy(self, y, x):
        check(self.L.vpx_axpy(ptr(y), ptr(x), y.numel(), _stream()), "vpx_axpy")


FLOW_FEATURES = 32  # channels of the flow generator's hidden layer (traj_gru.py:108-122, fixed by the reference)


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
        HW, F = H * W, FLOW_FEATURES
        lib = _Lib(dev, precision)
        Lc = lib.L
        ws = [t.contiguous() for t in (i2h_w, i2h_b, i2f_w, i2f_b, h2f_w, h2f_b, fl_w, fl_b, ret_w, ret_b)]
        i2h_w, i2h_b, i2f_w, i2f_b, h2f_w, h2f_b, fl_w, fl_b, ret_w, ret_b = ws
        k_i2h = int(i2h_w.shape[-1])
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
        i2h_all = None
        if x_tm is not None:                                     # input projection of ALL frames in one launch (:171-173)
            i2h_all = new(T, B, HW, 3 * C)
            lib.conv(x_tm, i2h_w, i2h_b, i2h_all, T * B, H, W, Cin, 3 * C, k_i2h)
        h_init = _nhwc(h0) if h0 is not None else torch.zeros(B, H, W, C, device=dev)
        hs = new(T, B, H, W, C)                                  # h_1 .. h_T, time-major
        flows = new(T, B, HW, 2 * L)
        f1 = new(T, B, HW, F)
        h2h = new(T, B, HW, 3 * C)
        gsave = new(T, B, HW, 3 * C) if need_grad else None
        warped = new(B, HW, L * C)
        for t in range(T):
            prev = h_init if t == 0 else hs[t - 1]
            # flow generator (:134-146): f1 = leaky(i2f(x_t) + h2f(h_{t-1})), flows = conv5x5(f1)
            lib.conv(prev, h2f_w, h2f_b, f1[t], B, H, W, C, F, 5, leaky=0.0 if x_tm is not None else slope)
            if x_tm is not None:
                lib.conv(x_tm[t], i2f_w, i2f_b, f1[t], B, H, W, Cin, F, 5, accumulate=True, leaky=slope)
            lib.conv(f1[t], fl_w, fl_b, flows[t], B, H, W, F, 2 * L, 5)
            # L warps of h_{t-1} along -flow (:148-162, :184-187), then the 1x1 `ret` convolution (:188)
            check(Lc.vpx_trajgru_warp_fwd(ptr(prev), ptr(flows[t]), ptr(warped), B, H, W, C, L, _stream()), "vpx_trajgru_warp_fwd")
            lib.conv(warped, ret_w, ret_b, h2h[t], B, H, W, L * C, 3 * C, 1)
            # gates + state update (:190-203)
            check(Lc.vpx_trajgru_gates_fwd(ptr(i2h_all[t]) if i2h_all is not None else None, HW * 3 * C, ptr(h2h[t]), ptr(prev),
                                           ptr(hs[t]), ptr(gsave[t]) if need_grad else None, B, HW, C, 0, float(slope), _stream()),
                  "vpx_trajgru_gates_fwd")
        out = hs.permute(1, 0, 4, 2, 3)                          # [B,T,C,H,W] view of the time-major slab
        hT = hs[T - 1].permute(0, 3, 1, 2)
        if need_grad:
            ctx.save_for_backward(x_tm, h_init, hs, flows, f1, h2h, gsave, *ws)
            ctx.geo = (B, T, Cin, C, H, W, L, float(slope), precision, h0 is not None)
        return out, hT

    @staticmethod
    def backward(ctx, dout, dhT):
        _sync_determinism()
        x_tm, h_init, hs, flows, f1, h2h, gsave, i2h_w, i2h_b, i2f_w, i2f_b, h2f_w, h2f_b, fl_w, fl_b, ret_w, ret_b = ctx.saved_tensors
        B, T, Cin, C, H, W, L, slope, precision, has_h0 = ctx.geo
        dev = hs.device
        HW, F = H * W, FLOW_FEATURES
        lib = _Lib(dev, precision)
        Lc = lib.L
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
        zeros = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)  # noqa: E731
        k_i2h = int(i2h_w.shape[-1])
        # gradient of the output sequence, time-major NHWC (a view when the consumer kept the slab's layout)
        dout_tm = None if dout is None else dout.permute(1, 0, 3, 4, 2).contiguous()
        carry = _nhwc(dhT).clone() if dhT is not None else zeros(B, H, W, C)
        di2h_all = new(T, B, HW, 3 * C) if x_tm is not None else None
        dx_i2f = new(T, B, HW, Cin) if x_tm is not None else None
        dh2h, dwarped, warped = new(B, HW, 3 * C), new(B, HW, L * C), new(B, HW, L * C)
        dflows, df1, df1s, dh_tmp = new(B, HW, 2 * L), new(B, HW, F), new(B, HW, F), new(B, H, W, C)
        dprev = new(B, H, W, C)
        # parameter gradients accumulate over the steps
        g = {n: zeros(*t.shape) for n, t in (("i2f_w", i2f_w), ("h2f_w", h2f_w), ("fl_w", fl_w), ("ret_w", ret_w))}
        gb = {n: zeros(*t.shape) for n, t in (("f_b", i2f_b), ("fl_b", fl_b), ("ret_b", ret_b))}
        t_w = {n: torch.empty_like(v) for n, v in g.items()}
        t_b = {n: torch.empty_like(v) for n, v in gb.items()}
        lws = torch.empty(Lc.vpx_leaky_bwd_workspace_bytes(F), dtype=torch.uint8, device=dev)
        det = torch.are_deterministic_algorithms_enabled()
        det_ws = torch.empty(Lc.vpx_trajgru_warp_bwd_det_workspace_bytes(B, H, W, C), dtype=torch.uint8, device=dev) if det else None
        for t in range(T - 1, -1, -1):
            prev = h_init if t == 0 else hs[t - 1]
            if dout_tm is not None:
                lib.axpy(carry, dout_tm[t])                       # total gradient of h_t
            check(Lc.vpx_trajgru_gates_bwd(ptr(carry), ptr(h2h[t]), ptr(prev), ptr(gsave[t]),
                                           ptr(di2h_all[t]) if di2h_all is not None else None, HW * 3 * C, ptr(dh2h), ptr(dprev),
                                           B, HW, C, 0, slope, _stream()), "vpx_trajgru_gates_bwd")
            # `ret` (1x1) backward needs the warped operand again: recomputed (a streaming kernel) instead of stored for all t
            check(Lc.vpx_trajgru_warp_fwd(ptr(prev), ptr(flows[t]), ptr(warped), B, H, W, C, L, _stream()), "vpx_trajgru_warp_fwd")
            lib.conv_bwd(warped, ret_w, dh2h, dwarped, t_w["ret_w"], t_b["ret_b"], B, H, W, L * C, 3 * C, 1)
            lib.axpy(g["ret_w"], t_w["ret_w"]); lib.axpy(gb["ret_b"], t_b["ret_b"])
            if det:   # integer (order-independent) scatter: bit-reproducible under torch.use_deterministic_algorithms(True)
                check(Lc.vpx_trajgru_warp_bwd_det(ptr(prev), ptr(flows[t]), ptr(dwarped), ptr(dprev), ptr(dflows), B, H, W, C, L,
                                                  ptr(det_ws), det_ws.numel(), _stream()), "vpx_trajgru_warp_bwd_det")
            else:
                check(Lc.vpx_trajgru_warp_bwd(ptr(prev), ptr(flows[t]), ptr(dwarped), ptr(dprev), ptr(dflows), B, H, W, C, L, _stream()),
                      "vpx_trajgru_warp_bwd")
            lib.conv_bwd(f1[t], fl_w, dflows, df1, t_w["fl_w"], t_b["fl_b"], B, H, W, F, 2 * L, 5)
            lib.axpy(g["fl_w"], t_w["fl_w"]); lib.axpy(gb["fl_b"], t_b["fl_b"])
            check(Lc.vpx_leaky_bwd(ptr(df1), ptr(f1[t]), slope, ptr(df1s), ptr(t_b["f_b"]), B * HW, F, ptr(lws), lws.numel(), _stream()),
                  "vpx_leaky_bwd")
            lib.axpy(gb["f_b"], t_b["f_b"])                       # the same sum is the gradient of BOTH flow-feature biases
            lib.conv_bwd(prev, h2f_w, df1s, dh_tmp, t_w["h2f_w"], None, B, H, W, C, F, 5)
            lib.axpy(g["h2f_w"], t_w["h2f_w"]); lib.axpy(dprev, dh_tmp)
            if x_tm is not None:
                lib.conv_bwd(x_tm[t], i2f_w, df1s, dx_i2f[t], t_w["i2f_w"], None, B, H, W, Cin, F, 5)
                lib.axpy(g["i2f_w"], t_w["i2f_w"])
            carry, dprev = dprev, carry                           # gradient of h_{t-1} through this step
        dx = d_i2h_w = d_i2h_b = None
        if x_tm is not None:
            d_i2h_w, d_i2h_b = torch.empty_like(i2h_w), torch.empty_like(i2h_b)
            dx_tm = new(T, B, H, W, Cin)
            lib.conv_bwd(x_tm, i2h_w, di2h_all, dx_tm, d_i2h_w, d_i2h_b, T * B, H, W, Cin, 3 * C, k_i2h)
            lib.axpy(dx_tm, dx_i2f)
            dx = dx_tm.permute(1, 0, 4, 2, 3)
        dh0 = carry.permute(0, 3, 1, 2) if has_h0 else None
        have_x = x_tm is not None
        return (dx, dh0, d_i2h_w, d_i2h_b, g["i2f_w"] if have_x else None, gb["f_b"] if have_x else None, g["h2f_w"], gb["f_b"].clone(),
                g["fl_w"], gb["fl_b"], g["ret_w"], gb["ret_b"], None, None, None, None, None, None)


def trajgru_seq(x, h0, params, *, seq_len, L, slope, state_hw, precision="f32"):
    """x: [B,T,Cin,H,W] or None; h0: [B,C,H,W] or None (zero state; not both None). params: the block's ten tensors in the
    order (i2h, i2f_conv1, h2f_conv1, flows_conv, ret) x (weight, bias). Returns (out [B,T,C,H,W], h_T [B,C,H,W])."""
    if x is None and h0 is None:
        raise ValueError("TrajGRU received 'None' both in input and state")
    if x is not None and x.shape[1] > seq_len:
        x = x[:, :seq_len]
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, h0, *params))
    return _TrajGRUSeqFn.apply(x, h0, *params, int(seq_len), int(L), float(slope), PRECISIONS[precision], need_grad, tuple(state_hw))
