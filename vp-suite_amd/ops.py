"""torch-level operators over the C ABI (include/vpx.h). Tensors keep the reference's LOGICAL shapes
([B,T,C,H,W], [B,C,H,W]) but live channels-last in memory (NHWC) so that kernels get coalesced channel rows and the
glue convolutions of the models (MIOpen, channels_last) exchange activations with them without a copy."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, ConvLSTMDesc, STLSTMDesc, check, ptr

VpxError = _lib.VpxError
PRECISIONS = {"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}


class KernelProfile:
    """Opt-in HIP-event bracket around every fused-cell library call (used by bench.py for the live roofline numbers).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (start_event, end_event, algorithmic_flops, algorithmic_bytes, cell_launches, tag, wide_read_bytes)

    def summary(self):
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        # wide_read_bytes: the part of the algorithmic reads that reaches HBM as wide (128-byte, 16 B per lane) coalesced requests — the
        # epilogue's cell-state reads; the operand stages arrive as 64-byte segments by LDS-DMA. The two are tallied differently by
        # FETCH_SIZE on gfx950 (profiles/r06_fetch_calibration.md): the PMC summaries need the split to turn the counter into bytes.
        return dict(ms=ms, flops=sum(r[2] for r in self.records), bytes=sum(r[3] for r in self.records),
                    launches=sum(r[4] for r in self.records), wide_read_bytes=sum(r[6] for r in self.records if len(r) > 6))


PROFILE = None  # set to a KernelProfile() to collect
BWD_WEIGHT_PACK_REUSE = True   #: ST-LSTM backward: keep the transposed weight packs between the steps of one cell (tests switch it off)

def _require_gpu(t: torch.Tensor, what: str):
    _sync_determinism()
    if not t.is_cuda:
        raise _lib.VpxError(f"{what}: tensors must live on the GPU (got device '{t.device}'). The hot path runs only as "
                            f"HIP kernels on MI355X; there is no CPU fallback.")
    if t.dtype != torch.float32:
        raise ValueError(f"{what}: expected float32 tensors, got {t.dtype}")


_det_state = None


def _sync_determinism():
    """Mirrors torch.use_deterministic_algorithms() into the library (deterministic mode forbids the K-split
    convolutions' floating-point atomics, see include/vpx.h). Called at the start of every op, before any size query."""
    global _det_state
    det = torch.are_deterministic_algorithms_enabled()
    if det != _det_state:
        _lib.lib().vpx_set_deterministic(int(det))
        _det_state = det


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_channels_last(t: torch.Tensor) -> torch.Tensor:
    """Same logical tensor ([..., C, H, W]) with NHWC memory. No copy if it already is."""
    nd = t.dim()
    perm = list(range(nd - 3)) + [nd - 2, nd - 1, nd - 3]
    inv = list(range(nd - 3)) + [nd - 1, nd - 3, nd - 2]
    return t.permute(perm).contiguous().permute(inv)


class _WorkspaceCache:
    """Workspaces (and with them packed weights) kept between inference calls while the owning weight tensor is unchanged.
    LRU with a byte budget; an entry whose weight tensor died, changed version or moved, or whose kernel-option epoch is
    stale, is dropped on the next insert — a ragged last batch, a varying horizon or an A/B option toggle cannot pin
    HBM beyond the budget."""

    def __init__(self, budget_bytes=4 << 30):
        import collections
        self.budget, self.bytes, self.ents = budget_bytes, 0, collections.OrderedDict()

    def get(self, key, w, addr, need_bytes, dev, exact=True):
        ent = self.ents.get(key)
        if ent is None:
            return None
        ref, version, a, ws, _ = ent
        ok = ref() is w and version == w._version and a == addr and ws.device == dev and \
            (ws.numel() == need_bytes if exact else ws.numel() >= need_bytes)
        if not ok:
            self._drop(key)
            return None
        self.ents.move_to_end(key)
        return ws

    def put(self, key, w, addr, ws, epoch):
        import weakref
        self._drop(key)
        for k in [k for k, (ref, version, a, _, ep) in self.ents.items()
                  if ref() is None or ref()._version != version or ref().data_ptr() != a or ep != epoch]:
            self._drop(k)
        if ws.numel() > self.budget:   # larger than the whole budget: never kept (the caller re-packs every call)
            return
        while self.ents and self.bytes + ws.numel() > self.budget:
            self._drop(next(iter(self.ents)))
        try:
            self.ents[key] = (weakref.ref(w), w._version, addr, ws, epoch)
            self.bytes += ws.numel()
        except TypeError:
            pass

    def _drop(self, key):
        ent = self.ents.pop(key, None)
        if ent is not None:
            self.bytes -= ent[3].numel()

    def clear(self):
        self.ents.clear()
        self.bytes = 0

    def __len__(self):
        return len(self.ents)


_cl_cache = {}


def _cached_channels_last(p: torch.Tensor) -> torch.Tensor:
    """Channels-last copy of a parameter tensor, re-used while the SAME tensor object is unchanged (identity checked
    through a weak reference — a recycled address can never hit — plus (_version, data_ptr)): the peephole tensors
    [1,Ch,H,W] of a ConvLSTM block are converted once per optimizer step instead of once per call. Writes through
    `.data` do not bump `_version`; code doing that must call `ops.clear_layout_cache()`."""
    import weakref
    ent = _cl_cache.get(id(p))
    if ent is not None:
        ref, version, addr, cl = ent
        if ref() is p and version == p._version and addr == p.data_ptr():
            return cl
    if len(_cl_cache) > 256:   # drop what died or went stale first; only a cache full of live entries is cleared
        for k in [k for k, (ref, version, addr, _) in _cl_cache.items()
                  if ref() is None or ref()._version != version or ref().data_ptr() != addr]:
            del _cl_cache[k]
        if len(_cl_cache) > 256:
            _cl_cache.clear()
    cl = to_channels_last(p.detach())
    try:
        _cl_cache[id(p)] = (weakref.ref(p), p._version, p.data_ptr(), cl)
    except TypeError:
        pass
    return cl


def clear_layout_cache():
    _cl_cache.clear()
    _clstm_ws.clear()
    _convq_ws.clear()


def is_channels_last(t: torch.Tensor) -> bool:
    nd = t.dim()
    perm = list(range(nd - 3)) + [nd - 2, nd - 1, nd - 3]
    return t.permute(perm).is_contiguous()


def new_channels_last(shape, device, dtype=torch.float32) -> torch.Tensor:
    """Uninitialised tensor of logical shape [..., C, H, W] with NHWC memory."""
    *lead, C, H, W = shape
    nd = len(shape)
    inv = list(range(nd - 3)) + [nd - 1, nd - 3, nd - 2]
    return torch.empty(*lead, H, W, C, device=device, dtype=dtype).permute(inv)


class SplitActivation:
    """An activation sequence in the library's split-bf16 operand format (include/vpx.h) — what a stage's last convolution
    hands to the recurrent block in inference so that no fp32 copy is written and no conversion pass runs. Not a tensor:
    only `ops.convlstm_seq` consumes it. `shape` is the logical [B, T, C, H, W]."""

    def __init__(self, buf, shape):
        self.buf, self.shape, self.device = buf, tuple(shape), buf.device


def convlstm_takes_split(B, T, Cin, Ch, H, W, k, gate_order, precision):
    """True when convlstm_seq on this problem (inference) consumes a SplitActivation input."""
    d = ConvLSTMDesc(B, T, Cin, Ch, H, W, k, k, gate_order, _lib.LAYOUT_NHWC, PRECISIONS[precision], 0)
    return bool(_lib.lib().vpx_convlstm_takes_split_input(ctypes.byref(d)))


_clstm_ws = _WorkspaceCache()
_CLSTM_WS_CACHE_LIMIT = 1 << 30   # bytes: larger workspaces (large batches) are not kept alive — there the repack is noise


def _kernel_options():
    """Epoch of the library's kernel-selection switches (they change which packs a workspace holds)."""
    return _lib.lib().vpx_option_epoch()


class _ConvLSTMSeqFn(torch.autograd.Function):
    """out, hT, cT = ConvLSTM over T steps. Replaces the python time loop of conv_lstm_hzzone.py:52-70 /
    conv_lstm_ndrplz.py:112-121 by ONE library call (T fused conv+gate launches on the current stream)."""

    @staticmethod
    def forward(ctx, x, h0, c0, W, b, Wci, Wcf, Wco, seq_len, gate_order, precision, in_channels, need_grad, out_split=False):
        ref = x.buf if isinstance(x, SplitActivation) else (x if x is not None else h0)
        _require_gpu(ref, "convlstm_seq")
        dev = ref.device
        Ch = W.shape[0] // 4
        kh, kw = int(W.shape[2]), int(W.shape[3])
        Cin = int(in_channels)
        if W.shape[1] != Cin + Ch:
            raise ValueError(f"convlstm_seq: weight has {W.shape[1]} input channels, expected {Cin}+{Ch}")
        x_split = isinstance(x, SplitActivation)
        if x_split:
            B, T, cx, H, Wd = x.shape
            if cx != Cin or T != seq_len or need_grad:
                raise ValueError("convlstm_seq: a SplitActivation input must match (Cin, seq_len) exactly and is inference-only")
            x = x.buf
        elif x is not None:
            B, T, cx, H, Wd = x.shape
            if cx != Cin or T < seq_len:
                raise ValueError(f"convlstm_seq: input shape {tuple(x.shape)} does not match Cin={Cin}, T>={seq_len}")
            x = to_channels_last(x[:, :seq_len]) if T != seq_len else to_channels_last(x)
        else:
            B, _, H, Wd = h0.shape
        T = int(seq_len)
        h0c = None if h0 is None else to_channels_last(h0)
        c0c = None if c0 is None else to_channels_last(c0)
        peep = Wci is not None
        wci = _cached_channels_last(Wci) if peep else None
        wcf = _cached_channels_last(Wcf) if peep else None
        wco = _cached_channels_last(Wco) if peep else None
        Wc = W.contiguous()
        bc = None if b is None else b.contiguous()
        if out_split and need_grad:
            raise ValueError("convlstm_seq: out_split is inference-only")
        d = ConvLSTMDesc(B, T, Cin, Ch, H, Wd, kh, kw, gate_order, _lib.LAYOUT_NHWC, precision,
                         (_lib.FLAG_SAVE_FOR_BWD if need_grad else 0) | (_lib.FLAG_X_SPLIT if x_split else 0) |
                         (_lib.FLAG_OUT_SPLIT if out_split else 0))
        L = _lib.lib()
        ws_bytes = L.vpx_convlstm_workspace_bytes(ctypes.byref(d))
        rs_bytes = L.vpx_convlstm_reserve_bytes(ctypes.byref(d))
        if ws_bytes == 0:
            check(-1 if "not implemented" not in L.vpx_last_error().decode() else -4, "vpx_convlstm_workspace_bytes")
        ws = None
        if not need_grad and ws_bytes <= _CLSTM_WS_CACHE_LIMIT:
            # inference: the block keeps its workspace, and with it the weight packs, while the weight tensor is unchanged
            # (identity by weak reference + version + address; kernel selection switches are part of the key)
            key = (id(W), B, T, Cin, Ch, H, Wd, kh, kw, gate_order, precision, x is None, h0 is None, x_split, out_split,
                   torch.are_deterministic_algorithms_enabled(), _kernel_options())
            ws = _clstm_ws.get(key, W, Wc.data_ptr(), ws_bytes, dev)
            if ws is not None:
                d.flags |= _lib.FLAG_WEIGHTS_PACKED
            else:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                _clstm_ws.put(key, W, Wc.data_ptr(), ws, _kernel_options())
        if ws is None:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        reserve = torch.empty(max(rs_bytes, 1), dtype=torch.uint8, device=dev)
        if out_split:   # the sequence in operand format (no fp32 copy exists); h_T separately in fp32
            out = torch.empty(B * T * H * Wd * Ch, dtype=torch.float32, device=dev)
            hT = new_channels_last((B, Ch, H, Wd), dev)
        else:
            out = new_channels_last((B, T, Ch, H, Wd), dev)
            hT = out[:, T - 1]  # h_T IS the last slice of the output slab: a view, no copy
        cT = new_channels_last((B, Ch, H, Wd), dev)
        if PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), ptr(x), ptr(h0c), ptr(c0c), ptr(Wc), ptr(bc), ptr(wci), ptr(wcf),
                                    ptr(wco), ptr(out), ptr(hT) if out_split else None, ptr(cT), ptr(reserve), rs_bytes, ptr(ws),
                                    ws_bytes, _stream())
        check(rc, "vpx_convlstm_seq_fwd")
        if PROFILE is not None:
            ev1.record()
            fl, by = convlstm_algorithmic_work(B, T, Cin if x is not None else 0, Ch, H, Wd, kh, kw,
                                               h0 is not None, peep)
            wide = 4.0 * H * Wd * Ch * B * (T if h0 is not None else T - 1)   # c_{t-1}, read by the fused step's epilogue in full lines
            PROFILE.records.append((ev0, ev1, fl, by, T, "convlstm_fwd", wide))
        if need_grad:
            ctx.save_for_backward(x, h0c, c0c, Wc, wci, wcf, wco, out, reserve)
            ctx.desc = d
            ctx.has_bias = b is not None
            ctx.rs_bytes = rs_bytes
        if out_split:
            return SplitActivation(out, (B, T, Ch, H, Wd)), hT, cT
        return out, hT, cT

    @staticmethod
    def backward(ctx, dout, dhT, dcT):
        _sync_determinism()
        x, h0c, c0c, Wc, wci, wcf, wco, out, reserve = ctx.saved_tensors
        d = ctx.desc
        dev = out.device
        L = _lib.lib()
        B, T, Cin, Ch, H, Wd = d.B, d.T, d.Cin, d.Ch, d.H, d.W
        needs = ctx.needs_input_grad
        dout = None if dout is None else to_channels_last(dout)
        dhT = None if dhT is None else to_channels_last(dhT)
        dcT = None if dcT is None else to_channels_last(dcT)
        dx = new_channels_last((B, T, Cin, H, Wd), dev) if (x is not None and needs[0]) else None
        dh0 = new_channels_last((B, Ch, H, Wd), dev) if (h0c is not None and needs[1]) else None
        dc0 = new_channels_last((B, Ch, H, Wd), dev) if (c0c is not None and needs[2]) else None
        dW = torch.empty_like(Wc) if needs[3] else None
        db = torch.empty(4 * Ch, device=dev) if (ctx.has_bias and needs[4]) else None
        peep = wci is not None
        dwci = new_channels_last((1, Ch, H, Wd), dev) if (peep and needs[5]) else None
        dwcf = new_channels_last((1, Ch, H, Wd), dev) if (peep and needs[6]) else None
        dwco = new_channels_last((1, Ch, H, Wd), dev) if (peep and needs[7]) else None
        ws_bytes = L.vpx_convlstm_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = L.vpx_convlstm_seq_bwd(ctypes.byref(d), ptr(x), ptr(h0c), ptr(c0c), ptr(Wc), ptr(wci), ptr(wcf), ptr(wco),
                                    ptr(out), ptr(reserve), ctx.rs_bytes, ptr(dout), ptr(dhT), ptr(dcT), ptr(dx),
                                    ptr(dh0), ptr(dc0), ptr(dW), ptr(db), ptr(dwci), ptr(dwcf), ptr(dwco), ptr(ws),
                                    ws_bytes, _stream())
        check(rc, "vpx_convlstm_seq_bwd")
        return dx, dh0, dc0, dW, db, dwci, dwcf, dwco, None, None, None, None, None


def convlstm_algorithmic_work(B, T, Cin, Ch, H, W, kh, kw, has_h0=True, peephole=True, dt=4):
    """Algorithmic FLOPs and bytes of a ConvLSTM sequence call: per cell-step and sample
    flops = 2*4Ch*(Cin+Ch)*k^2*H*W over the operand ranges that are not identically zero;
    bytes = dt*H*W*(Cin + 2Ch + 2Ch) (read x,h,c; write h,c) + per-step shared weights/bias/peepholes."""
    flops = 0.0
    nbytes = 0.0
    for t in range(T):
        kc = Cin + (Ch if (t > 0 or has_h0) else 0)
        flops += 2.0 * 4 * Ch * kc * kh * kw * H * W * B
        nbytes += dt * H * W * (Cin + 4 * Ch) * B
        nbytes += dt * (4 * Ch * (Cin + Ch) * kh * kw + 4 * Ch + (3 * Ch * H * W if peephole else 0))
    return flops, nbytes


def convlstm_writes_split(B, T, Cin, Ch, H, W, k, gate_order, precision):
    """True when convlstm_seq on this problem (inference) can hand its output sequence out as a SplitActivation (out_split=True)."""
    d = ConvLSTMDesc(B, T, Cin, Ch, H, W, k, k, gate_order, _lib.LAYOUT_NHWC, PRECISIONS[precision], 0)
    return bool(_lib.lib().vpx_convlstm_writes_split_output(ctypes.byref(d)))


def convlstm_seq(x, h0, c0, W, b, Wci=None, Wcf=None, Wco=None, *, seq_len, in_channels, gate_order=_lib.GATE_IFGO,
                 precision="f32", out_split=False):
    """x: [B,T,Cin,H,W] or None; h0/c0: [B,Ch,H,W] or None (not both x and h0 None). Returns (out [B,T,Ch,H,W], hT, cT);
    out_split (inference, where convlstm_writes_split says so): `out` is a SplitActivation — the sequence in the library's operand
    format for a consumer that reads it (conv2d_ex_from_split); no fp32 copy of it is written."""
    if x is None and h0 is None:
        raise ValueError("convlstm_seq: inputs and states must not both be None")
    # grad mode is always off INSIDE Function.forward, so decide here whether the forward must fill the reserve
    need_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (x, h0, c0, W, b, Wci, Wcf, Wco))
    if isinstance(x, SplitActivation) or out_split:   # autograd.Function.apply only takes / returns tensors: call the forward body directly
        if need_grad:
            raise ValueError("convlstm_seq: split-format input / output is inference-only")

        class _Ctx:   # (inference only: nothing is saved)
            needs_input_grad = (False,) * 13
        if x is not None and not isinstance(x, SplitActivation) and x.dim() == 5 and x.shape[1] > seq_len:
            x = x[:, :seq_len]
        return _ConvLSTMSeqFn.forward(_Ctx(), x, h0, c0, W, b, Wci, Wcf, Wco, int(seq_len), int(gate_order), PRECISIONS[precision],
                                      int(in_channels), False, bool(out_split))
    if x is not None and x.dim() == 5 and x.shape[1] > seq_len:
        x = x[:, :seq_len]  # sliced here, outside the Function: autograd pads dx back to x's shape
    return _ConvLSTMSeqFn.apply(x, h0, c0, W, b, Wci, Wcf, Wco, int(seq_len), int(gate_order), PRECISIONS[precision],
                                int(in_channels), need_grad)


class _Conv2dSameFn(torch.autograd.Function):
    """Stride-1 'same' convolution through the library's implicit-GEMM kernel, with explicit backward
    (data gradient = same kernel with transposed/flipped packing, weight gradient = MFMA wgrad kernel)."""

    @staticmethod
    def forward(ctx, x, w, bias, precision):
        _require_gpu(x, "conv2d_same")
        xs = to_channels_last(x)
        N, Ci, H, Wd = xs.shape
        Co, ci_w, kh, kw = w.shape
        if ci_w != Ci:
            raise ValueError(f"conv2d_same: weight expects {ci_w} input channels, input has {Ci}")
        wc = w.contiguous()
        bc = None if bias is None else bias.contiguous()
        L = _lib.lib()
        ws_bytes = L.vpx_conv2d_workspace_bytes(Ci, Co, kh, kw)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        y = new_channels_last((N, Co, H, Wd), x.device)
        rc = L.vpx_conv2d_nhwc_fwd(ptr(xs), ptr(wc), ptr(bc), ptr(y), N, H, Wd, Ci, Co, kh, kw, precision, ptr(ws),
                                   ws_bytes, _stream())
        check(rc, "vpx_conv2d_nhwc_fwd")
        ctx.save_for_backward(xs, wc)
        ctx.has_bias = bias is not None
        ctx.precision = precision
        return y

    @staticmethod
    def backward(ctx, dy):
        _sync_determinism()
        xs, wc = ctx.saved_tensors
        N, Ci, H, Wd = xs.shape
        Co, _, kh, kw = wc.shape
        dys = to_channels_last(dy)
        needs = ctx.needs_input_grad
        dx = new_channels_last((N, Ci, H, Wd), xs.device) if needs[0] else None
        dw = torch.empty_like(wc) if needs[1] else None
        db = torch.empty(Co, device=xs.device) if (ctx.has_bias and needs[2]) else None
        L = _lib.lib()
        ws_bytes = L.vpx_conv2d_bwd_workspace_bytes(N, H, Wd, Ci, Co, kh, kw)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xs.device)
        rc = L.vpx_conv2d_nhwc_bwd(ptr(xs), ptr(wc), ptr(dys), ptr(dx), ptr(dw), ptr(db), N, H, Wd, Ci, Co, kh, kw,
                                   ctx.precision, ptr(ws), ws_bytes, _stream())
        check(rc, "vpx_conv2d_nhwc_bwd")
        return dx, dw, db, None


def conv2d_same(x, w, bias=None, precision="f32"):
    """y = conv2d(x, w, bias, stride=1, padding=k//2) on a [N,C,H,W] tensor (PredRNN's 1x1 frame head,
    predrnn_v2.py:223, and any other stride-1 'same' convolution); differentiable."""
    return _Conv2dSameFn.apply(x, w, bias, PRECISIONS[precision])




class _ConvExFn(torch.autograd.Function):
    """Conv2d / ConvTranspose2d (stride 1 or 2) + bias + LeakyReLU in ONE library launch (4 for a stride-2 transposed
    conv) — the EF stage glue of ef_blocks.py:15-49. Forward and backward (LeakyReLU', bias / data / weight gradients)
    run in libvpx_hip. A layer whose backward the library does not implement (kernel smaller than its stride, negative slope)
    raises VpxError in the FORWARD of a call that will need gradients — not after a whole forward pass has been spent."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, transposed, slope, precision, out_pad=(0, 0)):
        kh, kw = int(w.shape[2]), int(w.shape[3])
        _require_gpu(x, "conv2d_ex")
        xs = to_channels_last(x)
        N, Ci, H, Wd = xs.shape
        kh, kw = int(w.shape[2]), int(w.shape[3])
        Co = int(w.shape[1] if transposed else w.shape[0])
        if int(w.shape[0] if transposed else w.shape[1]) != Ci:
            raise ValueError(f"conv2d_ex: weight {tuple(w.shape)} does not match {Ci} input channels")
        wc = w.contiguous()
        bc = None if bias is None else bias.contiguous()
        d = ConvDesc(N, H, Wd, Ci, Co, kh, kw, int(stride), int(padding), int(bool(transposed)), float(slope), precision,
                     int(out_pad[0]), int(out_pad[1]))
        L = _lib.lib()
        ho, wo = ctypes.c_int(0), ctypes.c_int(0)
        check(L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), "vpx_conv2d_ex_out_shape")
        y = new_channels_last((N, Co, ho.value, wo.value), x.device)
        x_sp = None
        if (ctx.needs_input_grad[1] and tuple(out_pad) == (0, 0) and L.vpx_conv2d_ex_bwd_uses_split(ctypes.byref(d))
                and L.vpx_conv2d_ex_takes_split(ctypes.byref(d))):
            # a training call whose weight gradient will want x in the split operand format anyway (wgrad2.hip, glue form): convert once,
            # run the forward on the split-input kernels (convq / c16 / first generation without its in-kernel conversion) and keep the copy
            x_sp = torch.empty(xs.numel(), dtype=torch.float32, device=x.device)
            check(L.vpx_split_convert(ptr(xs), ptr(x_sp), N * H * Wd, Ci, _stream()), "vpx_split_convert")
            ws_bytes = L.vpx_conv2d_ex_split_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            check(L.vpx_conv2d_ex_fwd_from_split(ctypes.byref(d), ptr(x_sp), 0, 0, 1, ptr(wc), ptr(bc), ptr(y), None, 0, ptr(ws), ws_bytes,
                                                 _stream()), "vpx_conv2d_ex_fwd_from_split")
        else:
            ws_bytes = L.vpx_conv2d_ex_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            check(L.vpx_conv2d_ex_fwd(ctypes.byref(d), ptr(xs), ptr(wc), ptr(bc), ptr(y), ptr(ws), ws_bytes, _stream()),
                  "vpx_conv2d_ex_fwd")
        ctx.x_sp = x_sp
        ctx.save_for_backward(xs, wc, y)
        ctx.cfg = (int(stride), int(padding), bool(transposed), float(slope), bias is not None)
        ctx.desc = d
        return y

    @staticmethod
    def backward(ctx, dy):
        _sync_determinism()
        xs, wc, y = ctx.saved_tensors
        stride, padding, transposed, slope, has_bias = ctx.cfg
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]]
        d = ctx.desc
        if d.kh >= d.stride and d.kw >= d.stride and slope >= 0.0:
            # library path: LeakyReLU' + bias gradient in one pass over dy, dx = the adjoint layer on the implicit-GEMM kernel,
            # dw on the MFMA weight-gradient kernel
            L = _lib.lib()
            dyc = to_channels_last(dy)
            ws_bytes = L.vpx_conv2d_ex_bwd_workspace_bytes(ctypes.byref(d))
            if ws_bytes == 0:
                check(-4 if b"not implemented" in L.vpx_last_error() else -1, "vpx_conv2d_ex_bwd_workspace_bytes")
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
            dx = new_channels_last(tuple(xs.shape), dy.device) if mask[0] else None
            dw = torch.empty_like(wc) if mask[1] else None
            db = torch.empty(d.Co, device=dy.device) if mask[2] else None
            check(L.vpx_conv2d_ex_bwd_ex(ctypes.byref(d), ptr(xs), ptr(ctx.x_sp), ptr(wc), ptr(y), ptr(dyc), ptr(dx), ptr(dw), ptr(db), ptr(ws),
                                         ws_bytes, _stream()), "vpx_conv2d_ex_bwd")
            ctx.x_sp = None
            return dx, dw, db, None, None, None, None, None, None
        # No second backend in the product path: a layer the library's glue backward does not implement fails loudly.
        raise _lib.VpxError(f"conv2d_ex backward: layer (k={d.kh}x{d.kw}, stride={stride}, transposed={bool(transposed)}, slope={slope}) "
                       f"is outside vpx_conv2d_ex_bwd (needs kernel >= stride and a non-negative LeakyReLU slope): unsupported")


def conv2d_ex(x, w, bias, stride, padding, transposed=False, leaky_slope=0.0, precision="f32", output_padding=(0, 0)):
    """Conv2d / ConvTranspose2d (stride 1 or 2; `output_padding` for transposed layers) + bias + LeakyReLU, differentiable."""
    kh, kw = int(w.shape[2]), int(w.shape[3])
    # (decided here: inside Function.forward grad mode is always off and needs_input_grad ignores torch.no_grad())
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, bias)):
        ok = kh >= stride and kw >= stride and leaky_slope >= 0.0
        if ok and x.dim() == 4:   # ... and the adjoint layer must be one the library runs (the same query the backward makes)
            N, Ci, H, Wd = x.shape
            Co = int(w.shape[1] if transposed else w.shape[0])
            d = ConvDesc(N, H, Wd, Ci, Co, kh, kw, int(stride), int(padding), int(bool(transposed)), float(leaky_slope), PRECISIONS[precision],
                         int(output_padding[0]), int(output_padding[1]))
            L = _lib.lib()
            ho, wo = ctypes.c_int(0), ctypes.c_int(0)
            if L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == 0:
                ok = L.vpx_conv2d_ex_bwd_workspace_bytes(ctypes.byref(d)) != 0
        if not ok:
            raise _lib.VpxError(f"conv2d_ex: layer (k={kh}x{kw}, stride={stride}, padding={padding}, transposed={bool(transposed)}, slope={leaky_slope}) has no "
                                f"backward in the library (vpx_conv2d_ex_bwd needs kernel >= stride, a non-negative LeakyReLU slope and an adjoint layer "
                                f"it implements: padding <= kernel - 1): unsupported in a call that requires gradients")
    return _ConvExFn.apply(x, w, bias, stride, padding, transposed, leaky_slope, PRECISIONS[precision], tuple(output_padding))


def conv2d_ex_split(x, w, bias, stride, padding, transposed=False, leaky_slope=0.0, precision="f32"):
    """conv2d_ex in inference with the output ONLY in the split-bf16 operand format: returns (buffer, (N, Co, Ho, Wo)).
    The buffer has the byte size of the fp32 NHWC output it replaces."""
    _require_gpu(x, "conv2d_ex_split")
    xs = to_channels_last(x)
    N, Ci, H, Wd = xs.shape
    kh, kw = int(w.shape[2]), int(w.shape[3])
    Co = int(w.shape[1] if transposed else w.shape[0])
    d = ConvDesc(N, H, Wd, Ci, Co, kh, kw, int(stride), int(padding), int(bool(transposed)), float(leaky_slope),
                 PRECISIONS[precision], 0, 0)
    L = _lib.lib()
    ho, wo = ctypes.c_int(0), ctypes.c_int(0)
    check(L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), "vpx_conv2d_ex_out_shape")
    ws_bytes = L.vpx_conv2d_ex_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    buf = torch.empty(N * ho.value * wo.value * Co, dtype=torch.float32, device=x.device)
    wc = w.contiguous()
    bc = None if bias is None else bias.contiguous()
    check(L.vpx_conv2d_ex_fwd_split(ctypes.byref(d), ptr(xs), ptr(wc), ptr(bc), None, ptr(buf), ptr(ws), ws_bytes, _stream()),
          "vpx_conv2d_ex_fwd_split")
    return buf, (N, Co, ho.value, wo.value)


def split_convert(x, native=True):
    """fp32 [..., C, H, W] tensor -> SplitActivation-style buffer in the split-bf16 operand format (vpx_split_convert for a
    channels-last 4-D batch; otherwise — and with native=False, the tests' cross-check — the same rounding with torch ops).
    Returns (buffer, logical shape)."""
    _require_gpu(x, "split_convert")
    xs = to_channels_last(x)
    C = xs.shape[-3]
    if C % 8:
        raise ValueError("split_convert: the channel count must be a multiple of 8")
    if native and xs.dim() == 4 and xs.dtype == torch.float32 and xs.is_contiguous(memory_format=torch.channels_last):
        buf = torch.empty(xs.numel(), device=xs.device, dtype=torch.float32)
        n, _, h, w = xs.shape
        check(_lib.lib().vpx_split_convert(ptr(xs), ptr(buf), n * h * w, C, _stream()), "vpx_split_convert")
        return buf, tuple(x.shape)
    # fp32 hi/lo split on the GPU with torch ops (same rounding as the kernels: round-to-nearest-even to bf16, twice)
    flat = xs.permute(*range(xs.dim() - 3), xs.dim() - 2, xs.dim() - 1, xs.dim() - 3).contiguous()   # [..., H, W, C]
    hi = flat.to(torch.bfloat16)
    lo = (flat - hi.to(torch.float32)).to(torch.bfloat16)
    g = flat.shape[:-1] + (C // 8, 8)
    buf = torch.stack([hi.reshape(g), lo.reshape(g)], dim=-2).contiguous()   # [..., C/8, 2, 8] bf16
    return buf.view(torch.float32).reshape(-1), tuple(x.shape)


_convq_ws = _WorkspaceCache()


def conv2d_ex_takes_split(N, H, W, Ci, Co, kh, kw, stride, padding, transposed, precision="bf16x3"):
    d = ConvDesc(N, H, W, Ci, Co, kh, kw, int(stride), int(padding), int(bool(transposed)), 0.0, PRECISIONS[precision], 0, 0)
    return bool(_lib.lib().vpx_conv2d_ex_takes_split(ctypes.byref(d)))


def conv2d_ex_prefers_split(N, H, W, Ci, Co, kh, kw, stride, padding, transposed, precision="bf16x3"):
    """True when the layer runs on the schedule-driven K = 32 kernel (convq) given split input — for stride-2 transposed layers
    that is worth converting an fp32 input first (all four output phases in one launch)."""
    d = ConvDesc(N, H, W, Ci, Co, kh, kw, int(stride), int(padding), int(bool(transposed)), 0.0, PRECISIONS[precision], 0, 0)
    return _lib.lib().vpx_conv2d_ex_takes_split(ctypes.byref(d)) == 2


def conv2d_ex_from_split(xbuf, xshape, w, bias, stride, padding, transposed=False, leaky_slope=0.0, precision="bf16x3",
                         out_split=False, out_fp32=True):
    """The stage-glue layer on an input that already is in the split-bf16 operand format (inference): xbuf / xshape =
    (buffer, (N, Ci, H, W)). Returns (y or None, ybuf or None, (N, Co, Ho, Wo)). The packed weights stay in a per-layer
    workspace while the weight tensor is unchanged (keyed on the tensor object, its version and address)."""
    import weakref
    N, Ci, H, Wd = xshape
    kh, kw = int(w.shape[2]), int(w.shape[3])
    Co = int(w.shape[1] if transposed else w.shape[0])
    d = ConvDesc(N, H, Wd, Ci, Co, kh, kw, int(stride), int(padding), int(bool(transposed)), float(leaky_slope),
                 PRECISIONS[precision], 0, 0)
    L = _lib.lib()
    ho, wo = ctypes.c_int(0), ctypes.c_int(0)
    check(L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), "vpx_conv2d_ex_out_shape")
    ws_bytes = L.vpx_conv2d_ex_split_workspace_bytes(ctypes.byref(d))
    if ws_bytes == 0:
        raise _lib.VpxError("conv2d_ex_from_split: layer not implemented on split input: " + L.vpx_last_error().decode())
    key = (id(w), N, H, Wd, Ci, Co, kh, kw, int(stride), int(padding), bool(transposed), precision, L.vpx_conv2d_ex_takes_split(ctypes.byref(d)),
           torch.are_deterministic_algorithms_enabled(), _kernel_options())
    packed = 0
    ws = _convq_ws.get(key, w, w.data_ptr(), ws_bytes, xbuf.device, exact=False)
    if ws is not None:
        packed = 1
    else:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xbuf.device)
        _convq_ws.put(key, w, w.data_ptr(), ws, _kernel_options())
    wc = w.contiguous()
    if wc.data_ptr() != w.data_ptr():
        packed = 0
    bc = None if bias is None else bias.contiguous()
    y = new_channels_last((N, Co, ho.value, wo.value), xbuf.device) if out_fp32 else None
    ybuf = torch.empty(N * ho.value * wo.value * Co, dtype=torch.float32, device=xbuf.device) if out_split else None
    check(L.vpx_conv2d_ex_fwd_from_split(ctypes.byref(d), ptr(xbuf), 0, 0, 1, ptr(wc), ptr(bc), ptr(y), ptr(ybuf), packed, ptr(ws),
                                         ws_bytes, _stream()), "vpx_conv2d_ex_fwd_from_split")
    return y, ybuf, (N, Co, ho.value, wo.value)


def conv_transpose2d_to_size(x, w, stride, padding, out_hw, precision="f32"):
    """nn.ConvTranspose2d(...)(x, output_size=out_hw) without bias (predrnn_v2.py:213-218): the output padding is whatever
    makes the result exactly out_hw (it must lie in [0, stride))."""
    kh, kw = int(w.shape[2]), int(w.shape[3])
    base = ((x.shape[-2] - 1) * stride - 2 * padding + kh, (x.shape[-1] - 1) * stride - 2 * padding + kw)
    op = (int(out_hw[0]) - base[0], int(out_hw[1]) - base[1])
    if min(op) < 0 or max(op) >= max(stride, 1):
        raise ValueError(f"requested output size {tuple(out_hw)} is not reachable (needs output padding {op}, stride {stride})")
    return conv2d_ex(x, w, None, stride, padding, True, 0.0, precision, op)


def glue_supported(kh, kw, stride, padding, transposed) -> bool:
    """Configurations vpx_conv2d_ex_fwd implements. Anything else is refused when a model is built (models/ef_conv_lstm._validate_stage):
    the product has no second convolution backend."""
    if stride not in (1, 2) or kh > 7 or kw > 7 or padding < 0:
        return False
    if transposed and stride == 1:
        return kh - 1 - padding >= 0 and kw - 1 - padding >= 0
    if transposed and stride == 2:
        return kh >= 2 and kw >= 2
    return True


class _DecoupleFn(torch.autograd.Function):
    """mean_{b,ch} |cos(normalize(A*delta_c), normalize(A*delta_m))| over H*W  (predrnn_v2.py:197-198, 209-211)."""

    @staticmethod
    def forward(ctx, delta_c, delta_m, adapter_w, precision):
        _require_gpu(delta_c, "decouple_term")
        dc, dm = to_channels_last(delta_c), to_channels_last(delta_m)
        B, Ch, H, Wd = dc.shape
        A = adapter_w.reshape(Ch, Ch).contiguous()
        L = _lib.lib()
        ws_bytes = L.vpx_decouple_workspace_bytes(B, Ch, H, Wd)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dc.device)
        value = torch.empty((), device=dc.device)
        rc = L.vpx_decouple_fwd(ptr(dc), ptr(dm), ptr(A), ptr(value), B, Ch, H, Wd, precision, ptr(ws), ws_bytes, _stream())
        check(rc, "vpx_decouple_fwd")
        ctx.save_for_backward(dc, dm, A)
        ctx.precision = precision
        ctx.wshape = tuple(adapter_w.shape)
        return value

    @staticmethod
    def backward(ctx, dvalue):
        _sync_determinism()
        dc, dm, A = ctx.saved_tensors
        B, Ch, H, Wd = dc.shape
        needs = ctx.needs_input_grad
        if needs[0] and needs[1]:
            gg = new_channels_last((2 * B, Ch, H, Wd), dc.device)  # adjacent: one adjoint conv for the pair
            g_dc, g_dm = gg[:B], gg[B:]
        else:
            g_dc = new_channels_last((B, Ch, H, Wd), dc.device) if needs[0] else None
            g_dm = new_channels_last((B, Ch, H, Wd), dc.device) if needs[1] else None
        g_A = torch.empty_like(A) if needs[2] else None
        L = _lib.lib()
        ws_bytes = L.vpx_decouple_workspace_bytes(B, Ch, H, Wd)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dc.device)
        dv = dvalue.contiguous().reshape(1)
        rc = L.vpx_decouple_bwd(ptr(dc), ptr(dm), ptr(A), ptr(dv), ptr(g_dc), ptr(g_dm), ptr(g_A), B, Ch, H, Wd,
                                ctx.precision, ptr(ws), ws_bytes, _stream())
        check(rc, "vpx_decouple_bwd")
        return g_dc, g_dm, (None if g_A is None else g_A.reshape(ctx.wshape)), None


def decouple_term(delta_c, delta_m, adapter_w, precision="f32"):
    """`precision`: arithmetic of the adapter contractions — pass the model's operand mode (default: exact fp32)."""
    return _DecoupleFn.apply(delta_c, delta_m, adapter_w, PRECISIONS[precision])


class _DecoupleBatchFn(torch.autograd.Function):
    """The decoupling term of ALL K layer-steps of a pass in one library call each way. `slab` is [2, K*B, Ch, H, W] (channels-last):
    slab[0] holds the K steps' delta_c one after the other, slab[1] their delta_m; `deltas` = (dc_0, dm_0, dc_1, dm_1, ...) are the
    steps' own output tensors — views of those slots — passed so that autograd has an edge to every step; the data is read from the
    slab. mean over (K*B, channel) of |cos| = the mean over the K steps of the per-step means (equal B): predrnn_v2.py:197-211, 229."""

    @staticmethod
    def forward(ctx, adapter_w, precision, slab, K, B, *deltas):
        _require_gpu(slab, "decouple_term_batched")
        _, KB, Ch, H, Wd = slab.shape
        if KB != K * B or len(deltas) != 2 * K:
            raise ValueError("decouple_term_batched: the slab does not hold K steps of B samples / 2K delta tensors")
        n = B * Ch * H * Wd
        for k in (0, K - 1):   # the steps wrote where the slab expects them (first and last: the rest follow the same arithmetic)
            if deltas[2 * k].data_ptr() != slab[0].data_ptr() + 4 * k * n or deltas[2 * k + 1].data_ptr() != slab[1].data_ptr() + 4 * k * n:
                raise ValueError("decouple_term_batched: a step's delta tensor is not its slot of the slab")
        dc, dm = slab[0], slab[1]
        if not (is_channels_last(dc) and dm.data_ptr() == dc.data_ptr() + 4 * K * n):
            raise ValueError("decouple_term_batched: slab must be one dense channels-last block")
        A = adapter_w.reshape(Ch, Ch).contiguous()
        L = _lib.lib()
        ws_bytes = L.vpx_decouple_workspace_bytes(KB, Ch, H, Wd)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=slab.device)
        value = torch.empty((), device=slab.device)
        check(L.vpx_decouple_fwd(ptr(dc), ptr(dm), ptr(A), ptr(value), KB, Ch, H, Wd, precision, ptr(ws), ws_bytes, _stream()), "vpx_decouple_fwd")
        ctx.save_for_backward(slab, A)
        ctx.geo = (K, B, precision, tuple(adapter_w.shape))
        return value

    @staticmethod
    def backward(ctx, dvalue):
        _sync_determinism()
        slab, A = ctx.saved_tensors
        K, B, precision, wshape = ctx.geo
        _, KB, Ch, H, Wd = slab.shape
        gg = new_channels_last((2 * KB, Ch, H, Wd), slab.device)   # adjacent: one adjoint conv for the pair
        g_dc, g_dm = gg[:KB], gg[KB:]
        g_A = torch.empty_like(A) if ctx.needs_input_grad[0] else None
        L = _lib.lib()
        ws_bytes = L.vpx_decouple_workspace_bytes(KB, Ch, H, Wd)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=slab.device)
        dv = dvalue.contiguous().reshape(1)
        check(L.vpx_decouple_bwd(ptr(slab[0]), ptr(slab[1]), ptr(A), ptr(dv), ptr(g_dc), ptr(g_dm), ptr(g_A), KB, Ch, H, Wd, precision,
                                 ptr(ws), ws_bytes, _stream()), "vpx_decouple_bwd")
        per_step = []
        for k in range(K):   # every step's gradients are views of the two slabs: no copies
            per_step += [g_dc[k * B:(k + 1) * B], g_dm[k * B:(k + 1) * B]]
        return (None if g_A is None else g_A.reshape(wshape), None, None, None, None, *per_step)


def decouple_term_batched(slab, adapter_w, precision, K, B, deltas):
    """mean over the K layer-steps of decouple_term(delta_c_k, delta_m_k) in ONE call (see _DecoupleBatchFn)."""
    return _DecoupleBatchFn.apply(adapter_w, PRECISIONS[precision], slab, int(K), int(B), *deltas)


def _is_dense(t: torch.Tensor) -> bool:
    """True when the tensor's elements tile one memory block without gaps or overlap (any dimension order)."""
    expect = 1
    for size, stride in sorted(((s, st) for s, st in zip(t.shape, t.stride()) if s > 1), key=lambda p: p[1]):
        if stride != expect:
            return False
        expect *= size
    return True


class _MSELossFn(torch.autograd.Function):
    """scale * mean_{b,t} sum_{c,h,w} (pred - target)^2 with d/dpred produced in the same pass
    (base_measure.py:55-57, image_wise.py:25, loss_provider.py:48-51)."""

    @staticmethod
    def forward(ctx, pred, target, scale):
        _require_gpu(pred, "mse_loss")
        if pred.ndim != 5 or target.ndim != 5:
            raise ValueError("Mean Squared Error (MSE) / L2 Loss expects 5-D inputs!")
        if pred.shape != target.shape:
            raise ValueError("Output images and target images are of different shape!")
        p = pred if _is_dense(pred) else pred.contiguous()
        # the sum runs over memory order, so the target only has to share the prediction's memory layout
        tg = target if (target.stride() == p.stride() and target.dtype == p.dtype) else torch.empty_like(p).copy_(target)
        L = _lib.lib()
        ws_bytes = L.vpx_mse_loss_workspace_bytes()
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=p.device)
        loss = torch.empty((), device=p.device)
        need = ctx.needs_input_grad[0]
        g = torch.empty_like(p) if need else None
        rc = L.vpx_mse_loss(ptr(p), ptr(tg), p.numel(), p.shape[0] * p.shape[1], float(scale), ptr(loss), ptr(g),
                            ptr(ws), ws_bytes, _stream())
        check(rc, "vpx_mse_loss")
        if need:
            ctx.save_for_backward(g)
        ctx.need = need
        return loss

    @staticmethod
    def backward(ctx, dloss):
        if not ctx.need:
            return None, None, None
        (g,) = ctx.saved_tensors  # out of place: a second backward through the same node (retain_graph) stays correct
        return g * dloss, None, None


def mse_loss(pred, target, scale: float = 1.0):
    return _MSELossFn.apply(pred, target, scale)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """One torch.optim.Adam update of flat fp32 buckets, in place (vpsuite.py:353; torch/optim/adam.py semantics)."""
    _require_gpu(param, "adam_step")
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == param.numel()):
            raise ValueError("adam_step: buckets must be contiguous float32 GPU tensors of equal size")
    rc = _lib.lib().vpx_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr),
                                  float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                  float(grad_scale), _stream())
    check(rc, "vpx_adam_step")


class STWorkspace:
    """Per-cell workspace that lets consecutive steps of one forward pass skip the weight repack
    (VPX_FLAG_WEIGHTS_PACKED): valid while the weights' version counters and the problem shape are unchanged."""

    def __init__(self):
        self.buf = None
        self.key = None
        self.bwd = None  # the backward's own holder (different workspace layout, transposed packs)

    def backward_holder(self):
        if self.bwd is None:
            self.bwd = STWorkspace()
        return self.bwd

    def get(self, nbytes, device, key):
        """Returns (buffer, packed_valid)."""
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.key = None
        valid = self.key == key
        self.key = key
        return self.buf, valid


_shadow_epoch = 0


def new_shadow_epoch():
    """Ends the life of every split-format shadow attached so far. A model calls this at the start of each forward pass: a shadow then
    only ever describes a tensor produced earlier in the SAME pass, inside the model's own time loop — memory no user code gets
    to write through an alias the version counter cannot see (`.data`, dlpack, a raw-pointer kernel) while the shadow lives."""
    global _shadow_epoch
    _shadow_epoch += 1


def _attach_shadow(t, buf):
    """Records `buf` as the split-format copy of the tensor `t` a library call just wrote: (buffer, version, address, shape, epoch)."""
    t._vpx_sp = (buf, t._version, t.data_ptr(), tuple(t.shape), _shadow_epoch)


def invalidate_shadow(t):
    """Drops a tensor's split-format shadow. Code that writes into a tensor WITHOUT bumping its version counter — `.data.copy_()`,
    a raw-pointer kernel, dlpack aliases — must call this (or `ops.clear_layout_cache()` for everything)."""
    if getattr(t, "_vpx_sp", None) is not None:
        t._vpx_sp = None


def _shadow_of(t, tl, channels):
    """The split-format copy a previous library call attached to tensor `t`, if it still describes the memory `tl` that is about to
    be handed to the library as an operand with `channels` channels: same shadow epoch (new_shadow_epoch), same tensor version, same
    address, same device and the same logical shape [B, channels, H, W] as when it was attached. The attribute lives on the very
    object the library returned — views, clones and `.data` aliases never carry it."""
    sp = getattr(t, "_vpx_sp", None)
    if sp is None:
        return None
    buf, version, addr, shape, epoch = sp
    if epoch != _shadow_epoch or version != t._version or addr != tl.data_ptr() or buf.device != tl.device or shape != tuple(tl.shape) or shape[1] != channels \
            or buf.numel() != tl.numel():
        return None
    return buf


class STWeightBank:
    """Deferred weight gradients of ONE ST-LSTM cell over the T steps of one training forward (include/vpx.h: dg8_out,
    vpx_stlstm_wgrad_batch). The owner (PredRNN_V2.forward) lays the cell's operands out as dense time-major slabs in the split operand
    format and hands every step its slots explicitly:
        g       [T][B*HW*8Ch]   dG8 of step t — written by the step's backward
        sources five tensors, each the slot of step 0 of a slab whose slot t (stride = one slot) is that operand of step t:
                x, h, m (inputs of the step) and c_new, m_new (its outputs)
    `weights()` passes the five weight tensors through an identity autograd node; the steps take THOSE. Their backward deposits dG8 and
    returns no weight gradient; when autograd reaches the identity node — every step of the cell has run its backward by then — `run()`
    computes the five gradients of all T steps in one launch and returns them as the node's input gradients."""

    def __init__(self, weights5, B, Cin, Ch, H, W, k, T, precision):
        self.w5 = tuple(weights5)
        self.geo = (int(B), int(Cin), int(Ch), int(H), int(W), int(k), int(T), int(precision))
        dev = self.w5[0].device
        self.g = torch.empty(T, B * H * W * 8 * Ch, dtype=torch.float32, device=dev)
        self.sources = None
        self.deposited = set()

    @staticmethod
    def available(B, Cin, Ch, H, W, k, precision):
        d = STLSTMDesc(B, Cin, Ch, H, W, k, 0, _lib.LAYOUT_NHWC, PRECISIONS[precision], _lib.FLAG_SAVE_FOR_BWD)
        L = _lib.lib()
        return bool(L.vpx_stlstm_defers_wgrad(ctypes.byref(d))) and bool(L.vpx_stlstm_uses_split(ctypes.byref(d)))

    def set_sources(self, x, h, m, c_new, m_new):
        """Each argument: the slot-0 tensor (1-D, one slot long) of the operand's slab; slots t = 1 .. T-1 follow it in memory."""
        self.sources = (x, h, m, c_new, m_new)

    def weights(self):
        return _BankFn.apply(self, *self.w5)

    def run(self):
        B, Cin, Ch, H, W, k, T, prec = self.geo
        if self.sources is None:
            raise _lib.VpxError("STWeightBank: set_sources was never called")
        for t in range(T):
            if t not in self.deposited:     # a step whose outputs reached no loss: its dG8 is zero
                self.g[t].zero_()
        d = STLSTMDesc(T * B, Cin, Ch, H, W, k, 0, _lib.LAYOUT_NHWC, prec, _lib.FLAG_SAVE_FOR_BWD)
        L = _lib.lib()
        nb = L.vpx_stlstm_wgrad_batch_workspace_bytes(ctypes.byref(d))
        if nb == 0:
            raise _lib.VpxError("STWeightBank: vpx_stlstm_wgrad_batch is not available for this cell")
        ws = torch.empty(nb, dtype=torch.uint8, device=self.g.device)
        dWs = [torch.empty_like(w) for w in self.w5]
        src = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in self.sources])
        check(L.vpx_stlstm_wgrad_batch(ctypes.byref(d), ptr(self.g), src, *[ptr(g) for g in dWs], ptr(ws), nb, _stream()), "vpx_stlstm_wgrad_batch")
        self.deposited.clear()
        return dWs


class _BankFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bank, *w5):
        ctx.bank = bank
        ctx.set_materialize_grads(False)
        return tuple(w.view_as(w) for w in w5)

    @staticmethod
    def backward(ctx, *grads):
        _sync_determinism()
        dWs = ctx.bank.run()
        for i, g in enumerate(grads):       # (a consumer that did not defer — none today — adds its gradient the ordinary way)
            if g is not None:
                dWs[i] = dWs[i] + g
        return (None, *dWs)


class _STLSTMStepFn(torch.autograd.Function):
    """(h_new, c_new, m_new, delta_c, delta_m) = ST-LSTM cell step (predrnn.py:57-83) in one library call.
    `ln` = () or the 8 LayerNorm tensors (x_gamma, x_beta, h_gamma, h_beta, m_gamma, m_beta, o_gamma, o_beta)."""

    @staticmethod
    def forward(ctx, x, h, c, m, Wx, Wh, Wm, Wo, Wlast, precision, need_grad, wsholder, use_shadows, slots, delta_out, *ln):
        _require_gpu(x, "stlstm_step")
        dev = x.device
        B, Cin, H, Wd = x.shape
        Ch = h.shape[1]
        k = int(Wx.shape[-1])
        xs, hs, cs, ms = (to_channels_last(t) for t in (x, h, c, m))
        W5 = [w.contiguous() for w in (Wx, Wh, Wm, Wo, Wlast)]
        lnc = [t.contiguous() for t in ln]
        use_ln = len(lnc) == 8
        flags = _lib.FLAG_SAVE_FOR_BWD if need_grad else 0
        d = STLSTMDesc(B, Cin, Ch, H, Wd, k, int(use_ln), _lib.LAYOUT_NHWC, precision, flags)
        L = _lib.lib()
        ws_bytes = L.vpx_stlstm_workspace_bytes(ctypes.byref(d))
        if ws_bytes == 0:
            check(-4 if b"not implemented" in L.vpx_last_error() else -1, "vpx_stlstm_workspace_bytes")
        rs_bytes = L.vpx_stlstm_reserve_bytes(ctypes.byref(d))
        key = (B, Cin, Ch, H, Wd, k, precision, flags, _det_state, _kernel_options(), tuple((w.data_ptr(), w._version) for w in W5),
               tuple((t.data_ptr(), t._version) for t in lnc))
        if wsholder is not None:
            ws, packed = wsholder.get(ws_bytes, dev, key)
        else:
            ws, packed = torch.empty(ws_bytes, dtype=torch.uint8, device=dev), False
        if packed:
            d.flags |= _lib.FLAG_WEIGHTS_PACKED
        reserve = torch.empty(max(rs_bytes, 1), dtype=torch.uint8, device=dev)
        outs = [new_channels_last((B, Ch, H, Wd), dev) for _ in range(3)]
        if delta_out is not None:   # the caller's slab slots (decouple_term_batched: one decoupling tail over all the steps of a pass)
            for t in delta_out:
                if tuple(t.shape) != (B, Ch, H, Wd) or not is_channels_last(t) or t.dtype != torch.float32 or t.device != dev:
                    raise ValueError("stlstm_step: delta_out must hold two channels-last float32 [B, Ch, H, W] tensors on the step's device")
            outs += [delta_out[0], delta_out[1]]
        else:
            dd = new_channels_last((2 * B, Ch, H, Wd), dev)  # delta_c | delta_m adjacent: the decoupling tail runs the pair as one conv
            outs += [dd[:B], dd[B:]]
        ln_arr = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in lnc]) if use_ln else None
        # split-format shadows (vpx.h): operands a previous step left in the kernels' operand format are handed back instead of being
        # converted again (h_new is the next step's h and the next layer's x, m_new the next layer's m), and this step's h_new / c_new /
        # m_new come back with shadows of their own
        sp_in, sp_out, shadows = [None] * 5, None, None
        bank = None
        if slots is not None:
            # explicit operand slots (STWeightBank mode): the owner's slabs hold x, h, m in the split format and receive h_new, c_new, m_new
            bank, bank_t, slot_in, slot_out, convert_x = slots
            if use_ln or not need_grad or not L.vpx_stlstm_uses_split(ctypes.byref(d)):
                raise _lib.VpxError("stlstm_step: operand slots need the split-operand kernels in a call that saves for the backward")
            sp_in[:3] = list(slot_in)
            if convert_x:   # an input without a split copy (the first layer's frame): converted into its slot of the owner's x slab
                check(L.vpx_split_convert(ptr(xs), ptr(sp_in[0]), B * H * Wd, Cin, _stream()), "vpx_split_convert")
            sp_out = list(slot_out)
            for buf, n in zip(sp_in[:3] + sp_out, (Cin, Ch, Ch, Ch, Ch, Ch)):
                if buf.numel() != B * H * Wd * n or not buf.is_contiguous() or buf.dtype != torch.float32:
                    raise ValueError("stlstm_step: an operand slot does not hold B*H*W*C elements")
        elif use_shadows and not use_ln and L.vpx_stlstm_uses_split(ctypes.byref(d)):
            sp_in[:3] = [_shadow_of(x, xs, Cin), _shadow_of(h, hs, Ch), _shadow_of(m, ms, Ch)]
            sp_out = [torch.empty(B * H * Wd * Ch, dtype=torch.float32, device=dev) for _ in range(3)]
        if sp_out is not None:
            shadows = _lib.STLSTMShadows((ctypes.c_void_p * 5)(*[None if t is None else t.data_ptr() for t in sp_in]),
                                         (ctypes.c_void_p * 3)(*[t.data_ptr() for t in sp_out]), None)
        if PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = L.vpx_stlstm_step_fwd_ex(ctypes.byref(d), ptr(xs), ptr(hs), ptr(cs), ptr(ms), *[ptr(w) for w in W5], ln_arr,
                                      *[ptr(o) for o in outs], ptr(reserve), rs_bytes, ptr(ws), ws_bytes, _stream(),
                                      None if shadows is None else ctypes.byref(shadows))
        check(rc, "vpx_stlstm_step_fwd")
        if PROFILE is not None:
            ev1.record()
            fl, by = stlstm_algorithmic_work(B, Cin, Ch, H, Wd, k)
            PROFILE.records.append((ev0, ev1, fl, by, 4, "stlstm_fwd", 0.0))
        if sp_out is not None and bank is None:
            for t, buf in zip(outs[:3], sp_out):
                _attach_shadow(t, buf)
        if need_grad:
            ctx.sp = None if sp_out is None else (sp_in[0], sp_in[1], sp_in[2], sp_out[1], sp_out[2])   # x, h, m, c_new, m_new
            ctx.save_for_backward(xs, hs, cs, ms, outs[1], outs[2], *W5, reserve, *lnc)
            d.flags = flags
            ctx.desc = d
            ctx.rs_bytes = rs_bytes
            ctx.use_ln = use_ln
            ctx.bwd_holder = wsholder.backward_holder() if (wsholder is not None and not use_ln) else None
            ctx.wkey = key
            ctx.bank = None if bank is None else (bank, bank_t)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dh_new, dc_new, dm_new, ddc, ddm):
        _sync_determinism()
        saved = ctx.saved_tensors
        xs, hs, cs, ms, c_new, m_new, Wx, Wh, Wm, Wo, Wlast, reserve = saved[:12]
        lnc = list(saved[12:])
        d = ctx.desc
        dev = xs.device
        L = _lib.lib()
        gin = [None if g is None else to_channels_last(g) for g in (dh_new, dc_new, dm_new, ddc, ddm)]
        needs = ctx.needs_input_grad
        dx = new_channels_last(tuple(xs.shape), dev) if needs[0] else None
        dh = new_channels_last(tuple(hs.shape), dev) if needs[1] else None
        dc = new_channels_last(tuple(cs.shape), dev) if needs[2] else None
        dm = new_channels_last(tuple(ms.shape), dev) if needs[3] else None
        bank = getattr(ctx, "bank", None)
        dWs = [torch.empty_like(w) if (needs[4 + i] and bank is None) else None for i, w in enumerate((Wx, Wh, Wm, Wo, Wlast))]
        dln = [torch.empty_like(t) if needs[15 + i] else None for i, t in enumerate(lnc)]
        ln_arr = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in lnc]) if ctx.use_ln else None
        dln_arr = (ctypes.c_void_p * 8)(*[None if t is None else t.data_ptr() for t in dln]) if ctx.use_ln else None
        ws_bytes = L.vpx_stlstm_workspace_bytes(ctypes.byref(d))
        flags0 = d.flags
        if ctx.bwd_holder is not None and BWD_WEIGHT_PACK_REUSE:
            # (which transposed packs a call makes depends on the data gradients it is asked for)
            ws, packed = ctx.bwd_holder.get(ws_bytes, dev, (ctx.wkey, dx is not None, dh is not None, dm is not None, bank is not None))
            if packed:
                d.flags = flags0 | _lib.FLAG_WEIGHTS_PACKED
        else:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        shadows = None
        if getattr(ctx, "sp", None) is not None:   # the forward's split shadows: the weight-gradient kernel stages them as they are
            shadows = _lib.STLSTMShadows((ctypes.c_void_p * 5)(*[None if t is None else t.data_ptr() for t in ctx.sp]), (ctypes.c_void_p * 3)(),
                                         None if bank is None else bank[0].g[bank[1]].data_ptr())
        rc = L.vpx_stlstm_step_bwd_ex(ctypes.byref(d), ptr(xs), ptr(hs), ptr(cs), ptr(ms), ptr(c_new), ptr(m_new), ptr(Wx),
                                      ptr(Wh), ptr(Wm), ptr(Wo), ptr(Wlast), ln_arr, ptr(reserve), ctx.rs_bytes,
                                      *[ptr(g) for g in gin], ptr(dx), ptr(dh), ptr(dc), ptr(dm), *[ptr(g) for g in dWs],
                                      dln_arr, ptr(ws), ws_bytes, _stream(), None if shadows is None else ctypes.byref(shadows))
        d.flags = flags0
        check(rc, "vpx_stlstm_step_bwd")
        if bank is not None:
            bank[0].deposited.add(bank[1])
        return (dx, dh, dc, dm, *dWs, None, None, None, None, None, None, *dln)


def stlstm_algorithmic_work(B, Cin, Ch, H, W, k, dt=4):
    """flops = 2*[(7Ch*Cin + 9Ch^2)*k^2 + 2Ch^2]*H*W per sample; bytes = dt*H*W*(Cin + 3Ch + 5Ch) per
    sample + the weights once per step."""
    flops = 2.0 * ((7 * Ch * Cin + 9 * Ch * Ch) * k * k + 2 * Ch * Ch) * H * W * B
    nbytes = dt * H * W * (Cin + 8 * Ch) * B + dt * ((7 * Ch * Cin + 9 * Ch * Ch) * k * k + 2 * Ch * Ch)
    return flops, nbytes


def stlstm_step(x, h, c, m, Wx, Wh, Wm, Wo, Wlast, precision="f32", wsholder=None, ln=(), use_shadows=False, slots=None, delta_out=None):
    """ln: () or the 8 LayerNorm parameter tensors [C,H,W] (x_gamma, x_beta, h_.., m_.., o_..) of the LayerNorm variant.
    use_shadows: hand h_new / c_new / m_new out with split-format shadows and take the shadows of x / h / m where a previous step of
    the same shadow epoch attached them (no second conversion). Opt-in: the caller vouches that nothing writes those tensors behind
    the version counter's back between the steps (PredRNN_V2.forward: the tensors never leave its loop).
    slots: (bank, t, (x_sp, h_sp, m_sp), (h_new_sp, c_new_sp, m_new_sp), convert_x) — STWeightBank mode: the weights must be
    `bank.weights()`, the operands' split copies live in the owner's slabs (convert_x: x has none yet — converted into x_sp here), the
    weight gradients of the cell's steps are computed together when autograd reaches the bank.
    delta_out: (delta_c, delta_m) — caller tensors [B, Ch, H, W] (channels-last) the step writes its two deltas into instead of
    allocating them (slots of the slab `decouple_term_batched` reads)."""
    ln = tuple(ln)
    if len(ln) not in (0, 8):
        raise ValueError("stlstm_step: ln must hold 0 or 8 tensors")
    need_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (x, h, c, m, Wx, Wh, Wm, Wo, Wlast) + ln)
    return _STLSTMStepFn.apply(x, h, c, m, Wx, Wh, Wm, Wo, Wlast, PRECISIONS[precision], need_grad, wsholder, bool(use_shadows), slots, delta_out, *ln)


class _LayerNormCHWFn(torch.autograd.Function):
    """nn.LayerNorm([C,H,W]) on a [B,C,H,W] tensor (channels-last memory) through the library's LayerNorm kernels (double-precision
    statistics over fixed chunks, explicit backward) — the action-conditional ST-LSTM cell's per-convolution normalisation
    (predrnn.py:105-135)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, "layer_norm_chw")
        xs = to_channels_last(x)
        B, C, H, Wd = xs.shape
        # keyed on the parameter itself (a `weight[None]` temporary is a new object every call and can never hit)
        g = _cached_channels_last(weight)
        b = _cached_channels_last(bias)
        L = _lib.lib()
        ws_bytes = L.vpx_layernorm_workspace_bytes(B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        y = new_channels_last((B, C, H, Wd), x.device)
        need = any(ctx.needs_input_grad)
        xhat = new_channels_last((B, C, H, Wd), x.device) if need else None
        stats = torch.empty(B, 2, device=x.device)
        check(L.vpx_layernorm_fwd(ptr(xs), ptr(g), ptr(b), ptr(y), ptr(xhat), ptr(stats), B, C * H * Wd, ptr(ws), ws_bytes, _stream()),
              "vpx_layernorm_fwd")
        if need:
            ctx.save_for_backward(xhat, stats, g)
            ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        _sync_determinism()
        xhat, stats, g = ctx.saved_tensors
        B, C, H, Wd = xhat.shape
        dys = to_channels_last(dy)
        L = _lib.lib()
        ws_bytes = L.vpx_layernorm_workspace_bytes(B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
        dx = new_channels_last((B, C, H, Wd), dy.device)
        dg = new_channels_last((1, C, H, Wd), dy.device)
        db = new_channels_last((1, C, H, Wd), dy.device)
        check(L.vpx_layernorm_bwd(ptr(dys), ptr(xhat), ptr(stats), ptr(g), ptr(dx), ptr(dg), ptr(db), B, H * Wd, C, ptr(ws), ws_bytes,
                                  _stream()), "vpx_layernorm_bwd")
        return dx, dg.reshape(ctx.wshape), db.reshape(ctx.wshape)


def layer_norm_chw(x, weight, bias, eps=1e-5):
    """F.layer_norm(x, [C,H,W], weight, bias, eps) for a [B,C,H,W] tensor, on the library's kernels (eps is fixed at 1e-5)."""
    if abs(float(eps) - 1e-5) > 1e-12:
        raise ValueError("layer_norm_chw: the library's LayerNorm kernels use eps = 1e-5 (nn.LayerNorm's default)")
    return _LayerNormCHWFn.apply(x, weight, bias)


class _ACSTStepFn(torch.autograd.Function):
    """(h_new, c_new, m_new, delta_c, delta_m) = action-conditional ST-LSTM cell step (predrnn.py:139-169) in ONE library call each
    way (vpx_acstlstm_step_fwd / _bwd): six biased convolutions, optional LayerNorms, the conv_h(h) * conv_a(a) product, both gate
    groups, the state updates and the output gate. `tensors` = the 12 parameters (conv_x, conv_h, conv_a, conv_m, conv_o, conv_last)
    x (weight, bias), then — with layer_norm — the 10 LayerNorm tensors (x, h, a, m, o) x (weight, bias)."""

    @staticmethod
    def forward(ctx, x, h, c, m, a, precision, forget_bias, *tensors):
        _require_gpu(x, "acstlstm_step")
        dev = x.device
        B, Cin, H, Wd = x.shape
        Ch = h.shape[1]
        prm = [t.contiguous() for t in tensors[:12]]
        lnc = [t.contiguous() for t in tensors[12:]]
        if len(prm) != 12 or len(lnc) not in (0, 10):
            raise ValueError("acstlstm_step: 12 parameters, then either no or ten LayerNorm tensors")
        k = int(prm[0].shape[-1])
        need = any(ctx.needs_input_grad)
        d = _lib.ACSTLSTMDesc(B, Cin, Ch, H, Wd, k, int(bool(lnc)), precision, _lib.FLAG_SAVE_FOR_BWD if need else 0, float(forget_bias))
        L = _lib.lib()
        ws_bytes = L.vpx_acstlstm_workspace_bytes(ctypes.byref(d))
        if ws_bytes == 0:
            check(-1, "vpx_acstlstm_workspace_bytes")
        rs_bytes = L.vpx_acstlstm_reserve_bytes(ctypes.byref(d))
        xs, hs, cs, ms, as_ = (to_channels_last(t) for t in (x, h, c, m, a))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        reserve = torch.empty(max(rs_bytes, 1), dtype=torch.uint8, device=dev)
        outs = [new_channels_last((B, Ch, H, Wd), dev) for _ in range(5)]
        p_arr = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in prm])
        ln_arr = (ctypes.c_void_p * 10)(*[t.data_ptr() for t in lnc]) if lnc else None
        check(L.vpx_acstlstm_step_fwd(ctypes.byref(d), ptr(xs), ptr(hs), ptr(cs), ptr(ms), ptr(as_), p_arr, ln_arr, *[ptr(o) for o in outs],
                                      ptr(reserve), rs_bytes, ptr(ws), ws_bytes, _stream()), "vpx_acstlstm_step_fwd")
        if need:
            ctx.save_for_backward(xs, hs, cs, ms, as_, reserve, *prm, *lnc)
            ctx.desc, ctx.rs_bytes = d, rs_bytes
        return tuple(outs)

    @staticmethod
    def backward(ctx, dh_new, dc_new, dm_new, ddc, ddm):
        _sync_determinism()
        saved = ctx.saved_tensors
        xs, hs, cs, ms, as_, reserve = saved[:6]
        prm, lnc = list(saved[6:18]), list(saved[18:])
        d, dev, L = ctx.desc, xs.device, _lib.lib()
        gin = [None if g is None else to_channels_last(g) for g in (dh_new, dc_new, dm_new, ddc, ddm)]
        if gin[0] is None:
            gin[0] = torch.zeros_like(hs)
        needs = ctx.needs_input_grad
        dins = [new_channels_last(tuple(t.shape), dev) if needs[i] else None for i, t in enumerate((xs, hs, cs, ms, as_))]
        dprm = [torch.empty_like(t) if needs[7 + i] else None for i, t in enumerate(prm)]
        dln = [torch.empty_like(t) if needs[19 + i] else None for i, t in enumerate(lnc)]
        p_arr = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in prm])
        dp_arr = (ctypes.c_void_p * 12)(*[None if t is None else t.data_ptr() for t in dprm])
        ln_arr = (ctypes.c_void_p * 10)(*[t.data_ptr() for t in lnc]) if lnc else None
        dln_arr = (ctypes.c_void_p * 10)(*[None if t is None else t.data_ptr() for t in dln]) if lnc else None
        ws_bytes = L.vpx_acstlstm_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(L.vpx_acstlstm_step_bwd(ctypes.byref(d), ptr(xs), ptr(hs), ptr(cs), ptr(ms), ptr(as_), p_arr, ln_arr, ptr(reserve), ctx.rs_bytes,
                                      *[ptr(g) for g in gin], *[ptr(t) for t in dins], dp_arr, dln_arr, ptr(ws), ws_bytes, _stream()),
              "vpx_acstlstm_step_bwd")
        return (*dins, None, None, *dprm, *dln)


def acstlstm_step(x, h, c, m, a, params, ln=(), precision="f32", forget_bias=1.0):
    """One step of the action-conditional ST-LSTM cell; `params`: the 12 convolution tensors, `ln`: () or the 10 LayerNorm tensors."""
    return _ACSTStepFn.apply(x, h, c, m, a, PRECISIONS[precision], float(forget_bias), *params, *ln)
