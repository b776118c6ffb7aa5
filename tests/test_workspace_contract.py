"""Workspace contract of the C ABI, checked WITHOUT a GPU (VPX_OPT_DRY_RUN, include/vpx.h).

Every entry point that is handed a workspace is called with fake (never dereferenced) pointers and a workspace of EXACTLY the
size its `*_workspace_bytes` query returns. In a dry run the library does all of its host-side work — kernel selection, carving,
and the bounds check of everything it would write into the workspace (weight packs, K-slice slabs, operand conversions, partial
sums) — and issues no HIP call. A sizing rule that disagrees with the launch code returns VPX_ERR_WORKSPACE (-2) here instead of
writing past the caller's buffer on the GPU box.

Regression: round 4's intermittent abort of the GPU suite (GPUTEST_r04: SIGABRT inside the default-size `trajgru` forward).
`vpx_conv2d_workspace_bytes` sized the weight pack with the stage size of a 4-group N tile (32 channels for 5x5 taps: 25 chunks per
32 input channels) while `vpx_conv2d_nhwc_fwd_ex` packs with the stage size of the tiling it actually picks (1 group for Co <= 32:
16-channel stages, 13 chunks each = 26 per 32 channels): the flow generator's 5x5 layers (h2f_conv1 64|96 -> 32, flows_conv
32 -> 26) wrote 4-12 KB past a torch allocation. `test_trajgru_flow_generator_layers` pins exactly those layers."""
import ctypes
import itertools

import pytest

from vp_suite_amd import _lib
from vp_suite_amd._lib import ConvDesc, ConvLSTMDesc, STLSTMDesc

OK, E_ARG, E_WS, E_LAUNCH, E_UNSUPPORTED = 0, -1, -2, -3, -4
WS_BASE = 0x7F0000000000            # fake workspace address (256-byte aligned; never dereferenced in a dry run)
WS_BASE_ODD = WS_BASE + 0x40        # ... and one that is not 256-byte aligned (the carver rounds up; the query's slack covers it)


def _fake(i):                        # distinct fake tensor addresses far away from the workspace
    return ctypes.c_void_p(0x100000000000 + i * (1 << 36))


@pytest.fixture(scope="module")
def L():
    lib = _lib.lib()
    prev = lib.vpx_set_option(_lib.OPT_DRY_RUN, 1)
    yield lib
    lib.vpx_set_option(_lib.OPT_DRY_RUN, prev)
    lib.vpx_set_deterministic(0)


def _ok(L, rc, what, allow_unsupported=True):
    if rc == OK or (allow_unsupported and rc == E_UNSUPPORTED):
        return
    raise AssertionError(f"{what}: rc={rc}: {L.vpx_last_error().decode()}")


GEOS = [(1, 16, 16), (6, 64, 64), (8, 16, 16), (6, 32, 32), (128, 16, 16), (2, 67, 83), (40, 32, 32), (4, 128, 128)]


@pytest.mark.parametrize("det", [0, 1])
def test_plain_conv_entry_points(L, det):
    """vpx_conv2d_nhwc_fwd / _fwd_ex / _bwd on every (Ci, Co, k) the models produce and a grid around them."""
    L.vpx_set_deterministic(det)
    cis = [1, 3, 8, 13, 16, 26, 32, 48, 64, 96, 128, 192, 256, 288, 13 * 64, 13 * 96]
    cos = [1, 3, 16, 26, 32, 33, 64, 96, 97, 128, 192, 288, 384]
    n = 0
    for Ci, Co, k, prec in itertools.product(cis, cos, (1, 3, 5, 7), (0, 1, 2)):
        if Ci * k * k > 13 * 96 * 9:      # (outside anything the models build; keeps the sweep short)
            continue
        nb = L.vpx_conv2d_workspace_bytes(Ci, Co, k, k)
        assert nb > 0
        for (N, H, W) in GEOS[:6]:
            for base in (WS_BASE, WS_BASE_ODD):
                rc = L.vpx_conv2d_nhwc_fwd_ex(_fake(1), _fake(2), _fake(3), _fake(4), N, H, W, Ci, Co, k, k, prec, 0, 0.0,
                                              ctypes.c_void_p(base), nb, None)
                _ok(L, rc, f"fwd_ex Ci={Ci} Co={Co} k={k} prec={prec} geo={(N, H, W)}")
            rc = L.vpx_conv2d_nhwc_fwd_ex(_fake(1), _fake(2), _fake(3), _fake(4), N, H, W, Ci, Co, k, k, prec, 1, 0.2,
                                          ctypes.c_void_p(WS_BASE), nb, None)
            _ok(L, rc, f"fwd_ex(acc, leaky) Ci={Ci} Co={Co} k={k} prec={prec} geo={(N, H, W)}")
            rc = L.vpx_conv2d_nhwc_fwd(_fake(1), _fake(2), _fake(3), _fake(4), N, H, W, Ci, Co, k, k, prec, ctypes.c_void_p(WS_BASE), nb, None)
            _ok(L, rc, f"fwd Ci={Ci} Co={Co} k={k} prec={prec} geo={(N, H, W)}")
            nbw = L.vpx_conv2d_bwd_workspace_bytes(N, H, W, Ci, Co, k, k)
            rc = L.vpx_conv2d_nhwc_bwd(_fake(1), _fake(2), _fake(3), _fake(4), _fake(5), _fake(6), N, H, W, Ci, Co, k, k, prec,
                                       ctypes.c_void_p(WS_BASE_ODD), nbw, None)
            _ok(L, rc, f"bwd Ci={Ci} Co={Co} k={k} prec={prec} geo={(N, H, W)}")
            n += 1
    assert n > 5000


def test_trajgru_flow_generator_layers(L):
    """The layers behind GPUTEST_r04's abort (default EF-TrajGRU: traj_gru.py:108-132 with ef_traj_gru.py:31-75), every operand mode."""
    L.vpx_set_deterministic(0)
    for (C, H) in ((64, 64), (96, 32), (96, 16)):
        for (Ci, Co, k) in ((C, 32, 5), (32, 26, 5), (13 * C, 3 * C, 1), (C, 3 * C, 3)):
            for prec in (0, 1, 2):
                nb = L.vpx_conv2d_workspace_bytes(Ci, Co, k, k)
                rc = L.vpx_conv2d_nhwc_fwd_ex(_fake(1), _fake(2), _fake(3), _fake(4), 2, H, H, Ci, Co, k, k, prec, 0, 0.0,
                                              ctypes.c_void_p(WS_BASE), nb, None)
                _ok(L, rc, f"Ci={Ci} Co={Co} k={k} prec={prec}", allow_unsupported=False)


def test_undersized_workspace_is_refused_not_overrun(L):
    nb = L.vpx_conv2d_workspace_bytes(96, 32, 5, 5)
    rc = L.vpx_conv2d_nhwc_fwd_ex(_fake(1), _fake(2), _fake(3), _fake(4), 2, 32, 32, 96, 32, 5, 5, 1, 0, 0.0, ctypes.c_void_p(WS_BASE), nb - 4096, None)
    assert rc == E_WS


CLSTM_BLOCKS = [(16, 64, 64, 64), (64, 96, 32, 32), (96, 96, 16, 16), (96, 96, 32, 32), (96, 64, 64, 64), (64, 64, 64, 64),
                (16, 64, 128, 128), (64, 96, 64, 64), (3, 8, 12, 10), (3, 64, 67, 83), (4, 8, 16, 16), (8, 8, 8, 8), (12, 12, 4, 4)]


@pytest.mark.parametrize("det", [0, 1])
@pytest.mark.parametrize("prec", [0, 1, 2])
def test_convlstm_seq(L, det, prec):
    L.vpx_set_deterministic(det)
    for (Cin, Ch, H, W), B, T, k, layout, gate in itertools.product(CLSTM_BLOCKS, (1, 2, 4, 32, 128), (1, 10), (3, 5), (0, 1), (0, 1)):
        if B * H * W > 128 * 64 * 64 or (layout == 1 and B > 4) or (k == 5 and B > 4):
            continue
        for save in (0, 1):
            for opt in (None, (_lib.OPT_CELL2, 0), (_lib.OPT_CELL2, 2), (_lib.OPT_CELL3, 0), (_lib.OPT_EXPERIMENT, 4096), (_lib.OPT_EXPERIMENT, 8192),
                        (_lib.OPT_EXPERIMENT, 32), (_lib.OPT_MFMA_SHAPE, 0)):
                if opt is not None and (save or B > 32 or T == 1):
                    continue
                prev = L.vpx_set_option(*opt) if opt else None
                try:
                    d = ConvLSTMDesc(B, T, Cin, Ch, H, W, k, k, gate, layout, prec, _lib.FLAG_SAVE_FOR_BWD if save else 0)
                    nb = L.vpx_convlstm_workspace_bytes(ctypes.byref(d))
                    if nb == 0:
                        continue
                    rs = L.vpx_convlstm_reserve_bytes(ctypes.byref(d))
                    tag = f"convlstm {(Cin, Ch, H, W)} B={B} T={T} k={k} layout={layout} prec={prec} save={save} opt={opt}"
                    for (x, h0) in ((_fake(1), _fake(2)), (None, _fake(2)), (_fake(1), None)):
                        rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), x, h0, None if h0 is None else _fake(3), _fake(4), _fake(5), _fake(6), _fake(7), _fake(8),
                                                    _fake(9), _fake(10), _fake(11), _fake(12), rs, ctypes.c_void_p(WS_BASE), nb, None)
                        _ok(L, rc, tag + f" fwd x={x is not None} h0={h0 is not None}")
                        if save:
                            rc = L.vpx_convlstm_seq_bwd(ctypes.byref(d), x, h0, None if h0 is None else _fake(3), _fake(4), _fake(6), _fake(7), _fake(8),
                                                        _fake(9), _fake(12), rs, _fake(13), _fake(14), _fake(15),
                                                        None if x is None else _fake(16), None if h0 is None else _fake(17), None if h0 is None else _fake(18),
                                                        _fake(19), _fake(20), _fake(21), _fake(22), _fake(23), ctypes.c_void_p(WS_BASE_ODD), nb, None)
                            _ok(L, rc, tag + f" bwd x={x is not None} h0={h0 is not None}")
                    if not save and L.vpx_convlstm_takes_split_input(ctypes.byref(d)):
                        d.flags |= _lib.FLAG_X_SPLIT
                        if L.vpx_convlstm_writes_split_output(ctypes.byref(d)):
                            d.flags |= _lib.FLAG_OUT_SPLIT
                        rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), _fake(1), _fake(2), _fake(3), _fake(4), _fake(5), _fake(6), _fake(7), _fake(8),
                                                    _fake(9), _fake(10), _fake(11), None, 0, ctypes.c_void_p(WS_BASE), nb, None)
                        _ok(L, rc, tag + " fwd (split in/out)")
                finally:
                    if opt:
                        L.vpx_set_option(opt[0], prev)


ST_CELLS = [(16, 128, 16, 16, 5), (128, 128, 16, 16, 5), (48, 128, 32, 32, 5), (128, 128, 32, 32, 5), (16, 16, 8, 8, 5), (3, 8, 12, 10, 3),
            (16, 64, 16, 16, 3), (64, 64, 16, 16, 5), (16, 128, 16, 16, 3)]
ST_EXPERIMENTS = (0, 64, 128, 256, 512, 1024, 2048, 64 | 128 | 256 | 512)


@pytest.mark.parametrize("det", [0, 1])
@pytest.mark.parametrize("prec", [0, 1, 2])
def test_stlstm_step(L, det, prec):
    L.vpx_set_deterministic(det)
    for (Cin, Ch, H, W, k), B, ln, layout, exp in itertools.product(ST_CELLS, (1, 2, 4, 8, 16, 128, 256), (0, 1), (0, 1), ST_EXPERIMENTS):
        if (layout == 1 or ln == 1 or exp) and B > 8:
            continue
        prev = L.vpx_set_option(_lib.OPT_EXPERIMENT, exp)
        try:
            for save in (0, 1):
                for packed in (0, 1):
                    d = STLSTMDesc(B, Cin, Ch, H, W, k, ln, layout, prec, (_lib.FLAG_SAVE_FOR_BWD if save else 0) | (_lib.FLAG_WEIGHTS_PACKED if packed else 0))
                    nb = L.vpx_stlstm_workspace_bytes(ctypes.byref(d))
                    if nb == 0:
                        continue
                    rs = L.vpx_stlstm_reserve_bytes(ctypes.byref(d))
                    lnarr = (ctypes.c_void_p * 8)(*[0x200000000000 + i * (1 << 30) for i in range(8)]) if ln else None
                    tag = f"stlstm {(Cin, Ch, H, W, k)} B={B} ln={ln} layout={layout} prec={prec} exp={exp} save={save} packed={packed}"
                    rc = L.vpx_stlstm_step_fwd(ctypes.byref(d), *[_fake(i) for i in range(1, 10)], lnarr, *[_fake(i) for i in range(10, 15)],
                                               _fake(15), rs, ctypes.c_void_p(WS_BASE), nb, None)
                    _ok(L, rc, tag + " fwd")
                    if save:
                        dlnarr = (ctypes.c_void_p * 8)(*[0x300000000000 + i * (1 << 30) for i in range(8)]) if ln else None
                        for (wantx, wanth) in ((1, 1), (0, 1), (1, 0)):
                            rc = L.vpx_stlstm_step_bwd(ctypes.byref(d), *[_fake(i) for i in range(1, 12)], lnarr, _fake(15), rs,
                                                       *[_fake(i) for i in range(16, 21)],
                                                       _fake(21) if wantx else None, _fake(22) if wanth else None, _fake(23), _fake(24),
                                                       *[_fake(i) for i in range(25, 30)], dlnarr, ctypes.c_void_p(WS_BASE_ODD), nb, None)
                            _ok(L, rc, tag + f" bwd dx={wantx} dh={wanth}")
                        if not ln and layout == 0 and L.vpx_stlstm_defers_wgrad(ctypes.byref(d)):
                            # deferred weight gradients: the step writes dG8 to the caller's slot; the batch call takes T*B images
                            sh = _lib.STLSTMShadows((ctypes.c_void_p * 5)(*[0x400000000000 + i * (1 << 34) for i in range(5)]), (ctypes.c_void_p * 3)(),
                                                    0x500000000000)
                            rc = L.vpx_stlstm_step_bwd_ex(ctypes.byref(d), *[_fake(i) for i in range(1, 12)], lnarr, _fake(15), rs,
                                                          *[_fake(i) for i in range(16, 21)], _fake(21), _fake(22), _fake(23), _fake(24),
                                                          None, None, None, None, None, dlnarr, ctypes.c_void_p(WS_BASE), nb, None, ctypes.byref(sh))
                            _ok(L, rc, tag + " bwd (deferred weight gradients)", allow_unsupported=False)
                            for T in (1, 19, 39):
                                dT = STLSTMDesc(T * B, Cin, Ch, H, W, k, 0, 0, prec, _lib.FLAG_SAVE_FOR_BWD)
                                nbb = L.vpx_stlstm_wgrad_batch_workspace_bytes(ctypes.byref(dT))
                                assert nbb > 0
                                rc = L.vpx_stlstm_wgrad_batch(ctypes.byref(dT), _fake(1), (ctypes.c_void_p * 5)(*[0x400000000000 + i * (1 << 34) for i in range(5)]),
                                                              *[_fake(i) for i in range(25, 30)], ctypes.c_void_p(WS_BASE_ODD), nbb, None)
                                _ok(L, rc, tag + f" wgrad_batch T={T}", allow_unsupported=False)
        finally:
            L.vpx_set_option(_lib.OPT_EXPERIMENT, prev)


def test_decouple_tail(L):
    for det in (0, 1):
        L.vpx_set_deterministic(det)
        for (B, Ch, H, W), prec, exp in itertools.product(((1, 16, 8, 8), (2, 128, 16, 16), (8, 128, 32, 32), (128, 128, 16, 16), (2, 8, 12, 10)), (0, 1, 2), (0, 512)):
            prev = L.vpx_set_option(_lib.OPT_EXPERIMENT, exp)
            try:
                nb = L.vpx_decouple_workspace_bytes(B, Ch, H, W)
                rc = L.vpx_decouple_fwd(_fake(1), _fake(2), _fake(3), _fake(4), B, Ch, H, W, prec, ctypes.c_void_p(WS_BASE), nb, None)
                _ok(L, rc, f"decouple fwd {(B, Ch, H, W)} prec={prec}")
                for adjacent in (0, 1):
                    dc = _fake(1)
                    dm = ctypes.c_void_p(dc.value + B * H * W * Ch * 4) if adjacent else _fake(2)
                    gc = _fake(5)
                    gm = ctypes.c_void_p(gc.value + B * H * W * Ch * 4) if adjacent else _fake(6)
                    rc = L.vpx_decouple_bwd(dc, dm, _fake(3), _fake(4), gc, gm, _fake(7), B, Ch, H, W, prec, ctypes.c_void_p(WS_BASE_ODD), nb, None)
                    _ok(L, rc, f"decouple bwd {(B, Ch, H, W)} prec={prec} adjacent={adjacent}")
            finally:
                L.vpx_set_option(_lib.OPT_EXPERIMENT, prev)


# the stage glue of EF-ConvLSTM / EF-TrajGRU (ef_conv_lstm.py:36-65, ef_traj_gru.py:37-44) and PredRNN's action / frame convolutions
GLUE = [  # (Ci, Co, k, stride, pad, transposed)
    (1, 16, 3, 1, 1, 0), (3, 16, 3, 1, 1, 0), (64, 64, 3, 2, 1, 0), (96, 96, 3, 2, 1, 0), (64, 96, 3, 2, 1, 0), (96, 96, 4, 2, 1, 1), (64, 64, 4, 2, 1, 1),
    (96, 64, 4, 2, 1, 1), (64, 16, 3, 1, 1, 1), (64, 16, 3, 1, 1, 0), (16, 16, 3, 1, 1, 0), (16, 1, 1, 1, 0, 0), (16, 3, 1, 1, 0, 0), (4, 8, 3, 2, 1, 0),
    (12, 12, 4, 2, 1, 1), (8, 4, 3, 1, 1, 1), (16, 128, 5, 1, 2, 0), (128, 16, 1, 1, 0, 0), (2, 16, 5, 2, 2, 0), (16, 16, 5, 2, 2, 1), (32, 48, 7, 2, 3, 0),
]


@pytest.mark.parametrize("det", [0, 1])
def test_stage_glue(L, det):
    L.vpx_set_deterministic(det)
    for (Ci, Co, k, s, p, tr), (N, H, W), prec, slope, exp in itertools.product(GLUE, GEOS + [(1280, 16, 16), (40, 64, 64)], (0, 1, 2), (0.0, 0.2), (0, 16, 1 << 29)):
        if N * H * W * max(Ci, Co) > 1 << 31 or (exp and prec != 1):
            continue
        prev = L.vpx_set_option(_lib.OPT_EXPERIMENT, exp)
        try:
            d = ConvDesc(N, H, W, Ci, Co, k, k, s, p, tr, slope, prec, 0, 0)
            ho, wo = ctypes.c_int(0), ctypes.c_int(0)
            if L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) != OK:
                continue
            tag = f"glue {(Ci, Co, k, s, p, tr)} geo={(N, H, W)} prec={prec} slope={slope} exp={exp}"
            nb = L.vpx_conv2d_ex_workspace_bytes(ctypes.byref(d))
            rc = L.vpx_conv2d_ex_fwd(ctypes.byref(d), _fake(1), _fake(2), _fake(3), _fake(4), ctypes.c_void_p(WS_BASE), nb, None)
            _ok(L, rc, tag + " fwd")
            if Co % 8 == 0:
                rc = L.vpx_conv2d_ex_fwd_split(ctypes.byref(d), _fake(1), _fake(2), _fake(3), None, _fake(5), ctypes.c_void_p(WS_BASE_ODD), nb, None)
                _ok(L, rc, tag + " fwd_split")
            if L.vpx_conv2d_ex_takes_split(ctypes.byref(d)):
                nbs = L.vpx_conv2d_ex_split_workspace_bytes(ctypes.byref(d))
                for packed in (0, 1):
                    rc = L.vpx_conv2d_ex_fwd_from_split(ctypes.byref(d), _fake(1), 0, 0, 1, _fake(2), _fake(3), _fake(4), _fake(5) if Co % 8 == 0 else None, packed,
                                                        ctypes.c_void_p(WS_BASE), nbs, None)
                    _ok(L, rc, tag + f" fwd_from_split packed={packed}")
            nbb = L.vpx_conv2d_ex_bwd_workspace_bytes(ctypes.byref(d))
            if nbb:
                rc = L.vpx_conv2d_ex_bwd(ctypes.byref(d), _fake(1), _fake(2), _fake(4), _fake(6), _fake(7), _fake(8), _fake(9), ctypes.c_void_p(WS_BASE_ODD), nbb, None)
                _ok(L, rc, tag + " bwd")
                if L.vpx_conv2d_ex_bwd_uses_split(ctypes.byref(d)):   # the weight gradient on split operands, the forward's copy of x handed back
                    rc = L.vpx_conv2d_ex_bwd_ex(ctypes.byref(d), _fake(1), _fake(10), _fake(2), _fake(4), _fake(6), _fake(7), _fake(8), _fake(9),
                                                ctypes.c_void_p(WS_BASE), nbb, None)
                    _ok(L, rc, tag + " bwd_ex")
                    rc = L.vpx_conv2d_ex_bwd_ex(ctypes.byref(d), _fake(1), _fake(10), _fake(2), _fake(4), _fake(6), None, _fake(8), None,
                                                ctypes.c_void_p(WS_BASE_ODD), nbb, None)
                    _ok(L, rc, tag + " bwd_ex (dw only)")
        finally:
            L.vpx_set_option(_lib.OPT_EXPERIMENT, prev)


def test_trajgru_sequence(L):
    """vpx_trajgru_seq_fwd / _bwd over the default EF-TrajGRU blocks (ef_traj_gru.py:31-75) and small / ragged shapes, every operand mode."""
    from vp_suite_amd._lib import TrajGRUDesc
    blocks = [(16, 64, 64, 64, 13), (64, 96, 32, 32, 13), (96, 96, 16, 16, 13), (96, 96, 32, 32, 13), (96, 64, 64, 64, 13), (4, 8, 16, 16, 3), (6, 12, 9, 11, 5)]
    for det in (0, 1):
        L.vpx_set_deterministic(det)
        for (Cin, C, H, W, nl), B, T, prec, save in itertools.product(blocks, (1, 2, 8), (1, 4), (0, 1, 2), (0, 1)):
            d = TrajGRUDesc(B, T, Cin, C, H, W, nl, 3, prec, _lib.FLAG_SAVE_FOR_BWD if save else 0, 0.2)
            nb, rs = L.vpx_trajgru_workspace_bytes(ctypes.byref(d)), L.vpx_trajgru_reserve_bytes(ctypes.byref(d))
            assert nb > 0 and (rs > 0) == bool(save)
            params = (ctypes.c_void_p * 10)(*[0x200000000000 + i * (1 << 32) for i in range(10)])
            dparams = (ctypes.c_void_p * 10)(*[0x300000000000 + i * (1 << 32) for i in range(10)])
            tag = f"trajgru {(Cin, C, H, W, nl)} B={B} T={T} prec={prec} save={save} det={det}"
            for (x, h0) in ((_fake(1), _fake(2)), (None, _fake(2)), (_fake(1), None)):
                rc = L.vpx_trajgru_seq_fwd(ctypes.byref(d), x, h0, params, _fake(3), _fake(4), rs, ctypes.c_void_p(WS_BASE), nb, None)
                _ok(L, rc, tag + " fwd", allow_unsupported=False)
                if save:
                    rc = L.vpx_trajgru_seq_bwd(ctypes.byref(d), x, h0, params, _fake(3), _fake(4), rs, _fake(5), _fake(6), None if x is None else _fake(7),
                                               None if h0 is None else _fake(8), dparams, ctypes.c_void_p(WS_BASE_ODD), nb, None)
                    _ok(L, rc, tag + " bwd", allow_unsupported=False)
    d = TrajGRUDesc(2, 3, 4, 6, 8, 8, 3, 3, 0, 0, 0.2)   # C % 4 != 0
    assert L.vpx_trajgru_workspace_bytes(ctypes.byref(d)) == 0


def test_acstlstm_step(L):
    """vpx_acstlstm_step_fwd / _bwd (action-conditional ST-LSTM cell, predrnn.py:86-169): with / without LayerNorm, every operand mode, odd
    map sizes, frozen parameters and unwanted data gradients."""
    from vp_suite_amd._lib import ACSTLSTMDesc
    shapes = [(4, 8, 16, 16, 3), (16, 32, 32, 32, 5), (3, 6, 9, 11, 3), (64, 64, 16, 16, 5), (1, 4, 7, 5, 1), (12, 20, 33, 17, 7)]
    for det in (0, 1):
        L.vpx_set_deterministic(det)
        for (Cin, Ch, H, W, k), B, ln, prec, save in itertools.product(shapes, (1, 2, 8), (0, 1), (0, 1, 2), (0, 1)):
            d = ACSTLSTMDesc(B, Cin, Ch, H, W, k, ln, prec, _lib.FLAG_SAVE_FOR_BWD if save else 0, 1.0)
            nb, rs = L.vpx_acstlstm_workspace_bytes(ctypes.byref(d)), L.vpx_acstlstm_reserve_bytes(ctypes.byref(d))
            assert nb > 0 and (rs > 0) == bool(save)
            params = (ctypes.c_void_p * 12)(*[0x200000000000 + i * (1 << 32) for i in range(12)])
            lnp = (ctypes.c_void_p * 10)(*[0x280000000000 + i * (1 << 32) for i in range(10)]) if ln else None
            tag = f"acstlstm {(Cin, Ch, H, W, k)} B={B} ln={ln} prec={prec} save={save} det={det}"
            for base in (WS_BASE, WS_BASE_ODD):
                rc = L.vpx_acstlstm_step_fwd(ctypes.byref(d), *[_fake(i) for i in range(1, 6)], params, lnp, *[_fake(i) for i in range(6, 11)], _fake(11), rs,
                                             ctypes.c_void_p(base), nb, None)
                _ok(L, rc, tag + " fwd", allow_unsupported=False)
            if save:
                for frozen in (False, True):
                    dparams = (ctypes.c_void_p * 12)(*[None if (frozen and i % 3 == 0) else 0x300000000000 + i * (1 << 32) for i in range(12)])
                    dln = (ctypes.c_void_p * 10)(*[None if (frozen and i % 4 == 1) else 0x380000000000 + i * (1 << 32) for i in range(10)]) if ln else None
                    douts = [_fake(12)] + [None if frozen else _fake(13 + i) for i in range(4)]
                    dins = [None if (frozen and i in (0, 3)) else _fake(20 + i) for i in range(5)]
                    rc = L.vpx_acstlstm_step_bwd(ctypes.byref(d), *[_fake(i) for i in range(1, 6)], params, lnp, _fake(11), rs, *douts, *dins, dparams, dln,
                                                 ctypes.c_void_p(WS_BASE_ODD), nb, None)
                    _ok(L, rc, tag + f" bwd frozen={frozen}", allow_unsupported=False)
                assert L.vpx_acstlstm_step_bwd(ctypes.byref(d), *[_fake(i) for i in range(1, 6)], params, lnp, _fake(11), rs, *douts, *dins, dparams, dln,
                                               ctypes.c_void_p(WS_BASE), nb - 4096, None) == E_WS
    d = ACSTLSTMDesc(2, 4, 8, 8, 8, 4, 0, 0, 0, 1.0)   # even kernel size
    assert L.vpx_acstlstm_workspace_bytes(ctypes.byref(d)) == 0


def test_small_workspaces(L):
    """LayerNorm, MSE."""
    for B in (1, 2, 8, 128):
        nb = L.vpx_layernorm_workspace_bytes(B)
        _ok(L, L.vpx_layernorm_fwd(_fake(1), _fake(2), _fake(3), _fake(4), _fake(5), _fake(6), B, 128 * 16 * 16, ctypes.c_void_p(WS_BASE), nb, None), "ln fwd", False)
        _ok(L, L.vpx_layernorm_bwd(_fake(1), _fake(2), _fake(3), _fake(4), _fake(5), _fake(6), _fake(7), B, 256, 128, ctypes.c_void_p(WS_BASE), nb, None), "ln bwd", False)
    nb = L.vpx_mse_loss_workspace_bytes()
    _ok(L, L.vpx_mse_loss(_fake(1), _fake(2), 128 * 10 * 64 * 64, 1280, 1.0, _fake(3), _fake(4), ctypes.c_void_p(WS_BASE), nb, None), "mse", False)


def test_random_shapes_fuzz():
    """tools/fuzz_contract.py: 4 000 random descriptors (odd map sizes, channel counts that are multiples of nothing, every kernel size,
    layout, operand mode and option bit) through every entry point that takes a workspace, in its own process (it switches the library
    into a dry run). Round 5: its first run found a 7x7 ConvLSTM data gradient over 4 * 288 gate channels whose 8-wave form needs 168 KB
    of LDS — refused at the launch with an 'invalid argument'; the layouts now fall back to the 4-wave form (conv_fits_lds)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for seed in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_contract.py"), "2000", str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
