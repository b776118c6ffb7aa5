"""GPU parity tests (run on the MI355X box: pytest -m gpu): the HIP ConvLSTM path, called through the C ABI
(include/vpx.h), against the reference-generated golden vectors and against the pinned oracle on seeded inputs.

Tolerance: north_star demands 1e-4 relative (fp32); the reference's own tests use atol=1e-4 (_convlstm_hzzone.py:91).
The exact-fp32 MFMA path is held to 1e-5 of the tensor's max magnitude."""
import ctypes

import numpy as np
import pytest
import torch

import golden_cases as gc
from golden_util import load_golden, name_seed, seeded_rand, seeded_randn

pytestmark = pytest.mark.gpu

RTOL = 1e-5


from parity import relmax as _relmax   # max|a - b| / max|b|, recorded (tests/parity.py)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", list(gc.HZZONE_CASES))
@pytest.mark.parametrize("mode", ["full", "states", "noinput"])
def test_hzzone_block_vs_golden(vpx, dev, tag, mode):
    Cin, Ch, H, W, k, B, T, _ = gc.HZZONE_CASES[tag]
    inp = {n: v.to(dev) for n, v in gc.hzzone_inputs(tag, Cin, Ch, H, W, k, B, T).items()}
    g = load_golden(f"hzzone_{tag}_{mode}")
    x = None if mode == "noinput" else inp["x"]
    h0 = None if mode == "full" else inp["h0"]
    c0 = None if mode == "full" else inp["c0"]
    out, hT, cT = vpx.ops.convlstm_seq(x, h0, c0, inp["W"], inp["b"], inp["Wci"], inp["Wcf"], inp["Wco"], seq_len=T,
                                       in_channels=Cin)
    assert out.shape == (B, T, Ch, H, W)
    assert _relmax(out, g["out"]) < RTOL and _relmax(hT, g["hT"]) < RTOL and _relmax(cT, g["cT"]) < RTOL


@pytest.mark.parametrize("tag", list(gc.NDRPLZ_CELL_CASES))
def test_ndrplz_cell_vs_golden(vpx, dev, tag):
    Cin, Ch, H, W, kh, kw, bias, B = gc.NDRPLZ_CELL_CASES[tag]
    inp = {n: v.to(dev) for n, v in gc.ndrplz_cell_inputs(tag, Cin, Ch, H, W, kh, kw, bias, B).items()}
    g = load_golden(f"ndrplz_cell_{tag}")
    out, hT, cT = vpx.ops.convlstm_seq(inp["x"][:, None], inp["h"], inp["c"], inp["W"], inp["b"] if bias else None,
                                       seq_len=1, in_channels=Cin, gate_order=vpx._lib.GATE_IFOG)
    assert _relmax(hT, g["h_next"]) < RTOL and _relmax(cT, g["c_next"]) < RTOL
    assert _relmax(out[:, 0], g["h_next"]) < RTOL


def test_c_abi_nchw_layout(vpx, dev):
    """Calls the C ABI directly on reference-layout (NCHW) buffers: the drop-in form of the boundary."""
    L = vpx._lib.lib()
    Cin, Ch, H, W, k, B, T, _ = gc.HZZONE_CASES["tiny"]
    inp = {n: v.to(dev).contiguous() for n, v in gc.hzzone_inputs("tiny", Cin, Ch, H, W, k, B, T).items()}
    g = load_golden("hzzone_tiny_states")
    d = vpx._lib.ConvLSTMDesc(B, T, Cin, Ch, H, W, k, k, vpx._lib.GATE_IFGO, vpx._lib.LAYOUT_NCHW, vpx._lib.PREC_F32, 0)
    ws_bytes = L.vpx_convlstm_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty(B, T, Ch, H, W, device=dev)
    hT = torch.empty(B, Ch, H, W, device=dev)
    cT = torch.empty(B, Ch, H, W, device=dev)
    p = vpx._lib.ptr
    rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), p(inp["x"]), p(inp["h0"]), p(inp["c0"]), p(inp["W"]), p(inp["b"]),
                                p(inp["Wci"]), p(inp["Wcf"]), p(inp["Wco"]), p(out), p(hT), p(cT), None, 0, p(ws),
                                ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    torch.cuda.synchronize()
    assert _relmax(out, g["out"]) < RTOL and _relmax(hT, g["hT"]) < RTOL and _relmax(cT, g["cT"]) < RTOL


def test_c_abi_error_paths(vpx, dev):
    L = vpx._lib.lib()
    d = vpx._lib.ConvLSTMDesc(1, 1, 4, 4, 8, 8, 4, 4, 0, 0, 0, 0)  # even kernel size
    assert L.vpx_convlstm_workspace_bytes(ctypes.byref(d)) == 0
    d = vpx._lib.ConvLSTMDesc(1, 1, 4, 4, 8, 8, 3, 3, 0, 0, 0, 0)
    rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), *([None] * 11), None, 0, None, 0, None)
    assert rc == -1 and b"NULL" in L.vpx_last_error()
    with pytest.raises(ValueError):
        vpx.ops.convlstm_seq(None, None, None, torch.zeros(16, 8, 3, 3, device=dev), None, seq_len=1, in_channels=4)


# the six ConvLSTM block shapes of convlstm-shi at 64x64 (ef_conv_lstm.py:31-33) + the "64x64x64ch" headline cell
REAL_SHAPES = [(16, 64, 64, 64), (64, 96, 32, 32), (96, 96, 16, 16), (96, 96, 32, 32), (96, 64, 64, 64),
               (64, 64, 64, 64)]


@pytest.mark.parametrize("Cin,Ch,H,W", REAL_SHAPES)
def test_real_block_shapes_vs_oracle(vpx, dev, Cin, Ch, H, W):
    from oracle import torch_ref as tr
    B, T, k = 2, 3, 3
    tag = f"real.{Cin}.{Ch}.{H}"
    Wt = seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * 9))
    b = seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1)
    pw = [seeded_randn((1, Ch, H, W), name_seed(tag + n), 0.1) for n in ("ci", "cf", "co")]
    x = seeded_rand((B, T, Cin, H, W), name_seed(tag + "x"))
    h0 = seeded_randn((B, Ch, H, W), name_seed(tag + "h0"), 0.5)
    c0 = seeded_randn((B, Ch, H, W), name_seed(tag + "c0"), 0.5)
    with torch.no_grad():
        ro, (rh, rc) = tr.convlstm_hzzone_seq(x, (h0, c0), T, Wt, b, *pw, padding=1)
    out, hT, cT = vpx.ops.convlstm_seq(x.to(dev), h0.to(dev), c0.to(dev), Wt.to(dev), b.to(dev),
                                       *[p.to(dev) for p in pw], seq_len=T, in_channels=Cin)
    assert _relmax(out, ro) < RTOL and _relmax(cT, rc) < RTOL and _relmax(hT, rh) < RTOL
    # zero-input form used by forecaster.rnn3 (ef_blocks.py:109-110)
    with torch.no_grad():
        ro2, _ = tr.convlstm_hzzone_seq(None, (h0, c0), T, Wt, b, *pw, padding=1)
    out2, _, _ = vpx.ops.convlstm_seq(None, h0.to(dev), c0.to(dev), Wt.to(dev), b.to(dev), *[p.to(dev) for p in pw],
                                      seq_len=T, in_channels=Cin)
    assert _relmax(out2, ro2) < RTOL


def test_conv2d_same_vs_torch(vpx, dev):
    for (Ci, Co, k, H, W) in [(128, 16, 1, 16, 16), (24, 40, 3, 9, 21), (7, 130, 5, 12, 10)]:
        x = seeded_randn((3, Ci, H, W), name_seed(f"c2d.x{Ci}{k}"))
        w = seeded_randn((Co, Ci, k, k), name_seed(f"c2d.w{Ci}{k}"), 1.0 / np.sqrt(Ci * k * k))
        b = seeded_randn((Co,), name_seed(f"c2d.b{Ci}{k}"), 0.1)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=k // 2).float()
        y = vpx.ops.conv2d_same(x.to(dev), w.to(dev), b.to(dev))
        assert _relmax(y, ref) < RTOL


# ------------------------------------------------------------------------------------------------------------------
# BPTT
# ------------------------------------------------------------------------------------------------------------------
GRTOL = 2e-5  # gradients: relative to the tensor's max magnitude


@pytest.mark.parametrize("tag", ["tiny", "k5"])
@pytest.mark.parametrize("mode", ["full", "states", "noinput"])
def test_hzzone_block_bwd_vs_golden(vpx, dev, tag, mode):
    Cin, Ch, H, W, k, B, T, _ = gc.HZZONE_CASES[tag]
    inp = {n: v.to(dev) for n, v in gc.hzzone_inputs(tag, Cin, Ch, H, W, k, B, T).items()}
    g = load_golden(f"hzzone_{tag}_{mode}")
    leaves = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h0", "c0", "W", "b", "Wci", "Wcf", "Wco")}
    x = None if mode == "noinput" else leaves["x"]
    h0 = None if mode == "full" else leaves["h0"]
    c0 = None if mode == "full" else leaves["c0"]
    out, hT, cT = vpx.ops.convlstm_seq(x, h0, c0, leaves["W"], leaves["b"], leaves["Wci"], leaves["Wcf"], leaves["Wco"],
                                       seq_len=T, in_channels=Cin)
    assert _relmax(out, g["out"]) < RTOL
    loss = (out * inp["g_out"]).sum() + (hT * inp["g_hT"]).sum() + (cT * inp["g_cT"]).sum()
    loss.backward()
    names = {"dW": "W", "db": "b", "dWci": "Wci", "dWcf": "Wcf", "dWco": "Wco", "dx": "x", "dh0": "h0", "dc0": "c0"}
    for gname, lname in names.items():
        if gname in g:
            assert leaves[lname].grad is not None, gname
            assert _relmax(leaves[lname].grad, g[gname]) < GRTOL, gname


@pytest.mark.parametrize("tag", list(gc.NDRPLZ_CELL_CASES))
def test_ndrplz_cell_bwd_vs_golden(vpx, dev, tag):
    Cin, Ch, H, W, kh, kw, bias, B = gc.NDRPLZ_CELL_CASES[tag]
    inp = {n: v.to(dev) for n, v in gc.ndrplz_cell_inputs(tag, Cin, Ch, H, W, kh, kw, bias, B).items()}
    g = load_golden(f"ndrplz_cell_{tag}")
    lv = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h", "c", "W", "b")}
    _, hT, cT = vpx.ops.convlstm_seq(lv["x"][:, None], lv["h"], lv["c"], lv["W"], lv["b"] if bias else None,
                                     seq_len=1, in_channels=Cin, gate_order=vpx._lib.GATE_IFOG)
    ((hT * inp["g_h"]).sum() + (cT * inp["g_c"]).sum()).backward()
    assert _relmax(lv["x"].grad, g["dx"]) < GRTOL and _relmax(lv["h"].grad, g["dh"]) < GRTOL
    assert _relmax(lv["c"].grad, g["dc"]) < GRTOL and _relmax(lv["W"].grad, g["dW"]) < GRTOL
    if bias:
        assert _relmax(lv["b"].grad, g["db"]) < GRTOL


@pytest.mark.parametrize("Cin,Ch,H,W", [(16, 64, 64, 64), (96, 96, 16, 16), (96, 64, 32, 32)])
def test_real_block_shapes_bwd_vs_oracle(vpx, dev, Cin, Ch, H, W):
    """Real block shapes: gradients against autograd through the pinned torch restatement (fp32 CPU)."""
    from oracle import torch_ref as tr
    B, T, k = 2, 3, 3
    tag = f"realbwd.{Cin}.{Ch}.{H}"
    P = {"W": seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * 9)),
         "b": seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1),
         "Wci": seeded_randn((1, Ch, H, W), name_seed(tag + "ci"), 0.1),
         "Wcf": seeded_randn((1, Ch, H, W), name_seed(tag + "cf"), 0.1),
         "Wco": seeded_randn((1, Ch, H, W), name_seed(tag + "co"), 0.1),
         "x": seeded_rand((B, T, Cin, H, W), name_seed(tag + "x")),
         "h0": seeded_randn((B, Ch, H, W), name_seed(tag + "h0"), 0.5),
         "c0": seeded_randn((B, Ch, H, W), name_seed(tag + "c0"), 0.5)}
    g_out = seeded_randn((B, T, Ch, H, W), name_seed(tag + "go"))
    g_c = seeded_randn((B, Ch, H, W), name_seed(tag + "gc"))
    ref = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    ro, (rh, rc) = tr.convlstm_hzzone_seq(ref["x"], (ref["h0"], ref["c0"]), T, ref["W"], ref["b"], ref["Wci"],
                                          ref["Wcf"], ref["Wco"], padding=1)
    ((ro * g_out).sum() + (rc * g_c).sum()).backward()
    lv = {n: v.to(dev).requires_grad_(True) for n, v in P.items()}
    out, hT, cT = vpx.ops.convlstm_seq(lv["x"], lv["h0"], lv["c0"], lv["W"], lv["b"], lv["Wci"], lv["Wcf"], lv["Wco"],
                                       seq_len=T, in_channels=Cin)
    ((out * g_out.to(dev)).sum() + (cT * g_c.to(dev)).sum()).backward()
    for n in P:
        assert _relmax(lv[n].grad, ref[n].grad) < 5e-5, n


def test_c_abi_nchw_bwd(vpx, dev):
    """Forward + backward through the raw C ABI on reference-layout (NCHW) buffers."""
    L = vpx._lib.lib()
    Cin, Ch, H, W, k, B, T, _ = gc.HZZONE_CASES["tiny"]
    inp = {n: v.to(dev).contiguous() for n, v in gc.hzzone_inputs("tiny", Cin, Ch, H, W, k, B, T).items()}
    g = load_golden("hzzone_tiny_states")
    d = vpx._lib.ConvLSTMDesc(B, T, Cin, Ch, H, W, k, k, vpx._lib.GATE_IFGO, vpx._lib.LAYOUT_NCHW, vpx._lib.PREC_F32,
                              vpx._lib.FLAG_SAVE_FOR_BWD)
    ws_bytes = L.vpx_convlstm_workspace_bytes(ctypes.byref(d))
    rs_bytes = L.vpx_convlstm_reserve_bytes(ctypes.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    rs = torch.empty(rs_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty(B, T, Ch, H, W, device=dev)
    hT, cT = torch.empty(B, Ch, H, W, device=dev), torch.empty(B, Ch, H, W, device=dev)
    p = vpx._lib.ptr
    rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), p(inp["x"]), p(inp["h0"]), p(inp["c0"]), p(inp["W"]), p(inp["b"]),
                                p(inp["Wci"]), p(inp["Wcf"]), p(inp["Wco"]), p(out), p(hT), p(cT), p(rs), rs_bytes,
                                p(ws), ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    G = {n: torch.empty_like(inp[m]) for n, m in (("dx", "x"), ("dh0", "h0"), ("dc0", "c0"), ("dW", "W"), ("db", "b"),
                                                   ("dWci", "Wci"), ("dWcf", "Wcf"), ("dWco", "Wco"))}
    rc = L.vpx_convlstm_seq_bwd(ctypes.byref(d), p(inp["x"]), p(inp["h0"]), p(inp["c0"]), p(inp["W"]), p(inp["Wci"]),
                                p(inp["Wcf"]), p(inp["Wco"]), p(out), p(rs), rs_bytes, p(inp["g_out"]), p(inp["g_hT"]),
                                p(inp["g_cT"]), p(G["dx"]), p(G["dh0"]), p(G["dc0"]), p(G["dW"]), p(G["db"]),
                                p(G["dWci"]), p(G["dWcf"]), p(G["dWco"]), p(ws), ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    torch.cuda.synchronize()
    assert _relmax(out, g["out"]) < RTOL
    for n in G:
        assert _relmax(G[n], g[n]) < GRTOL, n


# ------------------------------------------------------------------------------------------------------------------
# split-bf16 operand mode ("bf16x3"): fp32-level accuracy, held to the north-star tolerance (1e-4 relative)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", list(gc.HZZONE_CASES))
def test_hzzone_block_bf16x3_vs_golden(vpx, dev, tag):
    Cin, Ch, H, W, k, B, T, with_grads = gc.HZZONE_CASES[tag]
    inp = {n: v.to(dev) for n, v in gc.hzzone_inputs(tag, Cin, Ch, H, W, k, B, T).items()}
    g = load_golden(f"hzzone_{tag}_states")
    lv = {n: inp[n].clone().requires_grad_(True) for n in ("x", "h0", "c0", "W", "b", "Wci", "Wcf", "Wco")}
    out, hT, cT = vpx.ops.convlstm_seq(lv["x"], lv["h0"], lv["c0"], lv["W"], lv["b"], lv["Wci"], lv["Wcf"], lv["Wco"],
                                       seq_len=T, in_channels=Cin, precision="bf16x3")
    assert _relmax(out, g["out"]) < 2e-5 and _relmax(cT, g["cT"]) < 2e-5
    if with_grads:
        ((out * inp["g_out"]).sum() + (hT * inp["g_hT"]).sum() + (cT * inp["g_cT"]).sum()).backward()
        for gname, lname in {"dW": "W", "db": "b", "dWci": "Wci", "dx": "x", "dh0": "h0", "dc0": "c0"}.items():
            assert _relmax(lv[lname].grad, g[gname]) < 1e-4, gname


@pytest.mark.parametrize("Cin,Ch,H,W", [(16, 64, 64, 64), (96, 96, 16, 16), (64, 64, 64, 64)])
def test_real_block_shapes_bf16x3_vs_oracle(vpx, dev, Cin, Ch, H, W):
    from oracle import torch_ref as tr
    B, T, k = 2, 4, 3
    tag = f"real.{Cin}.{Ch}.{H}"
    Wt = seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * 9))
    b = seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1)
    pw = [seeded_randn((1, Ch, H, W), name_seed(tag + n), 0.1) for n in ("ci", "cf", "co")]
    x = seeded_rand((B, T, Cin, H, W), name_seed(tag + "x"))
    with torch.no_grad():
        ro, (rh, rc) = tr.convlstm_hzzone_seq(x, None, T, Wt, b, *pw, padding=1)
        out, hT, cT = vpx.ops.convlstm_seq(x.to(dev), None, None, Wt.to(dev), b.to(dev), *[p.to(dev) for p in pw],
                                           seq_len=T, in_channels=Cin, precision="bf16x3")
    assert _relmax(out, ro) < 2e-5 and _relmax(cT, rc) < 2e-5


@pytest.mark.parametrize("Cin,Ch,H,W", [(16, 64, 32, 32), (96, 96, 16, 16)])
def test_real_block_shapes_bwd_bf16x3_vs_oracle(vpx, dev, Cin, Ch, H, W):
    """bf16x3 backward (dgrad through the conv kernel, wgrad through the transposing-LDS-read kernel) vs fp32 autograd."""
    from oracle import torch_ref as tr
    B, T, k = 2, 3, 3
    tag = f"realbwd3.{Cin}.{Ch}.{H}"
    P = {"W": seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * 9)),
         "b": seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1),
         "Wci": seeded_randn((1, Ch, H, W), name_seed(tag + "ci"), 0.1),
         "Wcf": seeded_randn((1, Ch, H, W), name_seed(tag + "cf"), 0.1),
         "Wco": seeded_randn((1, Ch, H, W), name_seed(tag + "co"), 0.1),
         "x": seeded_rand((B, T, Cin, H, W), name_seed(tag + "x")),
         "h0": seeded_randn((B, Ch, H, W), name_seed(tag + "h0"), 0.5),
         "c0": seeded_randn((B, Ch, H, W), name_seed(tag + "c0"), 0.5)}
    g_out = seeded_randn((B, T, Ch, H, W), name_seed(tag + "go"))
    ref = {n: v.clone().requires_grad_(True) for n, v in P.items()}
    ro, _ = tr.convlstm_hzzone_seq(ref["x"], (ref["h0"], ref["c0"]), T, ref["W"], ref["b"], ref["Wci"], ref["Wcf"],
                                   ref["Wco"], padding=1)
    (ro * g_out).sum().backward()
    lv = {n: v.to(dev).requires_grad_(True) for n, v in P.items()}
    out, hT, cT = vpx.ops.convlstm_seq(lv["x"], lv["h0"], lv["c0"], lv["W"], lv["b"], lv["Wci"], lv["Wcf"], lv["Wco"],
                                       seq_len=T, in_channels=Cin, precision="bf16x3")
    (out * g_out.to(dev)).sum().backward()
    for n in P:
        assert _relmax(lv[n].grad, ref[n].grad) < 1e-4, n


def test_eight_wave_workgroup_variant_vs_oracle(vpx, dev):
    """Launches with >= 512 workgroups use the 8-wave (16x16 tile) form of the bf16x3 kernel; make sure that path is
    exercised (B=16 on a 64x64 map = 512 workgroups) and matches the fp32 oracle, ragged image edges included."""
    from oracle import torch_ref as tr
    for (Cin, Ch, H, W, B) in [(16, 64, 64, 64, 16), (8, 32, 72, 40, 24)]:
        T, k = 2, 3
        tag = f"mw2.{Cin}.{Ch}.{H}"
        Wt = seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * 9))
        b = seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1)
        pw = [seeded_randn((1, Ch, H, W), name_seed(tag + n), 0.1) for n in ("ci", "cf", "co")]
        x = seeded_rand((B, T, Cin, H, W), name_seed(tag + "x"))
        g_out = seeded_randn((B, T, Ch, H, W), name_seed(tag + "go"))
        ref = {"x": x.clone().requires_grad_(True), "W": Wt.clone().requires_grad_(True)}
        ro, (rh, rc) = tr.convlstm_hzzone_seq(ref["x"], None, T, ref["W"], b, *pw, padding=1)
        (ro * g_out).sum().backward()
        lv = {"x": x.to(dev).requires_grad_(True), "W": Wt.to(dev).requires_grad_(True)}
        out, hT, cT = vpx.ops.convlstm_seq(lv["x"], None, None, lv["W"], b.to(dev), *[p.to(dev) for p in pw],
                                           seq_len=T, in_channels=Cin, precision="bf16x3")
        assert _relmax(out, ro) < 2e-5 and _relmax(cT, rc) < 2e-5
        (out * g_out.to(dev)).sum().backward()   # the data-gradient conv takes the 8-wave form as well
        assert _relmax(lv["x"].grad, ref["x"].grad) < 1e-4 and _relmax(lv["W"].grad, ref["W"].grad) < 1e-4


@pytest.mark.parametrize("k", [1, 3, 5])
@pytest.mark.parametrize("Cin,with_x,with_h0", [(16, True, True), (3, True, False), (16, False, True)])
def test_weight_gradient_forms_bf16x3_vs_oracle(vpx, dev, k, Cin, with_x, with_h0):
    """Every launch form of the bf16x3 weight gradient against fp32 autograd of the oracle: 1x1 (single tap), 3x3 (tap-group
    kernel: two LDS item buffers, multi-item walks per K slice, column tiles pairing x and h halves), 5x5 (8-wave / 128-row
    form), a ragged channel count (scalar-load path), no input tensor, zero vs given initial state, ragged image edges."""
    from oracle import torch_ref as tr
    B, T, Ch, H, W = 12, 3, 32, 40, 24
    tag = f"wgforms.{k}.{Cin}.{int(with_x)}.{int(with_h0)}"
    Wt = seeded_randn((4 * Ch, Cin + Ch, k, k), name_seed(tag + "W"), 1.0 / np.sqrt((Cin + Ch) * k * k))
    b = seeded_randn((4 * Ch,), name_seed(tag + "b"), 0.1)
    pw = [seeded_randn((1, Ch, H, W), name_seed(tag + n), 0.1) for n in ("ci", "cf", "co")]
    x = seeded_rand((B, T, Cin, H, W), name_seed(tag + "x")) if with_x else None
    st = (seeded_randn((B, Ch, H, W), name_seed(tag + "h0"), 0.5), seeded_randn((B, Ch, H, W), name_seed(tag + "c0"), 0.5)) \
        if with_h0 else None
    g_out = seeded_randn((B, T, Ch, H, W), name_seed(tag + "go"))
    rW, rb = Wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ro, _ = tr.convlstm_hzzone_seq(x, st, T, rW, rb, *pw, padding=k // 2)
    (ro * g_out).sum().backward()
    lW, lb = Wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    out, hT, cT = vpx.ops.convlstm_seq(None if x is None else x.to(dev), None if st is None else st[0].to(dev),
                                       None if st is None else st[1].to(dev), lW, lb, *[p.to(dev) for p in pw],
                                       seq_len=T, in_channels=Cin, precision="bf16x3")
    assert _relmax(out, ro) < 2e-5
    (out * g_out.to(dev)).sum().backward()
    # without an input tensor the x columns of dW get no contribution on either side (exact zeros)
    assert _relmax(lW.grad, rW.grad) < 1e-4 and _relmax(lb.grad, rb.grad) < 1e-4
    if not with_x:
        assert torch.count_nonzero(lW.grad[:, :Cin]) == 0
