"""The second-generation fused cell kernel (csrc/cell2.hip: pre-split bf16x3 operands, LDS-DMA staging, 32x16-pixel tiles)
against (a) the first-generation kernel — same operand split and products, equal up to fp32 summation order —
and (b) the pinned oracle (oracle/torch_ref.convlstm_hzzone_seq = conv_lstm_hzzone.py:38-70), forward and all
gradients. `vpx_set_option(VPX_OPT_CELL2, 2)` forces the kernel at batch sizes the default rule would hand to the
first-generation kernel; ragged maps (H, W not multiples of the tile), absent inputs / states and both gate orders are
covered."""
import numpy as np
import pytest
import torch

from golden_util import name_seed, seeded_rand, seeded_randn

pytestmark = pytest.mark.gpu


from parity import relmax as _relmax   # max|a - b| / max|b|, recorded (tests/parity.py)


@pytest.fixture
def cell2_switch(vpx):
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_CELL2, 1)

    def set_mode(v):
        L.vpx_set_option(vpx._lib.OPT_CELL2, v)
    yield set_mode
    L.vpx_set_option(vpx._lib.OPT_CELL2, prev)


CASES = {  # tag: (Cin, Ch, H, W, B, T, with_x, with_state, peephole, gate_order)
    "enc1_ragged": (16, 64, 40, 48, 3, 3, True, False, True, 0),
    "enc2": (64, 96, 32, 32, 2, 3, True, False, True, 0),
    "fore1_states": (96, 64, 64, 64, 2, 2, True, True, True, 0),
    "fore3_noinput": (96, 96, 32, 32, 2, 3, False, True, True, 0),
    "ifog_nopeep": (32, 48, 24, 20, 2, 3, True, True, False, 1),
}


def _inputs(tag):
    Cin, Ch, H, W, B, T, with_x, with_state, peep, order = CASES[tag]
    p = f"cell2.{tag}."
    d = {"W": seeded_randn((4 * Ch, Cin + Ch, 3, 3), name_seed(p + "W"), 1.0 / np.sqrt((Cin + Ch) * 9.0)),
         "b": seeded_randn((4 * Ch,), name_seed(p + "b"), 0.1),
         "x": seeded_rand((B, T, Cin, H, W), name_seed(p + "x")) if with_x else None,
         "h0": seeded_randn((B, Ch, H, W), name_seed(p + "h0"), 0.5) if with_state else None,
         "c0": seeded_randn((B, Ch, H, W), name_seed(p + "c0"), 0.5) if with_state else None,
         "g_out": seeded_randn((B, T, Ch, H, W), name_seed(p + "g_out")),
         "g_cT": seeded_randn((B, Ch, H, W), name_seed(p + "g_cT"))}
    for n in ("Wci", "Wcf", "Wco"):
        d[n] = seeded_randn((1, Ch, H, W), name_seed(p + n), 0.1) if peep else None
    return d


def _run(vpx, tag, grads, precision="bf16x3"):
    Cin, Ch, H, W, B, T, with_x, with_state, peep, order = CASES[tag]
    inp = {k: (None if v is None else v.cuda()) for k, v in _inputs(tag).items()}
    leaves = {}
    if grads:
        for k in ("x", "h0", "c0", "W", "b", "Wci", "Wcf", "Wco"):
            if inp[k] is not None:
                leaves[k] = inp[k].clone().requires_grad_(True)
                inp[k] = leaves[k]
    out, hT, cT = vpx.ops.convlstm_seq(inp["x"], inp["h0"], inp["c0"], inp["W"], inp["b"], inp["Wci"], inp["Wcf"], inp["Wco"],
                                       seq_len=T, in_channels=Cin, gate_order=order, precision=precision)
    if grads:
        ((out * inp["g_out"]).sum() + (cT * inp["g_cT"]).sum()).backward()
    return out.detach(), hT.detach(), cT.detach(), {k: v.grad for k, v in leaves.items()}


def _oracle(tag):
    from oracle import torch_ref as tr
    Cin, Ch, H, W, B, T, with_x, with_state, peep, order = CASES[tag]
    inp = _inputs(tag)
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("x", "h0", "c0", "W", "b", "Wci", "Wcf", "Wco") if inp[k] is not None}
    z = torch.zeros(1, Ch, H, W)
    Wm, bm = leaves["W"], leaves["b"]
    if order == 1:  # reference rows are (i, f, o, g): the hzzone restatement wants (i, f, g, o)
        perm = torch.cat([torch.arange(0, 2 * Ch), torch.arange(3 * Ch, 4 * Ch), torch.arange(2 * Ch, 3 * Ch)])
        Wm, bm = Wm[perm], bm[perm]
    states = (leaves["h0"], leaves["c0"]) if with_state else None
    x = leaves.get("x")
    out, (hT, cT) = tr.convlstm_hzzone_seq(x, states, T, Wm, bm, leaves.get("Wci", z), leaves.get("Wcf", z), leaves.get("Wco", z))
    ((out * inp["g_out"]).sum() + (cT * inp["g_cT"]).sum()).backward()
    return out.detach(), hT.detach(), cT.detach(), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("tag", list(CASES))
def test_cell2_bit_identical_to_first_generation_and_matches_oracle(vpx, cell2_switch, tag):
    cell2_switch(0)
    o1, h1, c1, _ = _run(vpx, tag, grads=False)
    cell2_switch(2)
    o2, h2, c2, _ = _run(vpx, tag, grads=False)
    # same operand split, same products; the first generation sums a 32/64-channel stage tap by tap where it picks larger
    # stages, the second always 16 channels per tap: equal up to fp32 summation order
    assert _relmax(o2, o1) < 2e-6 and _relmax(c2, c1) < 2e-6 and _relmax(h2, h1) < 2e-6
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5 and _relmax(h2, rh) < 2e-5


CASES.update({  # plain-bf16 shapes: odd stage counts (a five-step last period), the shortest K loops, a long one, many tiles per CU
    "plain_s5": (16, 64, 32, 32, 2, 3, True, True, True, 0),
    "plain_s3": (16, 32, 32, 16, 3, 2, True, False, False, 1),
    "plain_s2_noinput": (32, 32, 16, 32, 2, 3, False, True, True, 0),
    "plain_s14": (96, 128, 32, 32, 2, 2, True, True, True, 0),
    "plain_s7_many_tiles": (48, 64, 64, 64, 24, 2, True, False, True, 0),
})


@pytest.mark.parametrize("tag", ["enc2", "fore1_states", "fore3_noinput", "plain_s5", "plain_s3", "plain_s2_noinput", "plain_s14",
                                 "plain_s7_many_tiles"])
def test_plain_bf16_on_the_q_kernel_matches_first_generation_bf16(vpx, cell2_switch, tag):
    """VPX_PREC_BF16 (BASELINE configs[1]'s literal dtype; hi parts only, one MFMA per product, NOT inside the 1e-4 bar) on
    cell2_kernel_q<.., 4, true> (inference, maps in whole 16x16 tiles) against the first-generation kernel's bf16 mode — the same
    operand rounding, fp32 summation order differs — and against the oracle at the mode's own tolerance, max|d| / max|ref| <= 1e-2."""
    cell2_switch(0)
    with torch.no_grad():
        o1, h1, c1, _ = _run(vpx, tag, grads=False, precision="bf16")
        cell2_switch(2)
        o2, h2, c2, _ = _run(vpx, tag, grads=False, precision="bf16")
        o3, _, _, _ = _run(vpx, tag, grads=False, precision="bf16")
    assert torch.equal(o2, o3)
    # recurrence: a one-ulp difference in a rounded h_t operand is a 2^-9 relative change of that operand; 3 steps stay below 1e-3
    assert _relmax(o2, o1) < 2e-3 and _relmax(c2, c1) < 2e-3 and _relmax(h2, h1) < 2e-3, (_relmax(o2, o1), _relmax(c2, c1))
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 1e-2 and _relmax(c2, rc) < 1e-2
    assert _relmax(o2, ro) > 1e-5   # it really is the reduced-precision mode


@pytest.mark.parametrize("tag", ["enc1_ragged", "fore1_states", "ifog_nopeep"])
def test_cell2_training_path_vs_oracle(vpx, cell2_switch, tag):
    """Forward with the saved-for-backward reserve filled by cell2, BPTT on top of it: every gradient against autograd."""
    cell2_switch(2)
    out, hT, cT, g = _run(vpx, tag, grads=True)
    ro, rh, rc, rg = _oracle(tag)
    assert _relmax(out, ro) < 2e-5
    for k in rg:
        assert _relmax(g[k], rg[k]) < 5e-5, k


def test_set_option_contract(vpx):
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_CELL2, 0)
    assert L.vpx_set_option(vpx._lib.OPT_CELL2, prev) == 0
    assert L.vpx_set_option(12345, 1) < 0 and b"unknown option" in L.vpx_last_error()


# ---- the small-grid sliced step (csrc/cell3.hip): 16x16-pixel tiles x 8-channel slices, weights resident in LDS ----
CASES3 = {  # eligible shapes: channels in 16s (<= 96), maps in whole 16x16 tiles, a grid below 256 first-generation workgroups
    "enc2_b4": (64, 96, 32, 32, 4, 3, True, False, True, 0),
    "enc3_states": (96, 96, 16, 16, 3, 3, True, True, True, 0),
    "fore1_noinput": (96, 96, 16, 16, 4, 3, False, True, True, 0),
    "ifog_nopeep_small": (32, 48, 32, 16, 2, 4, True, True, False, 1),
    "t1_single_step": (16, 64, 32, 32, 2, 1, True, True, True, 0),
}
CASES.update(CASES3)


@pytest.fixture
def cell3_switch(vpx):
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_CELL3, 1)

    def set_mode(v):
        L.vpx_set_option(vpx._lib.OPT_CELL3, v)
    yield set_mode
    L.vpx_set_option(vpx._lib.OPT_CELL3, prev)


@pytest.mark.parametrize("tag", list(CASES3))
def test_cell3_matches_split_path_and_oracle_and_is_reproducible(vpx, cell3_switch, tag):
    cell3_switch(0)
    o1, h1, c1, _ = _run(vpx, tag, grads=False)      # K-split convolution + pointwise gate kernel (or the fused first generation)
    cell3_switch(1)
    o2, h2, c2, _ = _run(vpx, tag, grads=False)
    assert _relmax(o2, o1) < 2e-6 and _relmax(c2, c1) < 2e-6 and _relmax(h2, h1) < 2e-6
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5 and _relmax(h2, rh) < 2e-5
    o3, h3, c3, _ = _run(vpx, tag, grads=False)      # no atomics on this path: bit-identical from run to run
    assert torch.equal(o2, o3) and torch.equal(c2, c3)


@pytest.mark.parametrize("tag", ["enc2_b4", "enc3_states", "ifog_nopeep_small"])
def test_cell3_training_path_vs_oracle(vpx, cell3_switch, tag):
    """cell3 fills the saved-for-backward reserve (gates, cell states); BPTT on top of it against autograd."""
    cell3_switch(1)
    out, hT, cT, g = _run(vpx, tag, grads=True)
    ro, rh, rc, rg = _oracle(tag)
    assert _relmax(out, ro) < 2e-5
    for k in rg:
        assert _relmax(g[k], rg[k]) < 5e-5, k


@pytest.mark.parametrize("tag,force2", [("enc2_b4", 0), ("fore1_states", 2)])
def test_c_abi_nchw_layout_on_operand_format_kernels(vpx, cell2_switch, cell3_switch, tag, force2):
    """The raw C ABI on reference-layout (NCHW) buffers, bf16x3, at shapes the small-grid kernel (cell3) resp. the second-generation
    kernel (cell2 forward, conv2 data gradient) take: forward + backward against the oracle's autograd."""
    import ctypes
    L = vpx._lib.lib()
    cell3_switch(1)
    cell2_switch(force2 if force2 else 1)
    Cin, Ch, H, W, B, T, with_x, with_state, peep, order = CASES[tag]
    inp = {k: (None if v is None else v.cuda().contiguous()) for k, v in _inputs(tag).items()}
    d = vpx._lib.ConvLSTMDesc(B, T, Cin, Ch, H, W, 3, 3, order, vpx._lib.LAYOUT_NCHW, vpx._lib.PREC_BF16X3, vpx._lib.FLAG_SAVE_FOR_BWD)
    ws_bytes, rs_bytes = L.vpx_convlstm_workspace_bytes(ctypes.byref(d)), L.vpx_convlstm_reserve_bytes(ctypes.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    rs = torch.empty(rs_bytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(B, T, Ch, H, W, device="cuda")
    hT, cT = torch.empty(B, Ch, H, W, device="cuda"), torch.empty(B, Ch, H, W, device="cuda")
    p = vpx._lib.ptr
    rc = L.vpx_convlstm_seq_fwd(ctypes.byref(d), p(inp["x"]), p(inp["h0"]), p(inp["c0"]), p(inp["W"]), p(inp["b"]), p(inp["Wci"]),
                                p(inp["Wcf"]), p(inp["Wco"]), p(out), p(hT), p(cT), p(rs), rs_bytes, p(ws), ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    names = [("dx", "x"), ("dh0", "h0"), ("dc0", "c0"), ("dW", "W"), ("db", "b"), ("dWci", "Wci"), ("dWcf", "Wcf"), ("dWco", "Wco")]
    G = {n: (torch.empty_like(inp[m]) if inp[m] is not None else None) for n, m in names}
    rc = L.vpx_convlstm_seq_bwd(ctypes.byref(d), p(inp["x"]), p(inp["h0"]), p(inp["c0"]), p(inp["W"]), p(inp["Wci"]), p(inp["Wcf"]),
                                p(inp["Wco"]), p(out), p(rs), rs_bytes, p(inp["g_out"]), None, p(inp["g_cT"]), *[p(G[n]) for n, _ in names],
                                p(ws), ws_bytes, None)
    assert rc == 0, L.vpx_last_error()
    torch.cuda.synchronize()
    ro, rh, rc_, rg = _oracle(tag)
    assert _relmax(out, ro) < 2e-5 and _relmax(cT, rc_) < 2e-5 and _relmax(hT, rh) < 2e-5
    for n, m in names:
        if G[n] is not None and m in rg:
            assert _relmax(G[n], rg[m]) < 5e-5, n


# ---- the 16x16x32 MFMA form of the second-generation main loop (cell2_kernel_q: fused step and conv2 data gradient) ----
CASESQ = {  # full 32x16 tiles and whole 32-channel tiles for the fused step; the K = 32 steps pair taps over the PRESENT stages
    "q_enc1_oddx": (16, 64, 64, 32, 2, 3, True, False, True, 0),      # x-only pack at t = 0 (S = 1), odd x|h boundary (S = 5)
    "q_odd_total_states": (48, 64, 32, 32, 2, 2, True, True, True, 0),  # S = 7: the last period has no odd stage
    "q_enc2": (64, 96, 32, 32, 2, 3, True, False, True, 0),
    "q_fore3_noinput": (96, 96, 32, 32, 2, 3, False, True, True, 0),   # h-only pack
    "q_ifog_nopeep": (32, 32, 32, 48, 2, 3, True, True, False, 1),
    "q_t1_nostate": (32, 64, 32, 16, 3, 1, True, False, True, 0),      # a single step, x only
    "q_h16_enc3": (96, 96, 16, 16, 3, 2, True, False, True, 0),        # 16-row maps: the half tile only (one tile = one image)
    "q_h48": (32, 32, 48, 32, 2, 2, True, True, True, 0),              # H % 32 == 16: half tiles inside the image, 32-row tiles not
}
CASES.update(CASESQ)


@pytest.fixture
def shape_switch(vpx):
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_MFMA_SHAPE, 0)

    def set_mode(v):
        L.vpx_set_option(vpx._lib.OPT_MFMA_SHAPE, v)
    yield set_mode
    L.vpx_set_option(vpx._lib.OPT_MFMA_SHAPE, prev)


@pytest.mark.parametrize("tag", list(CASESQ))
def test_mfma_16x16x32_form_matches_32x32x16_form_and_oracle(vpx, cell2_switch, shape_switch, tag):
    cell2_switch(2)
    shape_switch(0)
    o1, h1, c1, _ = _run(vpx, tag, grads=False)
    shape_switch(1)
    o2, h2, c2, _ = _run(vpx, tag, grads=False)
    assert _relmax(o2, o1) < 2e-6 and _relmax(c2, c1) < 2e-6 and _relmax(h2, h1) < 2e-6   # same products, other summation order
    assert not torch.equal(o2, o1)    # ... and it really is another kernel
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5 and _relmax(h2, rh) < 2e-5
    o3, _, c3, _ = _run(vpx, tag, grads=False)
    assert torch.equal(o2, o3) and torch.equal(c2, c3)   # no atomics: bit-identical run to run


@pytest.mark.parametrize("tag", ["q_enc1_oddx", "q_odd_total_states", "q_ifog_nopeep", "enc1_ragged", "fore1_states"])
def test_mfma_16x16x32_training_path_vs_oracle(vpx, cell2_switch, shape_switch, tag):
    """Forward on the q-form cell where it applies (enc1_ragged: ragged map, so only the data gradient takes the q form —
    its epilogue handles partial tiles), conv2 data gradient on the q form: every gradient against autograd."""
    cell2_switch(2)
    shape_switch(1)
    out, hT, cT, g = _run(vpx, tag, grads=True)
    ro, rh, rc, rg = _oracle(tag)
    assert _relmax(out, ro) < 2e-5
    for k in rg:
        assert _relmax(g[k], rg[k]) < 5e-5, k


# ---- the half tile (cell2_kernel_q<.., 4>: 16x16-pixel tiles, two workgroups per CU, weight ring of two chunks in halves) ----
@pytest.fixture
def experiment_switch(vpx):
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 0)

    def set_mode(v):
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, v)
    yield set_mode
    L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)


@pytest.mark.parametrize("tag", list(CASESQ))
def test_small_grid_32_column_tiles_match_the_half_tile(vpx, cell2_switch, shape_switch, experiment_switch, tag):
    """c3 (c5_kernel<4, 3> — the product's choice — and c5_kernel<2, 3>, VPX_OPT_EXPERIMENT bit 13; round 4): the fused step on
    16x16-pixel tiles x (4 gates x 16 | 8 channels), 8-channel stages, four stage buffers — what inference runs where the half tile
    has at most 256 workgroups (every case here). Same operand split and products, other summation order: 2e-6 against the half tile
    (bit 12), 2e-5 against the oracle, bit-identical run to run; x-only / h-only / x + h packs, both gate orders, with and without
    peepholes, odd stage counts."""
    cell2_switch(2)
    shape_switch(1)
    ro, rh, rc, _ = _oracle(tag)
    with torch.no_grad():
        experiment_switch(4096)
        o1, h1, c1, _ = _run(vpx, tag, grads=False)
        for bits in (0, 8192):   # 64-column tiles, 32-column tiles
            experiment_switch(bits)
            o2, h2, c2, _ = _run(vpx, tag, grads=False)
            o3, _, _, _ = _run(vpx, tag, grads=False)
            assert torch.equal(o2, o3), bits
            assert _relmax(o2, o1) < 2e-6 and _relmax(c2, c1) < 2e-6 and _relmax(h2, h1) < 2e-6, (bits, _relmax(o2, o1), _relmax(c2, c1))
            assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5 and _relmax(h2, rh) < 2e-5, bits


@pytest.mark.parametrize("tag", list(CASESQ))
def test_half_tile_bit_identical_to_full_tile(vpx, cell2_switch, shape_switch, experiment_switch, tag):
    """Same products in the same order per output element: the two tile forms must agree bit for bit."""
    cell2_switch(2)
    shape_switch(1)
    experiment_switch(4)
    o1, h1, c1, _ = _run(vpx, tag, grads=False)
    experiment_switch(2)
    o2, h2, c2, _ = _run(vpx, tag, grads=False)
    assert torch.equal(o2, o1) and torch.equal(c2, c1) and torch.equal(h2, h1)
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5


@pytest.mark.parametrize("tag", ["q_enc1_oddx", "q_odd_total_states", "q_ifog_nopeep", "enc1_ragged", "fore1_states"])
def test_half_tile_training_path(vpx, cell2_switch, shape_switch, experiment_switch, tag):
    cell2_switch(2)
    shape_switch(1)
    experiment_switch(4)
    out1, _, _, g1 = _run(vpx, tag, grads=True)
    experiment_switch(2)
    out2, _, _, g2 = _run(vpx, tag, grads=True)
    assert torch.equal(out1, out2)
    for k in g1:
        if g1[k] is not None:
            assert torch.equal(g1[k], g2[k]), k
    _, _, _, rg = _oracle(tag)
    for k in rg:
        assert _relmax(g2[k], rg[k]) < 5e-5, k


@pytest.mark.parametrize("tag", ["enc2_b4", "enc3_states", "t1_single_step"])
def test_hoisted_projection_on_convq_matches_first_generation_launch(vpx, cell3_switch, experiment_switch, tag):
    """Small-grid path (cell3): W_x * x_t of all frames as one convq launch (default) against the first-generation launch
    (VPX_OPT_EXPERIMENT bit 5): same bf16x3 products, other summation order."""
    cell3_switch(1)
    experiment_switch(32)
    o1, h1, c1, _ = _run(vpx, tag, grads=False)
    experiment_switch(0)
    o2, h2, c2, _ = _run(vpx, tag, grads=False)
    assert _relmax(o2, o1) < 2e-6 and _relmax(c2, c1) < 2e-6
    assert not torch.equal(o2, o1)
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5


# ---- the eight-wave half tile (csrc/cell2x.hip, round 6): 64-register wave tiles, four waves per SIMD; VPX_OPT_EXPERIMENT bit 15 selects
#      it, bit 16 swaps its wave split (columns <-> rows) ----
X_FORM, X_SWAP = 32768, 65536


@pytest.mark.parametrize("tag", list(CASESQ) + ["plain_s14", "plain_s7_many_tiles"])
def test_x_form_bit_identical_to_half_tile(vpx, cell2_switch, shape_switch, experiment_switch, tag):
    """cell2_kernel_x (both wave splits) multiplies the same operand pieces in the same order per output element as the four-wave half
    tile: bit-identical h, c and output sequence; and the oracle's bound on top."""
    cell2_switch(2)
    shape_switch(1)
    with torch.no_grad():
        experiment_switch(0)
        o1, h1, c1, _ = _run(vpx, tag, grads=False)
        for bits in (X_FORM, X_FORM | X_SWAP):
            experiment_switch(bits)
            o2, h2, c2, _ = _run(vpx, tag, grads=False)
            assert torch.equal(o2, o1) and torch.equal(c2, c1) and torch.equal(h2, h1), (bits, _relmax(o2, o1), _relmax(c2, c1))
    ro, rh, rc, _ = _oracle(tag)
    assert _relmax(o2, ro) < 2e-5 and _relmax(c2, rc) < 2e-5 and _relmax(h2, rh) < 2e-5


@pytest.mark.parametrize("tag", ["enc2", "fore1_states", "fore3_noinput", "plain_s5", "plain_s3", "plain_s2_noinput", "plain_s14",
                                 "plain_s7_many_tiles"])
def test_x_form_plain_bf16_bit_identical_to_half_tile(vpx, cell2_switch, experiment_switch, tag):
    """VPX_PREC_BF16 on cell2_kernel_x<true, *> (both wave splits; the copies are divided between the two wave groups) against
    cell2_kernel_q<.., 4, true>."""
    cell2_switch(2)
    with torch.no_grad():
        experiment_switch(0)
        o1, h1, c1, _ = _run(vpx, tag, grads=False, precision="bf16")
        for bits in (X_FORM, X_FORM | X_SWAP):
            experiment_switch(bits)
            o2, h2, c2, _ = _run(vpx, tag, grads=False, precision="bf16")
            assert torch.equal(o2, o1) and torch.equal(c2, c1) and torch.equal(h2, h1), (bits, _relmax(o2, o1), _relmax(c2, c1))


@pytest.mark.parametrize("tag", ["q_enc1_oddx", "q_odd_total_states", "q_ifog_nopeep", "fore1_states"])
def test_x_form_training_path_vs_oracle(vpx, cell2_switch, shape_switch, experiment_switch, tag):
    """The x form fills the saved-for-backward reserve (gates, c, split h): every gradient against autograd, both wave splits."""
    cell2_switch(2)
    shape_switch(1)
    ro, rh, rc, rg = _oracle(tag)
    for bits in (X_FORM, X_FORM | X_SWAP):
        experiment_switch(bits)
        out, hT, cT, g = _run(vpx, tag, grads=True)
        assert _relmax(out, ro) < 2e-5
        for k in rg:
            assert _relmax(g[k], rg[k]) < 5e-5, (bits, k)
