"""The schedule-driven K = 32 convolution (csrc/convq.hip) behind vpx_conv2d_ex_fwd_from_split: the EF stage-glue layers of
convlstm-shi (ef_blocks.py:15-49, ef_conv_lstm.py:36-65) and their adjoint shapes on split-format input, against
torch.nn.functional convolutions in fp32 on the same GPU (the oracle's op for these layers) at the glue's tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import name_seed, seeded_rand, seeded_randn

pytestmark = pytest.mark.gpu


from parity import relmax as _relmax   # max|a - b| / max|b|, recorded (tests/parity.py)


CASES = {  # tag: (N, Ci, Co, H, W, k, stride, pad, transposed, slope)
    "deconv2_t2k4": (3, 96, 96, 32, 32, 4, 2, 1, True, 0.2),
    "deconv1_t2k4_small": (2, 96, 96, 16, 16, 4, 2, 1, True, 0.2),
    "conv3_s2k3": (3, 96, 96, 32, 32, 3, 2, 1, False, 0.2),
    "conv2_s2k3_co64": (2, 64, 64, 64, 64, 3, 2, 1, False, 0.2),
    "deconv3_t1k3_co16": (2, 64, 16, 64, 64, 3, 1, 1, True, 0.2),
    "plain_k3_co96": (2, 32, 96, 40, 24, 3, 1, 1, False, 0.0),
    "adj_t2k3": (2, 64, 64, 16, 16, 3, 2, 1, True, 0.0),
    "adj_s2k4": (2, 96, 96, 32, 32, 4, 2, 1, False, 0.0),
    "ragged_s2k3": (2, 32, 48, 37, 21, 3, 2, 1, False, 0.2),
    "ragged_t2k4": (1, 16, 40, 19, 9, 4, 2, 1, True, 0.0),
    "one_stage_k3": (1, 16, 32, 32, 16, 3, 1, 1, False, 0.0),
    # 16 output channels: conv16.hip (weights resident in registers, whole K of a tile in LDS); > 256 tiles = several per workgroup
    "c16_plain_ci32_ragged": (2, 32, 16, 37, 21, 3, 1, 1, False, 0.2),
    "c16_t_ci16": (1, 16, 16, 20, 40, 3, 1, 1, True, 0.0),
    "c16_plain_ci48_many_tiles": (70, 48, 16, 24, 24, 3, 1, 1, False, 0.2),
    "c16_t_ci64_many_tiles": (40, 64, 16, 30, 34, 3, 1, 1, True, 0.2),
}


@pytest.mark.parametrize("tag", list(CASES))
def test_glue_layer_on_split_input_vs_torch(vpx, tag):
    N, Ci, Co, H, W, k, s, p, tr, slope = CASES[tag]
    x = seeded_rand((N, Ci, H, W), name_seed(f"convq.{tag}.x")).cuda() - 0.3
    wshape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
    w = (seeded_randn(wshape, name_seed(f"convq.{tag}.w"), 1.0 / np.sqrt(Ci * k * k))).cuda()
    b = seeded_randn((Co,), name_seed(f"convq.{tag}.b"), 0.1).cuda()
    assert vpx.ops.conv2d_ex_takes_split(N, H, W, Ci, Co, k, k, s, p, tr)
    xbuf, _ = vpx.ops.split_convert(x)
    want_split = Co % 8 == 0
    y, ybuf, shp = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3", out_split=want_split)
    ref = F.conv_transpose2d(x, w, b, stride=s, padding=p) if tr else F.conv2d(x, w, b, stride=s, padding=p)
    if slope:
        ref = F.leaky_relu(ref, slope)
    assert tuple(y.shape) == tuple(ref.shape) == shp
    assert _relmax(y, ref) < 2e-5, tag
    if want_split:   # the split copy decodes to the fp32 output (hi + lo, to bf16x2 precision)
        sb, _ = vpx.ops.split_convert(y)
        assert torch.equal(sb.view(torch.int32), ybuf.view(torch.int32))
    # second call: packed weights re-used from the layer's workspace
    y2, _, _ = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
    assert torch.equal(y, y2)
    # an in-place update of the weight (an optimizer step) invalidates the pack: the third call follows the new values
    w.mul_(0.5)
    y3, _, _ = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
    ref3 = F.conv_transpose2d(x, w, b, stride=s, padding=p) if tr else F.conv2d(x, w, b, stride=s, padding=p)
    if slope:
        ref3 = F.leaky_relu(ref3, slope)
    assert _relmax(y3, ref3) < 2e-5, tag


@pytest.mark.parametrize("tag", ["deconv2_t2k4", "conv3_s2k3", "conv2_s2k3_co64", "deconv3_t1k3_co16", "ragged_s2k3", "ragged_t2k4"])
def test_half_tile_and_full_tile_agree_bit_for_bit(vpx, tag):
    """convq on 16x16-pixel tiles (two workgroups per CU, ring of two weight chunks: the default) and on 32x16 tiles (VPX_OPT_EXPERIMENT
    bit 4): the same products in the same order per output element."""
    N, Ci, Co, H, W, k, s, p, tr, slope = CASES[tag]
    x = seeded_rand((N, Ci, H, W), name_seed(f"convq.{tag}.x")).cuda() - 0.3
    wshape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
    w = (seeded_randn(wshape, name_seed(f"convq.{tag}.w"), 1.0 / np.sqrt(Ci * k * k))).cuda()
    b = seeded_randn((Co,), name_seed(f"convq.{tag}.b"), 0.1).cuda()
    xbuf, _ = vpx.ops.split_convert(x)
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 0)
    try:
        y4, _, _ = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 16)
        y8, _, _ = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
    finally:
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
    assert torch.equal(y4, y8)


@pytest.mark.parametrize("tag", ["deconv3_t1k3_co16", "c16_plain_ci32_ragged", "c16_plain_ci48_many_tiles", "c16_t_ci64_many_tiles"])
def test_sixteen_column_kernel_vs_first_generation(vpx, tag):
    """conv16.hip against the first-generation kernel on the same split input (VPX_OPT_EXPERIMENT bit 28): the same bf16x3 products,
    summed in a different order — equal to fp32 rounding of the sums."""
    N, Ci, Co, H, W, k, s, p, tr, slope = CASES[tag]
    x = seeded_rand((N, Ci, H, W), name_seed(f"convq.{tag}.x")).cuda() - 0.3
    wshape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
    w = (seeded_randn(wshape, name_seed(f"convq.{tag}.w"), 1.0 / np.sqrt(Ci * k * k))).cuda()
    b = seeded_randn((Co,), name_seed(f"convq.{tag}.b"), 0.1).cuda()
    xbuf, _ = vpx.ops.split_convert(x)
    L = vpx._lib.lib()
    prev = L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 0)
    try:
        y16, _, _ = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, 1 << 28)
        y1, _, _ = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
    finally:
        L.vpx_set_option(vpx._lib.OPT_EXPERIMENT, prev)
    assert _relmax(y16, y1) < 2e-6, tag


@pytest.mark.parametrize("tag", ["deconv3_t1k3_co16", "c16_plain_ci32_ragged", "deconv1_t2k4_small", "deconv2_t2k4"])
def test_strided_sequence_source_through_the_c_abi(vpx, tag):
    """vpx_conv2d_ex_fwd_from_split on a source that is a [B][T] sequence with a gap between the samples (x_bstride, x_tstride, x_nT:
    image n at (n / x_nT) * x_bstride + (n % x_nT) * x_tstride) — what a recurrent block's output slab looks like to the glue. Called
    through the C ABI (the Python wrapper only hands over dense batches); equal to the dense call bit for bit, on both split-input
    kernels (conv16.hip, convq.hip)."""
    import ctypes
    from vp_suite_amd._lib import ConvDesc
    N, Ci, Co, H, W, k, s, p, tr, slope = CASES[tag]
    B, T = 2, 3
    N = B * T
    x = seeded_rand((N, Ci, H, W), name_seed(f"convq.seq.{tag}.x")).cuda() - 0.3
    wshape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
    w = (seeded_randn(wshape, name_seed(f"convq.{tag}.w"), 1.0 / np.sqrt(Ci * k * k))).cuda()
    b = seeded_randn((Co,), name_seed(f"convq.{tag}.b"), 0.1).cuda()
    xbuf, _ = vpx.ops.split_convert(x)
    dense, _, shp = vpx.ops.conv2d_ex_from_split(xbuf, (N, Ci, H, W), w, b, s, p, tr, slope, "bf16x3")
    img = H * W * Ci                                     # floats per image in the split format (same byte count as fp32)
    slab = torch.full((B, T + 1, img), float("nan"), device="cuda")   # one unused image slot per sample: the gap
    slab[:, :T] = xbuf.view(B, T, img)
    L = vpx._lib.lib()
    d = ConvDesc(N, H, W, Ci, Co, k, k, s, p, int(tr), float(slope), vpx._lib.PREC_BF16X3, 0, 0)
    ws_bytes = L.vpx_conv2d_ex_split_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(dense)
    pv = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.vpx_conv2d_ex_fwd_from_split(ctypes.byref(d), pv(slab), (T + 1) * img * 4, img * 4, T, pv(w), pv(b), pv(y), None, 0, pv(ws), ws_bytes,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.vpx_last_error().decode()
    assert torch.equal(y, dense)


def test_split_convert_entry_point_matches_the_torch_restatement(vpx):
    """vpx_split_convert (fp32 channels-last -> operand format) against the same two roundings written with torch ops."""
    x = seeded_randn((3, 40, 19, 23), name_seed("split_convert.x"), 2.0).cuda()
    a, shp = vpx.ops.split_convert(x)
    b, _ = vpx.ops.split_convert(x, native=False)
    assert shp == (3, 40, 19, 23) and torch.equal(a.view(torch.int32), b.view(torch.int32))
    with pytest.raises(ValueError):
        vpx.ops.split_convert(torch.zeros(1, 12, 4, 4, device="cuda"))


def test_convlstm_block_hands_its_output_over_in_operand_format(vpx):
    """VPX_FLAG_OUT_SPLIT: the second-generation cell writes its output sequence ONLY in the split-bf16 operand format ([B][T][HW][Ch]);
    decoded, it equals the fp32 sequence of a normal call bit for bit (same kernel, same arithmetic), h_T / c_T come out in fp32."""
    Cin, Ch, H, W, B, T = 64, 96, 32, 32, 32, 3
    x = seeded_rand((B, T, Cin, H, W), name_seed("osplit.x")).cuda()
    Wt = seeded_randn((4 * Ch, Cin + Ch, 3, 3), name_seed("osplit.W"), 1.0 / np.sqrt((Cin + Ch) * 9.0)).cuda()
    b = seeded_randn((4 * Ch,), name_seed("osplit.b"), 0.1).cuda()
    peep = [seeded_randn((1, Ch, H, W), name_seed(f"osplit.p{i}"), 0.1).cuda() for i in range(3)]
    assert vpx.ops.convlstm_writes_split(B, T, Cin, Ch, H, W, 3, 0, "bf16x3")
    with torch.no_grad():
        out, hT, cT = vpx.ops.convlstm_seq(x, None, None, Wt, b, *peep, seq_len=T, in_channels=Cin, precision="bf16x3")
        sp, hT2, cT2 = vpx.ops.convlstm_seq(x, None, None, Wt, b, *peep, seq_len=T, in_channels=Cin, precision="bf16x3", out_split=True)
    assert isinstance(sp, vpx.ops.SplitActivation) and sp.shape == (B, T, Ch, H, W)
    assert torch.equal(hT2, hT) and torch.equal(cT2, cT)
    want, _ = vpx.ops.split_convert(out)          # [B, T, H, W, C/8, 2, 8] bf16 of the fp32 sequence
    assert torch.equal(sp.buf.view(torch.int32), want.view(torch.int32))
    # small grids (cell3_kernel, which keeps h_t in operand format for its own recurrence) hand the sequence out the same way — with
    # operand-format INPUT as well where the hoisted projection runs on the schedule-driven kernel; decoded: bit for bit the fp32 call
    xs, ps = x[:2, :, :, :16, :16].contiguous(), [p[..., :16, :16].contiguous() for p in peep]
    assert vpx.ops.convlstm_writes_split(2, T, Cin, Ch, 16, 16, 3, 0, "bf16x3")
    h0 = seeded_randn((2, Ch, 16, 16), name_seed("osplit.h0"), 0.3).cuda()
    c0 = seeded_randn((2, Ch, 16, 16), name_seed("osplit.c0"), 0.3).cuda()
    with torch.no_grad():
        out_s, hT_s, cT_s = vpx.ops.convlstm_seq(xs, h0, c0, Wt, b, *ps, seq_len=T, in_channels=Cin, precision="bf16x3")
        sp_s, hT_s2, cT_s2 = vpx.ops.convlstm_seq(xs, h0, c0, Wt, b, *ps, seq_len=T, in_channels=Cin, precision="bf16x3", out_split=True)
        assert isinstance(sp_s, vpx.ops.SplitActivation) and sp_s.shape == (2, T, Ch, 16, 16)
        assert torch.equal(hT_s2, hT_s) and torch.equal(cT_s2, cT_s)
        want_s, _ = vpx.ops.split_convert(out_s)
        assert torch.equal(sp_s.buf.view(torch.int32), want_s.view(torch.int32))
        if vpx.ops.convlstm_takes_split(2, T, Cin, Ch, 16, 16, 3, 0, "bf16x3"):
            xbuf, _ = vpx.ops.split_convert(xs.reshape(2 * T, Cin, 16, 16))
            out_x, hT_x, cT_x = vpx.ops.convlstm_seq(vpx.ops.SplitActivation(xbuf, (2, T, Cin, 16, 16)), h0, c0, Wt, b, *ps, seq_len=T,
                                                     in_channels=Cin, precision="bf16x3")
            assert torch.equal(out_x, out_s) and torch.equal(hT_x, hT_s) and torch.equal(cT_x, cT_s)
    # exact-fp32 operands have no operand format: no split output, and asking for it is an error
    assert not vpx.ops.convlstm_writes_split(2, T, Cin, Ch, 16, 16, 3, 0, "f32")
    with pytest.raises(Exception):
        vpx.ops.convlstm_seq(xs, None, None, Wt, b, *ps, seq_len=T, in_channels=Cin, precision="f32", out_split=True)
