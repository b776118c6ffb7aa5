"""Random-shape sweep of the workspace contract (CPU, dry run; optionally on the ASan host build: VPX_LIB=/tmp/asan/libvpx_asan.so with
LD_PRELOAD of the ASan runtime, see tools/asan_host.sh). Every entry point that takes a workspace gets random descriptors — odd map
sizes, channel counts that are multiples of nothing, every kernel size / layout / operand mode / option bit — with a workspace of exactly
the queried size. Any return code other than OK / ARG / UNSUPPORTED is a finding. usage: python tools/fuzz_contract.py [N] [seed]"""
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vp_suite_amd import _lib                                   # noqa: E402
from vp_suite_amd._lib import ConvDesc, ConvLSTMDesc, STLSTMDesc, STLSTMShadows, TrajGRUDesc   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = _lib.lib()
L.vpx_set_option(_lib.OPT_DRY_RUN, 1)
WS = [0x7F0000000000, 0x7F0000000040, 0x7F0000000010]
fk = lambda i: ctypes.c_void_p(0x100000000000 + i * (1 << 36))   # noqa: E731
OKS = (0, -1, -4)
findings = []
chs = [1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 24, 26, 31, 32, 33, 40, 48, 64, 72, 96, 100, 128, 160, 192, 256, 288]
dims = [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 20, 24, 31, 32, 33, 47, 48, 64, 67, 83, 96, 128]
EXP = [0, 1, 4, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 64 | 128 | 256 | 512, 1024 | 2048]


def note(what, rc):
    if rc not in OKS:
        findings.append((what, rc, L.vpx_last_error().decode()))
        print("FINDING", what, rc, L.vpx_last_error().decode(), flush=True)


for it in range(N):
    L.vpx_set_deterministic(rng.random() < 0.3)
    L.vpx_set_option(_lib.OPT_EXPERIMENT, rng.choice(EXP))
    L.vpx_set_option(_lib.OPT_CELL2, rng.choice([1, 1, 0, 2]))
    L.vpx_set_option(_lib.OPT_CELL3, rng.choice([1, 1, 0]))
    L.vpx_set_option(_lib.OPT_MFMA_SHAPE, rng.choice([1, 1, 0]))
    ws = ctypes.c_void_p(rng.choice(WS))
    kind = rng.randrange(6)
    B, H, W = rng.choice([1, 2, 3, 4, 6, 8, 16, 32, 128, 256]), rng.choice(dims), rng.choice(dims)
    if B * H * W > 1 << 21:
        continue
    prec = rng.randrange(3)
    if kind == 0:      # plain convolutions
        Ci, Co, k = rng.choice(chs), rng.choice(chs), rng.choice([1, 3, 5, 7])
        nb = L.vpx_conv2d_workspace_bytes(Ci, Co, k, k)
        note(f"fwd_ex {(B, H, W, Ci, Co, k, prec)}", L.vpx_conv2d_nhwc_fwd_ex(fk(1), fk(2), fk(3), fk(4), B, H, W, Ci, Co, k, k, prec, rng.randrange(2), rng.choice([0.0, 0.2]), ws, nb, None))
        note(f"fwd {(B, H, W, Ci, Co, k, prec)}", L.vpx_conv2d_nhwc_fwd(fk(1), fk(2), fk(3), fk(4), B, H, W, Ci, Co, k, k, prec, ws, nb, None))
        nbw = L.vpx_conv2d_bwd_workspace_bytes(B, H, W, Ci, Co, k, k)
        note(f"bwd {(B, H, W, Ci, Co, k, prec)}", L.vpx_conv2d_nhwc_bwd(fk(1), fk(2), fk(3), fk(4), fk(5), fk(6), B, H, W, Ci, Co, k, k, prec, ws, nbw, None))
    elif kind == 1:    # ConvLSTM
        Cin, Ch, k, T = rng.choice(chs), rng.choice(chs), rng.choice([1, 3, 5, 7]), rng.choice([1, 2, 3, 10, 20])
        if B * T * H * W * max(Cin, 4 * Ch) > 1 << 31:
            continue
        save = rng.randrange(2)
        d = ConvLSTMDesc(B, T, Cin, Ch, H, W, k, rng.choice([k, k, 3]), rng.randrange(2), rng.randrange(2), prec, _lib.FLAG_SAVE_FOR_BWD if save else 0)
        nb = L.vpx_convlstm_workspace_bytes(ctypes.byref(d))
        if nb == 0:
            continue
        rs = L.vpx_convlstm_reserve_bytes(ctypes.byref(d))
        x, h0 = rng.choice([(fk(1), fk(2)), (None, fk(2)), (fk(1), None)])
        peep = rng.randrange(2)
        pp = [fk(6), fk(7), fk(8)] if peep else [None, None, None]
        note(f"convlstm fwd {(B, T, Cin, Ch, H, W, k, d.kw, d.gate_order, d.layout, prec, save)}",
             L.vpx_convlstm_seq_fwd(ctypes.byref(d), x, h0, None if h0 is None else fk(3), fk(4), fk(5), *pp, fk(9), fk(10), fk(11), fk(12), rs, ws, nb, None))
        if save:
            dpp = [fk(21), fk(22), fk(23)] if peep else [None, None, None]
            note(f"convlstm bwd {(B, T, Cin, Ch, H, W, k, d.kw, d.gate_order, d.layout, prec)}",
                 L.vpx_convlstm_seq_bwd(ctypes.byref(d), x, h0, None if h0 is None else fk(3), fk(4), *pp, fk(9), fk(12), rs, fk(13), fk(14), fk(15),
                                        None if x is None else fk(16), None if h0 is None else fk(17), None if h0 is None else fk(18), fk(19), fk(20), *dpp, ws, nb, None))
    elif kind == 2:    # ST-LSTM
        Cin, Ch, k = rng.choice(chs), rng.choice(chs), rng.choice([1, 3, 5, 7])
        ln, layout, save, packed = rng.randrange(2), rng.randrange(2), rng.randrange(2), rng.randrange(2)
        d = STLSTMDesc(B, Cin, Ch, H, W, k, ln, layout, prec, (_lib.FLAG_SAVE_FOR_BWD if save else 0) | (_lib.FLAG_WEIGHTS_PACKED if packed else 0))
        nb = L.vpx_stlstm_workspace_bytes(ctypes.byref(d))
        if nb == 0:
            continue
        rs = L.vpx_stlstm_reserve_bytes(ctypes.byref(d))
        lnarr = (ctypes.c_void_p * 8)(*[0x200000000000 + i * (1 << 30) for i in range(8)]) if ln else None
        sh = STLSTMShadows((ctypes.c_void_p * 5)(*[rng.choice([None, 0x400000000000 + i * (1 << 34)]) for i in range(5)]),
                           (ctypes.c_void_p * 3)(*[rng.choice([None, 0x480000000000 + i * (1 << 34)]) for i in range(3)]), None)
        tag = f"stlstm {(B, Cin, Ch, H, W, k, ln, layout, prec, save, packed)}"
        note(tag + " fwd", L.vpx_stlstm_step_fwd_ex(ctypes.byref(d), *[fk(i) for i in range(1, 10)], lnarr, *[fk(i) for i in range(10, 15)], fk(15), rs, ws, nb, None, ctypes.byref(sh)))
        if save:
            dln = (ctypes.c_void_p * 8)(*[0x300000000000 + i * (1 << 30) for i in range(8)]) if ln else None
            defer = (not ln) and layout == 0 and L.vpx_stlstm_defers_wgrad(ctypes.byref(d)) and rng.randrange(2)
            sh.dg8_out = 0x500000000000 if defer else None
            dws = [None] * 5 if defer else [fk(i) for i in range(25, 30)]
            note(tag + f" bwd defer={bool(defer)}", L.vpx_stlstm_step_bwd_ex(ctypes.byref(d), *[fk(i) for i in range(1, 12)], lnarr, fk(15), rs, *[fk(i) for i in range(16, 21)],
                                                                             rng.choice([None, fk(21)]), rng.choice([None, fk(22)]), fk(23), fk(24), *dws, dln, ws, nb, None, ctypes.byref(sh)))
            if defer:
                T = rng.choice([1, 5, 39])
                dT = STLSTMDesc(T * B, Cin, Ch, H, W, k, 0, 0, prec, _lib.FLAG_SAVE_FOR_BWD)
                nbb = L.vpx_stlstm_wgrad_batch_workspace_bytes(ctypes.byref(dT))
                note(tag + f" wgrad_batch T={T}", L.vpx_stlstm_wgrad_batch(ctypes.byref(dT), fk(1), (ctypes.c_void_p * 5)(*[0x400000000000 + i * (1 << 34) for i in range(5)]),
                                                                           *[fk(i) for i in range(25, 30)], ws, nbb, None))
    elif kind == 3:    # stage glue
        Ci, Co, k, s, tr = rng.choice(chs), rng.choice(chs), rng.choice([1, 2, 3, 4, 5, 7]), rng.choice([1, 2]), rng.randrange(2)
        p = rng.choice([0, 1, 2, 3])
        d = ConvDesc(B, H, W, Ci, Co, k, rng.choice([k, k, 3]), s, p, tr, rng.choice([0.0, 0.2]), prec, rng.randrange(2) if tr else 0, rng.randrange(2) if tr else 0)
        ho, wo = ctypes.c_int(0), ctypes.c_int(0)
        if L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) != 0 or B * ho.value * wo.value * Co > 1 << 29:
            continue
        tag = f"glue {(B, H, W, Ci, Co, d.kh, d.kw, s, p, tr, d.out_pad_h, d.out_pad_w, prec)}"
        nb = L.vpx_conv2d_ex_workspace_bytes(ctypes.byref(d))
        note(tag + " fwd", L.vpx_conv2d_ex_fwd(ctypes.byref(d), fk(1), fk(2), fk(3), fk(4), ws, nb, None))
        if Co % 8 == 0:
            note(tag + " fwd_split", L.vpx_conv2d_ex_fwd_split(ctypes.byref(d), fk(1), fk(2), fk(3), None, fk(5), ws, nb, None))
        if L.vpx_conv2d_ex_takes_split(ctypes.byref(d)):
            nbs = L.vpx_conv2d_ex_split_workspace_bytes(ctypes.byref(d))
            note(tag + " from_split", L.vpx_conv2d_ex_fwd_from_split(ctypes.byref(d), fk(1), 0, 0, 1, fk(2), fk(3), fk(4), fk(5) if Co % 8 == 0 else None, rng.randrange(2), ws, nbs, None))
        nbb = L.vpx_conv2d_ex_bwd_workspace_bytes(ctypes.byref(d))
        if nbb:
            note(tag + " bwd", L.vpx_conv2d_ex_bwd(ctypes.byref(d), fk(1), fk(2), fk(4), fk(6), rng.choice([None, fk(7)]), fk(8), fk(9), ws, nbb, None))
    elif kind == 4:    # decoupling tail
        Ch = rng.choice(chs)
        nb = L.vpx_decouple_workspace_bytes(B, Ch, H, W)
        note(f"decouple fwd {(B, Ch, H, W, prec)}", L.vpx_decouple_fwd(fk(1), fk(2), fk(3), fk(4), B, Ch, H, W, prec, ws, nb, None))
        adj = rng.randrange(2)
        dc = fk(1); dm = ctypes.c_void_p(dc.value + B * H * W * Ch * 4) if adj else fk(2)
        gc = fk(5); gm = ctypes.c_void_p(gc.value + B * H * W * Ch * 4) if adj else fk(6)
        note(f"decouple bwd {(B, Ch, H, W, prec, adj)}", L.vpx_decouple_bwd(dc, dm, fk(3), fk(4), gc, gm, fk(7), B, Ch, H, W, prec, ws, nb, None))
    else:              # TrajGRU
        Cin, C, nl, T, k = rng.choice(chs), rng.choice([4, 8, 12, 16, 24, 64, 96]), rng.choice([1, 3, 5, 13]), rng.choice([1, 3, 10]), rng.choice([1, 3, 5])
        if B * T * H * W * nl * C > 1 << 29:
            continue
        save = rng.randrange(2)
        d = TrajGRUDesc(B, T, Cin, C, H, W, nl, k, prec, _lib.FLAG_SAVE_FOR_BWD if save else 0, 0.2)
        nb, rs = L.vpx_trajgru_workspace_bytes(ctypes.byref(d)), L.vpx_trajgru_reserve_bytes(ctypes.byref(d))
        if nb == 0:
            continue
        params = (ctypes.c_void_p * 10)(*[0x200000000000 + i * (1 << 32) for i in range(10)])
        dparams = (ctypes.c_void_p * 10)(*[0x300000000000 + i * (1 << 32) for i in range(10)])
        x, h0 = rng.choice([(fk(1), fk(2)), (None, fk(2)), (fk(1), None)])
        tag = f"trajgru {(B, T, Cin, C, H, W, nl, k, prec, save)}"
        note(tag + " fwd", L.vpx_trajgru_seq_fwd(ctypes.byref(d), x, h0, params, fk(3), fk(4), rs, ws, nb, None))
        if save:
            note(tag + " bwd", L.vpx_trajgru_seq_bwd(ctypes.byref(d), x, h0, params, fk(3), fk(4), rs, rng.choice([None, fk(5)]), rng.choice([None, fk(6)]),
                                                    None if x is None else rng.choice([None, fk(7)]), None if h0 is None else fk(8), dparams, ws, nb, None))
print(f"{N} iterations, {len(findings)} findings")
sys.exit(1 if findings else 0)
