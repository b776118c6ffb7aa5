"""CPU-side tests: host logic of the product package, the C-ABI library's exports, and loud failure without a GPU.
No compute calls into the HIP library here (there is no GPU in the build container)."""
import ctypes
import json
import os
import pickle
import re

import pytest
import torch

import golden_cases as gc
from golden_util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(vpx):
    """Every function declared in include/vpx.h is exported by libvpx_hip.so (and bound in _lib.py)."""
    hdr = open(os.path.join(ROOT, "include", "vpx.h")).read()
    declared = sorted(set(re.findall(r"\b(vpx_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = vpx._lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vpx.h but not exported"
    assert sorted(vpx._lib.EXPORTED_SYMBOLS) == declared
    assert L.vpx_version() == 100


def test_workspace_queries_run_without_gpu(vpx):
    L = vpx._lib.lib()
    d = vpx._lib.ConvLSTMDesc(4, 10, 64, 64, 64, 64, 3, 3, 0, 0, 0, 1)
    assert L.vpx_convlstm_workspace_bytes(ctypes.byref(d)) > 0
    # reserve = gates (4Ch) + cell states (Ch) per step, fp32
    assert L.vpx_convlstm_reserve_bytes(ctypes.byref(d)) >= 4 * 10 * 64 * 64 * 64 * 5 * 4
    bad = vpx._lib.ConvLSTMDesc(4, 10, 64, 64, 64, 64, 2, 2, 0, 0, 0, 0)
    assert L.vpx_convlstm_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"odd" in L.vpx_last_error()


def test_workspace_queries_follow_the_kernel_selection(vpx):
    """The size queries (pure host logic, no GPU) reflect which kernel a descriptor gets: the second-generation cell keeps split
    copies of its operands in the reserve and takes split input; the small-grid kernel needs the hoisted input projection of
    all steps in the workspace — it takes split input where that projection runs on the schedule-driven kernel (input channels
    in 16s) and fp32 otherwise, and hands its output sequence out in split format either way (it keeps h_t that way for its own
    recurrence); nothing of that with SAVE_FOR_BWD; the backward of either carries dG of all steps in split format too."""
    L = vpx._lib.lib()
    BF16X3, SAVE = vpx._lib.PREC_BF16X3, vpx._lib.FLAG_SAVE_FOR_BWD
    def desc(B, T, Cin, Ch, H, W, flags):
        return vpx._lib.ConvLSTMDesc(B, T, Cin, Ch, H, W, 3, 3, 0, vpx._lib.LAYOUT_NHWC, BF16X3, flags)
    big, small = desc(128, 10, 64, 64, 64, 64, 0), desc(4, 10, 64, 96, 32, 32, 0)
    assert L.vpx_convlstm_takes_split_input(ctypes.byref(big)) == 1       # cell2: 2048 workgroups
    assert L.vpx_convlstm_takes_split_input(ctypes.byref(small)) == 1     # cell3, 64 input channels: hoisted projection on convq
    assert L.vpx_convlstm_takes_split_input(ctypes.byref(desc(4, 10, 8, 96, 32, 32, 0))) == 0    # 8 input channels: first-generation projection, fp32
    assert L.vpx_convlstm_writes_split_output(ctypes.byref(big)) == 1 and L.vpx_convlstm_writes_split_output(ctypes.byref(small)) == 1
    for dd in (desc(128, 10, 64, 64, 64, 64, SAVE), desc(4, 10, 64, 96, 32, 32, SAVE)):       # training: fp32 in, fp32 out
        assert L.vpx_convlstm_takes_split_input(ctypes.byref(dd)) == 0 and L.vpx_convlstm_writes_split_output(ctypes.byref(dd)) == 0
    n_state, n_x = 128 * 64 * 64 * 64, 128 * 10 * 64 * 64 * 64
    rs = L.vpx_convlstm_reserve_bytes(ctypes.byref(desc(128, 10, 64, 64, 64, 64, SAVE)))
    assert rs >= 10 * n_state * 5 * 4 + n_x * 4 + n_state * 4 + 10 * n_state * 4   # gates + c, then x, h0, h_1..h_T split
    ws_small = L.vpx_convlstm_workspace_bytes(ctypes.byref(small))
    assert ws_small >= 4 * (4 * 10 * 32 * 32 * 96) * 4                    # W_x * x of all steps: [B,T,HW,4Ch] fp32
    ws_fwd, ws_bwd = L.vpx_convlstm_workspace_bytes(ctypes.byref(big)), L.vpx_convlstm_workspace_bytes(ctypes.byref(desc(128, 10, 64, 64, 64, 64, SAVE)))
    assert ws_bwd >= ws_fwd and ws_bwd >= 2 * 10 * n_state * 16          # dG of all steps, fp32 and split


def test_conv_desc_layout_and_shape_calculus(vpx):
    """The ctypes mirror of vpx_conv_desc matches the C struct (every field influences the C side's answer), and the
    output-shape rules are those of nn.Conv2d / nn.ConvTranspose2d (ef_blocks.py:15-49 builds exactly these layers)."""
    import torch
    L = vpx._lib.lib()
    ho, wo = ctypes.c_int(), ctypes.c_int()
    for tr, k, s, p, op, H, W in ((0, 3, 2, 1, 0, 64, 64), (0, 3, 2, 1, 0, 17, 23), (1, 4, 2, 1, 0, 16, 16), (1, 3, 2, 1, 1, 9, 12),
                                  (1, 3, 1, 1, 0, 8, 8), (0, 1, 1, 0, 0, 5, 7)):
        d = vpx._lib.ConvDesc(2, H, W, 6, 5, k, k, s, p, tr, 0.2, 0, op, op)
        assert L.vpx_conv2d_ex_out_shape(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == 0
        layer = (torch.nn.ConvTranspose2d(6, 5, k, s, p, output_padding=op) if tr else torch.nn.Conv2d(6, 5, k, s, p))
        ref = layer(torch.zeros(1, 6, H, W)).shape
        assert (ho.value, wo.value) == (ref[2], ref[3]), (tr, k, s, p, op)
        assert L.vpx_conv2d_ex_workspace_bytes(ctypes.byref(d)) > 0
        assert L.vpx_conv2d_ex_bwd_workspace_bytes(ctypes.byref(d)) > 0
    bad = vpx._lib.ConvDesc(2, 8, 8, 6, 5, 3, 3, 2, 1, 1, 0.0, 0, 2, 0)   # output padding must stay below the stride
    assert L.vpx_conv2d_ex_out_shape(ctypes.byref(bad), ctypes.byref(ho), ctypes.byref(wo)) != 0
    assert b"output padding" in L.vpx_last_error()
    bad = vpx._lib.ConvDesc(2, 8, 8, 6, 5, 3, 3, 3, 1, 0, 0.0, 0, 0, 0)   # stride 3 is not implemented
    assert L.vpx_conv2d_ex_workspace_bytes(ctypes.byref(bad)) == 0


def test_no_cpu_fallback(vpx):
    """The product path must fail loudly on CPU tensors instead of silently computing on the host."""
    from vp_suite_amd.model_blocks import ConvLSTM
    blk = ConvLSTM("cpu", 3, 8, 12, 10, 3, 1, 1)
    with pytest.raises(vpx.VpxError, match="no CPU fallback"):
        blk(torch.rand(2, 4, 3, 12, 10), None, 4)


def test_unsupported_glue_fails_when_the_model_is_built_and_cpu_tensors_are_named_as_such(vpx):
    """no stock fallback exists, so (i) a stage convolution outside vpx_conv2d_ex is refused at CONSTRUCTION, (ii) a CPU tensor
    raises an error that says 'GPU', not 'unsupported layer', (iii) a layer without a library backward fails in the forward of a call
    that needs gradients."""
    from collections import OrderedDict
    from vp_suite_amd.models import MODEL_CLASSES, ef_conv_lstm
    with pytest.raises(vpx.VpxError, match="unsupported layer configuration"):
        ef_conv_lstm._stage(OrderedDict({"conv1_leaky_1": [3, 8, 3, 3, 1]}))          # stride 3
    with pytest.raises(vpx.VpxError, match="unsupported layer configuration"):
        MODEL_CLASSES["convlstm-shi"]("cpu", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], enc_conv_k=[3, 9, 3],
                                      enc_conv_p=[1, 4, 1])                             # 9x9 kernel
    st = ef_conv_lstm._stage(OrderedDict({"conv1_leaky_1": [3, 8, 3, 2, 1]}))
    with pytest.raises(vpx.VpxError, match="must live on the GPU"):
        ef_conv_lstm._run_stage(st, torch.rand(2, 3, 8, 8), "f32")

    with pytest.raises(vpx.VpxError, match="no backward"):                             # 1x1 kernel with stride 2: forward-only in the library
        vpx.ops.conv2d_ex(torch.rand(1, 4, 8, 8), torch.rand(4, 4, 1, 1, requires_grad=True), None, 2, 0, False, 0.0, "f32")
    with torch.no_grad(), pytest.raises(vpx.VpxError, match="must live on the GPU"):   # inference: the layer itself is fine, the device is not
        vpx.ops.conv2d_ex(torch.rand(1, 4, 8, 8), torch.rand(4, 4, 1, 1, requires_grad=True), None, 2, 0, False, 0.0, "f32")


def test_predrnn_training_slabs_have_byte_limits(vpx, monkeypatch):
    """The two round-5 slab schemes of PredRNN_V2 fall back to the per-step paths above their byte limits (no unbounded allocation)."""
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["predrnn-pp"]("cpu", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0])
    monkeypatch.setattr(vpx.ops.STWeightBank, "available", staticmethod(lambda *a: True))
    built = {}
    import vp_suite_amd.models.predrnn_v2 as pv
    monkeypatch.setattr(pv, "_CellBanks", lambda model, geo, T: built.setdefault("geo", (geo, T)) and "banks")
    assert m._weight_banks(4, 19) == "banks" and built["geo"][1] == 19
    px = 4 * 16 * 16
    need = sum(4 * px * (19 * 8 * 128 + 3 * 20 * 128) for _ in range(3)) + 4 * px * 19 * 16
    m.BANK_BYTES_LIMIT = need
    assert m._weight_banks(4, 19) == "banks"
    m.BANK_BYTES_LIMIT = need - 1
    assert m._weight_banks(4, 19) is None
    m.defer_weight_gradients = False
    m.BANK_BYTES_LIMIT = 1 << 60
    assert m._weight_banks(4, 19) is None


def test_workspace_cache_never_keeps_an_entry_larger_than_its_budget(vpx):
    C = vpx.ops._WorkspaceCache(budget_bytes=1000)
    w = torch.zeros(4)
    C.put(("small",), w, w.data_ptr(), torch.empty(400, dtype=torch.uint8), 0)
    C.put(("huge",), w, w.data_ptr(), torch.empty(4000, dtype=torch.uint8), 0)
    assert list(C.ents) == [("small",)] and C.bytes == 400


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vp-suite_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M)
                assert "/root/reference" not in src


@pytest.mark.parametrize("tag,kw", [("tiny", gc.EF_TINY_KW), ("tiny3", gc.EF_TINY3_KW),
                                    ("full_c1", dict(img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0]))])
def test_ef_state_dict_contract(vpx, tag, kw):
    """Parameter names, shapes and ORDER equal the reference's (fixture stores its state_dict table)."""
    from vp_suite_amd.models import MODEL_CLASSES
    g = load_golden(f"ef_{tag}")
    shapes = json.loads(str(g["sd_shapes"]))
    m = MODEL_CLASSES["convlstm-shi"]("cpu", **kw)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(list(sd[k].shape) == shapes[k] for k in shapes)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    # peepholes are real parameters (CPU behaviour of the reference), zero-initialised
    assert "encoder.rnn1.Wci" in dict(m.named_parameters()) and float(m.encoder.rnn1.Wci.abs().sum()) == 0.0


def test_ef_config_and_kwarg_rules(vpx):
    from vp_suite_amd.models import MODEL_CLASSES
    EF = MODEL_CLASSES["convlstm-shi"]
    m = EF("cpu", img_shape=(3, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0])
    cfg = m.config
    assert cfg["img_c"] == 3 and cfg["NAME"] == "EF-ConvLSTM (Shi et al.)" and cfg["enc_rnn_state_h"] == [64, 32, 16]
    assert cfg["dec_rnn_state_h"] == [16, 32, 64]
    assert not any(isinstance(v, (torch.Tensor, torch.nn.Module)) for v in cfg.values())
    with pytest.raises(ValueError, match="missing required parameter"):
        EF("cpu", img_shape=(3, 64, 64), action_size=0)
    with pytest.raises(ValueError, match="mismatching types"):
        EF("cpu", img_shape=(3, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], num_layers="3")
    with pytest.raises(AttributeError, match="doesn't match"):
        EF("cpu", img_shape=(3, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], enc_c=[1, 2, 3])
    with pytest.raises(AttributeError, match="wrong output size"):
        EF("cpu", img_shape=(3, 50, 50), action_size=0, tensor_value_range=[0.0, 1.0])
    # 128x128 (BASELINE config C4) is accepted by the shape calculus
    m128 = EF("cpu", img_shape=(3, 128, 128), action_size=0, tensor_value_range=[0.0, 1.0])
    assert m128.enc_rnn_state_h == [128, 64, 32]
    assert sum(p.numel() for p in m128.parameters()) == 12764003


def test_model_is_picklable_whole(vpx):
    """The reference checkpoints by pickling the whole module (vpsuite.py:394)."""
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["convlstm-shi"]("cpu", **gc.EF_TINY_KW)
    m2 = pickle.loads(pickle.dumps(m))
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys())


def test_unpack_data_and_mse(vpx):
    from vp_suite_amd.measure import PredictionLossProvider, mse_measure
    from vp_suite_amd.models import MODEL_CLASSES
    m = MODEL_CLASSES["convlstm-shi"]("cpu", **gc.EF_TINY_KW)
    frames = torch.rand(2, 6, 1, 16, 16)
    data = {"frames": frames, "actions": torch.zeros(2, 5, 0)}
    cfg = {"device": "cpu", "context_frames": 3, "pred_frames": 2}
    x, y, _ = m.unpack_data(data, cfg)
    assert x.shape == (2, 3, 1, 16, 16) and torch.equal(y, frames[:, 3:5])
    xr, yr, _ = m.unpack_data(data, cfg, reverse=True)
    assert torch.equal(xr, torch.flip(frames, dims=[1])[:, :3])
    xc, yc, _ = m.unpack_data(data, cfg, complete=True)
    assert xc.shape[1] == 5 and torch.equal(yc, frames[:, 3:5])
    a, b = torch.rand(2, 3, 1, 4, 4), torch.rand(2, 3, 1, 4, 4)
    want = ((a - b) ** 2).sum(dim=(2, 3, 4)).mean()
    assert torch.allclose(mse_measure(a, b), want)
    disp, total = PredictionLossProvider({"device": "cpu", "losses_and_scales": {"mse": 2.0}}).get_losses(a, b)
    assert torch.allclose(total, 2 * want) and torch.allclose(disp["mse"], want)
    with pytest.raises(ValueError):
        mse_measure(a[0], b[0])


def test_channels_last_helpers(vpx):
    t = torch.rand(2, 3, 5, 4, 6)
    cl = vpx.ops.to_channels_last(t)
    assert cl.shape == t.shape and torch.equal(cl, t) and vpx.ops.is_channels_last(cl)
    assert cl.permute(0, 1, 3, 4, 2).is_contiguous()
    n = vpx.ops.new_channels_last((2, 3, 5, 4, 6), "cpu")
    assert n.shape == t.shape and vpx.ops.is_channels_last(n)
    fl, by = vpx.ops.convlstm_algorithmic_work(1, 1, 64, 64, 64, 64, 3, 3)
    assert abs(fl - 2.416e9) < 1e7  # BASELINE.md §4 headline cell: 2.416 GFLOP per sample-step
    assert abs(by - (5.243e6 + 4.326e6)) < 2e4


def test_peephole_modes_and_gpu_reference_checkpoints(vpx):
    """on GPU devices the reference's peepholes are plain tensors (conv_lstm_hzzone.py:30-32) — absent from
    state_dict, never trained. Such checkpoints must load (strict), and train_peepholes=False reproduces the behaviour."""
    from vp_suite_amd.models import MODEL_CLASSES
    EF = MODEL_CLASSES["convlstm-shi"]
    m = EF("cpu", **gc.EF_TINY_KW)
    full = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    gpu_ckpt = {k: v for k, v in full.items() if k.split(".")[-1] not in ("Wci", "Wcf", "Wco")}
    assert len(gpu_ckpt) == len(full) - 18
    m.load_state_dict(gpu_ckpt)  # strict: missing peephole keys are tolerated, values stay zero
    assert float(m.encoder.rnn2.Wcf.abs().sum()) == 0.0
    assert torch.equal(m.encoder.rnn2._conv.weight, full["encoder.rnn2._conv.weight"])
    with pytest.raises(RuntimeError):  # anything else missing is still an error
        m.load_state_dict({k: v for k, v in gpu_ckpt.items() if k != "encoder.rnn2._conv.bias"})

    fixed = EF("cpu", train_peepholes=False, **gc.EF_TINY_KW)
    assert list(fixed.state_dict().keys()) == list(gpu_ckpt.keys())
    assert not any(n.split(".")[-1] in ("Wci", "Wcf", "Wco") for n, _ in fixed.named_parameters())
    assert fixed.encoder.rnn1.Wci.requires_grad is False and fixed.config["train_peepholes"] is False
    fixed.load_state_dict(gpu_ckpt)
    fixed.load_state_dict(full)  # a CPU-built reference checkpoint: values taken, still not trainable
    assert torch.equal(fixed.forecaster.rnn3.Wco, full["forecaster.rnn3.Wco"])
    pickle.loads(pickle.dumps(fixed))


def test_flat_adam_relinks_broken_views(vpx):
    """FlatAdam._relink (CPU-checkable part of ADVICE r1): a None / foreign .grad is re-homed into the flat bucket with
    its values before the update reads the bucket."""
    from vp_suite_amd.train import FlatAdam
    lin = torch.nn.Linear(3, 2)
    opt = FlatAdam.from_module(lin, lr=1e-3)
    base = opt.flat_grad.untyped_storage().data_ptr()
    lin.weight.grad = None
    lin.bias.grad = torch.full((2,), 7.0)
    opt._relink()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in lin.parameters())
    assert float(opt.flat_grad[:6].abs().sum()) == 0.0 and torch.equal(opt.flat_grad[6:], torch.full((2,), 7.0))
    assert all(p.data.untyped_storage().data_ptr() == opt.flat_param.untyped_storage().data_ptr() for p in lin.parameters())


def test_bench_spawns_ranks_before_touching_the_gpu(vpx, monkeypatch):
    """`python bench.py --gpus N` (no launcher) must start N ranks under torch.distributed.run as a CHILD process."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--mode", "train"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--mode", "train"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    names = [s.name for s in bench.extras_for(1)]
    assert {"infer_b4", "infer_b32", "train_b32", "train_b128", "infer_b128_f32", "predrnn_infer_b128"} <= set(names)
    assert any(s.mode == "train" for s in bench.extras_for(8))


def test_workspace_cache_is_bounded_and_drops_stale_entries(vpx):
    """ops._WorkspaceCache (inference workspaces + weight packs): LRU under a byte budget; entries whose weight died, changed
    version or belong to an older kernel-option epoch are dropped on insert (up to 64 x 1 GiB could stay pinned)."""
    C = vpx.ops._WorkspaceCache(budget_bytes=1000)
    w1, w2 = torch.zeros(4), torch.zeros(4)
    a = torch.empty(400, dtype=torch.uint8)
    C.put(("k1",), w1, w1.data_ptr(), a, 0)
    assert C.get(("k1",), w1, w1.data_ptr(), 400, a.device) is a
    assert C.get(("k1",), w1, w1.data_ptr(), 300, a.device) is None and len(C) == 0      # size mismatch drops it
    C.put(("k1",), w1, w1.data_ptr(), a, 0)
    C.put(("k2",), w2, w2.data_ptr(), torch.empty(400, dtype=torch.uint8), 0)
    C.get(("k1",), w1, w1.data_ptr(), 400, a.device)                                        # k1 most recent
    C.put(("k3",), w2, w2.data_ptr(), torch.empty(400, dtype=torch.uint8), 0)               # budget: evicts k2 (LRU)
    assert list(C.ents) == [("k1",), ("k3",)] and C.bytes == 800
    w1.add_(1)                                                                              # version bump: k1 is stale
    assert C.get(("k1",), w1, w1.data_ptr(), 400, a.device) is None and C.bytes == 400
    C.put(("k4",), w1, w1.data_ptr(), torch.empty(100, dtype=torch.uint8), 1)               # new option epoch: k3 (epoch 0) goes
    assert list(C.ents) == [("k4",)] and C.bytes == 100
    del w1
    C.put(("k5",), w2, w2.data_ptr(), torch.empty(100, dtype=torch.uint8), 1)               # dead weakref: k4 goes
    assert list(C.ents) == [("k5",)]
    # LayerNorm parameters ([C,H,W]) hit the layout cache on the parameter object itself
    p = torch.nn.Parameter(torch.rand(3, 4, 5))
    assert vpx.ops._cached_channels_last(p) is vpx.ops._cached_channels_last(p)


def test_bench_final_line_fits_the_driver_capture(vpx):
    """The LAST stdout line of bench.py is what the driver parses, from a bounded capture (BENCH_r03.json: a 27 KB line gave
    "parsed": null; r02's 10.5 KB line parsed). With every extra present the compact line must stay below bench.LINE_BUDGET,
    round-trip through json, and still carry the contract's keys, `roofline` and `cpu_baseline`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ps = {"flops": 156.27e9 * 1200, "bytes": 321452715 * 1200, "ms": 388.0, "launches": 1200}
    head = bench.Spec("headline")
    full = {"metric": "predicted frames/sec (whole node), MovingMNIST 64x64 10->10", "value": 53211.12, "unit": "frames/s",
            "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 24.0512, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3", "data": "synthetic",
            "config": {"workload": head.workload(), "mode": "infer", "semantics": "VPModel.forward under no_grad" * 4,
                       "per_gpu_batch": 128, "global_batch": 128, "parallelism": "dp1", "ranks": 1, "backend": "single process",
                       "prewarm": {"seconds": 0.5, "steps": 23, "what": "x" * 100}},
            "roofline": bench.roofline(head, ps), "extras": [],
            "cpu_baseline": {"value": 173.69, "unit": "predicted frames/s", "cores": 16, "kind": "port",
                             "all_cores": {"cores": 128, "value": 18.96}, "thread_scan": {"128": 18.96, "32": 120.1, "16": 173.69},
                             "sample": "oracle/torch_ref.ef_convlstm_forward " + "y" * 400}}
    for es in bench.extras_for(1):
        rf = bench.roofline(es, ps)
        full["extras"].append({"name": es.name, "workload": es.workload(), "mode": es.mode, "semantics": "s" * 200,
                               "dtype": es.precision, "per_gpu_batch": es.batch, "global_batch": es.batch, "n_gpus": 1,
                               "steps": 128, "timed_region_s": 3.004, "ms_per_step": 23.4712, "value": 54538.11,
                               "unit": "cell steps x samples/s" if es.cell else "frames/s", "roofline": rf})
    full["extras"].append({"name": "broken", "error": "RuntimeError: " + "z" * 500})
    assert len(json.dumps(full)) > 20000          # the record that broke the r03 parse
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_BUDGET <= 8192, len(line)
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k
    assert got["value"] == full["value"] and got["config"]["workload"] == head.workload()
    rf = got["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    # traffic is a citation of a committed PMC pass, never presented as live: its source file is named next to it
    # (round 6: ... and only while the loaded library is the one that was measured — else null + "stale: <file> ...")
    src = rf["traffic_source"]
    assert (rf["traffic"] is None) == (src is None or src.startswith("stale:"))
    if src and not src.startswith("stale:"):
        assert os.path.exists(os.path.join(ROOT, src))
    cb = got["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and len(cb["sample"]) <= 360
    names = [e["name"] for e in got["extras"]]
    assert names == [e["name"] for e in full["extras"]] and got["extras_file"] == "bench_extras.json"
    e0 = got["extras"][0]
    assert set(e0) <= {"name", "dtype", "ms_per_step", "value", "unit", "frac", "hbm_frac", "traffic"} and e0["frac"] is not None
    # degenerate case: far too many extras still yields a parseable headline under the budget
    full["extras"] = full["extras"] * 8
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_BUDGET and json.loads(line)["roofline"]["frac"] == rf["frac"]


def test_ef_trajgru_registry_and_state_dict_contract(vpx):
    """"trajgru" is registered under the reference's key; parameter names / shapes / order equal the reference's."""
    from vp_suite_amd.models import MODEL_CLASSES
    from vp_suite_amd.model_blocks.traj_gru import Activation
    g = load_golden("ef_trajgru_tiny")
    shapes = json.loads(str(g["sd_shapes"]))
    m = MODEL_CLASSES["trajgru"]("cpu", **gc.EF_TRAJGRU_TINY_KW)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and all(list(sd[k].shape) == shapes[k] for k in shapes)
    assert m.encoder.rnn1.ret.weight.shape == (3 * 8, 3 * 8, 1, 1)   # L * C inputs, 3C outputs
    assert Activation("leaky", 0.3).negative_slope == 0.3 and float(Activation("relu")(torch.tensor(-1.0))) == 0.0
    with pytest.raises(NotImplementedError):
        Activation("swish")(torch.zeros(1))
    pickle.loads(pickle.dumps(m))


def test_tools_compile_and_referenced_tools_exist():
    """Every tools/*.py byte-compiles, every tools/*.sh passes `bash -n`, and every tools/ path named in DESIGN.md / README.md /
    INTEGRATION.md exists (profiles/README.md also names tools of earlier rounds that were pruned: it says so next to each)."""
    import glob, py_compile, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "tools", "*.py")):
        py_compile.compile(f, doraise=True)
    for f in glob.glob(os.path.join(root, "tools", "*.sh")):
        assert subprocess.run(["bash", "-n", f]).returncode == 0, f
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(root, doc)).read()
        for t in set(re.findall(r"tools/[A-Za-z0-9_]+\.(?:py|sh)", text)):
            assert os.path.exists(os.path.join(root, t)), (doc, t)


def test_bench_drops_traffic_measured_on_another_library(vpx, tmp_path, monkeypatch):
    """`roofline.traffic` cites a committed PMC summary only while the loaded library IS the one that was measured
    (the file's `lib_sha16` = sha256(libvpx_hip.so)[:16]); another build — or a pre-round-6 file without the field — gives null and says why."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench.loaded_lib_sha16()
    assert sha is not None and len(sha) == 16
    good, stale, old = tmp_path / "r06_pmc_x.json", tmp_path / "r06_pmc_y.json", tmp_path / "r05_pmc_z.json"
    good.write_text(json.dumps({"lib_sha16": sha, "hbm_traffic_bytes_per_launch": {"total": 123.4}}))
    stale.write_text(json.dumps({"lib_sha16": "0" * 16, "hbm_traffic_bytes_per_launch": {"total": 123.4}}))
    old.write_text(json.dumps({"hbm_traffic_bytes_per_launch": {"total": 123.4}}))
    assert bench.traffic_from_file(str(good), sha) == (123, "profiles/r06_pmc_x.json")
    for f in (stale, old):
        t, why = bench.traffic_from_file(str(f), sha)
        assert t is None and why.startswith("stale:")
    # the committed round-5 files predate the field: the headline's traffic must read null until this round's passes are committed
    t, why = bench.measured_traffic(bench.Spec("headline"))
    assert t is None or why.startswith("profiles/r06")


def test_bench_two_ranks_end_to_end_on_gloo():
    """`python bench.py --gpus 2` end to end without a GPU (--stub: CPU stand-in for the model, gloo): the launcher
    child, RANK / WORLD_SIZE plumbing, EXACTLY one JSON line (rank 0 only), n_gpus / ranks / backend fields, whole-job value, and the
    MAX-over-ranks timing — rank 1 sleeps 30 ms per step, rank 0 5 ms: the line must report the slow rank's time."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, VPX_BENCH_STUB_MS="5,30", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "4", "--warmup", "1",
                        "--mode", "train", "--batch", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["global_batch"] == 16
    assert out["config"]["backend"] == "gloo (stub)" and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert out["steps"] == 4 and out["warmup"] == 1
    assert 30.0 <= out["ms_per_step"] < 80.0, out["ms_per_step"]            # the slowest rank's step, not rank 0's 5 ms
    assert out["value"] == pytest.approx(2 * 8 * 10 * 4 / (out["ms_per_step"] * 4e-3), rel=1e-3)   # whole-job frames / the bracketed time
