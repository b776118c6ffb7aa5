#!/usr/bin/env python3
"""A/B of the round-4 ST-LSTM kernels inside one process: predrnn-pp forward (and training step) timed under VPX_OPT_EXPERIMENT masks
(64: first-generation weight gradients, 128: data gradients, 256: forward launches, 512: 1x1 layers). BB, IMG, CH, PRED, LAYERS, MODE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd.models import MODEL_CLASSES
B = int(os.environ.get("BB", 4)); IMG = int(os.environ.get("IMG", 128)); CH = int(os.environ.get("CH", 3))
PRED = int(os.environ.get("PRED", 30)); LAYERS = int(os.environ.get("LAYERS", 4)); MODE = os.environ.get("MODE", "infer")
MASKS = [int(x) for x in os.environ.get("MASKS", "0,960,256,512").split(",")]
L = v._lib.lib()
torch.manual_seed(0)
m = MODEL_CLASSES["predrnn-pp"]("cuda", img_shape=(CH, IMG, IMG), action_size=0, tensor_value_range=[0.0, 1.0], num_layers=LAYERS,
                                cell_precision="bf16x3").cuda()
x = torch.rand(B, 10 + PRED, CH, IMG, IMG, device="cuda")
if MODE == "train":
    from vp_suite_amd.train import DataParallelTrainer
    tr = DataParallelTrainer(m, lr=1e-4, world_size=1)
def step():
    if MODE == "train":
        tr.step(x, x[:, 10:], PRED)
    else:
        with torch.no_grad():
            m(x, pred_frames=PRED)
res = {k: [] for k in MASKS}
for rnd in range(3):
    for k in MASKS:
        L.vpx_set_option(v._lib.OPT_EXPERIMENT, k)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        n = 3 if MODE == "train" else 6
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / n * 1e3)
L.vpx_set_option(v._lib.OPT_EXPERIMENT, 0)
for k in MASKS:
    print(f"B={B} {CH}x{IMG}x{IMG} 10->{PRED} L={LAYERS} {MODE} mask {k:4d}: best {min(res[k]):8.2f} ms  all {[round(t, 2) for t in res[k]]}")
