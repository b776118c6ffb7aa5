"""Probe: capture one EF_ConvLSTM forward into a HIP graph (torch.cuda.CUDAGraph) and compare replay time / outputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd.models import MODEL_CLASSES
B = int(os.environ.get("BB", 4))
dev = "cuda"
torch.manual_seed(0)
m = MODEL_CLASSES["convlstm-shi"](dev, img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision="bf16x3").to(dev).eval()
x = torch.rand(B, 10, 1, 64, 64, device=dev)
with torch.no_grad():
    for _ in range(300):
        ref, _ = m(x, pred_frames=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ref, _ = m(x, pred_frames=10)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            m(x, pred_frames=10)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out, _ = m(x, pred_frames=10)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20
print(f"B={B} eager {eager*1e3:.3f} ms  graph {graph*1e3:.3f} ms  maxdiff {(out-ref).abs().max().item():.2e}")
