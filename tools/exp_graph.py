"""Experiment: the B = 4 inference step (convlstm-shi, 64x64, 10 -> 10: BASELINE configs[0] as worded) as a captured HIP graph
(torch.cuda.CUDAGraph over the model's forward) against the eager call sequence: is the small-batch step bound by host-side
launch work? Prints ms per step for both and checks that the graph's output equals the eager one."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd.models import MODEL_CLASSES
dev = torch.device("cuda:0")
B = int(os.environ.get("BB", 4)); IMG = int(os.environ.get("IMG", 64)); CH = int(os.environ.get("CH", 1)); PRED = int(os.environ.get("PRED", 10))
MODEL = os.environ.get("MODEL", "convlstm-shi")
torch.manual_seed(0)
kw = dict(img_shape=(CH, IMG, IMG), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision=os.environ.get("PREC", "bf16x3"))
model = MODEL_CLASSES[MODEL](str(dev), **kw).to(dev)
frames = torch.rand(B, 10 + PRED, CH, IMG, IMG, device=dev)
x = frames if model.NEEDS_COMPLETE_INPUT else frames[:, :10]

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    eager = lambda: model(x, pred_frames=PRED)
    ref = eager()[0].clone()
    t_e = timeit(eager)
    static_x = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(static_x, pred_frames=PRED)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(static_x, pred_frames=PRED)
    def replay():
        static_x.copy_(x); g.replay()
    replay(); torch.cuda.synchronize()
    print("graph == eager:", bool(torch.equal(out[0], ref)), float((out[0] - ref).abs().max()))
    t_g = timeit(replay)
print(f"{MODEL} B={B} {IMG}x{IMG}x{CH} 10->{PRED}: eager {t_e:.3f} ms/step ({B * PRED / t_e * 1e3:.0f} frames/s), graph {t_g:.3f} ms/step ({B * PRED / t_g * 1e3:.0f} frames/s)")
