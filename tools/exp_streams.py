"""Developer experiment (round 6): is a small-batch forward GPU-bound on ONE kernel at a time? Two independent B=4 forwards of convlstm-shi
on two HIP streams against the same two one after the other; plus the host time of a forward (enqueue only) against its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd.models import MODEL_CLASSES
B = int(os.environ.get("BB", 4))
img = int(os.environ.get("IMG", 64)); ch = int(os.environ.get("CH", 1)); pred = int(os.environ.get("PRED", 10))
kw = dict(img_shape=(ch, img, img), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision="bf16x3")
ms = [MODEL_CLASSES["convlstm-shi"]("cuda", **kw).cuda() for _ in range(2)]
xs = [torch.rand(B, 10, ch, img, img, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def fwd(i):
    with torch.no_grad():
        return ms[i](xs[i], pred_frames=pred)
for i in range(2):
    for _ in range(3): fwd(i)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): fwd(0); fwd(1)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_seq = time.perf_counter() - t0
# two streams, one host thread
for i in range(2):
    with torch.cuda.stream(streams[i]):
        for _ in range(2): fwd(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    for i in range(2):
        with torch.cuda.stream(streams[i]): fwd(i)
t_host2 = time.perf_counter() - t0
torch.cuda.synchronize()
t_par = time.perf_counter() - t0
print(f"B={B} {ch}x{img}x{img} 10->{pred}: sequential {t_seq / (2 * N) * 1e3:.3f} ms per forward (host enqueue {t_host / (2 * N) * 1e3:.3f} ms); two streams {t_par / (2 * N) * 1e3:.3f} ms per forward "
      f"(host enqueue {t_host2 / (2 * N) * 1e3:.3f} ms) -> x{t_seq / t_par:.2f}")
# graph replay of one forward vs two graphs on two streams
gs, outs = [], []
for i in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(streams[i]):
        fwd(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=streams[i]):
            outs.append(fwd(i))
    gs.append(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): gs[0].replay(); gs[1].replay()
torch.cuda.synchronize()
t_g = time.perf_counter() - t0
print(f"   graph replay (both on the replaying stream): {t_g / (2 * N) * 1e3:.3f} ms per forward")
t0 = time.perf_counter()
for _ in range(N):
    for i in range(2):
        with torch.cuda.stream(streams[i]): gs[i].replay()
torch.cuda.synchronize()
t_g2 = time.perf_counter() - t0
print(f"   graph replay on two streams: {t_g2 / (2 * N) * 1e3:.3f} ms per forward -> x{t_g / t_g2:.2f}")
