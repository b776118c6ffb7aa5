"""Developer tool: event-timed glue calls inside a B=4 EF_ConvLSTM forward (no profiler attached)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd import ops
import bench
dev = torch.device("cuda:0")
B = int(os.environ.get("BB", 4))
spec = bench.Spec("dbg", batch=B)
runner = bench.Runner(spec, dev, 0, 1, False)
acc = collections.defaultdict(list)
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record()
        acc[name + str(tuple(a[1]) if name == "conv2d_ex_from_split" else "")].append((e0, e1))
        return r
    setattr(ops, name, g)
for n in ("conv2d_ex_from_split", "split_convert", "conv2d_ex"):
    wrap(n)
for _ in range(5):
    runner.step()
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
for _ in range(10):
    runner.step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
for k, evs in acc.items():
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{k:60s} n={len(ts)} median {ts[len(ts)//2]*1e3:8.1f} us  max {ts[-1]*1e3:8.1f} us")
