"""B=4 (or BB) inference step of convlstm-shi, A/B inside one process: workspace / weight-pack cache on vs off."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd.models import MODEL_CLASSES
dev = torch.device("cuda:0")
B = int(os.environ.get("BB", 4))
torch.manual_seed(0)
m = MODEL_CLASSES["convlstm-shi"]("cuda:0", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision="bf16x3").to(dev)
x = torch.rand(B, 10, 1, 64, 64, device=dev)
res = {}
with torch.no_grad():
    for rnd in range(5):
        for mode in (0, 1):
            v.ops._CLSTM_WS_CACHE_LIMIT = (1 << 30) if mode else 0
            v.ops.clear_layout_cache()
            for _ in range(3): m(x, pred_frames=10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): m(x, pred_frames=10)
            torch.cuda.synchronize()
            res.setdefault(mode, []).append((time.perf_counter() - t0) / 20)
for mode in (0, 1):
    r = sorted(res[mode])
    print(f"B={B} cache={'on' if mode else 'off'}: median {r[len(r)//2]*1e3:.3f} ms best {r[0]*1e3:.3f} ms -> {B*10/r[len(r)//2]:.0f} frames/s", flush=True)
