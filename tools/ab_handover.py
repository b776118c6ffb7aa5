"""Inference step of convlstm-shi with / without the operand-format handover between recurrent blocks and stage glue
(models.ef_conv_lstm.SPLIT_HANDOVER), one process, interleaved rounds. BB = batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vp_suite_amd as v
from vp_suite_amd.models import MODEL_CLASSES, ef_conv_lstm
dev = torch.device("cuda:0")
B = int(os.environ.get("BB", 128))
torch.manual_seed(0)
m = MODEL_CLASSES["convlstm-shi"]("cuda:0", img_shape=(1, 64, 64), action_size=0, tensor_value_range=[0.0, 1.0], cell_precision="bf16x3").to(dev)
x = torch.rand(B, 10, 1, 64, 64, device=dev)
res, outs = {}, {}
with torch.no_grad():
    for rnd in range(5):
        for mode in (0, 1):
            ef_conv_lstm.SPLIT_HANDOVER = bool(mode)
            for _ in range(2): outs[mode] = m(x, pred_frames=10)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): m(x, pred_frames=10)
            torch.cuda.synchronize()
            res.setdefault(mode, []).append((time.perf_counter() - t0) / 10)
print("max |difference| of the predictions:", float((outs[0] - outs[1]).abs().max()))
for mode in (0, 1):
    r = sorted(res[mode])
    print(f"B={B} handover={'on' if mode else 'off'}: median {r[len(r)//2]*1e3:.3f} ms best {r[0]*1e3:.3f} ms -> {B*10/r[len(r)//2]:.0f} frames/s", flush=True)
