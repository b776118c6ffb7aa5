"""debug: localise the peephole-gradient mismatch of the 128x128 EF model against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import vp_suite_amd
from vp_suite_amd.models import MODEL_CLASSES
from vp_suite_amd.measure import PredictionLossProvider
from golden_util import fill_state_dict_, name_seed, seeded_rand
from oracle import torch_ref as tr
prec = os.environ.get("PREC", "bf16x3")
img = int(os.environ.get("IMG", 128)); B = int(os.environ.get("BB", 4)); T = int(os.environ.get("TT", 4)); Pn = int(os.environ.get("PP", 3))
m = MODEL_CLASSES["convlstm-shi"]("cuda", action_size=0, tensor_value_range=[0.0, 1.0], img_shape=(3, img, img), cell_precision=prec)
fill_state_dict_(m, name_seed("ef.c4train"))
m = m.cuda().train()
frames = seeded_rand((B, T + Pn, 3, img, img), name_seed("ef.c4train.x"))
lp = PredictionLossProvider({"device": "cuda", "losses_and_scales": {"mse": 1.0}})
pred, _ = m(frames[:, :T].cuda(), pred_frames=Pn)
_, loss = lp.get_losses(pred, frames[:, T:].cuda())
loss.backward()
sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
rp = tr.ef_convlstm_forward(sd, frames[:, :T], Pn)
rl = tr.mse_measure(rp, frames[:, T:])
rl.backward()
print("loss", float(loss), float(rl))
for k, p in m.named_parameters():
    g, r = p.grad.detach().cpu().numpy(), sd[k].grad.numpy()
    e = np.abs(g - r)
    rel = e.max() / (np.abs(r).max() + 1e-30)
    if rel > 1e-4 or k.endswith("Wci"):
        idx = np.unravel_index(e.argmax(), e.shape)
        print(f"{k:32s} rel {rel:.3e} max|ref| {np.abs(r).max():.3e} rms|ref| {np.sqrt((r**2).mean()):.3e} at {idx} got {g[idx]:.5e} want {r[idx]:.5e}")
        if e.ndim == 4 and e.shape[0] == 1:
            em = e[0].max(axis=0)  # [H, W]
            H, W = em.shape
            print("   border rows/cols max err:", em[0].max(), em[-1].max(), em[:, 0].max(), em[:, -1].max(), " interior:", em[2:-2, 2:-2].max())
            ys, xs = np.where(em > 0.5 * em.max())
            print("   #pixels with err > half max:", len(ys), " examples:", list(zip(ys[:8].tolist(), xs[:8].tolist())))
