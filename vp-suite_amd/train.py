"""Batch-sharded data-parallel training harness for the hot path — NEW functionality (the reference is single-process,
SURVEY.md §2.1), restating the training semantics of the reference around it:

  loss      MSE summed over (c,h,w), averaged over t then b, scale 1.0 (+ model losses)   base_measure.py:57, base_model.py:168-171
  optimizer Adam(lr) + ReduceLROnPlateau(patience=5, factor=0.2, min_lr=1e-6) on val MSE    vpsuite.py:353-355
  loop      zero_grad -> backward -> step                                                   base_model.py:174-176

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests). The only
exchange is ONE all-reduce of ONE flat fp32 gradient bucket per step: every parameter's .grad is a view into the
bucket, so there is no gather/scatter copy around the collective. Because the loss is a batch mean, all-reduce(sum)/W
of the per-shard gradients equals the single-process gradient of the global batch."""
import torch
import torch.distributed as dist

from .measure import PredictionLossProvider


def shard_batch(t: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """Contiguous batch shard of rank `rank` (global batch must be divisible by the world size)."""
    b = t.shape[0]
    if b % world_size:
        raise ValueError(f"global batch {b} is not divisible by world size {world_size}")
    per = b // world_size
    return t[rank * per:(rank + 1) * per]


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) (vpsuite.py:353) as ONE HIP kernel over flat buckets: `params` are views into
    `flat_param`, their .grad views into `flat_grad`. A torch Optimizer (param_groups / state_dict / zero_grad), so LR
    schedulers such as ReduceLROnPlateau (vpsuite.py:354) drive it unchanged."""

    def __init__(self, params, flat_param, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_param, self.flat_grad = flat_param, flat_grad
        self.exp_avg = torch.zeros_like(flat_param)
        self.exp_avg_sq = torch.zeros_like(flat_param)
        self.steps = 0
        self.grad_scale = 1.0

    @classmethod
    def from_module(cls, module, lr=1e-3, **kw):
        """Re-homes the module's trainable parameters and their gradients into two flat buckets and returns the
        optimizer over them — the drop-in for `torch.optim.Adam(model.parameters(), lr=lr)` (vpsuite.py:353)."""
        params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in params)
        dev = params[0].device
        flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat_p[off:off + n].copy_(p.reshape(-1))
                p.data = flat_p[off:off + n].view_as(p)
                p.grad = flat_g[off:off + n].view_as(p)
                off += n
        return cls(params, flat_p, flat_g, lr=lr, **kw)

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        self.steps += 1
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.steps, g["lr"], g["betas"],
                      g["eps"], g["weight_decay"], self.grad_scale)
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that everything keyed
        # on (data_ptr, _version) — packed-weight and layout caches of the cells — sees new values
        torch.autograd.graph.increment_version(g["params"])
        return loss

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()  # grads stay views of the bucket


class DataParallelTrainer:
    def __init__(self, model, lr: float = 1e-4, world_size: int = None, losses_and_scales=None, device=None,
                 force_collectives: bool = False):
        self.model = model
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = device if device is not None else self.params[0].device
        self.loss_provider = PredictionLossProvider({"device": dev, "losses_and_scales": losses_and_scales or {"mse": 1.0}})
        # one flat gradient bucket; parameter grads are views into it
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat_grad[off:off + n].view_as(p)
            off += n
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        self.fused = torch.device(dev).type == "cuda"
        if self.fused:
            # parameters become views of ONE flat bucket as well, so the update is one kernel over (param, grad, m, v)
            self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in self.params:
                    n = p.numel()
                    self.flat_param[off:off + n].copy_(p.reshape(-1))
                    p.data = self.flat_param[off:off + n].view_as(p)
                    off += n
        if self.collectives:
            self.broadcast_parameters()
        self.optimizer = FlatAdam(self.params, self.flat_param, self.flat_grad, lr=lr) if self.fused \
            else torch.optim.Adam(self.params, lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, patience=5, factor=0.2, min_lr=1e-6)

    def broadcast_parameters(self, src: int = 0):
        with torch.no_grad():
            if getattr(self, "fused", False):
                dist.broadcast(self.flat_param, src=src)  # one message for the whole model
                return
            for p in self.params:
                dist.broadcast(p.data, src=src)

    def loss(self, predictions, targets, model_losses):
        _, total = self.loss_provider.get_losses(predictions, targets)
        if model_losses is not None:
            for value in model_losses.values():
                total = total + value
        return total

    def reduce_gradients(self):
        if self.collectives:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            if self.world > 1:
                if self.fused:
                    self.optimizer.grad_scale = 1.0 / self.world  # folded into the update kernel
                else:
                    self.flat_grad.div_(self.world)

    def step(self, x, target, pred_frames: int, **fwd_kwargs):
        """One optimisation step on this rank's shard. Returns the local loss tensor (no host sync)."""
        self.flat_grad.zero_()
        predictions, model_losses = self.model(x, pred_frames=pred_frames, **fwd_kwargs)
        total = self.loss(predictions, target, model_losses)
        total.backward()
        self.reduce_gradients()
        self.optimizer.step()
        return total.detach()

    @torch.no_grad()
    def validate(self, batches, pred_frames: int):
        """Mean validation MSE over `batches` of (x, target), averaged over ranks; steps the LR scheduler."""
        self.model.eval()
        vals = [self.loss_provider.get_losses(self.model(x, pred_frames=pred_frames)[0], y)[0]["mse"] for x, y in batches]
        self.model.train()
        v = torch.stack(vals).mean()
        if self.world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            v = v / self.world
        self.scheduler.step(v.item())
        return v
