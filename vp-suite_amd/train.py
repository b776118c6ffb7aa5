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
        if self.collectives:
            self.broadcast_parameters()
        self.optimizer = torch.optim.Adam(self.params, lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, patience=5, factor=0.2, min_lr=1e-6)

    def broadcast_parameters(self, src: int = 0):
        with torch.no_grad():
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
