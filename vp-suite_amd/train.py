"""Batch-sharded data-parallel training harness for the hot path — NEW functionality (the reference is single-process,
SURVEY.md §2.1), restating the training semantics of the reference around it:

  loss      MSE summed over (c,h,w), averaged over t then b, scale 1.0 (+ model losses)   base_measure.py:57, base_model.py:168-171
  optimizer Adam(lr) + ReduceLROnPlateau(patience=5, factor=0.2, min_lr=1e-6) on val MSE    vpsuite.py:353-355
  loop      zero_grad -> backward -> step                                                   base_model.py:174-176
  PredRNN   forward + time-reversed forward averaged, training_iteration += 1               predrnn_v2.py:319-365
            (through the model's `training_loss` hook, so the trainer runs each model's own train_iter semantics)

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests). The only
exchange is the all-reduce of the fp32 gradients: every parameter's .grad is a view into ONE flat buffer, so there is no
gather/scatter copy around the collective. The buffer is cut into per-block buckets (encoder.stage1, encoder.rnn1, ...,
contiguous ranges in parameter order); a bucket's all-reduce is launched asynchronously from the backward pass (autograd
post-accumulate hooks) and runs on RCCL's stream under the BPTT of the blocks still to come. Launch order is FIXED —
strictly descending bucket index, the order in which the backward pass finishes the blocks: bucket i goes out once its own
gradients are complete AND every bucket above it has gone out, so every rank enqueues the same collectives in the same order
whatever order its hooks fire in (a rank-dependent order would hang or mix up RCCL messages); what is still missing at the
end of the backward pass goes out in the same descending order. The last wait sits in front of the optimizer step
(`bucketed=False`: one all-reduce of the whole buffer after the backward pass). Because the loss is a batch mean, all-reduce(sum)/W of the per-shard gradients equals
the single-process gradient of the global batch."""
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


def _link_views(params, flat, attr):
    """Makes `p.<attr>` (data or grad) of every parameter a view into the flat bucket, preserving current values."""
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            view = flat[off:off + n].view_as(p)
            if attr == "data":
                view.copy_(p.data)
                p.data = view
            else:
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view
            off += n


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) (vpsuite.py:353) as ONE HIP kernel over flat buckets: `params` are views into
    `flat_param`, their .grad views into `flat_grad`. A torch Optimizer (param_groups / state_dict / zero_grad), so LR
    schedulers such as ReduceLROnPlateau (vpsuite.py:354) drive it unchanged. One parameter group only.

    One deliberate difference from torch.optim.Adam: EVERY parameter of the bucket is stepped in every iteration with the
    bucket-wide step count. torch skips a parameter whose .grad is None; here such a parameter is re-linked to a zeroed slice
    of the gradient bucket and takes a zero-gradient Adam step (its moments decay, remaining momentum still moves it, weight
    decay applies). All parameters of the shipped models receive a gradient in every iteration, so the trajectories coincide
    (tests/test_gpu_models.py pins them against torch.optim.Adam); a model with a branch that is unused in some iterations
    should freeze that branch (requires_grad = False) or use torch.optim.Adam."""

    def __init__(self, params, flat_param, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        if params and isinstance(params[0], dict):
            raise ValueError("FlatAdam: a single parameter group only (the flat buckets carry one set of hyper-parameters)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
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
        _link_views(params, flat_p, "data")
        _link_views(params, flat_g, "grad")
        return cls(params, flat_p, flat_g, lr=lr, **kw)

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise ValueError("FlatAdam: a single parameter group only")
        super().add_param_group(param_group)

    def _relink(self):
        """The update reads flat_grad and writes flat_param, so both must still back the parameters. `zero_grad(
        set_to_none=True)` (torch's default on a plain Optimizer), `p.grad = None` or `model.to()` break the link and
        autograd then accumulates into fresh tensors: detect that here and repair it instead of stepping on stale data."""
        gbase = self.flat_grad.untyped_storage().data_ptr()
        pbase = self.flat_param.untyped_storage().data_ptr()
        off = 0
        for p in self.param_groups[0]["params"]:
            n = p.numel()
            if p.data.untyped_storage().data_ptr() != pbase:
                if p.device != self.flat_param.device:
                    raise RuntimeError("FlatAdam: a parameter left the device of its flat bucket (model.to() after the "
                                       "optimizer was built); rebuild the optimizer with FlatAdam.from_module")
                view = self.flat_param[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view
            if p.grad is None:
                self.flat_grad[off:off + n].zero_()
                p.grad = self.flat_grad[off:off + n].view_as(p)
            elif p.grad.untyped_storage().data_ptr() != gbase:
                view = self.flat_grad[off:off + n].view_as(p)
                view.copy_(p.grad)
                p.grad = view
            off += n

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._relink()
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

    def state_dict(self):
        sd = super().state_dict()
        sd["flat_adam"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "steps": self.steps}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fa = state_dict.pop("flat_adam", None)
        super().load_state_dict(state_dict)
        if fa is not None:
            self.exp_avg.copy_(fa["exp_avg"])
            self.exp_avg_sq.copy_(fa["exp_avg_sq"])
            self.steps = int(fa["steps"])


class DataParallelTrainer:
    """`all_reduce` is the collective of the single exchange step (default torch.distributed.all_reduce(SUM)); tests
    replace it to execute the world>1 arithmetic (sum of shard gradients, 1/W folded into the update) on one device."""

    def __init__(self, model, lr: float = 1e-4, world_size: int = None, losses_and_scales=None, device=None,
                 force_collectives: bool = False, all_reduce=None, broadcast=None, bucketed: bool = True,
                 seed: int = None, rank: int = None):
        self.model = model
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = device if device is not None else self.params[0].device
        self.loss_provider = PredictionLossProvider({"device": dev, "losses_and_scales": losses_and_scales or {"mse": 1.0}})
        # returns None (done) or a handle with .wait(): the default launches asynchronously
        self._all_reduce = all_reduce if all_reduce is not None else \
            (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
        self._broadcast = broadcast if broadcast is not None else (lambda t, src: dist.broadcast(t, src=src))
        # one flat gradient bucket; parameter grads are views into it
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p in self.params:
            p.grad = None
        _link_views(self.params, self.flat_grad, "grad")
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        self.fused = torch.device(dev).type == "cuda"
        # per-block gradient buckets: contiguous ranges of the flat buffer, one per top-level block of the model
        self.bucketed = bool(bucketed)
        self.buckets = []          # [offset, numel, n_params, name]
        self._bucket_of = {}       # id(param) -> bucket index
        names = {id(p): n for n, p in model.named_parameters()}
        off = 0
        for p in self.params:
            parts = names.get(id(p), "").split(".")
            key = ".".join(parts[:max(1, min(2, len(parts) - 1))])   # "encoder.rnn1._conv.weight" -> "encoder.rnn1"; "conv.weight" -> "conv"
            if not self.buckets or self.buckets[-1][3] != key:
                self.buckets.append([off, 0, 0, key])
            self.buckets[-1][1] += p.numel()
            self.buckets[-1][2] += 1
            self._bucket_of[id(p)] = len(self.buckets) - 1
            off += p.numel()
        self._pending = []         # gradients still missing per bucket (this step)
        self._handles = []
        self._launched = []
        self._next = -1            # the bucket whose turn it is (descending); -1 = nothing outstanding
        self._hooks = []           # RemovableHandles: close() takes the hooks off the model again
        if self.collectives and self.bucketed:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._grad_ready))
        if self.fused:
            # parameters become views of ONE flat bucket as well, so the update is one kernel over (param, grad, m, v)
            self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
            _link_views(self.params, self.flat_param, "data")
        if self.collectives:
            self.broadcast_parameters()
        self.optimizer = FlatAdam(self.params, self.flat_param, self.flat_grad, lr=lr) if self.fused \
            else torch.optim.Adam(self.params, lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, patience=5, factor=0.2, min_lr=1e-6)
        if seed is not None:
            self.seed_rank_rng(seed)

    def seed_rank_rng(self, seed: int):
        """Per-rank random stream = seed + rank: the replicas hold identical parameters (broadcast) but must draw DIFFERENT
        scheduled-sampling masks for their shards (predrnn_v2.py:252-317 draws them from the global generator), as the
        single-process run draws independent rows for every sample of the global batch."""
        torch.manual_seed(int(seed) + self.rank)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(int(seed) + self.rank)

    def broadcast_parameters(self, src: int = 0):
        with torch.no_grad():
            if getattr(self, "fused", False):
                self._broadcast(self.flat_param, src)  # one message for the whole model
                torch.autograd.graph.increment_version(self.params)
                return
            for p in self.params:
                self._broadcast(p.data, src)

    def loss(self, predictions, targets, model_losses):
        _, total = self.loss_provider.get_losses(predictions, targets)
        if model_losses is not None:
            for value in model_losses.values():
                total = total + value
        return total

    def close(self):
        """Takes the gradient hooks off the model (a discarded trainer must not launch collectives from a later backward)."""
        for h in self._hooks:
            h.remove()
        self._hooks, self._pending, self._next = [], [], -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _launch_bucket(self, i):
        off, n = self.buckets[i][0], self.buckets[i][1]
        h = self._all_reduce(self.flat_grad[off:off + n])
        self._launched[i] = True
        if h is not None:
            self._handles.append(h)

    def _launch_ready(self, force: bool = False):
        """Launches buckets in strictly descending index order: `_next` goes out when complete (or `force`), then the one
        below it, ... — never a lower bucket before a higher one, so the sequence of collectives is rank-independent."""
        while self._next >= 0 and (force or self._pending[self._next] <= 0):
            self._launch_bucket(self._next)
            self._next -= 1

    def _grad_ready(self, p):
        """Post-accumulate hook: p's gradient of this step is final."""
        if not self._pending:
            return
        self._pending[self._bucket_of[id(p)]] -= 1
        self._launch_ready()

    def backward_shard(self, x, target, pred_frames: int, **fwd_kwargs):
        """Forward + loss + backward on this rank's shard; leaves the shard's gradient in `flat_grad`. Models with their
        own training semantics (PredRNN-V2: scheduled sampling, reversed pass) provide `training_loss`."""
        self.flat_grad.zero_()
        self._pending = [b[2] for b in self.buckets] if (self.collectives and self.bucketed) else []
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._next = len(self.buckets) - 1 if self._pending else -1
        hook = getattr(self.model, "training_loss", None)
        if hook is not None:
            total = hook(x, target, pred_frames, self.loss_provider, **fwd_kwargs)
        else:
            predictions, model_losses = self.model(x, pred_frames=pred_frames, **fwd_kwargs)
            total = self.loss(predictions, target, model_losses)
        total.backward()
        return total.detach()

    def reduce_gradients(self):
        if self.collectives:
            if self.bucketed:
                # buckets whose hooks did not all fire (a parameter without a gradient this step) go out now, same order
                if len(self._launched) != len(self.buckets):
                    self._launched = [False] * len(self.buckets)
                if self._next < 0 and not all(self._launched):
                    self._next = len(self.buckets) - 1   # reduce_gradients without backward_shard: everything, descending
                    self._pending = [0] * len(self.buckets)
                self._launch_ready(force=True)
            else:
                h = self._all_reduce(self.flat_grad)
                if h is not None:
                    self._handles.append(h)
            for h in self._handles:
                h.wait()
            self._handles, self._pending, self._next = [], [], -1
            if self.world > 1:
                if self.fused:
                    self.optimizer.grad_scale = 1.0 / self.world  # folded into the update kernel
                else:
                    self.flat_grad.div_(self.world)

    def step(self, x, target, pred_frames: int, **fwd_kwargs):
        """One optimisation step on this rank's shard. Returns the local loss tensor (no host sync)."""
        total = self.backward_shard(x, target, pred_frames, **fwd_kwargs)
        self.reduce_gradients()
        self.optimizer.step()
        return total

    @torch.no_grad()
    def validate(self, batches, pred_frames: int):
        """Mean validation MSE over `batches` of (x, target), averaged over ranks; steps the LR scheduler."""
        self.model.eval()
        vals = [self.loss_provider.get_losses(self.model(x, pred_frames=pred_frames)[0], y)[0]["mse"] for x, y in batches]
        self.model.train()
        v = torch.stack(vals).mean()
        if self.world > 1:
            v = v.clone()
            h = self._all_reduce(v)   # asynchronous by default: the sum must have landed before it is divided and read
            if h is not None:
                h.wait()
            v = v / self.world
        self.scheduler.step(v.item())
        return v
