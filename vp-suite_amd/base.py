"""VPModel / VPModelBlock: the drop-in surface of the reference (vp_suite/base/base_model.py:11-216,
base_model_block.py:4-13), restated. Same constructor contract, class constants, config property, unpack_data,
train_iter / eval_iter loop semantics."""
import torch
from torch import nn

from .utils import get_public_attrs, set_from_kwarg

try:
    from tqdm import tqdm as _progress
except Exception:  # pragma: no cover
    def _progress(it):
        return it


class VPModelBlock(nn.Module):
    NAME: str = __name__
    PAPER_REFERENCE = None
    CODE_REFERENCE = None
    MATCHES_REFERENCE: str = None


class VPModel(nn.Module):
    NON_CONFIG_VARS = ["functions", "model_dir", "dump_patches", "training"]

    NAME = None
    PAPER_REFERENCE = None
    CODE_REFERENCE = None
    MATCHES_REFERENCE: str = None
    REQUIRED_ARGS = ["img_shape", "action_size", "tensor_value_range"]
    CAN_HANDLE_ACTIONS = False
    TRAINABLE = True
    NEEDS_COMPLETE_INPUT = False
    MIN_CONTEXT_FRAMES = 1

    model_dir = None
    img_shape = None
    action_size = None
    action_conditional = False
    tensor_value_range = None

    def __init__(self, device: str, **model_kwargs):
        super().__init__()
        # per-instance copy: subclasses extend this list in __init__ (the reference mutates the class list)
        self.NON_CONFIG_VARS = list(type(self).NON_CONFIG_VARS)
        self.device = device
        for name in self.REQUIRED_ARGS:
            if name == "tensor_value_range":
                rng = model_kwargs.get(name, (0, 0))
                if type(rng) not in (tuple, list) or len(rng) != 2:
                    raise ValueError("value for argument 'tensor_value_range' needs to be tuple or list with 2 elems")
            set_from_kwarg(self, model_kwargs, name, required=True)
            if name == "img_shape":
                self.img_c, self.img_h, self.img_w = self.img_shape
        for name in model_kwargs:
            if name not in self.REQUIRED_ARGS:
                set_from_kwarg(self, model_kwargs, name)

    @property
    def config(self):
        attrs = get_public_attrs(self, "config", non_config_vars=self.NON_CONFIG_VARS, model_mode=True)
        c, h, w = self.img_shape
        attrs.update({"img_h": h, "img_w": w, "img_c": c, "NAME": self.NAME})
        return attrs

    def unpack_data(self, data, config: dict, reverse: bool = False, complete: bool = False):
        """data: {"frames": [b,T,c,h,w] (or [T,c,h,w]), "actions": [b,T-1,a]} -> (input, target, actions)."""
        frames = data["frames"].to(config["device"])
        actions = data["actions"].to(config["device"])
        if frames.ndim == 4:
            frames, actions = frames.unsqueeze(0), actions.unsqueeze(0)
        if reverse:
            frames, actions = torch.flip(frames, dims=[1]), torch.flip(actions, dims=[1])
        t_in, t_pred = config["context_frames"], config["pred_frames"]
        if self.NEEDS_COMPLETE_INPUT or complete:
            inp = frames[:, :t_in + t_pred]
            return inp, inp[:, t_in:].clone(), actions
        inp, target = torch.split(frames[:, :t_in + t_pred], [t_in, t_pred], dim=1)
        return inp, target, actions

    def pred_1(self, x, **kwargs):
        raise NotImplementedError

    def forward(self, x, pred_frames: int = 1, **kwargs):
        preds = []
        for _ in range(pred_frames):
            nxt = self.pred_1(x, **kwargs).unsqueeze(dim=1)
            preds.append(nxt)
            x = torch.cat([x, nxt], dim=1)
        return torch.cat(preds, dim=1), None

    def _total_loss(self, predictions, targets, model_losses, loss_provider):
        _, total = loss_provider.get_losses(predictions, targets)
        if model_losses is not None:
            for value in model_losses.values():
                total = total + value
        return total

    def training_loss(self, inp, targets, pred_frames, loss_provider, **fwd_kwargs):
        """The scalar one training iteration differentiates (base_model.py:165-171): prediction losses + model losses.
        Models with their own iteration semantics override it (PredRNN-V2); `train.DataParallelTrainer` calls it too,
        so data-parallel training runs exactly what `train_iter` runs."""
        predictions, model_losses = self(inp, pred_frames=pred_frames, **fwd_kwargs)
        return self._total_loss(predictions, targets, model_losses, loss_provider)

    def train_iter(self, config, loader, optimizer, loss_provider, epoch):
        """One pass over `loader`: forward, loss (+ model losses), zero_grad, backward, optimizer step
        (base_model.py:162-179)."""
        loop = _progress(loader)
        for data in loop:
            inp, targets, actions = self.unpack_data(data, config)
            total = self.training_loss(inp, targets, config["pred_frames"], loss_provider, actions=actions)
            optimizer.zero_grad()
            total.backward()
            optimizer.step()
            if hasattr(loop, "set_postfix"):
                loop.set_postfix(loss=total.item())

    def eval_iter(self, config, loader, loss_provider):
        """Validation pass under no_grad; returns ({loss: mean value}, indicator loss) (base_model.py:194-216)."""
        self.eval()
        per_batch, indicators = [], []
        with torch.no_grad():
            for data in _progress(loader):
                inp, targets, actions = self.unpack_data(data, config)
                predictions, _ = self(inp, pred_frames=config["pred_frames"], actions=actions)
                values, _ = loss_provider.get_losses(predictions, targets)
                per_batch.append(values)
                indicators.append(values[config["val_rec_criterion"]])
        indicator = torch.stack(indicators).mean()
        means = {k: torch.stack([v[k] for v in per_batch]).mean().item() for k in per_batch[0]}
        self.train()
        return means, indicator
