"""The one loss the hot path's harness needs: MSE with the reference's reduction (sum over c,h,w -> mean over t -> mean
over b; vp_suite/base/base_measure.py:57, measure/image_wise.py:19-27) and the provider contract of
measure/loss_provider.py:30-53 (returns (display dict, scaled total))."""
import torch


def mse_measure(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    if pred.ndim != 5 or target.ndim != 5:
        raise ValueError("Mean Squared Error (MSE) / L2 Loss expects 5-D inputs!")
    if pred.is_cuda:  # one HIP pass producing the value and d/dpred (train_tail.hip)
        from . import ops
        return ops.mse_loss(pred, target)
    # host tensors (data-pipeline side checks, gloo tests): the reference's own expression
    return ((pred - target) ** 2).sum(dim=(4, 3, 2)).mean(dim=1).mean(dim=0)


class PredictionLossProvider:
    """config: {"device": ..., "losses_and_scales": {"mse": scale}}. Only "mse" is available in this build."""

    def __init__(self, config: dict):
        self.device = config["device"]
        scales = dict(config.get("losses_and_scales", {"mse": 1.0}))
        unknown = [k for k in scales if k != "mse"]
        if unknown:
            raise NotImplementedError(f"losses {unknown} are outside the hot-path scope of this build (only 'mse')")
        self.losses = {k: (mse_measure, s) for k, s in scales.items()}

    def get_losses(self, pred, target):
        if pred.shape != target.shape:
            raise ValueError("Output images and target images are of different shape!")
        display, total = {}, torch.zeros((), device=pred.device)
        for key, (fn, scale) in self.losses.items():
            val = fn(pred, target)
            total = total + scale * val
            display[key] = val
        return display, total
