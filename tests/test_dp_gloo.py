"""Multi-process CPU test (gloo, world_size 2) of the data-parallel harness: gradient all-reduce (per-block buckets launched
from the backward pass, or one flat bucket after it), parameter broadcast and batch sharding reproduce the single-process
update on the global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _TinyPredictor(torch.nn.Module):
    """Stand-in with the VPModel call contract (forward(x, pred_frames) -> (pred, model_losses)); plain torch ops so it
    runs on CPU. The HIP models themselves have no CPU path."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(1, 4, 3, padding=1)
        self.head = torch.nn.Conv2d(4, 1, 1)
        self.unused = torch.nn.Linear(2, 2)   # never gets a gradient: its bucket must still go out (as zeros)

    def forward(self, x, pred_frames=1, **kw):
        last = x[:, -1]
        preds = []
        for _ in range(pred_frames):
            last = self.head(torch.tanh(self.conv(last)))
            preds.append(last)
        reg = {"reg": 1e-3 * sum((p ** 2).sum() for n, p in self.named_parameters() if not n.startswith("unused"))}
        return torch.stack(preds, dim=1), reg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, frames, ret, bucketed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vp_suite_amd  # noqa: F401
    from vp_suite_amd.train import DataParallelTrainer, shard_batch
    torch.manual_seed(100 + rank)  # deliberately different init per rank: the broadcast must fix it
    model = _TinyPredictor()
    tr = DataParallelTrainer(model, lr=1e-2, world_size=world, device="cpu", bucketed=bucketed)
    launches = []
    inner = tr._all_reduce
    tr._all_reduce = lambda t: (launches.append(t.numel()), inner(t))[1]
    mine = shard_batch(frames, rank, world)
    for _ in range(3):
        tr.step(mine[:, :3], mine[:, 3:], pred_frames=2)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        ret["params"] = flat.clone()
        ret["equal_across_ranks"] = bool(all(torch.equal(gathered[0], g) for g in gathered))
        ret["launches"] = list(launches)
        ret["buckets"] = [(b[3], b[1]) for b in tr.buckets]
    dist.destroy_process_group()


@pytest.mark.parametrize("bucketed", [True, False])
def test_dp_matches_single_process(bucketed):
    import vp_suite_amd  # noqa: F401
    from vp_suite_amd.train import DataParallelTrainer
    torch.manual_seed(7)
    frames = torch.rand(4, 5, 1, 8, 8)
    # single process, global batch, init = rank 0's init
    torch.manual_seed(100)
    ref_model = _TinyPredictor()
    tr = DataParallelTrainer(ref_model, lr=1e-2, world_size=1, device="cpu")
    for _ in range(3):
        tr.step(frames[:, :3], frames[:, 3:], pred_frames=2)
    want = torch.cat([p.detach().reshape(-1) for p in ref_model.parameters()])

    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, frames, ret, bucketed), nprocs=2, join=True)
    assert ret["equal_across_ranks"]
    names = [n for n, _ in ret["buckets"]]
    assert names == ["conv", "head", "unused"]
    total = sum(n for _, n in ret["buckets"])
    if bucketed:   # three steps x one launch per block, in the order the backward pass finishes them (head's gradients come first)
        assert len(ret["launches"]) == 9 and sum(ret["launches"]) == 3 * total
        assert ret["launches"][0] == dict(ret["buckets"])["head"] and ret["launches"][2] == dict(ret["buckets"])["unused"]
    else:
        assert ret["launches"] == [total] * 3
    assert torch.allclose(ret["params"], want, rtol=1e-5, atol=1e-7), float((ret["params"] - want).abs().max())


def test_shard_batch_rejects_ragged():
    import vp_suite_amd  # noqa: F401
    from vp_suite_amd.train import shard_batch
    t = torch.arange(12).reshape(6, 2)
    assert torch.equal(shard_batch(t, 1, 3), t[2:4])
    with pytest.raises(ValueError):
        shard_batch(t, 0, 4)


def test_grads_are_views_of_the_flat_bucket():
    import vp_suite_amd  # noqa: F401
    from vp_suite_amd.train import DataParallelTrainer
    m = _TinyPredictor()
    tr = DataParallelTrainer(m, world_size=1, device="cpu")
    x = torch.rand(2, 5, 1, 8, 8)
    tr.step(x[:, :3], x[:, 3:], pred_frames=2)
    base = tr.flat_grad.untyped_storage().data_ptr()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in m.parameters())
    assert float(tr.flat_grad.abs().sum()) > 0
