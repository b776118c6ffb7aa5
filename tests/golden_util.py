"""Seeded tensor helpers shared by the golden-vector generator (tools/gen_golden.py) and the tests.

Golden fixtures store only what cannot be regenerated: the reference's OUTPUTS. Inputs, parameters and cotangents are
regenerated from integer seeds with torch's CPU generator (mt19937 -> machine independent), so fixtures stay small.
Every fixture also stores a float64 checksum of each regenerated tensor, so a generator drift is detected
(instead of silently comparing against the wrong numbers).
"""
import os
import zlib

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def name_seed(name: str, base: int = 0) -> int:
    """Stable 31-bit seed from a string (python's hash() is salted, so use crc32)."""
    return (zlib.crc32(name.encode()) + 7919 * base) & 0x7FFFFFFF


def seeded_randn(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale


def seeded_rand(shape, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.rand(tuple(shape), generator=g, dtype=torch.float32)


def fan_in_scale(shape) -> float:
    """1/sqrt(fan_in) for conv-like weights, 0.1 for everything else: keeps activations O(1) with seeded weights."""
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return 1.0 / float(np.sqrt(max(fan_in, 1)))
    return 0.1


def fill_state_dict_(module: torch.nn.Module, base_seed: int):
    """Overwrites every entry of module.state_dict() in place with seeded values, keyed by the entry's NAME
    (independent of construction order). Peephole tensors (Wci/Wcf/Wco) get 0.1*randn so the peephole path is
    exercised (the reference initialises them to zero, conv_lstm_hzzone.py:30-32)."""
    sd = module.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if not torch.is_floating_point(t):
                continue
            leaf = key.split(".")[-1]
            if leaf in ("Wci", "Wcf", "Wco"):
                scale = 0.1
            elif "LayerNorm" in key or leaf == "bias":
                scale = 0.1
            else:
                scale = fan_in_scale(t.shape)
            val = seeded_randn(t.shape, name_seed(key, base_seed), scale)
            # LayerNorm weight (gamma) around 1
            if leaf == "weight" and t.ndim == 3:
                val = 1.0 + val
            t.copy_(val)
    return module


def checksum(t) -> float:
    a = t.detach().cpu().double().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)
    idx = np.arange(a.size, dtype=np.float64).reshape(a.shape)
    return float((a * np.cos(0.37 * idx)).sum())


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_golden(name: str):
    with np.load(golden_path(name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def seeded_state_dict(golden: dict, base_seed: int):
    """Rebuilds the seeded parameter dict of a fixture (same values fill_state_dict_ wrote into the reference module)
    from the fixture's key/shape table."""
    import json
    shapes = json.loads(str(golden["sd_shapes"]))

    class _Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._t = {k: torch.zeros(v) for k, v in shapes.items()}

        def state_dict(self, *a, **k):
            return self._t
    h = _Holder()
    fill_state_dict_(h, base_seed)
    return h._t
