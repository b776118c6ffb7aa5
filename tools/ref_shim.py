"""Import the upstream reference (AIS-Bonn/vp-suite, mounted read-only at /root/reference) in THIS container only.

Test/fixture infrastructure — never shipped, never imported by the product path, never used on the GPU box.

The reference cannot be imported as-is here:
  * vp_suite/defaults.py:24-30 writes a json file next to itself at import time -> needs a writable copy;
  * vp_suite/base/base_dataset.py:11 imports torch._utils._accumulate (gone in torch 2.x);
  * the import chain pulls cv2 / torchvision / piqa / wandb / ... which are not installed.
None of the stubbed modules is touched at run time by the ConvLSTM / ST-LSTM hot path.
"""
import itertools
import os
import shutil
import sys
import tempfile
from unittest.mock import MagicMock

REFERENCE_ROOT = "/root/reference"

_STUBS = [
    "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.datasets",
    "torchvision.io", "cv2", "piqa", "piqa.lpips", "piqa.ssim", "wandb", "imageio", "torchfile", "moviepy",
    "moviepy.editor", "tfrecord", "tfrecord.tools", "tfrecord.tools.tfrecord2idx", "tfrecord.torch",
    "tfrecord.torch.dataset", "optuna",
]

_loaded = None


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vp_suite"))


def load_reference():
    """Returns the imported `vp_suite` package of the reference (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference not mounted at /root/reference (this only works in the build container)")
    import torch._utils
    if not hasattr(torch._utils, "_accumulate"):
        torch._utils._accumulate = itertools.accumulate
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = MagicMock(name=name)
    tmp = tempfile.mkdtemp(prefix="vpsuite_ref_")
    shutil.copytree(os.path.join(REFERENCE_ROOT, "vp_suite"), os.path.join(tmp, "vp_suite"))
    sys.path.insert(0, tmp)
    import vp_suite  # noqa: E402
    import vp_suite.models  # noqa: E402,F401
    import vp_suite.model_blocks  # noqa: E402,F401
    _loaded = vp_suite
    return vp_suite
