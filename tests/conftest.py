import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vpx():
    """The product package (import shim vp_suite_amd.py -> directory vp-suite_amd/)."""
    import vp_suite_amd
    return vp_suite_amd


# ---- parity record: every GPU parity test may log (name, metric, value, bound); the session leaves parity_r04.json behind ----
_PARITY = []


@pytest.fixture
def parity_log():
    """parity_log(name, got, ref, bound) -> max|got - ref| / max|ref| (the metric every tolerance in tests/ is stated in), recorded
    together with the plain max abs difference (the reference's own convention: np.allclose(rtol=0, atol=1e-4))."""
    import numpy as np

    def log(name, got, ref, bound):
        g = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
        r = ref.detach().cpu().numpy() if hasattr(ref, "detach") else np.asarray(ref)
        d = float(np.abs(g - r).max())
        rel = d / (float(np.abs(r).max()) + 1e-30)
        _PARITY.append({"name": name, "metric": "max|got-ref| / max|ref|", "value": rel, "bound": bound, "max_abs_diff": d,
                        "ref_max_abs": float(np.abs(r).max())})
        return rel
    return log


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_r04.json"), "w") as fh:
        json.dump({"what": "per-test parity figures of the -m gpu run (HIP path vs the pinned oracle on the same seeded inputs)",
                   "metric": "max|got-ref| / max|ref| over the tensor (max-normalised, not element-wise relative); max_abs_diff next to it",
                   "entries": _PARITY}, fh, indent=1)
