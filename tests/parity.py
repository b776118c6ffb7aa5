"""Parity record of the GPU suite (round 6): EVERY comparison a `-m gpu` test makes goes through relmax() below, which returns the figure
the tests' bounds are stated in — max|got - ref| / max|ref| over the tensor — and records it together with the plain max abs difference
and the ELEMENT-WISE relative error |got - ref| / max(|ref|, 1e-3 max|ref|) (maximum and 99.9th percentile; elements below a thousandth
of the tensor's scale are measured against that floor: their own magnitude is rounding noise of the sums). conftest.py names the running
test, writes gpurun_out/parity_r06.json at the end of the session and prints the worst entries into pytest's terminal summary."""
import numpy as np

ENTRIES = []
_current = {"test": None, "n": 0}


def set_current(nodeid):
    _current["test"], _current["n"] = nodeid, 0


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def record(name, got, ref, bound=None):
    g, r = _np(got).astype(np.float64, copy=False), _np(ref).astype(np.float64, copy=False)
    diff = np.abs(g - r)
    d = float(diff.max()) if diff.size else 0.0
    rmax = float(np.abs(r).max()) if r.size else 0.0
    rel = d / (rmax + 1e-30)
    ew = diff / np.maximum(np.abs(r), 1e-3 * rmax + 1e-30)
    flat = ew.reshape(-1)
    if flat.size > 4_000_000:   # (the quantile of a subsample: the maximum above is over everything)
        flat = flat[:: flat.size // 2_000_000]
    ENTRIES.append({"test": _current["test"], "name": name, "value": rel, "bound": bound, "max_abs_diff": d, "ref_max_abs": rmax,
                    "elementwise_rel_max": float(ew.max()) if ew.size else 0.0,
                    "elementwise_rel_p999": float(np.quantile(flat, 0.999)) if flat.size else 0.0, "numel": int(r.size)})
    return rel


def relmax(a, b):
    """max|a - b| / max|b| (b = the reference side), recorded under the running test's name + the call's ordinal inside it."""
    _current["n"] += 1
    return record(f"cmp{_current['n']}", a, b)
