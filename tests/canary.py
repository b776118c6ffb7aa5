"""Guard bands around every GPU tensor the tests and the package allocate from Python.

`install()` wraps torch.empty / zeros / empty_like / zeros_like / ones: a CUDA allocation becomes a flat byte buffer of
[guard | payload | guard], the guards (64 KiB each) filled with 0xFF, the tensor handed out a view of the payload. `check()` (run
after every GPU test by conftest.py) verifies every guard byte of every buffer allocated since the last check — also of tensors
that were freed in between: buffers are kept alive (up to a byte budget) until they have been checked, so a library call that
writes past ITS workspace cannot hide behind the allocator handing that memory to the next tensor.

Modes (environment VPX_CANARY): 0 off; 1 (default for -m gpu runs) guards; 2 additionally fills the payload of `empty` tensors
with 0xFF (fp32 / bf16 NaN): a kernel that reads memory nobody wrote — padding lanes multiplied by zero weights included —
turns its output into NaN instead of depending on what the allocator left there."""
import os

import torch

GUARD = 1 << 16
BUDGET = 24 << 30          # bytes of already-freed buffers kept for the next check
MODE = int(os.environ.get("VPX_CANARY", "1"))

_orig = {}
_pending = []              # [base uint8 tensor, payload bytes, tag]
_pending_bytes = 0
violations = []


def _is_cuda_device(device):
    if device is None:
        return False
    try:
        return torch.device(device).type == "cuda"
    except Exception:
        return False


def _shape_of(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(int(s) for s in args[0])
    return tuple(int(s) for s in args)


def _guarded(shape, dtype, device, fill):
    """fill: None (uninitialised / poisoned), or a byte-pattern-free value written with fill_ on the payload view."""
    global _pending_bytes
    dtype = dtype or torch.get_default_dtype()
    item = torch.empty((), dtype=dtype).element_size()
    n = 1
    for s in shape:
        n *= s
    nbytes = n * item
    pad = (-nbytes) % 256
    base = _orig["empty"](GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
    if MODE >= 2 and fill is None:
        base.fill_(0xFF)
    else:
        base[:GUARD].fill_(0xFF)
        base[GUARD + nbytes:].fill_(0xFF)
    t = base[GUARD:GUARD + nbytes].view(dtype).view(shape)
    if fill is not None:
        t.fill_(fill)
    _pending.append([base, nbytes, f"{tuple(shape)} {dtype}"])
    _pending_bytes += base.numel()
    while _pending_bytes > BUDGET and len(_pending) > 1:
        _check_entry(_pending.pop(0), sync=True)
    return t


def _factory(name, fill):
    def fn(*args, **kw):
        dev = kw.get("device")
        plain = set(kw) <= {"device", "dtype", "requires_grad"}
        if MODE and plain and _is_cuda_device(dev) and not kw.get("requires_grad", False):
            try:
                shape = _shape_of(args)
            except Exception:
                return _orig[name](*args, **kw)
            return _guarded(shape, kw.get("dtype"), dev, fill)
        return _orig[name](*args, **kw)
    return fn


def _like(name, fill):
    def fn(t, **kw):
        if MODE and not kw and isinstance(t, torch.Tensor) and t.is_cuda and t.layout == torch.strided and not t.requires_grad:
            # same sizes and (dense) strides as the stock op; non-dense inputs fall through
            order = sorted(range(t.dim()), key=lambda i: (t.stride(i), t.size(i)), reverse=True)
            expect, dense = 1, True
            for i in reversed(order):
                if t.size(i) != 1 and t.stride(i) != expect:
                    dense = False
                    break
                expect *= t.size(i)
            if dense and t.numel() > 0:
                flat = _guarded((t.numel(),), t.dtype, t.device, fill)
                return flat.as_strided(t.size(), t.stride())
        return _orig[name](t, **kw)
    return fn


def install():
    if not MODE or _orig:
        return
    for name, fill in (("empty", None), ("zeros", 0), ("ones", 1)):
        _orig[name] = getattr(torch, name)
        setattr(torch, name, _factory(name, fill))
    for name, fill in (("empty_like", None), ("zeros_like", 0)):
        _orig[name] = getattr(torch, name)
        setattr(torch, name, _like(name, fill))


def _check_entry(ent, sync):
    global _pending_bytes
    base, nbytes, tag = ent
    _pending_bytes -= base.numel()
    lo = (base[:GUARD] != 0xFF)
    hi = (base[GUARD + nbytes:] != 0xFF)
    bad = lo.any() | hi.any()
    if sync and bool(bad):
        _report(base, nbytes, tag, lo, hi)
    return bad, (base, nbytes, tag, lo, hi)


def _report(base, nbytes, tag, lo, hi):
    nlo, nhi = int(lo.sum()), int(hi.sum())
    first_hi = int(hi.nonzero()[0]) if nhi else -1
    last_hi = int(hi.nonzero()[-1]) if nhi else -1
    violations.append(f"guard band overwritten around tensor {tag} ({nbytes} payload bytes): {nlo} bytes before it, {nhi} bytes after it "
                      f"(offsets {first_hi}..{last_hi} past the end)")


def check():
    """Verifies and releases every buffer allocated since the last call; returns the list of violations found (and clears it)."""
    global _pending
    if not MODE:
        return []
    torch.cuda.synchronize()
    ents, _pending = _pending, []
    results = [_check_entry(e, sync=False) for e in ents]
    if results:
        anybad = torch.stack([r[0] for r in results]).any()
        if bool(anybad):
            for bad, info in results:
                if bool(bad):
                    _report(*info)
    out = list(violations)
    violations.clear()
    return out
