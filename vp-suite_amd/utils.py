"""Host-side helpers with the reference's names and semantics (vp_suite/utils/utils.py:113-156, 208-234;
vp_suite/utils/models.py:131-193), restated."""
import inspect

import torch
from torch import nn


def set_from_kwarg(obj, kwarg_dict, attr_name, default=None, required=False, choices=None, skip_unusable=False):
    """obj.<attr_name> = kwarg_dict[attr_name] (or the class-level default). A value whose type differs from the
    default's type is rejected with ValueError, as is a missing required key (reference: utils.py:128-156)."""
    if required and attr_name not in kwarg_dict:
        raise ValueError(f"missing required parameter '{attr_name}' for object '{obj.__class__}'")
    if skip_unusable and not hasattr(obj, attr_name):
        print(f"parameter '{attr_name}' is not usable for init of object '{obj.__class__}' -> skipping")
    if default is None and hasattr(obj, attr_name):
        default = getattr(obj, attr_name)
    value = kwarg_dict.get(attr_name, default)
    if default is not None and not isinstance(value, type(default)):
        raise ValueError(f"mismatching types for parameter '{attr_name}' for object '{obj.__class__}'")
    if choices is not None:
        for i, v in enumerate(value if isinstance(value, list) else [value]):
            if v not in choices:
                raise ValueError(f"entry {i} of parameter '{attr_name}' is not one of the acceptable choices ({choices})")
    setattr(obj, attr_name, value)


def get_public_attrs(obj, calling_method=None, non_config_vars=None, model_mode=False):
    """Public, non-constant, non-routine attributes of obj as a dict (modules/tensors dropped in model_mode)."""
    result = {}
    for name in set(dir(obj)):
        if name.startswith("_") or name[0].isupper() or name == calling_method:
            continue
        value = getattr(obj, name)
        if inspect.isroutine(value):
            continue
        if model_mode and isinstance(value, (nn.Module, torch.Tensor)):
            continue
        result[name] = value
    for key in non_config_vars or []:
        result.pop(key, None)
    return result


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


def conv_output_shape(h_w, kernel_size=1, stride=1, pad=0, dilation=1):
    """(H, W) after a Conv2d."""
    h_w, k, s, p = _pair(h_w), _pair(kernel_size), _pair(stride), _pair(pad)
    return tuple((h_w[i] + 2 * p[i] - dilation * (k[i] - 1) - 1) // s[i] + 1 for i in range(2))


def convtransp_output_shape(h_w, kernel_size=1, stride=1, pad=0, dilation=1):
    """(H, W) after a ConvTranspose2d, with the reference's formula (utils/models.py:190-191:
    (in-1)*stride - 2*pad + (k-1) + pad — it coincides with torch for the k/s/p triples the EF models use)."""
    h_w, k, s, p = _pair(h_w), _pair(kernel_size), _pair(stride), _pair(pad)
    return tuple((h_w[i] - 1) * s[i] - 2 * p[i] + (k[i] - 1) + p[i] for i in range(2))
