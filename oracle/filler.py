"""Oracle (test infrastructure): deterministic, name-seeded weight filler.

Golden fixtures carry inputs and expected outputs only, never weights: both the
reference modules (in ``tests/golden/make_golden.py``, build container only) and
the modules under test are filled by this function, keyed on the ``state_dict``
entry name and shape, so identical weights are reproduced anywhere from the
names alone.  Our own code; there is no reference counterpart.
"""
import zlib
import numpy as np
import torch


def fill_value(name, shape, dtype=torch.float32):
    rs = np.random.RandomState(zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        a = rs.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = rs.uniform(-0.2, 0.2, size=shape)
    elif len(shape) <= 1:
        is_norm = (".bn" in name or ".downsample.1" in name or "norm." in name) and leaf == "weight"
        a = rs.uniform(0.5, 1.5, size=shape) if is_norm else rs.uniform(-0.2, 0.2, size=shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        bound = np.sqrt(3.0 / fan_in) * 1.4
        a = rs.uniform(-bound, bound, size=shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float64)).to(dtype).reshape(shape)


def fill_state_dict(spec_or_sd):
    """spec: iterable of (name, shape) or a state_dict -> new dict name -> filled tensor."""
    if isinstance(spec_or_sd, dict):
        items = [(k, tuple(v.shape)) for k, v in spec_or_sd.items()]
    else:
        items = list(spec_or_sd)
    return {k: fill_value(k, s) for k, s in items}


def fill_module_(module):
    """In-place fill of every parameter/buffer of ``module`` (keys as in its state_dict)."""
    sd = module.state_dict()
    module.load_state_dict(fill_state_dict(sd), strict=True)
    return module
