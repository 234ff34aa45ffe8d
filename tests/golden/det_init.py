"""Deterministic parameter / input values shared by the fixture generators and the tests that replay them.

Large weight tensors (the 768->512 EdgeConv layer alone is 1.5 MB) are not stored in the fixtures: both
sides rebuild them from (name, shape) with 64-bit integer arithmetic only, so the values are identical
on every machine and numpy version (no libm, no Generator stream involved)."""
import zlib

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def det_values(name, shape, scale=1.0, offset=0.0):
    """float32 array of `shape`, uniform in offset + scale * [-0.5, 0.5), a pure function of (name, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * _M1 + np.uint64(zlib.crc32(name.encode()) + 1) * _M2
        z = (z ^ (z >> np.uint64(30))) * _M2      # splitmix64 finaliser
        z = (z ^ (z >> np.uint64(27))) * _M3
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5      # 24 bits: exact in fp32
    return (offset + scale * u).astype(np.float32).reshape(shape)


def det_state(module, prefix=""):
    """Fill a torch module's parameters / buffers in place from their state_dict names: weights uniform with the
    fan-in scaling of a default init (so activations stay O(1)), norm weights around 1, biases small, running
    statistics positive.  Returns the module."""
    import torch
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            leaf = k.rsplit(".", 1)[-1]
            if leaf == "running_var":
                val = det_values(prefix + k, tuple(v.shape), 1.0, 1.0)
            elif leaf == "running_mean":
                val = det_values(prefix + k, tuple(v.shape), 0.6, 0.0)
            elif v.ndim <= 1 and leaf == "weight":
                val = det_values(prefix + k, tuple(v.shape), 0.8, 1.0)
            elif v.ndim <= 1:
                val = det_values(prefix + k, tuple(v.shape), 0.4, 0.0)
            else:
                fan_in = int(np.prod(v.shape[1:]))
                val = det_values(prefix + k, tuple(v.shape), 2.0 * (3.0 / fan_in) ** 0.5, 0.0)
            v.copy_(torch.from_numpy(val).to(v.dtype))
    return module
