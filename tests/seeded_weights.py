"""Deterministic policy weights for the model goldens (test infrastructure).

``tests/golden/model_golden.npz`` was produced by the reference's ``mpinets/model.py`` holding the weights this
module generates (19 M parameters: too large to commit, so the fixture stores the parameter names, shapes and a
SHA-256 of the values, and the tests regenerate them).  Every tensor is a function of (seed, parameter name, shape)
only -- no dependence on construction order or on torch's RNG:

* matrices / 1x1 convolutions: U(-a, a), a = sqrt(3 / fan_in)  (unit-gain: activations stay O(1) through the ReLU /
  LeakyReLU stacks, so the 1e-5 tolerances of the parity tests are not met trivially by tiny numbers);
* the decoder's last matrix is scaled by 0.25 so that a rollout moves ~0.1-0.5 of the normalised joint range per step
  (some joints reach the clamp at +-1 within five steps, most do not);
* biases: U(-0.1, 0.1);  GroupNorm scale 1 + 0.1 N(0,1), shift 0.1 N(0,1).
"""
from __future__ import annotations

import hashlib
import zlib
from typing import Dict, Mapping, Sequence

import numpy as np


def seeded_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, np.ndarray]:
    out = {}
    for name in sorted(shapes):
        shape = tuple(int(s) for s in shapes[name])
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        is_norm = ".fc_layer.1." in name or ".fc_layer.4." in name
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            w = rng.uniform(-1.0, 1.0, shape) * np.sqrt(3.0 / fan_in)
            if name == "decoder.6.weight":
                w *= 0.25
        elif name.endswith(".weight") and is_norm:
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif is_norm:
            w = 0.1 * rng.standard_normal(shape)
        else:
            w = rng.uniform(-0.1, 0.1, shape)
        out[name] = w.astype(np.float32)
    return out


def digest(sd: Mapping[str, np.ndarray]) -> str:
    h = hashlib.sha256()
    for name in sorted(sd):
        h.update(name.encode())
        h.update(np.ascontiguousarray(sd[name], dtype=np.float32).tobytes())
    return h.hexdigest()
