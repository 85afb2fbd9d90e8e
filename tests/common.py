"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_NPZ = os.path.join(HERE, "golden", "golden.npz")
LITERAL_JSON = os.path.join(HERE, "golden", "literal.json")

GROUPS = [
    ("alt_bn128_g1", 0, 1), ("alt_bn128_g2", 0, 2), ("bls12_377_g1", 1, 1),
    ("bls12_377_g2", 1, 2), ("bw6_761_g1", 2, 1), ("bw6_761_g2", 2, 2),
    ("bls12_381_g1", 3, 1), ("bls12_381_g2", 3, 2),
]
GROUP_IDS = [g[0] for g in GROUPS]
MSM_SIZES = [1, 2, 3, 4, 5, 256, 257]
DIGIT_CS = [2, 3, 5, 8, 11, 12, 16, 17, 21]

_golden = None
_literal = None


def golden():
    global _golden
    if _golden is None:
        _golden = dict(np.load(GOLDEN_NPZ))
    return _golden


def literal():
    global _literal
    if _literal is None:
        with open(LITERAL_JSON) as f:
            _literal = json.load(f)
    return _literal


def small_scalars_mont(backend, curve, vals):
    """Fr(v) for small non-negative integers in Montgomery form, via an oracle backend."""
    fl = backend.sizes(curve, 1)["fr_bytes"] // 8
    plain = np.zeros((len(vals), fl), dtype=np.uint64)
    for i, v in enumerate(vals):
        plain[i, 0] = v
    return backend.fr_from_bigint(curve, plain)


def to_int(limbs):
    return sum(int(x) << (64 * i) for i, x in enumerate(limbs))
