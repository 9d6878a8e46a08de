"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "q4_golden.npz")


def golden():
    return np.load(GOLDEN)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rel_max_err(got, ref):
    """max_i |g_i - c_i| / max_i |c_i|  (per-element relative error is meaningless near 0)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    denom = np.max(np.abs(ref))
    return float(np.max(np.abs(got - ref)) / (denom if denom > 0 else 1.0))


def rel_err_rows(got, ref):
    """worst per-row rel_max_err (rows = activation columns n, which may differ wildly in scale)."""
    return max(rel_max_err(g, r) for g, r in zip(np.asarray(got), np.asarray(ref)))


_WCACHE = {}


def make_weights(port, qtype, M, K, seed, scale=0.02):
    key = (qtype, M, K, seed, scale)
    if key not in _WCACHE:
        if len(_WCACHE) > 3:
            _WCACHE.clear()
        rng = np.random.default_rng(seed)
        w = rng.standard_normal((M, K), dtype=np.float32) * np.float32(scale)
        _WCACHE[key] = port.quantize_q4(qtype, w)
    return _WCACHE[key].copy() if M * K < (1 << 22) else _WCACHE[key]


def make_x(N, K, seed):
    return np.random.default_rng(seed).normal(0, 1, (N, K)).astype(np.float32)
