"""Helpers shared by the tests: golden-fixture loading and tolerance checks."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def weights_of(fx, prefix="w:"):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g) * scale


def rel_err(a, b):
    """max |a-b| / max |b| : the 'within 1e-4 rel' metric of the north star."""
    a = torch.as_tensor(a).detach().to("cpu", torch.float64)
    b = torch.as_tensor(b).detach().to("cpu", torch.float64)
    denom = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / denom)


def elementwise_violations(a, b, rtol=1e-4, floor=1e-5):
    """Elements with |a - b| > rtol * |b| + floor * max|b|: the north star's 1e-4 taken ELEMENT-WISE, with an
    absolute floor (a sum of many fp32 terms that cancel to a small value cannot be right to 1e-4 of itself)
    at half the global tolerance the tests assert.  Returns (count, worst excess ratio)."""
    a = torch.as_tensor(a).detach().to("cpu", torch.float64)
    b = torch.as_tensor(b).detach().to("cpu", torch.float64)
    bound = rtol * b.abs() + floor * b.abs().max().clamp_min(1e-30)
    bad = (a - b).abs() > bound
    worst = float(((a - b).abs() / bound).max()) if a.numel() else 0.0
    return int(bad.sum()), worst


def assert_close(a, b, tol, what=""):
    """Two criteria: the global one (max |a - b| <= tol * max |b|) and, for tolerances at or below the north
    star's 1e-4, the element-wise one of elementwise_violations (so small-magnitude outputs are checked in
    relative terms too, not only against the largest element)."""
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    if 0.0 < tol <= 1e-4 and torch.as_tensor(b).numel() > 1:
        n, worst = elementwise_violations(a, b)
        assert n == 0, f"{what}: {n} elements off by more than 1e-4 relative + 1e-5 * max (worst {worst:.2f}x the bound)"
