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


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
