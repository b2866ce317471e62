"""HIP VQ lookup (through the C ABI) against the CPU oracle and the golden vectors.

Indices must be bit-exact.  The only licence taken: a row whose two best codes are closer
than 4 ulp of the distance (|d| ~ 2 => 1e-6) may legitimately differ, because the reference's
own argmin then depends on MKL's summation order; such rows are counted
and must be absent on the seeded data used here (min margin 3.1e-5 at K = 8192).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded
from util import GOLDEN, assert_close, load_golden, weights_of

pytestmark = pytest.mark.gpu
TOL = 2e-5
TIE = 1e-6


def _check_indices(idx_dev, idx_ref, margin):
    idx_dev = idx_dev.cpu().numpy().reshape(-1)
    idx_ref = np.asarray(idx_ref).reshape(-1).astype(np.int64)
    bad = np.nonzero(idx_dev != idx_ref)[0]
    if len(bad):
        assert np.all(np.asarray(margin).reshape(-1)[bad] < TIE), f"{len(bad)} index mismatches away from ties"
    return len(bad)


def test_codebook_small_golden(device):
    from amk.models import Codebook

    fx = load_golden("codebook_small")
    K, C = fx["E"].shape
    cb = Codebook(K, C)
    cb.load_state_dict({"embedding.weight": torch.from_numpy(fx["E"])})
    cb = cb.to(device)
    z = torch.from_numpy(fx["z"]).to(device).requires_grad_(True)
    cot = torch.from_numpy(fx["cot"]).to(device)
    zq, idx, loss = cb(z)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == tuple(fx["idx"].shape)
    assert _check_indices(idx, fx["idx"], fx["margin"]) == 0
    assert_close(zq, fx["zq"], TOL, "z_q")
    assert_close(loss, fx["loss"], TOL, "loss")
    gz, gE = torch.autograd.grad((zq * cot).sum() + float(fx["loss_weight"]) * loss, [z, cb.embedding.weight])
    assert_close(gz, fx["gz"], TOL, "grad z")
    assert_close(gE, fx["gE"], TOL, "grad codebook")
    assert_close(cb.indices_to_embeddings(idx), fx["emb"], TOL, "indices_to_embeddings")


def test_codebook_config3_golden_indices(device):
    """K 8192 x C 32, N = 2 x 1024: indices the reference produced, bit-exact."""
    from amk import ops

    fx = load_golden("codebook_c3")
    K, C, B, T = (int(v) for v in fx["dims"])
    s_e, s_z = (int(v) for v in fx["seeds"])
    E = seeded((K, C), s_e).to(device)
    z = seeded((B, T, C), s_z).to(device)
    zq, idx, loss = ops.vq_lookup(z, E, 0.25)
    assert _check_indices(idx, fx["idx"], fx["margin"]) == 0
    assert_close(loss, fx["loss"], TOL, "loss")
    assert abs(float(zq.double().sum()) - float(fx["zq_sum"])) < 1e-2


@pytest.mark.parametrize("N,K,C", [(1, 32, 32), (100, 64, 32), (129, 512, 64), (1000, 2048, 32), (4096, 8192, 32),
                                   (300, 256, 128), (513, 1024, 256), (33, 32, 256),
                                   # codebook sizes that are not whole tiles (the reference takes any K)
                                   (257, 1000, 32), (64, 1, 32), (200, 5, 64), (4096, 8191, 32), (130, 3000, 32),
                                   (77, 100, 128), (40, 33, 256)])
def test_lookup_vs_oracle(device, N, K, C):
    from amk import ops

    z = seeded((N, C), 1000 + N)
    E = seeded((K, C), 2000 + K)
    zc = z.clone().requires_grad_(True)
    Ec = E.clone().requires_grad_(True)
    cot = seeded((N, C), 3000 + N)
    zq_r, idx_r, loss_r = ref_cpu.codebook_forward(zc, Ec, 0.25)
    gz_r, gE_r = torch.autograd.grad((zq_r * cot).sum() + 2.0 * loss_r, [zc, Ec])
    margin = ref_cpu.codebook_margin(z, E) if K > 1 else torch.ones(N)

    zd = z.to(device).requires_grad_(True)
    Ed = E.to(device).requires_grad_(True)
    zq, idx, loss = ops.vq_lookup(zd, Ed, 0.25)
    assert _check_indices(idx, idx_r.numpy(), margin.numpy()) == 0
    assert_close(zq, zq_r, TOL, "z_q")
    assert_close(loss, loss_r, TOL, "loss")
    gz, gE = torch.autograd.grad((zq * cot.to(device)).sum() + 2.0 * loss, [zd, Ed])
    assert_close(gz, gz_r, TOL, "grad z")
    assert_close(gE, gE_r, TOL, "grad codebook")


def test_indices_to_embeddings_rejects_out_of_range_indices(device):
    """nn.Embedding raises IndexError for an index outside the codebook (models/vitvqgan.py:173-176);
    the kernel never dereferences such an index and the binding raises from its count."""
    from amk import ops

    K, C = 100, 32
    E = seeded((K, C), 9).to(device)
    good = torch.tensor([[0, 5, 99]], device=device)
    want = torch.nn.functional.normalize(E[good.view(-1)], dim=-1).view(1, 3, C)
    assert_close(ops.vq_gather(good, E), want, 1e-6, "gather")
    for bad in ([[0, 100, 3]], [[-1, 2, 3]], [[1 << 40, 2, 3]]):
        with pytest.raises(IndexError, match="outside the codebook"):
            ops.vq_gather(torch.tensor(bad, device=device), E)
    torch.cuda.synchronize()   # no device fault behind the exception


def test_exact_ties_take_first_index(device):
    """Duplicate code rows: torch.argmin returns the first of equal minima; so must the kernel,
    whichever half-wave / codebook slice / tile the duplicates fall in."""
    from amk import ops

    K, C, N = 8192, 32, 256
    E = seeded((K, C), 77)
    z = seeded((N, C), 78)
    # make z[n] exactly proportional to a code and plant duplicates of that code further on
    for n in range(N):
        src = (n * 37) % 4000
        z[n] = E[src] * 1.5
        E[src + 4000 + (n % 5)] = E[src]          # a later duplicate (other slice / tile / half)
    _, idx_r, _ = ref_cpu.codebook_forward(z, E, 0.25)
    _, idx, _ = ops.vq_lookup(z.to(device), E.to(device), 0.25)
    assert torch.equal(idx.cpu(), idx_r)
    assert int(idx_r.max()) < 4000


def test_zero_vector_and_full_size_properties(device):
    """(1) an all-zero z row normalises to zero (eps clamp) without NaN; (2) at the full C3 size
    (N = 32*1024) every row's chosen code is at least as close as 64 random other codes
    (a size-independent optimality check), indices are in range, and the output is idempotent:
    quantising z_q again returns the same indices."""
    from amk import ops

    K, C = 8192, 32
    E = seeded((K, C), 5).to(device)
    z = seeded((8, C), 6)
    z[3] = 0
    zq, idx, loss = ops.vq_lookup(z.to(device), E, 0.25)
    assert torch.isfinite(zq).all() and torch.isfinite(loss)

    N = 32 * 1024
    g = torch.Generator().manual_seed(1234)
    zbig = torch.randn(N, C, generator=g).to(device)
    zq, idx, loss = ops.vq_lookup(zbig, E, 0.25)
    assert int(idx.min()) >= 0 and int(idx.max()) < K
    zn = torch.nn.functional.normalize(zbig, dim=-1)
    en = torch.nn.functional.normalize(E, dim=-1)
    best = ((zn - en[idx]) ** 2).sum(-1)
    probe = torch.randint(0, K, (N, 64), generator=g).to(device)
    other = ((zn[:, None, :] - en[probe]) ** 2).sum(-1)
    assert bool((best[:, None] <= other + 1e-6).all())
    _, idx2, _ = ops.vq_lookup(zq, E, 0.25)
    assert torch.equal(idx, idx2)


def test_vitvqgan_small_golden(device):
    """Whole ViT-VQGAN (reduced size) with the reference's weights: reconstruction, loss,
    encode_imgs indices (bit-exact), decode_indices and every parameter gradient."""
    from amk.models import ViTVQGAN

    fx = load_golden("vitvqgan_small")
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))["vitvqgan_small"]
    m = ViTVQGAN(meta["cfg"], meta["codebook"])
    res = m.load_state_dict(weights_of(fx), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"]
    m = m.to(device)
    imgs = torch.from_numpy(fx["imgs"]).to(device)
    rec, loss = m(imgs)
    idx = m.encode_imgs(imgs)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (2, m.num_patches)
    assert _check_indices(idx, fx["idx"], fx["margin"]) == 0
    assert_close(rec, fx["rec"], 5e-5, "reconstruction")
    assert_close(loss, fx["loss"], 5e-5, "codebook loss")
    assert_close(m.decode_indices(idx), fx["dec"], 5e-5, "decode_indices")
    total = torch.nn.functional.mse_loss(rec, imgs) + torch.nn.functional.l1_loss(rec, imgs) + loss
    params = dict(m.named_parameters())
    names = sorted(params)
    gs = torch.autograd.grad(total, [params[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        if "g:" + n in fx:
            assert_close(g, fx["g:" + n], 2e-4, f"grad {n}")


def test_vqgan_codebook_golden(device):
    """Conv-VQGAN codebook (models/vqgan.py:138-182; C = 256, channels-first, beta on the codebook term)
    against the reference's own outputs."""
    from amk.models.vqgan import Codebook

    fx = load_golden("vqgan_codebook")
    K, C = fx["E"].shape
    cb = Codebook(K, C, beta=float(fx["beta"]))
    cb.load_state_dict({"embedding.weight": torch.from_numpy(fx["E"])})
    cb = cb.to(device)
    z = torch.from_numpy(fx["z"]).to(device).requires_grad_(True)
    cot = torch.from_numpy(fx["cot"]).to(device)
    zq, idx, loss = cb(z)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == tuple(fx["idx"].shape)
    assert _check_indices(idx, fx["idx"], fx["margin"]) == 0
    assert_close(zq, fx["zq"], TOL, "z_q")
    assert_close(loss, fx["loss"], TOL, "loss")
    gz, gE = torch.autograd.grad((zq * cot).sum() + float(fx["loss_weight"]) * loss, [z, cb.embedding.weight])
    assert_close(gz, fx["gz"], TOL, "grad z")
    assert_close(gE, fx["gE"], TOL, "grad codebook")
    emb = cb.indices_to_embeddings(idx.view(z.shape[0], -1))
    assert_close(emb, fx["emb"], 0.0, "indices_to_embeddings")


@pytest.mark.parametrize("N,K,C", [(4096, 64, 32), (1000, 2048, 32), (513, 100, 256)])
def test_reproducible_codebook_gradient(device, N, K, C):
    """Under torch.use_deterministic_algorithms the codebook gradient -- a scatter-add of N rows into K codes, f32
    atomics by default -- is built from per-row contributions added in a fixed order: equal to the oracle within
    tolerance and bit-identical from call to call (few codes, many rows per code: the case atomics reorder)."""
    from amk import ops

    z = seeded((N, C), 1100 + N)
    E = seeded((K, C), 2100 + K)
    cot = seeded((N, C), 3100 + N)
    zc, Ec = z.clone().requires_grad_(True), E.clone().requires_grad_(True)
    zq_r, _, loss_r = ref_cpu.codebook_forward(zc, Ec, 0.25)
    _, gE_r = torch.autograd.grad((zq_r * cot).sum() + 2.0 * loss_r, [zc, Ec])

    def grads():
        zd, Ed = z.to(device).requires_grad_(True), E.to(device).requires_grad_(True)
        zq, _, loss = ops.vq_lookup(zd, Ed, 0.25)
        return torch.autograd.grad((zq * cot.to(device)).sum() + 2.0 * loss, [zd, Ed])

    torch.use_deterministic_algorithms(True)
    try:
        g1, g2 = grads(), grads()
    finally:
        torch.use_deterministic_algorithms(False)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert_close(g1[1], gE_r, TOL, "grad codebook (ordered)")
    # the ordered path must leave torch's global switches as it found them (warn-only mode stays warn-only)
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        g3 = grads()
        assert torch.are_deterministic_algorithms_enabled() and torch.is_deterministic_algorithms_warn_only_enabled()
    finally:
        torch.use_deterministic_algorithms(False)
    assert torch.equal(g3[1], g1[1])
