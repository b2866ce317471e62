"""Seeded random shape / mask sweeps of every HIP path against the CPU oracle: ragged lengths
(1 .. a few tiles, not multiples of any tile size), random key masks, both attention backward
paths, random expert counts.  Deterministic (fixed seeds) so a failure names its case."""
import random

import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded, seeded_params
from util import assert_close

pytestmark = pytest.mark.gpu
TOL = 3e-5


def _attn_cases(n=24):
    rng = random.Random(20261004)
    out = []
    for i in range(n):
        B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 5])
        I = rng.choice([1, 2, 31, 33, 63, 64, 65, 97, 127, 129, 191, 257])
        J = rng.choice([1, 3, 32, 47, 64, 66, 100, 128, 130, 193, 300])
        mask = rng.choice(["none", "none", "key", "causal", "both"])
        if mask in ("causal", "both"):
            J = I
        out.append((i, B, H, I, J, mask, rng.choice([False, True])))
    return out


@pytest.mark.parametrize("case", _attn_cases(), ids=lambda c: f"{c[0]}-B{c[1]}H{c[2]}I{c[3]}J{c[4]}-{c[5]}-{'det' if c[6] else 'fused'}")
def test_attention_random(device, case):
    from amk import ops

    i, B, H, I, J, mask, det = case
    D = 64
    g = torch.Generator().manual_seed(1000 + i)
    q, k, v = (torch.randn(B, H, T, D, generator=g) for T in (I, J, J))
    cot = torch.randn(B, H, I, D, generator=g)
    km = cm = None
    if mask in ("key", "both"):
        km = torch.rand(B, J, generator=g) > 0.3
    if mask in ("causal", "both"):
        cm = torch.ones(I, J).triu(1).bool()
        if I > 4:
            cm[I // 2, :] = True  # one fully masked row
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref = ref_cpu.attention_core(qc, kc, vc, D ** -0.5, km, cm)
    g_ref = torch.autograd.grad((o_ref * cot).sum(), [qc, kc, vc])
    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = det
    try:
        qd, kd, vd = (t.to(device).requires_grad_(True) for t in (q, k, v))
        o = ops.attention(qd, kd, vd, D ** -0.5, key_mask=None if km is None else km.to(device),
                          causal_mask=None if cm is None else cm.to(device))
        gs = torch.autograd.grad((o * cot.to(device)).sum(), [qd, kd, vd])
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old
    assert_close(o, o_ref, TOL, "o")
    for name, a, b in zip(("dq", "dk", "dv"), gs, g_ref):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a.cpu() - b).abs().max()) <= TOL * max(scale, float(g_ref[2].abs().max()) * 1e-2), name


def _vq_cases(n=10):
    rng = random.Random(7)
    return [(i, rng.choice([1, 5, 127, 128, 129, 1000, 2500]), rng.choice([32, 96, 256, 1024, 4096]), rng.choice([32, 64]))
            for i in range(n)]


@pytest.mark.parametrize("case", _vq_cases(), ids=lambda c: f"{c[0]}-N{c[1]}K{c[2]}C{c[3]}")
def test_vq_random(device, case):
    from amk import ops

    i, N, K, C = case
    z, E = seeded((N, C), 500 + i), seeded((K, C), 600 + i)
    zq_r, idx_r, loss_r = ref_cpu.codebook_forward(z, E, 0.25)
    margin = ref_cpu.codebook_margin(z, E)
    zq, idx, loss = ops.vq_lookup(z.to(device), E.to(device), 0.25)
    bad = idx.cpu() != idx_r
    assert not bool((bad & (margin > 1e-6)).any()), f"{int(bad.sum())} index mismatches away from ties"
    assert_close(loss, loss_r, TOL, "loss")
    if not bool(bad.any()):
        assert_close(zq, zq_r, TOL, "z_q")


def _moe_cases(n=8):
    rng = random.Random(11)
    return [(i, rng.choice([1, 2, 3]), rng.choice([1, 7, 65, 130]), rng.choice([64, 100, 256]), rng.choice([2, 3, 6, 17]), rng.choice([1, 2]))
            for i in range(n)]


@pytest.mark.parametrize("case", _moe_cases(), ids=lambda c: f"{c[0]}-B{c[1]}T{c[2]}D{c[3]}E{c[4]}k{c[5]}")
def test_moe_random(device, case):
    from amk.models import MoELayer

    i, B, T, D, E, k = case
    shapes = {"gate.weight": (E, D), "gate.bias": (E,)}
    for e in range(E):
        shapes[f"experts.{e}.weight"] = (D, D)
        shapes[f"experts.{e}.bias"] = (D,)
    w = seeded_params(shapes, 700 + i)
    x, cot = seeded((B, T, D), 800 + i), seeded((B, T, D), 900 + i)
    xr = x.clone().requires_grad_(True)
    out_r, sel_r = ref_cpu.moe_layer(xr, w, E, k)
    (gx_r,) = torch.autograd.grad((out_r * cot).sum(), [xr])
    m = MoELayer(D, D, E, k)
    m.load_state_dict(w)
    m = m.to(device)
    xd = x.to(device).requires_grad_(True)
    out = m(xd)
    assert torch.equal(m.last_selected_experts.cpu(), sel_r)
    assert_close(out, out_r, TOL, "out")
    (gx,) = torch.autograd.grad((out * cot.to(device)).sum(), [xd])
    assert_close(gx, gx_r, TOL, "grad x")
