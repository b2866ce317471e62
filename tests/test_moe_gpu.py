"""Routing + grouped expert GEMM kernels (through the C ABI) against the golden vectors and the
CPU oracle: MoELayer and SwitchHeadAttention.  Expert ids must be bit-exact; values within 2e-5.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded, seeded_params
from util import assert_close, load_golden, weights_of

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _stacked_to_ref(module):
    """state_dict of an amk module (per-expert keys, as the reference names them) on the CPU."""
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def test_moe_small_golden(device):
    from amk.models import MoELayer

    fx = load_golden("moe_small")
    D, E, k = (int(v) for v in fx["dims"])
    m = MoELayer(D, D, E, k)
    m.load_state_dict(weights_of(fx), strict=True)
    m = m.to(device)
    x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
    out = m(x)
    assert tuple(out.shape) == tuple(fx["out"].shape)
    assert np.array_equal(m.last_selected_experts.cpu().numpy(), fx["sel"])  # bit-exact expert ids
    assert_close(out, fx["out"], TOL, "out")
    cot = torch.from_numpy(fx["cot"]).to(device)
    (out * cot).sum().backward()
    assert_close(x.grad, fx["gx"], TOL, "grad x")
    assert_close(m.gate.weight.grad, fx["g:gate.weight"], TOL, "grad gate.weight")
    assert_close(m.gate.bias.grad, fx["g:gate.bias"], TOL, "grad gate.bias")
    for e in range(E):
        for leaf, grad in (("weight", m.experts_weight.grad[e]), ("bias", m.experts_bias.grad[e])):
            key = f"g:experts.{e}.{leaf}"
            want = fx[key] if key in fx else np.zeros_like(grad.cpu().numpy())
            assert float((grad.cpu() - torch.from_numpy(want)).abs().max()) <= TOL * max(1.0, float(np.abs(want).max())), key


@pytest.mark.parametrize("B,T,D,E,k", [(2, 65, 1024, 32, 2), (3, 17, 128, 8, 3), (1, 1, 64, 4, 1), (2, 200, 256, 5, 2)])
def test_moe_vs_oracle(device, B, T, D, E, k):
    """Includes BASELINE.json configs[3] dims (D 1024, E 32, top-2, T 65)."""
    from amk.models import MoELayer

    shapes = {"gate.weight": (E, D), "gate.bias": (E,)}
    for e in range(E):
        shapes[f"experts.{e}.weight"] = (D, D)
        shapes[f"experts.{e}.bias"] = (D,)
    w = seeded_params(shapes, 60 + E)
    x = seeded((B, T, D), 61 + T)
    cot = seeded((B, T, D), 62 + T)
    wr = {n: v.clone().requires_grad_(True) for n, v in w.items()}
    xr = x.clone().requires_grad_(True)
    out_r, sel_r = ref_cpu.moe_layer(xr, wr, E, k)
    names = sorted(wr)
    g_r = torch.autograd.grad((out_r * cot).sum(), [xr] + [wr[n] for n in names], allow_unused=True)

    m = MoELayer(D, D, E, k)
    m.load_state_dict(w, strict=True)
    m = m.to(device)
    xd = x.to(device).requires_grad_(True)
    out = m(xd)
    # expert ids: bit-exact wherever the k-th and (k+1)-th logits are not within rounding
    logits = x @ w["gate.weight"].t() + w["gate.bias"]
    srt = torch.sort(logits, dim=-1, descending=True).values
    gap = (srt[..., :-1] - srt[..., 1:])[..., :k].min(-1).values if E > k else torch.ones(B, T)
    sel = m.last_selected_experts.cpu()
    bad = (sel != sel_r).any(-1)
    assert not bool((bad & (gap > 1e-5)).any()), "expert ids differ away from logit ties"
    assert int(bad.sum()) == 0
    assert_close(out, out_r, TOL, "out")
    (out * cot.to(device)).sum().backward()
    assert_close(xd.grad, g_r[0], TOL, "grad x")
    got = _grads_by_ref_name(m)
    for n, g in zip(names, g_r[1:]):
        want = g if g is not None else torch.zeros_like(w[n])
        assert_close_abs(got[n], want, TOL, n)


def _grads_by_ref_name(module):
    out = {}
    for name, p in module.named_parameters():
        stacked = getattr(module, "_stacked", {})
        if name in stacked:
            mod, leaf = stacked[name]
            for e in range(p.shape[0]):
                out[f"{mod}.{e}.{leaf}"] = p.grad[e] if p.grad is not None else None
        else:
            out[name] = p.grad
    return out


def assert_close_abs(a, b, tol, what):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-3)
    err = float((a - b).abs().max())
    assert err <= tol * scale * 5, f"{what}: abs err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("variant", ["self", "self_keymask"])
def test_switchhead_small_golden(device, variant):
    from amk.models import SwitchHeadAttention

    fx = load_golden("switchhead_small")
    dim, h, d, E, k = (int(v) for v in fx["dims"])
    m = SwitchHeadAttention(dim, h, d, num_experts=E, sel_experts=k)
    m.load_state_dict(weights_of(fx), strict=True)
    m = m.to(device)
    x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
    kw = {} if variant == "self" else dict(context_mask=torch.from_numpy(fx["keymask"]).to(device))
    out = m(x, **kw)
    assert tuple(out.shape) == tuple(fx[f"{variant}:out"].shape)
    assert np.array_equal(m.last_selected_v.cpu().numpy(), fx["sel_v"])
    assert np.array_equal(m.last_selected_out.cpu().numpy(), fx["sel_o"])
    assert_close(out, fx[f"{variant}:out"], TOL, "out")
    (out * torch.from_numpy(fx["cot"]).to(device)).sum().backward()
    assert_close(x.grad, fx[f"{variant}:gx"], TOL, "grad x")
    got = _grads_by_ref_name(m)
    assert got["W_d.0.weight"] is None  # no gradient, exactly as the reference
    for n, g in got.items():
        key = f"{variant}:g:{n}"
        if key in fx:
            assert_close_abs(g, torch.from_numpy(fx[key]), TOL, n)


@pytest.mark.parametrize("dense_z", [True, False], ids=["dense_z", "routed"])
@pytest.mark.parametrize("B,T,dim,h,E,k", [(2, 65, 1024, 8, 32, 2), (2, 65, 1024, 8, 5, 2), (1, 10, 512, 2, 5, 2)])
def test_switchhead_vs_oracle(device, monkeypatch, B, T, dim, h, E, k, dense_z):
    """ViTMoE layer dims (E 32 through the factory, E 5 standalone default) and README.md:133-142; the head / slot sums
    as one dense product over per-expert sums (the default at these sizes) and as routed GEMM + ordered combine."""
    from amk import ops
    from amk.models import SwitchHeadAttention

    monkeypatch.setattr(ops, "MOE_DENSE_Z", dense_z)
    assert ops._moe_dense_z(dim, 64, h * k, E) == dense_z
    d = 64
    shapes = {"q.0.weight": (h * d, dim), "k.0.weight": (h * d, dim), "W_s.0.weight": (h * E, dim), "W_d.0.weight": (h * E, dim)}
    for e in range(E):
        shapes[f"experts_v.{e}.weight"] = (d, dim)
        shapes[f"experts_out.{e}.weight"] = (dim, d)
    w = seeded_params(shapes, 70 + E)
    x = seeded((B, T, dim), 71 + T)
    cot = seeded((B, T, dim), 72 + T)
    wr = {n: v.clone().requires_grad_(True) for n, v in w.items()}
    xr = x.clone().requires_grad_(True)
    out_r, sel_v, sel_o = ref_cpu.switchhead_attention(xr, wr, h, d, E, k)
    names = sorted(wr)
    g_r = torch.autograd.grad((out_r * cot).sum(), [xr] + [wr[n] for n in names], allow_unused=True)

    m = SwitchHeadAttention(dim, h, d, num_experts=E, sel_experts=k)
    m.load_state_dict(w, strict=True)
    m = m.to(device)
    xd = x.to(device).requires_grad_(True)
    out = m(xd)
    assert tuple(out.shape) == (B, T, dim)
    assert torch.equal(m.last_selected_v.cpu(), sel_v)
    assert torch.equal(m.last_selected_out.cpu(), sel_o)
    assert_close(out, out_r, TOL, "out")
    (out * cot.to(device)).sum().backward()
    assert_close(xd.grad, g_r[0], TOL, "grad x")
    got = _grads_by_ref_name(m)
    for n, g in zip(names, g_r[1:]):
        if g is None:
            assert got[n] is None, n
        else:
            assert_close_abs(got[n], g, TOL, n)


@pytest.mark.parametrize("U,E,k", [(64 * 65 * 8, 32, 2), (3000, 1024, 8), (1500, 100, 3), (70, 5, 1), (5000, 64, 4)])
def test_routing_properties_full_size(device, U, E, k):
    """At U = 64*65*8 units (ViTMoE batch 64) and at the limits of the routing kernels (1024 experts, top-8; partial
    blocks): perm is a permutation, grouped by expert with ascending pair index inside each expert, offsets match a
    histogram of ids, gates = sigmoid."""
    from amk import ops

    g = torch.Generator().manual_seed(3)
    logits = torch.randn(U, E, generator=g).to(device)
    r = ops.moe_route(logits, k)
    ids, gate, offsets, perm = r["ids"].cpu(), r["gate"].cpu(), r["offsets"].cpu().long(), r["perm"].cpu().long()
    vals, ref_ids = torch.topk(logits.cpu(), k)
    assert torch.equal(ids, ref_ids)
    assert_close(gate, torch.sigmoid(vals), 1e-6, "gate")
    assert torch.equal(torch.sort(perm).values, torch.arange(U * k))
    hist = torch.bincount(ids.view(-1), minlength=E)
    assert torch.equal(offsets[1:] - offsets[:-1], hist)
    flat = ids.view(-1)
    for e in range(E):
        seg = perm[offsets[e]:offsets[e + 1]]
        assert bool((flat[seg] == e).all())
        assert bool((seg[1:] > seg[:-1]).all())


@pytest.mark.parametrize("G,fan,a_div,E,d,weighted", [(37, 16, 2, 32, 64, False), (5, 6, 3, 4, 8, True), (1, 1, 1, 1, 4, True),
                                                       (130, 300, 1, 7, 20, True)])
def test_expert_sums(device, G, fan, a_div, E, d, weighted):
    """amk_moe_expert_sums against an index_add over the pairs (float64)."""
    from amk import ops

    g = torch.Generator().manual_seed(G * 31 + fan)
    P = G * fan
    rows = (P + a_div - 1) // a_div
    A = torch.randn(rows, d + 4, generator=g).to(device)[:, :d]   # rows at a stride larger than d
    ids = torch.randint(0, E, (P,), generator=g).to(device)
    scale = torch.rand(P, generator=g).to(device) if weighted else None
    Z = ops._expert_sums(A, a_div, ids, ops._ptr(scale) if weighted else ops._NULL, G, fan, E, d)
    p = torch.arange(P, device=device)
    src = A.double()[p // a_div] * (scale.double()[:, None] if weighted else 1.0)
    ref = torch.zeros(G * E, d, dtype=torch.float64, device=device).index_add_(0, (p // fan) * E + ids, src)
    assert_close(Z.view(G * E, d), ref.float(), 2e-6, "expert sums")


@pytest.mark.parametrize("G,fan,E", [(4160, 16, 32), (37, 6, 64), (1, 1, 1), (1500, 300, 7), (2049, 2, 5)])
def test_distinct_lists(device, G, fan, E):
    """amk_moe_route_distinct: per expert the groups (ascending) in which some pair chose it, as virtual pairs g*E + e."""
    from amk import ops

    g = torch.Generator().manual_seed(G + fan + E)
    ids = torch.randint(0, E, (G * fan,), generator=g)
    offsets, perm = ops._route_distinct(ids.to(device), G, fan, E)
    offsets, perm = offsets.cpu().tolist(), perm.cpu().tolist()
    present = torch.zeros(G, E, dtype=torch.bool)
    present[torch.arange(G * fan) // fan, ids] = True
    assert offsets[0] == 0 and offsets[E] == int(present.sum())
    for e in range(E):
        want = [int(t) * E + e for t in torch.nonzero(present[:, e]).flatten()]
        assert perm[offsets[e]:offsets[e + 1]] == want, e


def test_combine_rows_reads_per_group_expert_rows(device):
    """amk_moe_combine_rows with v_div: pair p reads row (p / v_div) * E + ids[p]."""
    import ctypes

    from amk import lib as amk_lib
    from amk import ops

    G, H, k, E, N = 9, 4, 2, 8, 12
    U, fan = G * H, H * k
    g = torch.Generator().manual_seed(5)
    V = torch.randn(G * E, N, generator=g).to(device)
    logits = torch.randn(U, E, generator=g).to(device)
    ids, gate = ops._topk(logits, k)
    out = torch.empty(U, N, device=device)
    L = amk_lib.load()
    amk_lib.check(L.amk_moe_combine_rows(ops._ptr(V), ops._ptr(ids), ops._ptr(gate), U, 1, k, N, fan, E, ops._ptr(out),
                                         ops._stream()), "amk_moe_combine_rows")
    tok = (torch.arange(U * k, device=device) // fan).view(U, k)
    ref = (V.double()[tok * E + ids] * gate.double()[..., None]).sum(1)
    assert_close(out, ref.float(), 1e-6, "combine rows")
