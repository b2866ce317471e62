"""Random SwitchHeadAttention configurations: the distinct-row / dense-sum form against the per-pair routed form
(ops.MOE_DENSE_Z on / off) -- outputs, selections and all gradients must agree to f32 rounding."""
import os, sys, random
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch
from amk import ops
from amk.models import SwitchHeadAttention

dev = torch.device("cuda:0")
rng = random.Random(7)
worst = 0.0
n_distinct = 0
for it in range(60):
    dim = rng.choice([256, 260, 320, 512, 1024])
    h = rng.choice([1, 2, 3, 4, 8])
    k = rng.choice([1, 2, 3, 4])
    E = rng.choice([k, k + 1, 5, 8, 16, 31, 32, 64])
    E = max(E, k)
    B, T = rng.choice([1, 2, 5]), rng.choice([1, 7, 33, 65, 130])
    masked = rng.random() < 0.3
    torch.manual_seed(it)
    m = SwitchHeadAttention(dim, h, 64, num_experts=E, sel_experts=k).to(dev)
    x = torch.randn(B, T, dim, device=dev)
    cot = torch.randn(B, T, dim, device=dev)
    km = (torch.rand(B, T, device=dev) > 0.2) if masked else None
    if km is not None:
        km[:, 0] = True
    res = []
    for dz in (True, False):
        ops.MOE_DENSE_Z = dz
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        out = m(xi, context_mask=km)
        (out * cot).sum().backward()
        res.append((out.detach(), xi.grad, [p.grad for p in m.parameters()], m.last_selected_v.clone(), m.last_selected_out.clone()))
    ops.MOE_DENSE_Z = True
    used = ops._moe_dense_z(dim, 64, h * k, E) and E <= 64
    n_distinct += used
    (o1, g1, p1, sv1, so1), (o2, g2, p2, sv2, so2) = res
    assert torch.equal(sv1, sv2) and torch.equal(so1, so2)
    def rel(a, b):
        if a is None and b is None: return 0.0
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))
    errs = [rel(o1, o2), rel(g1, g2)] + [rel(a, b) for a, b in zip(p1, p2)]
    worst = max(worst, max(errs))
    assert max(errs) < 2e-5, (it, dim, h, k, E, B, T, errs)
    assert all(torch.isfinite(t).all() for t in (o1, g1))
print(f"60 configurations ({n_distinct} on the distinct-row form): worst relative difference {worst:.2e}")
