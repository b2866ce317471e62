"""AgentAttention kernels (through the C ABI) against the golden vectors and the CPU oracle."""
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded, seeded_params
from util import assert_close, load_golden, weights_of

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _abs_close(a, b, tol, what):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    scale = max(float(b.abs().max()), 1e-3)
    err = float((a - b).abs().max())
    assert err <= tol * scale * 5, f"{what}: abs err {err:.3e} (scale {scale:.3e})"


def test_agent_small_golden(device):
    from amk.models import AgentAttention

    fx = load_golden("agent_small")
    dim, h, d, agent_num = (int(v) for v in fx["dims"])
    m = AgentAttention(dim, h, d, agent_num=agent_num)
    res = m.load_state_dict(weights_of(fx), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.to(device)
    x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
    out = m(x)
    assert_close(out, fx["out"], TOL, "out")
    (out * torch.from_numpy(fx["cot"]).to(device)).sum().backward()
    assert_close(x.grad, fx["gx"], TOL, "grad x")
    for n, p in m.named_parameters():
        if n in ("bias1", "bias2"):
            # softmax rows are shift invariant: exact gradient 0 (the reference's autograd leaves ~1e-9 noise)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            assert float(abs(fx["g:" + n]).max()) < 1e-6
            continue
        _abs_close(p.grad, fx["g:" + n], TOL, f"grad {n}")


@pytest.mark.parametrize("B,T,dim,h,agent_num", [(2, 10, 384, 6, 47), (1, 1024, 384, 6, 47), (2, 65, 256, 4, 16), (1, 300, 128, 2, 4), (1, 37, 64, 1, 1),
                                                   (1, 513, 128, 2, 4), (2, 256, 192, 3, 9),
                                                   # 7 and 8 agents (the streaming backward's widest instantiation), 9 (the
                                                   # LDS-staged kernels for more than 8 agents)
                                                   (1, 200, 448, 7, 49), (2, 129, 512, 8, 64), (1, 140, 576, 9, 81)])
def test_agent_vs_oracle(device, B, T, dim, h, agent_num):
    """README.md:116-127 shape (2,10,384) h=6, the BASELINE.md T=1024 case, ragged bins (T % p != 0),
    a last chunk of one token (513 = 2*256 + 1) and an exactly full chunk."""
    from amk.models import AgentAttention

    d = 64
    m = AgentAttention(dim, h, d, agent_num=agent_num)
    shapes = {n: tuple(p.shape) for n, p in m.named_parameters()}
    w = seeded_params(shapes, 80 + h)
    m.load_state_dict(w, strict=True)
    x = seeded((B, T, dim), 81 + T)
    cot = seeded((B, T, dim), 82 + T)
    wr = {n: v.clone().requires_grad_(True) for n, v in w.items()}
    xr = x.clone().requires_grad_(True)
    out_r = ref_cpu.agent_attention(xr, wr, h, d, agent_num)
    names = sorted(wr)
    g_r = torch.autograd.grad((out_r * cot).sum(), [xr] + [wr[n] for n in names], allow_unused=True)

    m = m.to(device)
    xd = x.to(device).requires_grad_(True)
    out = m(xd)
    assert tuple(out.shape) == (B, T, dim)
    assert_close(out, out_r, TOL, "out")
    (out * cot.to(device)).sum().backward()
    assert_close(xd.grad, g_r[0], TOL, "grad x")
    params = dict(m.named_parameters())
    for n, g in zip(names, g_r[1:]):
        if n in ("bias1", "bias2"):
            continue
        _abs_close(params[n].grad, g, TOL, f"grad {n}")


def test_agent_full_size_properties(device):
    """B 64, T 1024, h 6 (the chip-filling case, 3072 workgroups per kernel): (1) one batch element
    equals the same element run alone (chunks / workgroups do not leak into each other);
    (2) forward and backward are bitwise reproducible (partials are folded in chunk order, no atomics);
    (3) one element against the CPU oracle."""
    from amk.models import AgentAttention

    torch.manual_seed(0)
    m = AgentAttention(384, 6, 64).to(device)
    x = torch.randn(64, 1024, 384, device=device, requires_grad=True)
    cot = torch.randn(64, 1024, 384, device=device)
    out = m(x)
    (g,) = torch.autograd.grad((out * cot).sum(), [x])
    out2 = m(x)
    (g2,) = torch.autograd.grad((out2 * cot).sum(), [x])
    assert torch.equal(out, out2) and torch.equal(g, g2)
    xs = x[17:18].detach().clone().requires_grad_(True)
    outs = m(xs)
    (gs,) = torch.autograd.grad((outs * cot[17:18]).sum(), [xs])
    assert_close(outs, out[17:18], 1e-6, "element alone vs in batch")
    assert_close(gs, g[17:18], 1e-6, "grad: element alone vs in batch")
    w = {n: p.detach().cpu() for n, p in m.named_parameters()}
    xr = x[17:18].detach().cpu().requires_grad_(True)
    out_r = ref_cpu.agent_attention(xr, w, 6, 64, 47)
    (g_r,) = torch.autograd.grad((out_r * cot[17:18].cpu()).sum(), [xr])
    assert_close(out[17:18], out_r, TOL, "vs oracle")
    assert_close(g[17:18], g_r, TOL, "grad vs oracle")
