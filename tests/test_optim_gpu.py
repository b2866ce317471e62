"""amk.optim.FlatAdam (csrc/optim.hip: global-norm clip + Adam / AdamW + gradient zeroing over the flat
buckets of GradReducer) against torch.optim.Adam / AdamW + clip_grad_norm_ on the same gradients.
Tolerance 1e-6 relative on parameters and moments after several steps (same arithmetic, fp32)."""
import copy

import pytest
import torch
import torch.nn as nn

from util import assert_close

pytestmark = pytest.mark.gpu


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(37, 53)          # odd sizes: segments end inside the 1-KiB alignment padding
        self.b = nn.Linear(53, 300)
        self.c = nn.Linear(300, 7, bias=False)
        self.unused = nn.Linear(5, 5)       # never reached: no gradient in any step
        self.big = nn.Parameter(torch.randn(70001))

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.relu(self.a(x))))) + self.big[:7] * self.big[7:14].sum()


@pytest.mark.parametrize("decoupled,wd", [(False, 0.0), (False, 0.05), (True, 0.05)])
@pytest.mark.parametrize("max_norm", [None, 0.5])
def test_flat_adam_matches_torch(device, decoupled, wd, max_norm):
    from amk.dp import GradReducer
    from amk.optim import FlatAdam

    torch.manual_seed(0)
    net = Net().to(device)
    ref = copy.deepcopy(net)
    red = GradReducer(net.parameters(), bucket_bytes=64 << 10)   # several buckets
    assert len(red.buckets) > 1
    opt = FlatAdam(red, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, decoupled=decoupled)
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    ropt = cls(ref.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        x = torch.randn(16, 37, generator=g).to(device) * (1.0 + step)
        red.begin(True)
        net(x).pow(2).mean().backward()
        red.finish(detach_unused=False)
        ref(x).pow(2).mean().backward()
        want_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm if max_norm else float("inf"))
        ropt.step()
        ropt.zero_grad()
        lr = 3e-3 * (0.5 if step == 2 else 1.0)   # a schedule changes lr between steps
        for grp in ropt.param_groups:
            grp["lr"] = 3e-3 * (0.5 if step + 1 == 2 else 1.0)
        norm = opt.step(max_norm=max_norm, lr=lr)
        assert_close(norm, want_norm, 2e-6, "global gradient norm")
        for b in red.buckets:
            assert float(b.flat.abs().max()) == 0.0                      # gradients left zeroed
        for p, v in zip([p for b in red.buckets for p in b.params], [v for b in red.buckets for v in b.views]):
            assert p.grad is v                                            # and re-attached to their bucket
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert_close(p, q, 1e-6, n)
        st = opt.state_of(p)
        if n.startswith("unused"):
            assert st["step"] == 0 and torch.equal(p, q)                 # skipped like .grad None, wd included
            continue
        rs = ropt.state[q]
        assert st["step"] == int(rs["step"])
        assert_close(st["exp_avg"], rs["exp_avg"], 2e-6, n + " exp_avg")
        assert_close(st["exp_avg_sq"], rs["exp_avg_sq"], 2e-6, n + " exp_avg_sq")


def test_flat_adam_bf16_shadow(device):
    """bf16_shadow=True: after every step ``p._amk_bf16`` is exactly ``p.to(bfloat16)`` (parameters without a gradient
    included: they keep their initial copy), and ops.linear under autocast reads it."""
    from amk import ops
    from amk.dp import GradReducer
    from amk.optim import FlatAdam

    torch.manual_seed(0)
    net = Net().to(device)
    red = GradReducer(net.parameters(), bucket_bytes=64 << 10)
    opt = FlatAdam(red, lr=3e-3, bf16_shadow=True)
    g = torch.Generator().manual_seed(1)
    for step in range(3):
        for p in net.parameters():
            assert p._amk_bf16.dtype == torch.bfloat16 and p._amk_bf16.shape == p.shape
            assert torch.equal(p._amk_bf16, p.detach().to(torch.bfloat16))
        x = torch.randn(16, 37, generator=g).to(device)
        red.begin(True)
        net(x).pow(2).mean().backward()
        red.finish(detach_unused=False)
        opt.step(max_norm=1.0)
    assert ops._w16(net.b.weight) is net.b.weight._amk_bf16
    with torch.no_grad():
        net.b.weight.mul_(2.0)
    # written behind the optimizer's back: the stale copy is not used (the in-place write bumped the tensor's version)
    assert ops._w16(net.b.weight) is not net.b.weight._amk_bf16
    assert torch.equal(ops._w16(net.b.weight), net.b.weight.detach().to(torch.bfloat16))
    opt.refresh_shadow()
    assert ops._w16(net.b.weight) is net.b.weight._amk_bf16
    assert torch.equal(net.b.weight._amk_bf16, net.b.weight.detach().to(torch.bfloat16))
