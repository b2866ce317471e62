"""The benchmarked step itself against the oracle: ONE VQGANTrainStep.step (amk/train.py -- both phases,
gradient penalty with a fixed eta, global-norm clip 1.0, Adam) on the GPU against oracle/train_step_cpu._step
(the CPU restatement of trainers/vitgqgan.py:139-189) from the same weights, images and eta.
Compared: the logged losses, the gradients as Adam saw them (its first-moment state = 0.1 * clipped
gradient after one step; tolerance 1e-4 relative, the north star's) and the updated generator +
discriminator parameters.  Adam's first update is -lr * g / (|g| + eps): +-lr wherever |g| >> eps, whatever
the size of g -- so an element whose gradient is zero to rounding (|g| ~ 1e-7 of the largest) may move the
other way on the two sides.  The parameters are therefore compared as updates: no element further apart
than such a flip (2 lr), and all but a 1e-3 fraction within 5 % of lr.  The attention backward runs on its
bitwise-reproducible path."""
import copy

import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_optimizer", [True, False])
def test_one_train_step_matches_the_oracle_step(device, fused_optimizer):
    from amk import ops
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep
    from oracle import train_step_cpu

    cfg = dict(dim=64, img_size=32, patch_size=4, n_heads=2, d_head=64, depth=2, mlp_dim=96, dropout=0.0)
    torch.manual_seed(0)
    model = ViTVQGAN(cfg, dict(codebook_size=256, codebook_dim=32))
    discr = NLayerDiscriminator(3, 8, 3)
    lr = 1e-3
    # ---- oracle side (CPU)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = sorted(w)
    for n in names:
        w[n].requires_grad_(True)
    discr_cpu = copy.deepcopy(discr)
    g_opt = torch.optim.Adam([w[n] for n in names], lr=lr)
    d_opt = torch.optim.Adam(discr_cpu.parameters(), lr=lr)
    g = torch.Generator().manual_seed(7)
    imgs = torch.rand(4, 3, 32, 32, generator=g)
    eta = torch.rand(4, 1, 1, 1, generator=g)
    want = train_step_cpu._step(w, names, cfg, discr_cpu, g_opt, d_opt, imgs, eta=eta)
    # ---- product side (GPU); warmup_steps=0: step 0 runs at the base learning rate
    model, discr = model.to(device), discr.to(device)
    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        trainer = VQGANTrainStep(model, discr, lr=lr, warmup_steps=0, decay_steps=100000, fused_optimizer=fused_optimizer)
        logs = trainer.step(imgs.to(device), eta=eta.to(device))
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old
    for k in ("d_loss", "g_loss", "l1", "l2", "codebook_loss", "loss"):
        assert abs(float(logs[k]) - want[k]) <= 1e-4 * max(1.0, abs(want[k])), (k, float(logs[k]), want[k])

    def first_moment(opt, p):
        return opt.state_of(p)["exp_avg"] if fused_optimizer else opt.state[p]["exp_avg"]

    got_g = dict(model.named_parameters())
    pairs = [("generator " + n, got_g[n], w[n], trainer.g_optim, g_opt) for n in names if n in got_g]  # not buffers
    pairs += [("discriminator " + n, p, q, trainer.d_optim, d_opt)
              for (n, p), q in zip(discr.named_parameters(), discr_cpu.parameters())]
    n_bad = n_all = 0
    for what, p, q, opt, ropt in pairs:
        if q not in ropt.state:   # no gradient on the oracle side either (none in this model)
            continue
        assert_close(first_moment(opt, p), ropt.state[q]["exp_avg"], 1e-4, what + " (gradient)")
        d = (p.detach().cpu() - q.detach()).abs()
        assert float(d.max()) <= 2.001 * lr, what
        n_bad += int((d > 0.05 * lr).sum())
        n_all += d.numel()
    assert n_all > 100000 and n_bad <= 1e-3 * n_all, (n_bad, n_all)


def test_direct_gradient_writes_match_accumulated_gradients(device):
    """GradReducer(direct_grads=True): the Linear layers' weight / bias gradients written straight into the bucket views
    equal the gradients autograd accumulates (same kernels, one element-wise add less), also over two accumulated
    micro-steps (the second contribution goes through autograd), and every parameter is marked as fired."""
    import copy

    from amk.dp import GradReducer
    from amk.models import ViTVQGAN

    vit = dict(dim=128, img_size=32, patch_size=8, n_heads=2, d_head=64, depth=1, mlp_dim=768, dropout=0.0)
    torch.manual_seed(0)
    a = ViTVQGAN(vit, dict(codebook_size=64, codebook_dim=32)).to(device)
    b = copy.deepcopy(a)
    ra, rb = GradReducer(a.parameters(), direct_grads=True), GradReducer(b.parameters(), direct_grads=False)
    if not ra.direct_grads:
        pytest.skip("AMK_DIRECT_GRADS=0 in the environment")
    assert not rb.direct_grads
    imgs = [torch.rand(4, 3, 32, 32, device=device) for _ in range(2)]
    for micro, img in enumerate(imgs):
        for net, red in ((a, ra), (b, rb)):
            red.begin(sync=micro == 1)
            rec, _ = net(img)
            ((rec - img) ** 2).mean().backward()
            red.finish(detach_unused=False)
        wrote = sum(sum(bk.direct) for bk in ra.buckets)
        assert (wrote > 0) == (micro == 0)      # first gradients are written in place, later ones accumulate through autograd
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert p.grad is not None and q.grad is not None, n
        assert float((p.grad - q.grad).abs().max()) <= 1e-6 * max(1.0, float(q.grad.abs().max())), n
    assert [f for bk in ra.buckets for f in bk.fired] == [f for bk in rb.buckets for f in bk.fired]
