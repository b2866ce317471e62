"""Host-side logic that needs no GPU: checkpoint-key compatibility of every drop-in class with the
reference's state_dict (taken from the golden fixtures the reference produced), the routing / mask /
schedule helpers, the gradient-bucket layout."""
import json
import math
import os

import pytest
import torch

from util import GOLDEN, load_golden, weights_of


def _meta(name):
    return json.load(open(os.path.join(GOLDEN, "golden_meta.json")))[name]


def _roundtrip(module, fx):
    ref = weights_of(fx)
    res = module.load_state_dict(ref, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    out = module.state_dict()
    assert set(out) == set(ref)                      # same key set as the reference's checkpoint
    for k in ref:
        assert torch.equal(out[k].cpu(), ref[k]), k  # and the values survive the stacking translation


def test_state_dict_keys_match_reference_checkpoints():
    from amk.models import (AgentAttention, BidirectionalDecoder, MoELayer, SoftmaxAttention, SwitchHeadAttention, ViT,
                            ViTMoE, ViTVQGAN)

    fx = load_golden("softmax_attention"); d = [int(v) for v in fx["dims"]]
    _roundtrip(SoftmaxAttention(d[0], d[1], d[2]), fx)
    fx = load_golden("moe_small"); d = [int(v) for v in fx["dims"]]
    _roundtrip(MoELayer(d[0], d[0], d[1], d[2]), fx)
    fx = load_golden("switchhead_small"); d = [int(v) for v in fx["dims"]]
    _roundtrip(SwitchHeadAttention(d[0], d[1], d[2], num_experts=d[3], sel_experts=d[4]), fx)
    fx = load_golden("agent_small"); d = [int(v) for v in fx["dims"]]
    _roundtrip(AgentAttention(d[0], d[1], d[2], agent_num=d[3]), fx)
    m = _meta("vitvqgan_small")
    _roundtrip(ViTVQGAN(m["cfg"], m["codebook"]), load_golden("vitvqgan_small"))
    _roundtrip(ViT(**_meta("vit_small")["cfg"]), load_golden("vit_small"))
    _roundtrip(ViTMoE(**_meta("vit_moe_small")["cfg"]), load_golden("vit_moe_small"))
    _roundtrip(BidirectionalDecoder(**_meta("muse_decoder_small")["cfg"]), load_golden("muse_decoder_small"))


def test_constructor_contracts():
    from amk.models import AgentAttention, MoELayer, ViT

    with pytest.raises(ValueError, match="num_heads"):
        AgentAttention(384, 8, 64, agent_num=47)       # the reference's einsum fails there too
    assert AgentAttention(384, 6, 64).pool_size == 6
    with pytest.raises(ValueError, match="output_dim"):
        MoELayer(64, 32, 4, 2)
    v = ViT(dim=64, image_size=32, patch_size=8, n_heads=1, d_head=64, depth=1, mlp_dim=128, dropout=0.0, num_classes=3)
    assert v.encoder.layers[0].feed_forward.ff[0].weight.shape == (0, 64)   # mult = dropout quirk


def test_vq_nsplit_and_masks():
    from amk import ops

    for N, K in [(1, 32), (2048, 8192), (32768, 8192), (10 ** 6, 8192), (128, 64), (5000, 96)]:
        s = ops.vq_nsplit(N, K)
        assert s >= 1 and K % (32 * s) == 0 and (s == 1 or K // s >= 256)
    m = ops._mask_u8(torch.ones(1, 1, 5, 7, dtype=torch.bool), (5, 7), "causal_mask")
    assert m.dtype == torch.uint8 and tuple(m.shape) == (5, 7) and m.is_contiguous()
    assert tuple(ops._mask_u8(torch.ones(1, 7, dtype=torch.bool), (5, 7), "causal_mask").shape) == (5, 7)  # broadcast rows
    with pytest.raises(RuntimeError):
        ops._mask_u8(torch.ones(2, 3, 4, dtype=torch.bool), (3, 4), "context_mask")
    assert ops._mask_u8(None, (2, 2), "x") is None


def test_lr_schedule_matches_timm_formula():
    from amk.train import cosine_warmup_lr

    base, t_init, warm = 1e-4, 100000, 50000
    assert cosine_warmup_lr(0, base, t_init, warm) == pytest.approx(1e-6)
    assert cosine_warmup_lr(25000, base, t_init, warm) == pytest.approx(1e-6 + 25000 * (base - 1e-6) / warm)
    for t in (50000, 75000, 100000):
        want = 5e-5 + 0.5 * (base - 5e-5) * (1 + math.cos(math.pi * t / t_init))
        assert cosine_warmup_lr(t, base, t_init, warm) == pytest.approx(want)


def test_grad_reducer_bucket_layout():
    """Reverse registration order, grads are views of the flat buckets, unused parameters included."""
    from amk.dp import GradReducer

    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    red = GradReducer(net.parameters(), bucket_bytes=16 * 4 * 4)
    params = list(net.parameters())
    assert red.buckets[0].params[0] is params[-1]                 # last registered parameter first
    assert sum(len(b.params) for b in red.buckets) == len(params)
    for b in red.buckets:
        for p, v in zip(b.params, b.views):
            assert p.grad.data_ptr() == v.data_ptr() and v.data_ptr() >= b.flat.data_ptr()
    red.begin()
    net[:2](torch.randn(3, 8)).sum().backward()                   # the last layer gets no gradient
    red.finish()
    # no gradient this step: .grad is detached (optimizers skip the parameter, as in the reference), zeros travel
    assert params[-1].grad is None and float(red.buckets[0].views[0].abs().sum()) == 0.0
    assert [id(p) for p in red.unused_parameters()] == [id(params[-1]), id(params[-2])]
    assert float(params[0].grad.abs().sum()) > 0.0
    red.zero_grad()
    assert all(float(b.flat.abs().sum()) == 0.0 for b in red.buckets)
    assert all(p.grad is not None for p in params)                # re-attached to the buckets
    # every parameter starts on a 1-KiB boundary of its bucket (amk.optim.FlatAdam walks 256-element segments)
    assert all(o % 256 == 0 for b in red.buckets for o in b.offsets)
    assert red.grads_nbytes() == sum(-(-p.numel() // 256) * 256 for p in params) * 4


def test_discriminator_conv_second_order_matches_nn_conv2d():
    """amk.models.discriminator.Conv2d restates the convolution's first and second derivatives with
    ordinary forward / data-gradient / weight-gradient calls; the gradient-penalty step must give the
    same parameter gradients as plain nn.Conv2d (trainers/vitgqgan.py:115-131 structure)."""
    import torch.nn as nn

    from amk.models.discriminator import NLayerDiscriminator, input_grad_only

    torch.manual_seed(0)
    ours = NLayerDiscriminator(3, 8, 3).double()
    ref = nn.Sequential()
    for i, m in enumerate(ours.model):
        if isinstance(m, nn.Conv2d):
            c = nn.Conv2d(m.in_channels, m.out_channels, 4, stride=m.stride, padding=1, bias=m.bias is not None).double()
            c.load_state_dict(m.state_dict())
            ref.append(c)
        elif isinstance(m, nn.BatchNorm2d):
            ref.append(nn.BatchNorm2d(m.num_features).double())
        else:
            ref.append(nn.LeakyReLU(0.2, True))
    real = torch.rand(3, 3, 64, 64, dtype=torch.double)
    fake = torch.rand(3, 3, 64, 64, dtype=torch.double)
    eta = torch.rand(3, 1, 1, 1, dtype=torch.double)

    def d_loss(net, guard):
        mixed = (eta * real + (1 - eta) * fake).requires_grad_(True)
        pred = net(mixed)
        with guard():
            (g,) = torch.autograd.grad(pred, mixed, grad_outputs=torch.ones_like(pred), create_graph=True, retain_graph=True)
        gp = ((g.norm(2, dim=1) - 1.0) ** 2).mean() * 10.0
        return torch.relu(1.0 - net(real)).mean() + torch.relu(1.0 + net(fake)).mean() + gp

    import contextlib
    la = d_loss(ours, input_grad_only)
    lb = d_loss(ref, contextlib.nullcontext)
    assert abs(float(la) - float(lb)) < 1e-10 * max(1.0, abs(float(lb)))
    ga = torch.autograd.grad(la, list(ours.parameters()))
    gb = torch.autograd.grad(lb, list(ref.parameters()))
    assert len(ga) == len(gb)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(b.abs().max()))


def test_checkpoint_format_round_trip(tmp_path):
    """save_ckpt writes the reference's {'step','state_dict','config'} file; model_factory.load_model
    and resume_from_checkpoint read it back (trainers/utils/base_trainer.py:92-115)."""
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.models.model_factory import load_model
    from amk.train import VQGANTrainStep

    vit = dict(dim=64, img_size=32, patch_size=8, n_heads=1, d_head=64, depth=1, mlp_dim=128, dropout=0.0)
    cb = dict(codebook_size=64, codebook_dim=32)
    torch.manual_seed(1)
    tr = VQGANTrainStep(ViTVQGAN(vit, cb), NLayerDiscriminator(3, 8, 3))
    tr.global_step = 37
    path = str(tmp_path / "vitvqgan.pt")
    tr.save_ckpt(path, config={"model": {"name": "vitvqgan"}})
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"step", "state_dict", "config"} and raw["step"] == 37
    assert "encoder.encoder.layers.0.self_attn.q.0.weight" in raw["state_dict"]
    assert "codebook.embedding.weight" in raw["state_dict"]
    torch.manual_seed(2)
    other = ViTVQGAN(vit, cb)
    load_model(other, path)
    for (n, a), b in zip(tr.model.state_dict().items(), other.state_dict().values()):
        assert torch.equal(a, b), n
    tr2 = VQGANTrainStep(ViTVQGAN(vit, cb), NLayerDiscriminator(3, 8, 3))
    assert tr2.resume_from_checkpoint(path) == {"model": {"name": "vitvqgan"}} and tr2.global_step == 37


def test_switchhead_form_rule(monkeypatch):
    """ops._moe_dense_z: the dense-sum / distinct-row forms apply where a row sums at least E/2 pairs into a wide output
    (SwitchHead at the ViTMoE layer: 16 pairs, 32 experts, 1024 wide, 64 per pair) and not to the top-2 MoE layer."""
    from amk import ops

    monkeypatch.setattr(ops, "MOE_DENSE_Z", True)
    assert ops._moe_dense_z(1024, 64, 16, 32)
    assert ops._moe_dense_z(512, 64, 4, 5)
    assert not ops._moe_dense_z(1024, 1024, 2, 32)      # MoELayer: two pairs per row
    assert not ops._moe_dense_z(1024, 64, 15, 32)       # fewer than E/2 pairs
    assert not ops._moe_dense_z(128, 64, 16, 32)        # narrow output
    assert not ops._moe_dense_z(256, 128, 16, 32)       # output less than 4x the contraction
    monkeypatch.setattr(ops, "MOE_DENSE_Z", False)
    assert not ops._moe_dense_z(1024, 64, 16, 32)


def test_cosine_and_constant_warmup_factors_match_transformers():
    """trainers/vit.py:33 and trainers/utils/scheduler.py:10-13 use transformers' schedules, stepped with an explicit
    index (`scheduler.step(self.global_step)`): lr = base * lambda(index)."""
    transformers = pytest.importorskip("transformers")
    from amk.train import constant_with_warmup_factor, cosine_with_warmup_factor

    for warm, total in ((0, 10), (5, 20), (7, 7), (3, 4)):
        p = [torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.AdamW(p, lr=1.0)
        sch = transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert opt.param_groups[0]["lr"] == cosine_with_warmup_factor(0, warm, total)   # LambdaLR starts at lambda(0)
            for g in range(total + 6):
                sch.step(g)
                assert opt.param_groups[0]["lr"] == pytest.approx(cosine_with_warmup_factor(g, warm, total), abs=1e-15), (warm, total, g)
            opt2 = torch.optim.AdamW(p, lr=1.0)
            sch2 = transformers.get_constant_schedule_with_warmup(opt2, num_warmup_steps=warm)
            for g in range(warm + 4):
                sch2.step(g)
                assert opt2.param_groups[0]["lr"] == pytest.approx(constant_with_warmup_factor(g, warm), abs=1e-15)


def test_classifier_step_follows_the_reference_loop_on_cpu():
    """ClassifierTrainStep (one rank, CPU tensors -> torch.optim.AdamW inside) against the loop of trainers/vit.py:66-76
    written out with torch's own pieces and transformers' scheduler: accumulation of 2, clip, AdamW weight decay 0.01,
    `scheduler.step(global_step)` on sync iterations only (accelerate skips optimizer and scheduler otherwise)."""
    transformers = pytest.importorskip("transformers")
    import copy
    import warnings

    from amk.train import ClassifierTrainStep

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 5))
    ref = copy.deepcopy(net)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, 3, 2, 2, generator=g) for _ in range(6)]
    ys = [torch.randint(0, 5, (4,), generator=g) for _ in range(6)]
    ts = ClassifierTrainStep(net, lr=1e-2, betas=(0.9, 0.95), warmup_steps=2, total_steps=6, max_grad_norm=0.5, accum_steps=2,
                             bucket_bytes=256)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sch = transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=6)
        for gs, (x, y) in enumerate(zip(xs, ys)):
            loss = ts.step(x, y)
            want = torch.nn.functional.cross_entropy(ref(x), y)
            (want / 2).backward()
            if (gs + 1) % 2 == 0:
                torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
                opt.step()
                sch.step(gs)
                opt.zero_grad()
            assert torch.allclose(loss, want, rtol=1e-6, atol=1e-7), gs
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    assert ts.global_step == 6
