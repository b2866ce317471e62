"""Whole models of BASELINE.json configs[1] and configs[3] (reduced size) on the HIP path with the
reference's weights, against the reference's logits / loss / parameter gradients; plus the
README shape contracts and build_model."""
import json
import os
import types

import pytest
import torch

from util import GOLDEN, assert_close, load_golden, weights_of

pytestmark = pytest.mark.gpu


def _meta(name):
    return json.load(open(os.path.join(GOLDEN, "golden_meta.json")))[name]


def _ref_named_grads(module):
    out = {}
    for prefix, sub in module.named_modules():
        stacked = getattr(sub, "_stacked", {})
        for pname, p in sub.named_parameters(recurse=False):
            full = f"{prefix}.{pname}" if prefix else pname
            if pname in stacked:
                mod, leaf = stacked[pname]
                for e in range(p.shape[0]):
                    key = f"{prefix}.{mod}.{e}.{leaf}" if prefix else f"{mod}.{e}.{leaf}"
                    out[key] = None if p.grad is None else p.grad[e]
            else:
                out[full] = p.grad
    return out


def _abs_close(a, b, tol, what):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    scale = max(float(b.abs().max()), 1e-4)
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: abs err {err:.3e} (scale {scale:.3e})"


def test_vit_small_golden(device):
    from amk.models import ViT

    fx, meta = load_golden("vit_small"), _meta("vit_small")
    m = ViT(**meta["cfg"])
    res = m.load_state_dict(weights_of(fx), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"]
    m = m.to(device)
    logits = m(torch.from_numpy(fx["imgs"]).to(device))
    assert_close(logits, fx["logits"], 5e-5, "logits")
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(fx["labels"]).to(device))
    assert_close(loss, fx["loss"], 5e-5, "loss")
    loss.backward()
    for n, g in _ref_named_grads(m).items():
        if "g:" + n in fx:
            _abs_close(g, fx["g:" + n], 2e-4, f"grad {n}")


def test_vit_moe_small_golden(device):
    from amk.models import ViTMoE

    fx, meta = load_golden("vit_moe_small"), _meta("vit_moe_small")
    m = ViTMoE(**meta["cfg"])
    res = m.load_state_dict(weights_of(fx), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"]
    m = m.to(device)
    logits = m(torch.from_numpy(fx["imgs"]).to(device))
    assert_close(logits, fx["logits"], 5e-5, "logits")
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(fx["labels"]).to(device))
    loss.backward()
    got = _ref_named_grads(m)
    for n, g in got.items():
        if "g:" + n in fx:
            _abs_close(g, fx["g:" + n], 3e-4, f"grad {n}")
        elif n.endswith("W_d.0.weight"):
            assert g is None  # un-weighted moe_out: no gradient, as in the reference


def test_readme_shapes_and_factory(device):
    """Output shapes the reference's README prints (README.md:108,125,141,156,182) and build_model."""
    from amk.models import AgentAttention, SoftmaxAttention, SwitchHeadAttention, ViT, ViTMoE, build_model

    x = torch.randn(2, 10, 512, device=device)
    assert tuple(SoftmaxAttention(512, 16, 64).to(device)(x).shape) == (2, 10, 512)
    assert tuple(SwitchHeadAttention(512, 2, 64, num_experts=5, sel_experts=2).to(device)(x).shape) == (2, 10, 512)
    xa = torch.randn(2, 10, 384, device=device)
    assert tuple(AgentAttention(384, 6, 64).to(device)(xa).shape) == (2, 10, 384)
    imgs = torch.randn(2, 3, 256, 256, device=device)
    vit = ViT(dim=1024, image_size=256, patch_size=32, n_heads=16, d_head=64, depth=2, mlp_dim=2048, num_classes=1000).to(device)
    assert tuple(vit(imgs).shape) == (2, 1000)
    ns = types.SimpleNamespace
    cfg = ns(model=ns(name="vit_moe", transformer=ns(dim=256, n_heads=4, patch_size=32, depth=1, n_experts=8, sel_experts=2,
                                                     dropout=0.0, num_classes=1000)),
             dataset=ns(preprocessing=ns(resolution=256)))
    vm = build_model(cfg).to(device)
    assert isinstance(vm, ViTMoE) and tuple(vm(imgs).shape) == (2, 1000)
    cfg = ns(model=ns(name="vitvqgan", transformer=ns(dim=64, patch_size=32, n_heads=1, d_head=64, depth=1, mlp_dim=64, dropout=0.0)),
             codebook=ns(codebook_dim=32, codebook_size=64), dataset=ns(preprocessing=ns(resolution=256)))
    vq = build_model(cfg).to(device)
    rec, loss = vq(imgs)
    assert tuple(rec.shape) == (2, 3, 256, 256) and loss.dim() == 0
    idx = vq.encode_imgs(imgs)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (2, vq.num_patches)
    assert tuple(vq.decode_indices(idx).shape) == (2, 3, 256, 256)


@pytest.mark.parametrize("variant", ["plain", "ctxmask"])
def test_muse_decoder_small_golden(device, variant):
    """BASELINE.json configs[4]: the masked-token decoder (self- + cross-attention on the HIP kernels)
    with the reference's weights: logits, cross-entropy loss, gradients."""
    from amk.models import BidirectionalDecoder

    fx, meta = load_golden("muse_decoder_small"), _meta("muse_decoder_small")
    m = BidirectionalDecoder(**meta["cfg"])
    res = m.load_state_dict(weights_of(fx), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"]
    m = m.to(device)
    ctx = torch.from_numpy(fx["context"]).to(device).requires_grad_(True)
    kw = {} if variant == "plain" else dict(context_mask=torch.from_numpy(fx["cmask"]).to(device))
    logits = m(torch.from_numpy(fx["ids"]).to(device), context=ctx, **kw)
    assert_close(logits, fx[f"{variant}:logits"], 5e-5, "logits")
    loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2), torch.from_numpy(fx["tgt"]).to(device), ignore_index=-1)
    assert_close(loss, fx[f"{variant}:loss"], 5e-5, "loss")
    loss.backward()
    _abs_close(ctx.grad, fx[f"{variant}:gctx"], 3e-4, "grad context")
    for n, p in m.named_parameters():
        if f"{variant}:g:{n}" in fx:
            _abs_close(p.grad, fx[f"{variant}:g:{n}"], 3e-4, f"grad {n}")


def test_muse_schedule_helpers_and_generate(device):
    """cosine schedule / logit filter against the reference's outputs; fill_mask invariants; one
    training loss and a full 18-step parallel decode (2 decoder passes per step) on the HIP path."""
    from amk.models import MUSE, ViTVQGAN
    from amk.models.muse import cosine_schedule, filter_logits

    fx = load_golden("muse_decoder_small")
    assert_close(cosine_schedule(torch.from_numpy(fx["cosine_t"])), fx["cosine_out"], 1e-6, "cosine_schedule")
    got = filter_logits(torch.from_numpy(fx["filter_in"]).to(device), p=0.9).cpu().numpy()
    assert (got == fx["filter_out"]).all()

    torch.manual_seed(0)
    vq = ViTVQGAN(dict(dim=64, img_size=32, patch_size=8, n_heads=1, d_head=64, depth=1, mlp_dim=64, dropout=0.0),
                  dict(codebook_size=64, codebook_dim=32))
    muse = MUSE(dim=64, vq=vq, text_dim=24, n_heads=1, d_head=64, depth=2, mult=4).to(device)
    assert not any(p.requires_grad for p in muse.vq.parameters())          # frozen quantiser
    tokens = torch.randint(0, 64, (5, 16), device=device)
    inp, tgt = muse.fill_mask(tokens)
    masked = inp == muse.mask_token_id
    assert bool((masked.sum(-1) >= 1).all())
    assert bool((tgt[~masked] == -1).all()) and bool((tgt[masked] == tokens[masked]).all()) and bool((inp[~masked] == tokens[~masked]).all())
    text_hidden = torch.randn(2, 7, 24, device=device)
    imgs = torch.rand(2, 3, 32, 32, device=device)
    loss = muse(text_hidden, imgs)
    loss.backward()
    assert torch.isfinite(loss) and muse.decoder.linear.weight.grad is not None
    out = muse.generate(text_hidden, timesteps=18)
    assert tuple(out.shape) == (2, 3, 32, 32)
    with pytest.raises(TypeError, match="CLIP"):
        muse(["a photo"], imgs)


def test_hip_graph_capture_of_a_train_step(device):
    """The C ABI is graph-safe (async on the caller's stream, no host sync, no allocation): a whole
    ViTMoE step -- routing, grouped GEMMs, fused attention forward/backward, AdamW -- captures into a
    HIP graph and replays with the same result as eager execution."""
    import copy

    from amk.graphs import GraphedStep
    from amk.models import ViTMoE

    torch.manual_seed(0)
    cfg = dict(dim=64, image_size=32, patch_size=8, n_heads=2, d_head=64, depth=2, n_experts=4, sel_experts=2, num_classes=10)
    eager = ViTMoE(**cfg).to(device)
    graphed = copy.deepcopy(eager)
    x = torch.randn(4, 3, 32, 32, device=device)
    y = torch.randint(0, 10, (4,), device=device)

    def make_step(model):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, foreach=True)

        def step(a, b):
            loss = torch.nn.functional.cross_entropy(model(a), b)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=False)
            return loss
        return step

    s_eager = make_step(eager)
    g = GraphedStep(make_step(graphed), [x, y], warmup=3)     # 3 eager warm-up steps; capture itself runs nothing
    for _ in range(3):
        s_eager(x, y)
    for _ in range(3):
        l_e = s_eager(x, y)
        l_g = g.replay(x, y)
    assert_close(l_g, l_e, 1e-4, "loss at the 6th step")
    for (n, a), (_, b) in zip(eager.named_parameters(), graphed.named_parameters()):
        if a.grad is not None:
            a, b = a.detach(), b.detach()
            assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max())), n


def test_train_step_shared_forward_is_the_same_step(device):
    """VQGANTrainStep(share_forward=True) = the reference-shaped step with the duplicated generator
    forward removed: same losses and same parameters after two steps (dropout 0)."""
    import copy

    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    vit = dict(dim=64, img_size=32, patch_size=8, n_heads=1, d_head=64, depth=1, mlp_dim=128, dropout=0.0)
    torch.manual_seed(0)
    model = ViTVQGAN(vit, dict(codebook_size=64, codebook_dim=32)).to(device)
    discr = NLayerDiscriminator(3, 8, 3).to(device)
    model2, discr2 = copy.deepcopy(model), copy.deepcopy(discr)
    imgs = torch.rand(4, 3, 32, 32, device=device)
    a = VQGANTrainStep(model, discr, warmup_steps=1, share_forward=False)
    b = VQGANTrainStep(model2, discr2, warmup_steps=1, share_forward=True)
    for _ in range(2):
        torch.manual_seed(5)   # the gradient penalty draws eta
        la = a.step(imgs)
        torch.manual_seed(5)
        lb = b.step(imgs)
        for k in la:
            assert float((la[k] - lb[k]).abs()) <= 1e-5 * max(1.0, float(la[k].abs())), k
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(p.abs().max())), n
    for (n, p), q in zip(discr.named_parameters(), discr2.parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(p.abs().max())), n


def test_vqgan_step_body_replays_as_hip_graph(device):
    """The whole GAN step (both phases, gradient penalty's double backward, MIOpen convolutions, fused
    Adam, clip) is capture-safe: two replays of the captured step_body = two eager steps.
    gp_lambda = 0 so that the penalty's random eta (a different Philox offset under capture) does not
    enter the comparison."""
    import copy

    from amk import ops
    from amk.graphs import GraphedStep
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    # Adam turns a last-bit difference of a near-zero gradient into a +-lr difference of the parameter, so the
    # comparison runs on the bitwise-reproducible attention backward (no dq atomics)
    monkey = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        _graph_replay_case(device, copy, GraphedStep, ViTVQGAN, NLayerDiscriminator, VQGANTrainStep)
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = monkey


def _graph_replay_case(device, copy, GraphedStep, ViTVQGAN, NLayerDiscriminator, VQGANTrainStep):
    vit = dict(dim=64, img_size=32, patch_size=8, n_heads=1, d_head=64, depth=1, mlp_dim=128, dropout=0.0)
    torch.manual_seed(0)
    model = ViTVQGAN(vit, dict(codebook_size=64, codebook_dim=32)).to(device)
    discr = NLayerDiscriminator(3, 8, 3).to(device)
    model2, discr2 = copy.deepcopy(model), copy.deepcopy(discr)
    imgs = torch.rand(4, 3, 32, 32, device=device)
    eager = VQGANTrainStep(model, discr, warmup_steps=1, gp_lambda=0.0, capturable=True)
    graphed = VQGANTrainStep(model2, discr2, warmup_steps=1, gp_lambda=0.0, capturable=True)
    graphed._set_lr()
    g = GraphedStep(lambda x: graphed.step_body(x), [imgs], warmup=1)   # runs one eager step, then captures
    graphed.global_step = 1
    eager.step(imgs)                                                     # the same one step
    for _ in range(2):
        le = eager.step(imgs)
        graphed._set_lr()
        lg = g.replay(imgs)
        graphed.global_step += 1
        for k in le:
            assert float((le[k] - lg[k]).abs()) <= 2e-5 * max(1.0, float(le[k].abs())), k
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        assert float((p - q).abs().max()) <= 2e-5 * max(1.0, float(p.abs().max())), n
    # the same through VQGANTrainStep.capture(): step() replays the graph (what bench.py times on one rank)
    torch.manual_seed(0)
    model3 = ViTVQGAN(vit, dict(codebook_size=64, codebook_dim=32)).to(device)
    discr3 = NLayerDiscriminator(3, 8, 3).to(device)
    model3.load_state_dict(model2.state_dict())
    discr3.load_state_dict(discr2.state_dict())
    model4, discr4 = copy.deepcopy(model3), copy.deepcopy(discr3)
    e2 = VQGANTrainStep(model3, discr3, warmup_steps=1, gp_lambda=0.0, capturable=True)
    c2 = VQGANTrainStep(model4, discr4, warmup_steps=1, gp_lambda=0.0, capturable=True)
    e2.step(imgs)
    c2.step(imgs)            # eager: also fixes the set of parameters with gradients
    c2.capture(imgs, warmup=0)
    for _ in range(2):
        le, lc = e2.step(imgs), c2.step(imgs)
        for k in le:
            assert float((le[k] - lc[k]).abs()) <= 2e-5 * max(1.0, float(le[k].abs())), k
    assert c2.global_step == e2.global_step == 3
    for (n, p), q in zip(model3.named_parameters(), model4.parameters()):
        assert float((p - q).abs().max()) <= 2e-5 * max(1.0, float(p.abs().max())), n


def test_accumulated_step_capture_matches_eager_micro_steps(device, monkeypatch):
    """VQGANTrainStep.capture_accumulated: one optimizer step over two micro-batches (the reference's shipped
    gradient_accumulation_steps: 2, cfg/vitvqgan.yaml:72-76, as accelerator.accumulate runs it) replayed as one HIP
    graph gives the parameters of the eager micro-steps."""
    import copy

    from amk import ops
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    monkeypatch.setattr(ops, "DETERMINISTIC_ATTENTION_BACKWARD", True)   # (no dq atomics: comparable to 2e-5)
    vit = dict(dim=64, img_size=32, patch_size=8, n_heads=1, d_head=64, depth=1, mlp_dim=128, dropout=0.0)
    torch.manual_seed(0)
    model = ViTVQGAN(vit, dict(codebook_size=64, codebook_dim=32)).to(device)
    discr = NLayerDiscriminator(3, 8, 3).to(device)
    model2, discr2 = copy.deepcopy(model), copy.deepcopy(discr)
    a, b = torch.rand(2, 3, 32, 32, device=device), torch.rand(2, 3, 32, 32, device=device)
    eager = VQGANTrainStep(model, discr, warmup_steps=1, gp_lambda=0.0, capturable=True)
    graphed = VQGANTrainStep(model2, discr2, warmup_steps=1, gp_lambda=0.0, capturable=True)
    eager.step_accumulated([a, b])
    graphed.step_accumulated([a, b])          # eager: also fixes the set of parameters with gradients
    graphed.capture_accumulated([a, b], warmup=0)
    for _ in range(2):
        le, lg = eager.step_accumulated([a, b]), graphed.step_accumulated([a, b])
        for k in le:
            assert float((le[k] - lg[k]).abs()) <= 2e-5 * max(1.0, float(le[k].abs())), k
    assert eager.global_step == graphed.global_step == 6
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        assert float((p - q).abs().max()) <= 2e-5 * max(1.0, float(p.abs().max())), n
