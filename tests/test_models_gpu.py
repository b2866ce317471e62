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
