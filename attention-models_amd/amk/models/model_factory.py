"""build_model(cfg): the reference's dispatch on ``cfg.model.name`` (models/model_factory.py:24-151)
for the models on the north-star path.  ``cfg`` is anything with attribute access (OmegaConf in the
reference; ``types.SimpleNamespace`` works).  muse / maskgit / vqgan / parti need the CLIP text
tower or the conv VQGAN, which are out of scope."""
import logging

import torch

from .vit import ViT
from .vit_moe import ViTMoE
from .vitvqgan import ViTVQGAN


def load_model(model, checkpoint):
    """strict=False load of a reference checkpoint ({'step','state_dict','config'})."""
    ckpt = torch.load(checkpoint, map_location="cpu", weights_only=True)
    model.load_state_dict(ckpt["state_dict"], strict=False)
    logging.info("Loaded checkpoint %s", checkpoint)


def freeze_model(model):
    for p in model.parameters():
        p.requires_grad = False


def build_model(cfg):
    name = cfg.model.name
    t = getattr(cfg.model, "transformer", None)
    if name == "vitvqgan":
        vit_params = dict(dim=t.dim, img_size=cfg.dataset.preprocessing.resolution, patch_size=t.patch_size,
                          n_heads=t.n_heads, d_head=t.d_head, depth=t.depth, mlp_dim=t.mlp_dim, dropout=t.dropout)
        codebook_params = dict(codebook_dim=cfg.codebook.codebook_dim, codebook_size=cfg.codebook.codebook_size)
        return ViTVQGAN(vit_params, codebook_params)
    if name == "vit":
        return ViT(dim=t.dim, image_size=cfg.dataset.preprocessing.resolution, patch_size=t.patch_size, depth=t.depth,
                   n_heads=t.n_heads, mlp_dim=t.mlp_dim, dropout=t.dropout, num_classes=t.num_classes)
    if name == "vit_moe":  # d_head is not forwarded by the reference either: default 64
        return ViTMoE(dim=t.dim, image_size=cfg.dataset.preprocessing.resolution, n_heads=t.n_heads,
                      patch_size=t.patch_size, depth=t.depth, n_experts=t.n_experts, sel_experts=t.sel_experts,
                      dropout=t.dropout, num_classes=t.num_classes)
    raise NotImplementedError(f"build_model: '{name}' is outside the MI355X hot-path build (vitvqgan, vit, vit_moe)")
