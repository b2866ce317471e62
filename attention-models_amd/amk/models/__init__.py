"""Same export list as the reference's ``models`` package (models/__init__.py:1-14) for the
classes on the hot path and their scaffolding."""
from .attention import AgentAttention, SoftmaxAttention, SwitchHeadAttention
from .model_factory import build_model
from .moe import MoELayer
from .maskgit import BiDirectionalTransformer, MaskGitTransformer
from .muse import MUSE, BidirectionalDecoder
from .vit import ViT
from .vit_moe import ViTMoE
from .vitvqgan import Codebook, ViTVQGAN
from . import vqgan  # conv-VQGAN codebook (models/vqgan.py:138-182): vqgan.Codebook

__all__ = ["SoftmaxAttention", "AgentAttention", "SwitchHeadAttention", "MoELayer", "Codebook", "ViTVQGAN",
           "ViT", "ViTMoE", "MUSE", "BidirectionalDecoder", "MaskGitTransformer", "BiDirectionalTransformer", "build_model"]
