"""Same export list as the reference's ``models`` package (models/__init__.py:1-14) for the
classes on the hot path and their scaffolding."""
from .attention import AgentAttention, SoftmaxAttention, SwitchHeadAttention
from .moe import MoELayer
from .vitvqgan import Codebook, ViTVQGAN

__all__ = ["SoftmaxAttention", "AgentAttention", "SwitchHeadAttention", "MoELayer", "Codebook", "ViTVQGAN"]
