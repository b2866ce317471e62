"""Conv-VQGAN codebook on the libamk.so VQ kernels.

Reference: models/vqgan.py:138-182.  Same l2-normalised nearest-neighbour lookup as the ViT-VQGAN
codebook with a channels-first ``(B, C, H, W)`` input, ``codebook_dim`` 256 by default
(README.md:246-249), uniform init, ``beta`` on the codebook term instead of the commitment term, flat
``(B*H*W,)`` indices and an ``indices_to_embeddings`` that returns the raw rows.  The convolutional
encoder / decoder of that model are not part of the hot path and are not built.
"""
import torch.nn as nn

from .. import ops


class Codebook(nn.Module):
    def __init__(self, codebook_size=1024, codebook_dim=256, beta=0.25):
        super().__init__()
        self.codebook_size = codebook_size
        self.codebook_dim = codebook_dim
        self.beta = beta
        self.embedding = nn.Embedding(codebook_size, codebook_dim)
        self.embedding.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)

    def forward(self, z):
        zt = z.permute(0, 2, 3, 1)                                  # 'b d h w -> b h w d'
        # the kernel's loss is  w * mean((zq.detach() - z)^2) + mean((zq - z.detach())^2); the reference's
        # weights here are (1, beta) = beta * (1/beta, 1)
        z_q, idx, loss = ops.vq_lookup(zt, self.embedding.weight, 1.0 / self.beta)
        return z_q.permute(0, 3, 1, 2), idx.reshape(-1), self.beta * loss

    def indices_to_embeddings(self, indices):
        e = self.embedding(indices)                                  # (B, T, C), no l2-norm here
        side = int(e.shape[1] ** 0.5)
        return e.view(e.shape[0], side, side, -1).permute(0, 3, 1, 2)
