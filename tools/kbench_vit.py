"""BASELINE.json configs[1]: ViT classifier (dim 1024, patch 32, 16 heads, depth 6 -> 65 tokens) training
step (forward, cross-entropy, backward, AdamW) on synthetic images.

    python tools/kbench_vit.py [batch ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attention-models_amd"))
import torch  # noqa: E402

from amk import tuning  # noqa: E402
from amk.models import ViT  # noqa: E402

tuning.enable_gemm_tuning(results_csv=None)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ViT(dim=1024, image_size=256, patch_size=32, n_heads=16, d_head=64, depth=6, mlp_dim=2048, dropout=0.0,
            num_classes=1000).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
print(f"ViT parameters: {sum(p.numel() for p in model.parameters())/1e6:.1f} M")
for B in ([int(b) for b in sys.argv[1:]] or (2, 64, 256)):
    imgs = torch.randn(B, 3, 256, 256, device=dev)
    labels = torch.randint(0, 1000, (B,), device=dev)

    def step():
        loss = torch.nn.functional.cross_entropy(model(imgs), labels)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"batch {B:3d}: {dt*1e3:7.2f} ms per step = {B/dt:8.1f} images/s")
